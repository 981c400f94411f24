#!/bin/bash
# Kernel trace of the mixed-precision optics configuration and one rank's share of BASELINE configs[4]
# (1000x1000x60 over 8 GPUs = 125 000 columns per rank).  Run on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/profile_mixed.sh r01_v8'
set -eo pipefail
tag=${1:-r01}
out=gpurun_out/prof_${tag}
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_mixed" -o trace -- \
    python3 bench.py --mode full --optics-precision mixed --steps 5 --warmup 2 --no-cpu > "$out/trace_mixed.log" 2>&1
echo "mixed trace done"
python3 bench.py --optics-precision mixed --steps 10 --warmup 2 > "$out/bench_mixed_200x200x40.json" 2> "$out/bench_mixed.err"
echo "mixed bench done"
timeout -k 10 600 python3 bench.py --ni 1000 --nj 125 --nk 60 --optics-precision mixed --steps 3 --warmup 1 --no-cpu \
    > "$out/bench_mixed_1000x125x60.json" 2> "$out/bench_share.err"
echo "config-4 share done"
