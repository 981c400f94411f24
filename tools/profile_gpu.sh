#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/profile_gpu.sh r01'
# Writes under gpurun_out/prof_<tag>/ ; tools/pmc_summary.py turns that into profiles/<tag>_*.
# Kernel trace and counter passes are separate runs (counters serialise the dispatches), and every
# --pmc pass carries only counters (no other trace domain).
set -eo pipefail
tag=${1:-r02}
shift || true
out=gpurun_out/prof_${tag}
mkdir -p "$out"
export TMPDIR=/tmp
# the bench workload to profile: default = bench.py's default (BASELINE configs[3]); extra arguments are passed on,
# e.g.  bash tools/profile_gpu.sh r02_c2 --config 2
cmd=(python3 bench.py --steps 3 --warmup 1 --no-cpu --host-path-rows 0 --cloudy-pass 0 "$@")
"${cmd[@]}" > "$out/bench.json" 2> "$out/bench.err" || echo "plain bench run failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- "${cmd[@]}" > "$out/trace.log" 2>&1
echo "trace done"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o pmc -- "${cmd[@]}" > "$out/pmc$i.log" 2>&1 || echo "pass $i ($set) failed"
    echo "pmc pass $i done"
done
