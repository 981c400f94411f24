#!/bin/bash
# Kernel durations of a short bench run (rocprofv3 --kernel-trace --stats):  bash tools/gpu_kstats.sh <tag> [bench args]
tag=$1; shift
out=gpurun_out/ks_${tag}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --host-path-rows 0 --cloudy-pass 0 "$@" > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print("%-60s calls %4s avg %10.3f ms  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
