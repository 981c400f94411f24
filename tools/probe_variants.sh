#!/bin/bash
# Development aid: time bench.py with lw_kernel.hip rebuilt under each set of -D probe macros (on the GPU box).
#   bash tools/probe_variants.sh <tag> "<flags of variant 1>" "<flags of variant 2>" ...
tag=$1; shift
out=gpurun_out/probe_${tag}; mkdir -p $out
export TMPDIR=/tmp
cd wrfchem-arc-interactions_amd/csrc
for v in "$@"; do
  touch lw_kernel.hip sw_kernel.hip optics_kernel.hip
  make -s EXTRA="$v" > ../../$out/build.log 2>&1 || { echo "build failed for [$v]"; tail -5 ../../$out/build.log; continue; }
  ( cd ../.. && rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --host-path-rows 0 --cloudy-pass 0 $BENCH_ARGS > $out/log.txt 2>&1
    f=$(find $out/t -name "*kernel_stats.csv" | head -1)
    echo "== [$v]"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ("lw_" in r["Name"] or "sw_col" in r["Name"] or "optics" in r["Name"]):
        print("   %-50s avg %8.3f ms" % (r["Name"][:50], float(r["AverageNs"]) / 1e6))
PY
    rm -rf $out/t )
done
touch lw_kernel.hip sw_kernel.hip optics_kernel.hip; make -s > /dev/null 2>&1
