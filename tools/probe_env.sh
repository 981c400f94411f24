#!/bin/bash
# Development aid: kernel times of the bench under different environments:  bash tools/probe_env.sh "A=1 B=2" "C=3" ...
out=gpurun_out/probe_env; mkdir -p $out
export TMPDIR=/tmp
for v in "$@"; do
  ( export $v; rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --host-path-rows 0 --cloudy-pass 0 > $out/log.txt 2>&1 )
  f=$(find $out/t -name "*kernel_stats.csv" | head -1)
  echo "== [$v]"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ("lw_" in r["Name"] or "sw_col" in r["Name"]) and "optics" not in r["Name"] and "fill" not in r["Name"]:
        print("   %-50s calls %3s avg %8.3f ms" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  rm -rf $out/t
done
