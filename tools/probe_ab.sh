#!/bin/bash
# Development aid: A/B two versions of one source file on the same GPU box.
#   [BENCH_ARGS="--cloudy-pass 1"] bash tools/probe_ab.sh <rounds> <file under csrc> <variant A> <variant B> [<variant C> ...]
rounds=$1; f=$2; shift 2
out=gpurun_out/probe_ab; mkdir -p $out
export TMPDIR=/tmp
cp wrfchem-arc-interactions_amd/csrc/$f /tmp/orig_$f
for r in $(seq $rounds); do
for v in "$@"; do
  cp $v wrfchem-arc-interactions_amd/csrc/$f
  make -s -C wrfchem-arc-interactions_amd/csrc > $out/build.log 2>&1 || { echo "build failed for $v"; tail -5 $out/build.log; continue; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --host-path-rows 0 --cloudy-pass 0 $BENCH_ARGS > $out/log.txt 2>&1
  fcsv=$(find $out/t -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$fcsv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ("lw_" in r["Name"] or "sw_col" in r["Name"] or "optics" in r["Name"]) and "fill" not in r["Name"]:
        print("   %-50s avg %8.3f ms" % (r["Name"][:50], float(r["AverageNs"]) / 1e6))
PY
  rm -rf $out/t
done
done
cp /tmp/orig_$f wrfchem-arc-interactions_amd/csrc/$f; make -s -C wrfchem-arc-interactions_amd/csrc > /dev/null 2>&1
