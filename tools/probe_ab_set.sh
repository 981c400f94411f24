#!/bin/bash
# Development aid: A/B sets of source files on the same GPU box -- each variant is a directory whose files are laid over csrc/.
#   [BENCH_ARGS="--cloudy-pass 1"] bash tools/probe_ab_set.sh <rounds> <dir A> <dir B> [<dir C> ...]
rounds=$1; shift
out=gpurun_out/probe_ab; mkdir -p $out /tmp/orig_set
export TMPDIR=/tmp
for v in "$@"; do for f in $v/*; do cp -n wrfchem-arc-interactions_amd/csrc/$(basename $f) /tmp/orig_set/; done; done
for r in $(seq $rounds); do
for v in "$@"; do
  cp $v/* wrfchem-arc-interactions_amd/csrc/
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
cp /tmp/orig_set/* wrfchem-arc-interactions_amd/csrc/; make -s -C wrfchem-arc-interactions_amd/csrc > /dev/null 2>&1
