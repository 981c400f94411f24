#!/bin/bash
set -e
L=wrfchem-arc-interactions_amd/librrtmg_hip.so
cp $L /tmp/lib_orig.so
for v in b2; do
  cp tools/tmpv/lib_$v.so $L
  echo "== $v"
  timeout -k 10 120 python bench.py --mode sw --steps 20 --warmup 3 --no-cpu 2>&1 | tail -1 | cut -c160-200
done
cp /tmp/lib_orig.so $L
