#!/bin/bash
# One GPU-box visit while developing: the named test files, then a short bench of the default workload.
#   gpurun --timeout 900 -- 'bash tools/gpu_check.sh <tag> "<pytest args>" [bench args]'
tag=$1; tests=$2; shift 2
mkdir -p gpurun_out
if [ -n "$tests" ]; then
  timeout -k 10 600 python -m pytest $tests -x -q > gpurun_out/${tag}_tests.log 2>&1
  rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/${tag}_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --host-path-rows 0 --cloudy-pass 0 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json")); r = d["roofline"]
print("step %.2f ms  sw %.2f  lw %.2f  optics %.2f (sw %.2f lw %.2f)" % (d["ms_per_step"], r["sw_kernel_ms"], r["lw_kernel_ms"], r["optics_stage_ms"], r["optics_kernel_ms"], r["optics_lw_kernel_ms"]))
PY
