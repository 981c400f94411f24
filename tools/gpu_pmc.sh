#!/bin/bash
# Quick counter passes for the kernels matching a regex (development aid; tools/profile_gpu.sh collects the full set):
#   bash tools/gpu_pmc.sh <tag> <kernel regex> [bench args]
tag=$1; re=$2; shift 2
out=gpurun_out/pmc_${tag}
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i + 1))
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-include-regex "$re" --output-format csv -d $out/p$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu --host-path-rows 0 --cloudy-pass 0 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 - $out <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print("[%s]" % k)
    for c in sorted(m):
        print("  %-22s %.4g" % (c, m[c]))
    if "GRBM_GUI_ACTIVE" in m and "SQ_WAVE_CYCLES" in m:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; issue slots = cycles x 1024 SIMDs / 4 cycles per wave64 instruction
        slots = m["GRBM_GUI_ACTIVE"] / 8 * 256
        print("  waves/SIMD ~ %.2f   VALU issue %.3f   cycles/XCD %.4g" % (m["SQ_WAVE_CYCLES"] * 8 / (m["GRBM_GUI_ACTIVE"] * 256),
              m.get("SQ_INSTS_VALU", 0) / slots, m["GRBM_GUI_ACTIVE"] / 8))
    if "FETCH_SIZE" in m:
        print("  HBM read %.2f GB  write %.2f GB" % (m["FETCH_SIZE"] * 1024 * 2 / 1e9, m["WRITE_SIZE"] * 1024 / 1e9))
PY
