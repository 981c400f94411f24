#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile_gpu.sh into profiles/.

    python tools/pmc_summary.py gpurun_out/prof_r01 r01

writes profiles/<tag>_kernel_stats.csv (copy of the --kernel-trace --stats summary),
profiles/<tag>_pmc.txt (mean counter value per dispatch of each rrtmg:: kernel) and refreshes
profiles/traffic_latest.json, the per-launch HBM byte counts bench.py reports as roofline.traffic
(FETCH_SIZE and WRITE_SIZE are in KiB; gfx950 reports half the fetched bytes, hence the x2 --
MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    os.makedirs(root, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(root, f"{tag}_kernel_stats.csv"))
    sums = defaultdict(lambda: defaultdict(float))
    cnts = defaultdict(lambda: defaultdict(int))
    meta = {}
    for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                if "rrtmg::" not in k:
                    continue
                # "void rrtmg::sw_column_kernel<false, false>(rrtmg::SwKernelArgs)" -> "rrtmg::sw_column_kernel":
                # the instantiations of one kernel (icloud, profile outputs) are reported under one name
                k = k.replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
                if k.startswith("void "):
                    k = k[5:]
                sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnts[k][r["Counter_Name"]] += 1
                meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                           r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    traffic = {}
    lines = [f"# rocprofv3 --pmc <set> -- python3 bench.py --steps 5 --warmup 2 --no-cpu (tools/profile_gpu.sh; one pass per",
             "# counter set); mean per dispatch.  HBM bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count) + WRITE_SIZE KiB x 1024."]
    for k in sorted(sums):
        g, wg, lds, vgpr, agpr, sgpr, scr = meta[k]
        lines.append(f"\n[{k}] grid={g} wg={wg} lds={lds} vgpr={vgpr} agpr={agpr} sgpr={sgpr} scratch={scr}")
        mean = {c: sums[k][c] / cnts[k][c] for c in sums[k]}
        for c in sorted(mean):
            lines.append(f"{c:32s} {mean[c]:.6g}")
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            rd, wr = mean["FETCH_SIZE"] * 1024 * 2, mean["WRITE_SIZE"] * 1024
            lines.append(f"# HBM bytes per launch: read {rd:.4g} + write {wr:.4g} = {rd + wr:.4g}")
            traffic[k.replace("rrtmg::", "")] = {"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr,
                                                 "fetch_size_kb": mean["FETCH_SIZE"], "write_size_kb": mean["WRITE_SIZE"]}
        if "SQ_BUSY_CYCLES" in mean and "SQ_INSTS_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs, and a wave64 fp64/fp32 VALU
            # instruction occupies its SIMD for 4 cycles
            peak = mean["GRBM_GUI_ACTIVE"] / 8 * 256 * 4 / 4
            lines.append(f"# VALU issue: {mean['SQ_INSTS_VALU']:.4g} wave-instructions / {peak:.4g} issue slots "
                         f"= {mean['SQ_INSTS_VALU'] / peak:.3f}")
            traffic.setdefault(k.replace("rrtmg::", ""), {}).update(
                valu_wave_instructions=mean["SQ_INSTS_VALU"], valu_issue_frac=mean["SQ_INSTS_VALU"] / peak)
    with open(os.path.join(root, f"{tag}_pmc.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if traffic:
        sw = traffic.get("sw_column_kernel", {})
        cols = None
        try:   # the un-profiled run of the same command: how many columns one step covered, and how many steps ran
            bj = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
            cols = bj["config"]["columns_rank0"]
            calls = bj["steps"] + bj["warmup"]
            shutil.copy(os.path.join(src, "bench.json"), os.path.join(root, f"{tag}_bench.json"))
            for k, v in traffic.items():
                # large tiles run in several column batches per step (scratch caps): columns per dispatch follow from
                # the dispatch count; the optics kernels are also called 3 more times each for their separate timing
                n = cnts["rrtmg::" + k].get("FETCH_SIZE") or max(cnts["rrtmg::" + k].values())
                per_step = n / (calls + (3 if k.startswith("optics") else 0))
                v["dispatches_per_step"] = per_step
                v["columns_per_dispatch"] = cols / per_step
        except Exception:
            pass
        try:   # the commit whose build was profiled (this script runs in the build container, where .git exists)
            import subprocess
            commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            commit = None
        out = {"source": f"profiles/{tag}_pmc.txt", "source_commit": commit,
               "collected": "builder-run rocprofv3 --pmc passes of bench.py (tools/profile_gpu.sh), not by the run that prints it",
               "columns_per_launch": cols, "kernels": traffic,
               "hbm_bytes_per_launch": sw.get("hbm_bytes_per_launch"),
               "fetch_size_kb": sw.get("fetch_size_kb"), "write_size_kb": sw.get("write_size_kb")}
        # traffic_latest.json is what bench.py reads: keep it on the default workload (pass --keep-latest for others)
        target = f"{tag}_traffic.json" if "--keep-latest" in sys.argv else "traffic_latest.json"
        with open(os.path.join(root, target), "w") as fh:
            json.dump(out, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
