#!/usr/bin/env python3
"""Benchmark of the hot path: aerosol + clean-sky RRTMG double call on MI355X.

A "step" is one pass of the HIP path over one batch of synthetic WRF columns
resident in HBM.  Workload at N=1 (BASELINE.json configs[1]):
200x200x40 synthetic columns, RRTMG-SW aerosol+clean double call, fp64.
With --gpus N each rank owns its own 200x200 slab of (i,j) columns (weak
scaling, no data-path collective); the only exchange is an all-reduce of the
per-tile flux sums (16 doubles) over RCCL.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
FP64_VALU_PEAK_TFLOPS = 78.6   # vendor fp64 vector peak


def algorithmic_bytes_per_column(nk, real_bytes=8):
    """Compulsory boundary traffic of the SW double call with the WRF-Chem compact aerosol form
    (DESIGN.md 'Measurement'): reads p3d,t3d,qv3d,pi3d (4K), p8w,t8w (2(K+1)), 12 aerosol arrays (12K),
    3 surface scalars; writes rthratensw (K) and 22 two-dimensional fields."""
    return ((4 + 12 + 1) * nk + 2 * (nk + 1) + 3 + 22) * real_bytes


def cpu_baseline(tab, st, ncol_sample):
    """Reference Fortran (oracle/_ref/ref_sw, built from /root/reference by oracle/build_ref.sh) timed on
    one host core over a bounded sample of the same workload; falls back to the C port if the binary is absent."""
    from tools.make_golden import sw_columns_from_state, run_ref_sw
    cols = sw_columns_from_state(st)
    cols = {k: v[:ncol_sample] for k, v in cols.items()}
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_sw")
    if os.path.exists(exe):
        t0 = time.perf_counter()
        _, stdout = run_ref_sw(tab, cols, clean=1)
        wall = time.perf_counter() - t0
        secs = float(stdout.split("seconds=")[1].split()[0])   # solver loop only, excludes table init and file I/O
        return {"value": ncol_sample / secs, "unit": "columns/s", "cores": 1, "kind": "reference",
                "sample": f"{ncol_sample} columns of the bench workload through rrtmg_sw (flang -O2, fp64), "
                          f"{secs:.2f} s solver time ({wall:.2f} s incl. I/O)"}
    from oracle import oracle as orc
    o = orc.SwOracle(tab)
    n = min(ncol_sample, 512)
    t0 = time.perf_counter()
    o.rrtmg_sw({k: v[:n] for k, v in cols.items()}, clean=1)
    secs = time.perf_counter() - t0
    return {"value": n / secs, "unit": "columns/s", "cores": 1, "kind": "port",
            "sample": f"{n} columns through the C restatement (gcc -O2), {secs:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ni", type=int, default=200)
    ap.add_argument("--nj", type=int, default=200)
    ap.add_argument("--nk", type=int, default=40)
    ap.add_argument("--night", type=float, default=0.0, help="fraction of night columns (0 for kernel timing)")
    ap.add_argument("--cpu-sample", type=int, default=6000)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import wrfchem_arc_amd as wa
    from tests.helpers import AER, IN2, IN3, OUT2, dims_of

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if not os.path.exists(wa.engine.LIB_PATH):
        ge.build_product()
    eng = wa.engine
    tmp = tempfile.mkdtemp(prefix="rrtmg_bench_")
    tab = os.path.join(tmp, "RRTMG_SW_DATA")
    wa.synth.write_sw_data(tab, seed=20260001, real_bytes=8)
    eng.rrtmg_swinit(tab, 8, device=local_rank)

    # each rank: its own (i,j) slab of the domain, seed = 20260000 + config# + rank
    st = wa.synth.make_wrf_state(args.ni, args.nj, args.nk, seed=20260002 + 1000 * rank, night_fraction=args.night)
    ncol = st.ncol
    d = dims_of(st)
    dev_in = {n: torch.from_numpy(np.ascontiguousarray(st.a2[n])).cuda() for n in IN2}
    dev_in.update({n: torch.from_numpy(np.ascontiguousarray(st.a3[n])).cuda() for n in IN3 + AER})
    dev_out = {"rthratensw": torch.zeros((st.nj, st.nk + 1, st.ni), dtype=torch.float64, device="cuda")}
    dev_out.update({n: torch.zeros((st.nj, st.ni), dtype=torch.float64, device="cuda") for n in OUT2})
    sums = torch.zeros(16, dtype=torch.float64, device="cuda")
    flux12 = ["swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
              "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln"]

    def step():
        eng.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **dev_in, **dev_out)
        if world > 1:
            # the only exchange on the path: per-tile flux / forcing sums for the domain means
            for i, n in enumerate(flux12):
                sums[i] = dev_out[n].sum()
            sums[12] = sums[2] - sums[0]
            sums[13] = float(ncol)
            dist.all_reduce(sums)

    for _ in range(args.warmup):
        step()
    eng.sw_status()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev0.record()
        eng.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **dev_in, **dev_out)
        ev1.record()
        if world > 1:
            for i, n in enumerate(flux12):
                sums[i] = dev_out[n].sum()
            sums[12] = sums[2] - sums[0]
            sums[13] = float(ncol)
            dist.all_reduce(sums)
        kernel_ms.append((ev0, ev1))
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.sw_status()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    kms = float(np.mean([a.elapsed_time(b) for a, b in kernel_ms]))   # HIP events on the launch stream

    if rank == 0:
        value = world * ncol * args.steps / elapsed
        bpc = algorithmic_bytes_per_column(args.nk)
        achieved = ncol * bpc / (kms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "columns/sec, aerosol+clean RRTMG SW+LW double-call; flux RMSE vs reference",
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.ni}x{args.nj}x{args.nk} synthetic column batch per GPU, RRTMG-SW "
                                   "aerosol+clean double-call only, fp64 (BASELINE configs[1]); LW not yet on the path",
                       "columns_per_gpu": ncol, "layers": args.nk, "night_fraction": args.night,
                       "aerosol_form": "WRF-Chem compact (tauaer/waer/gaer at 300/400/600/999 nm)",
                       "parallelism": f"(i,j) slabs over {world} GPU(s), RCCL all-reduce of 16 flux sums"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "sw_column_kernel", "kernel_ms": kms, "algorithmic_bytes_per_column": bpc,
                         "note": "fp64-VALU-bound path (SURVEY 8d): arithmetic intensity ~1e3 flop/B, "
                                 "see DESIGN.md for the fp64 ceiling"},
        }
        if world == 1 and not args.no_cpu:
            try:
                line["cpu_baseline"] = cpu_baseline(tab, st, min(args.cpu_sample, ncol))
            except Exception as e:   # never let the baseline leg kill the bench line
                line["cpu_baseline"] = {"value": None, "unit": "columns/s", "cores": 0, "kind": "reference",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
