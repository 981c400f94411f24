#!/usr/bin/env python3
"""Benchmark of the hot path: aerosol + clean-sky RRTMG SW+LW double call on MI355X.

A "step" is one pass of the HIP path -- MOSAIC optical averaging, then RRTMG_SWRAD + RRTMG_LWRAD, each with its
aerosol and its clean-sky call -- over one synthetic domain resident in HBM.  The default workload is the domain the
north-star target is quoted on, BASELINE.json configs[3]: 500x500x50, MOSAIC 8-bin optics, SW+LW, fp64.
`--config 1|2|4` select the other BASELINE configurations (1: 200x200x40 SW only, 2: 200x200x40 4-bin SW+LW,
4: 1000x1000x60 mixed-precision optics); `--ni/--nj/--nk/--nbin/--mode/--optics-precision` override single items.

With --gpus N the SAME domain is cut into N j-slabs, one rank per GPU.  `python3 bench.py --gpus N` starts its own
ranks: when no launcher environment is present (WORLD_SIZE unset) the parent -- before anything touches the GPU --
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (never an exec), relays
its output and exit code.  Launched by torch.distributed.run directly (RANK/LOCAL_RANK/WORLD_SIZE set) it is a rank.
`--dry-launch` walks the same launch path without a GPU (gloo): every rank reports its slab, rank 0 prints them.
The domain is cut
(tiles.slab_bounds, weighted by the cost of a row when --night > 0; reference tile model DRV:975-982): every rank builds
only its slab, there is no data-path collective, and the one exchange is an all-reduce (RCCL) of the 32 per-tile flux
sums -- "scaling": "strong".  `--scaling weak` gives every rank its own full domain instead.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6        # fp64 vector peak: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz
P_TOP = 5000.0                 # Pa, model top of the synthetic atmosphere

CONFIGS = {
    1: dict(ni=200, nj=200, nk=40, mode="sw", nbin=4, prec="f64"),
    2: dict(ni=200, nj=200, nk=40, mode="full", nbin=4, prec="f64"),
    3: dict(ni=500, nj=500, nk=50, mode="full", nbin=8, prec="f64"),
    4: dict(ni=1000, nj=1000, nk=60, mode="full", nbin=4, prec="mixed"),
}

# argument names of the two radiation entry points (the reference's dummy-argument names, SW:9901-9945, LW:11451-11486)
SW_IN2 = ["xcoszen", "albedo", "tsk"]
ST_IN3 = ["t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d"]
SW_AER = ["tauaer300", "tauaer400", "tauaer600", "tauaer999", "gaer300", "gaer400", "gaer600", "gaer999",
          "waer300", "waer400", "waer600", "waer999"]
SW_OUT2 = ["coszr", "gsw", "swcf", "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
           "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln",
           "swvisdir", "swvisdif", "swnirdir", "swnirdif", "swddir", "swddni", "swddif"]
LW_IN2 = ["emiss", "tsk"]
LW_AER = [f"tauaerlw{b}" for b in range(1, 17)]
LW_OUT2 = ["glw", "olr", "lwcf", "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
           "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"]
SW12 = ["swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
        "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln"]
LW12 = ["lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
        "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"]
CLD3 = ["cldfra3d", "qc3d", "qr3d", "qi3d", "qs3d"]
CLD2 = ["xland", "xice", "snow"]


def algorithmic_bytes_per_column(nk, mode, real_bytes=8):
    """Compulsory boundary traffic with the WRF-Chem compact aerosol form.
    SW: reads p3d,t3d,qv3d,pi3d (4K), p8w,t8w (2(K+1)), 12 aerosol arrays (12K), 3 surface scalars;
        writes rthratensw (K) and 22 two-dimensional fields.
    LW: reads the same 6 state arrays again (second launch), 16 aerosol arrays (16K), 2 scalars;
        writes rthratenlw (K) and 15 two-dimensional fields."""
    sw = (4 + 12 + 1) * nk + 2 * (nk + 1) + 3 + 22
    lw = (4 + 16 + 1) * nk + 2 * (nk + 1) + 2 + 15
    return (sw + (lw if mode != "sw" else 0)) * real_bytes, sw * real_bytes, lw * real_bytes


def optics_bytes_per_column(nk, nbin, real_bytes=8, part="all"):
    """Optical averaging: reads 10 tracers per section + alt + dz8w per cell, writes tau/omega/g at 4 wavelengths
    (optics_kernel) and 16 longwave optical depths (optics_lw_kernel) per cell; each kernel reads the tracers."""
    rd = 10 * nbin + 2
    if part == "sw":
        return (rd + 12) * nk * real_bytes
    if part == "lw":
        return (rd + 16) * nk * real_bytes
    return (2 * rd + 12 + 16) * nk * real_bytes


def dims_of(ni, nj, nk, tile=None):
    d = dict(ims=1, ime=ni, jms=1, jme=nj, kms=1, kme=nk + 1, its=1, ite=ni, jts=1, jte=nj, kts=1, kte=nk)
    if tile:
        d.update(tile)
    return d


def night_rows(ni, nj, seed, frac):
    """Per-row night mask of the global domain (any rank can evaluate any row): True = night column."""
    if frac <= 0.0:
        return np.zeros((nj, ni), dtype=bool)
    return np.stack([np.random.default_rng(seed + 104729 * j).uniform(size=ni) < frac for j in range(nj)])


# ----------------------------------------------------------------------------------------------------------------
# CPU-baseline leg: the only part of bench.py that touches oracle/ (and oracle/_ref, the reference Fortran compiled
# from /root/reference by oracle/build_ref.sh; the binaries travel, the sources do not).
# ----------------------------------------------------------------------------------------------------------------
def _ref_seconds(run, tab, cols, nproc):
    """Runs the reference harness on `cols` split over nproc processes; returns (solver seconds of the slowest
    process, wall seconds)."""
    n = len(next(iter(cols.values())))
    cuts = [n * p // nproc for p in range(nproc + 1)]
    parts = [{k: v[cuts[p]:cuts[p + 1]] for k, v in cols.items()} for p in range(nproc) if cuts[p + 1] > cuts[p]]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(parts)) as ex:
        outs = list(ex.map(lambda c: run(tab, c, clean=1)[1], parts))
    wall = time.perf_counter() - t0
    return max(float(o.split("seconds=")[1].split()[0]) for o in outs), wall


def cpu_baseline(wa, tab_sw, tab_lw, st, ms, mode, ncol_1t, ncol_per_core, ncol_optics):
    """Reference Fortran rrtmg_sw + rrtmg_lw (oracle/_ref, flang -O2, fp64) on a bounded sample of the bench
    workload: once on one core, once with one process per host core (columns are independent -- the reference's own
    parallelism is MPI patches / OpenMP tiles of columns, DRV:975-982).  The optical-averaging stage has no reference
    arithmetic: its CPU time comes from the C port of the engine's own specification (OpenMP over
    columns)."""
    from tools.make_golden import sw_columns_from_state, run_ref_sw
    from tools.make_golden_lw import lw_columns_from_state, run_ref_lw
    have_ref = all(os.path.exists(os.path.join(ROOT, "oracle", "_ref", b)) for b in ("ref_sw", "ref_lw"))
    if not have_ref:
        raise RuntimeError("oracle/_ref/ref_sw, ref_lw not built (oracle/build_ref.sh needs /root/reference)")
    host_cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(host_cpus, 64)      # processes / threads actually used (a GPU box shares its host; 64 bounds the load)
    # the sample: the first rows of the slab (the column lists are built for those rows only)
    import copy
    rows = min(st.nj, -(-max(ncol_1t, ncol_per_core * cores) // st.ni))
    sub = copy.copy(st)
    sub.nj = rows
    sub.a2 = {n: v[:rows] for n, v in st.a2.items()}
    sub.a3 = {n: v[:rows] for n, v in st.a3.items()}
    ncol = sub.ncol
    swc = sw_columns_from_state(sub)
    lwc = lw_columns_from_state(sub) if mode != "sw" else None

    def rate(n, nproc):
        n = min(n, ncol)
        s_sw, _ = _ref_seconds(run_ref_sw, tab_sw, {k: v[:n] for k, v in swc.items()}, nproc)
        s_lw = 0.0
        if lwc is not None:
            s_lw, _ = _ref_seconds(run_ref_lw, tab_lw, {k: v[:n] for k, v in lwc.items()}, nproc)
        return n, s_sw, s_lw

    n1, sw1, lw1 = rate(ncol_1t, 1)
    nall, swa, lwa = rate(ncol_per_core * cores, cores)
    spc1, spca = (sw1 + lw1) / n1, (swa + lwa) / nall
    # `value`: the reference Fortran alone (the radiation double call, the part the reference defines), all used cores
    res = {"value": 1.0 / spca, "value_1_core": 1.0 / spc1,
           "unit": "columns/s", "cores": cores, "host_cpus": host_cpus, "kind": "reference",
           "what": "reference rrtmg_sw + rrtmg_lw double call only; the optical averaging (no reference arithmetic) is "
                   "under `optics_port`, the two together under `whole_step_with_optics_port`",
           "threads_1": {"columns": n1, "sw_s": sw1, "lw_s": lw1},
           "threads_all": {"columns": nall, "processes": cores, "sw_s": swa, "lw_s": lwa}}
    sample = (f"reference rrtmg_sw{' + rrtmg_lw' if lwc is not None else ''} (oracle/_ref, flang -O2, fp64), solver time: "
              f"{n1} columns on 1 core (SW {sw1:.2f} s, LW {lw1:.2f} s); {nall} columns over {cores} processes, one "
              f"per host core (SW {swa:.2f} s, LW {lwa:.2f} s)")
    if ms is not None:
        from oracle import oracle_optics as oo
        lib = oo._lib()
        wl, ref, den = wa.engine_optics.table_info()
        tabs = oo.make_tables(ms.nbin, wa.engine_optics.MOSAIC_DLO_UM, wa.engine_optics.MOSAIC_DHI_UM, wl, ref, den)

        def opt_rate(nthreads, ncols):
            lib.omp_set_num_threads(nthreads)
            nj = max(1, min(st.nj, ncols // st.ni))
            t0 = time.perf_counter()
            oo.tile(tabs, ms.chem, ms.alt, ms.dz8w, ms.idx, dims_of(st.ni, st.nj, st.nk, dict(jte=nj)))
            return (time.perf_counter() - t0) / (nj * st.ni), nj * st.ni
        o1, no1 = opt_rate(1, ncol_optics)
        oa, noa = opt_rate(cores, ncol_optics * min(cores, 8))
        lib.omp_set_num_threads(1)
        res["optics_port"] = {"kind": "port", "unit": "columns/s", "value": 1.0 / oa, "value_1_thread": 1.0 / o1,
                              "threads": cores, "columns_1_thread": no1, "columns_all_threads": noa,
                              "what": "oracle/optics_oracle.c, the C port of the engine's own specification (OpenMP over "
                                      "columns): own-spec stage, never part of `value`"}
        res["whole_step_with_optics_port"] = {"unit": "columns/s", "value": 1.0 / (spca + oa),
                                              "value_1_thread": 1.0 / (spc1 + o1)}
        sample += (f"; separately, optical averaging through the C port of the engine's own specification: {no1} columns on "
                   f"1 thread, {noa} columns on {cores} OpenMP threads")
    res["sample"] = sample
    return res


def flux_rmse(wa, tab_sw, tab_lw, st, ms, sw_out, lw_out, nrows, mode):
    """The second half of the metric: root-mean-square difference between the device's fluxes / heating rates and the
    CPU restatement of the reference (pinned bit for bit on the reference solvers) over the first `nrows` rows of the
    rank-0 slab.  The CPU leg is independent of the device from the tracers on: its optical averaging is the oracle's
    own, so an error of the device's optics (either precision) shows up in the flux error."""
    from oracle import oracle as orc, oracle_lw, oracle_optics as oo
    tile = dict(jts=1, jte=nrows)
    a3 = dict(st.a3)
    against = "oracle/ (C restatement, pinned bit for bit on the reference solvers)"
    if ms is not None:
        wl, ref, den = wa.engine_optics.table_info()
        tabs = oo.make_tables(ms.nbin, wa.engine_optics.MOSAIC_DLO_UM, wa.engine_optics.MOSAIC_DHI_UM, wl, ref, den)
        oo._lib().omp_set_num_threads(len(os.sched_getaffinity(0)))
        opt = oo.tile(tabs, ms.chem, ms.alt, ms.dz8w, ms.idx, dims_of(st.ni, st.nj, st.nk, tile))
        oo._lib().omp_set_num_threads(1)
        for w, nm in enumerate(("300", "400", "600", "999"), start=1):
            for stem in ("tauaer", "waer", "gaer"):
                a = opt[f"{stem}{w}"].copy()
                a[a == -7777.0] = 0.0
                a3[stem + nm] = a
        for b in range(1, 17):
            a = opt[f"tauaerlw{b}"].copy()
            a[a == -7777.0] = 0.0
            a3[f"tauaerlw{b}"] = a
        against += "; optical averaging by the oracle's own Mie code from the same tracers (fp64)"
    res = {"columns": nrows * st.ni, "against": against}
    d = dims_of(st.ni, st.nj, st.nk, tile)

    def alloc(names2, name3):
        out = {name3: np.full((st.nj, st.nk + 1, st.ni), -7777.0)}
        out.update({n: np.full((st.nj, st.ni), -7777.0) for n in names2})
        return out

    def rms(got, ref, names, rows3=False):
        dd = []
        for n in names:
            g = got[n].cpu().numpy()
            dd.append((g[:nrows] - ref[n][:nrows]).ravel() if not rows3 else (g[:nrows, :st.nk] - ref[n][:nrows, :st.nk]).ravel())
        dd = np.concatenate(dd)
        return float(np.sqrt(np.mean(dd * dd))), float(np.abs(dd).max())

    ref = alloc(SW_OUT2, "rthratensw")
    arrays = {n: st.a2[n] for n in SW_IN2}
    arrays.update({n: a3[n] for n in ST_IN3 + SW_AER})
    orc.swrad_tile(orc.SwOracle(tab_sw), d, arrays, 1370.0, 1, 1, ref)
    day = st.a2["xcoszen"][:nrows] > 0
    for n in SW12:   # night columns: both sides write zeros
        ref[n][:nrows][~day] = 0.0
    res["sw_flux_rmse_Wm2"], res["sw_flux_maxabs_Wm2"] = rms(sw_out, ref, SW12)
    hr_g = sw_out["rthratensw"].cpu().numpy()[:nrows, :st.nk][np.broadcast_to(day[:, None, :], (nrows, st.nk, st.ni))]
    hr_r = ref["rthratensw"][:nrows, :st.nk][np.broadcast_to(day[:, None, :], (nrows, st.nk, st.ni))]
    res["sw_heating_rmse_K_per_day"] = float(np.sqrt(np.mean((hr_g - hr_r) ** 2))) * 86400.0
    res["sw_forcing_rmse_Wm2"] = float(np.sqrt(np.mean(
        ((sw_out["swuptcln"] - sw_out["swupt"]).cpu().numpy()[:nrows] - (ref["swuptcln"] - ref["swupt"])[:nrows]) ** 2)))
    if mode != "sw":
        nlay = oracle_lw.nlayers(st.nk + 1, P_TOP)
        refl = alloc(LW_OUT2, "rthratenlw")
        arrays = {n: st.a2[n] for n in LW_IN2}
        arrays.update({n: a3[n] for n in ST_IN3 + LW_AER})
        oracle_lw.lwrad_tile(oracle_lw.LwOracle(tab_lw), d, nlay, arrays, 1, 1, refl)
        res["lw_flux_rmse_Wm2"], res["lw_flux_maxabs_Wm2"] = rms(lw_out, refl, LW12)
        res["lw_heating_rmse_K_per_day"] = rms(lw_out, refl, ["rthratenlw"], rows3=True)[0] * 86400.0
        res["lw_forcing_rmse_Wm2"] = float(np.sqrt(np.mean(
            ((lw_out["lwuptcln"] - lw_out["lwupt"]).cpu().numpy()[:nrows] - (refl["lwuptcln"] - refl["lwupt"])[:nrows]) ** 2)))
    return res


def algorithmic_flops(wa, tab_sw, tab_lw, st, ms, mode, ncols=16):
    """Exact fp64 operation count of the CPU restatement per column of THIS workload (oracle built with
    -DORACLE_COUNT_FLOPS: every arithmetic operation of the column solvers goes through a counting type).
    Returns None when the counting build is not present."""
    try:
        from oracle import flopcount
    except Exception:
        return None
    return flopcount.per_column(wa, tab_sw, tab_lw, st, ms, mode, ncols)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv):
    """Parent of `python3 bench.py --gpus N`: starts the N ranks as a child process group through
    torch.distributed.run (one rank per GPU of this node, rendezvous on 127.0.0.1) and returns its exit code.  The
    parent has not imported torch or touched the GPU, and it does not exec: the ranks' stdout / stderr are inherited,
    so rank 0's JSON line is this command's output."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL (see README)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args, cfg, world, rank):
    """The launch path without a GPU: rendezvous over gloo, the slab of every rank (same cut rule as the timed path),
    one all-reduce of the 32-vector in which every rank has put its column count, one JSON line on rank 0."""
    import torch
    import torch.distributed as dist
    from importlib import import_module
    tiles = import_module("wrfchem-arc-interactions_amd.tiles")
    ni, nj = cfg["ni"], cfg["nj"]
    if world > 1:
        dist.init_process_group("gloo")
    night = night_rows(ni, nj, 20260009, args.night)
    weights = None
    if args.night > 0 and args.scaling == "strong":
        weights = slab_row_cost(cfg["mode"]) * ni + SW_MS * (~night).sum(axis=1)
    if args.scaling == "strong" or world == 1:
        js, je = tiles.slab_bounds(1, nj, world, rank, weights=weights)
    else:
        js, je = 1, nj
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "js": js, "je": je,
            "columns": max(0, je - js + 1) * ni}
    sums = torch.zeros(32, dtype=torch.float64)
    sums[13] = mine["columns"]
    tiles.allreduce_sums(sums)
    slabs = [mine]
    if world > 1:
        slabs = [None] * world
        dist.all_gather_object(slabs, mine)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "scaling": args.scaling if world > 1 else "strong",
                          "domain": [ni, nj, cfg["nk"]], "slabs": slabs, "columns_all_ranks": float(sums[13])}))


# shares of the step measured at the north-star workload (ms per 250 000 columns): they weight the rows of a slab
OPT_MS, LW_MS, SW_MS = 94.0, 59.0, 47.0


def slab_row_cost(mode):
    """Cost of a column that every column pays (longwave call, optical averaging); the shortwave call is paid by
    daylit columns only."""
    return (LW_MS if mode != "sw" else 0.0) + (OPT_MS if mode == "full" else 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=3,
                    help="BASELINE.json configs[n]; 3 = the north-star domain (500x500x50, 8-bin, SW+LW, fp64)")
    ap.add_argument("--ni", type=int)
    ap.add_argument("--nj", type=int)
    ap.add_argument("--nk", type=int)
    ap.add_argument("--mode", choices=["full", "swlw", "sw"])
    ap.add_argument("--nbin", type=int, help="MOSAIC sections of the optics stage (4 or 8)")
    ap.add_argument("--optics-precision", choices=["f64", "mixed"],
                    help="mixed: the Mie series of the optical averaging in packed fp32 (BASELINE configs[4])")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = one domain cut into j-slabs (default), weak = one domain per rank")
    ap.add_argument("--night", type=float, default=0.0, help="fraction of night columns (0 for kernel timing)")
    ap.add_argument("--clouds", type=float, default=0.0,
                    help="fraction of columns with 1-3 cloud decks (McICA path; 0 = cloud-free headline workload)")
    ap.add_argument("--cpu-sample", type=int, default=4000, help="columns of the 1-core reference run")
    ap.add_argument("--cpu-sample-per-core", type=int, default=2000, help="columns per process of the all-core run")
    ap.add_argument("--cpu-optics-sample", type=int, default=500)
    ap.add_argument("--host-path-rows", type=int, default=40,
                    help="rows of the domain pushed through the mem = 0 (host pointer) entry points; 0 = skip")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cloudy-pass", type=int, default=1,
                    help="N = 1: after the timed region, one short pass of the same domain with the SURVEY 8d full-domain "
                         "atmosphere (clouds in 20 %% of the columns, 10 %% night columns) -> key `cloudy`; 0 = skip")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 on a box with fewer GPUs than ranks (development): ranks share the devices (local_rank modulo "
                         "the device count) and the one all-reduce goes over gloo, since RCCL refuses two ranks on a device; "
                         "exercises launch, slabs, kernels on slabs and the reduction -- the number it prints is not a "
                         "scaling measurement")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch path only (no GPU, gloo): every rank computes its slab, rank 0 prints them as JSON")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` without a launcher: this process only starts the ranks
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    cfg = dict(CONFIGS[args.config])
    for k, a in (("ni", args.ni), ("nj", args.nj), ("nk", args.nk), ("mode", args.mode), ("nbin", args.nbin),
                 ("prec", args.optics_precision)):
        if a is not None:
            cfg[k] = a
    ni, nj, nk, mode, nbin, prec = cfg["ni"], cfg["nj"], cfg["nk"], cfg["mode"], cfg["nbin"], cfg["prec"]
    is_config = all(cfg[k] == v for k, v in CONFIGS[args.config].items())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.dry_launch:
        return dry_launch(args, cfg, world, rank)

    import torch
    import torch.distributed as dist
    import wrfchem_arc_amd as wa

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    rehearse = args.rehearse_on_one_gpu and world > 1
    if rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # one launch per kernel for the whole domain: an explicit opt-in (the engine's default scratch caps are modest,
    # 32 / 16 GiB and at most a quarter of the free memory; larger tiles then run in column batches)
    os.environ.setdefault("RRTMG_HIP_SW_SCRATCH_MB", "98304")
    os.environ.setdefault("RRTMG_HIP_LW_SCRATCH_MB", "49152")
    import __graft_entry__ as ge
    if not os.path.exists(wa.engine.LIB_PATH):
        ge.build_product()
    eng, engl = wa.engine, wa.engine_lw
    tmp = tempfile.mkdtemp(prefix="rrtmg_bench_")
    tab = os.path.join(tmp, "RRTMG_SW_DATA")
    tabl = os.path.join(tmp, "RRTMG_LW_DATA")
    wa.synth.write_sw_data(tab, seed=20260001, real_bytes=8)
    wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
    eng.rrtmg_swinit(tab, 8, device=local_rank)
    do_lw = mode != "sw"
    do_opt = mode == "full"
    if do_opt:
        wa.engine_optics.optics_init(nbin, device=local_rank)
        wa.engine_optics.optics_precision(prec)
    if do_lw:
        engl.rrtmg_lwinit(P_TOP, tabl, kme=nk + 1, device=local_rank)

    # ---- this rank's slab of the domain ----
    strong = args.scaling == "strong" or world == 1
    wseed, mseed, nseed, cseed = 20260002, 20260005, 20260009, 20260007
    if strong:
        night = night_rows(ni, nj, nseed, args.night)                   # cheap: any rank evaluates every row
        # cost of a row: every column pays the longwave call and the optics, only daylit columns the shortwave call
        # (OPT_MS / LW_MS / SW_MS above)
        weights = None
        if args.night > 0:
            weights = slab_row_cost(mode) * ni + SW_MS * (~night).sum(axis=1)
        js, je = wa.tiles.slab_bounds(1, nj, world, rank, weights=weights)   # WRF indices, inclusive
        j0, j1 = js - 1, je
    else:
        wseed += 1000003 * rank
        mseed += 1000003 * rank
        night = night_rows(ni, nj, nseed + 1000003 * rank, args.night)
        j0, j1 = 0, nj
    if do_opt:
        st, ms = wa.synth_mosaic.make_mosaic_rows(ni, nj, nk, j0, j1, nbin=nbin, seed=mseed, wrf_seed=wseed)
    else:
        st, ms = wa.synth.make_wrf_rows(ni, nj, nk, j0, j1, seed=wseed), None
    if args.night > 0:
        st.a2["xcoszen"] = np.where(night[j0:j1], -0.3, st.a2["xcoszen"])
    ncol = st.ncol
    ncol_job = ni * nj if strong else world * ni * nj
    d = dims_of(ni, st.nj, nk)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    sw_in = {n: dev(st.a2[n]) for n in SW_IN2}
    sw_in.update({n: dev(st.a3[n]) for n in ST_IN3 + SW_AER})
    sw_out = {"rthratensw": torch.zeros((st.nj, nk + 1, ni), dtype=torch.float64, device="cuda")}
    sw_out.update({n: torch.zeros((st.nj, ni), dtype=torch.float64, device="cuda") for n in SW_OUT2})
    lw_in = {n: dev(st.a2[n]) for n in LW_IN2}
    lw_in.update({n: sw_in[n] for n in ST_IN3})              # the same state arrays, not copies
    lw_in.update({n: dev(st.a3[n]) for n in LW_AER})
    lw_out = {"rthratenlw": torch.zeros((st.nj, nk + 1, ni), dtype=torch.float64, device="cuda")}
    lw_out.update({n: torch.zeros((st.nj, ni), dtype=torch.float64, device="cuda") for n in LW_OUT2})
    if do_opt:
        # MOSAIC tracers -> the aerosol inputs of the two radiation calls (written in place every step)
        opt_in = dict(chem=dev(ms.chem), alt=dev(ms.alt), dz8w=dev(ms.dz8w))
        opt_out = {}
        for w, nm in enumerate(("300", "400", "600", "999"), start=1):
            for stem in ("tauaer", "waer", "gaer"):
                opt_out[f"{stem}{w}"] = sw_in[stem + nm]
        for b in range(1, 17):
            opt_out[f"tauaerlw{b}"] = lw_in[f"tauaerlw{b}"]
        for t in opt_out.values():
            t.zero_()
    sums = torch.zeros(32, dtype=torch.float64, device="cuda")

    cld_flags, cld_in = dict(icloud=0), {}
    if args.clouds > 0.0:
        cld_flags = wa.synth_clouds.add_clouds(st, seed=cseed + 1000003 * rank, cloudy_fraction=args.clouds)
        cld_in = {n: dev(st.a3[n]) for n in CLD3}
        cld_in.update({n: dev(st.a2[n]) for n in CLD2})

    empty = ncol == 0

    def sw_call():
        if not empty:
            eng.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, **cld_flags, **d, **sw_in, **cld_in, **sw_out)

    def lw_call():
        if not empty:
            engl.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, **cld_flags, **d, **lw_in, **cld_in, **lw_out)

    def gather_sums():
        # the only exchange on the path: per-tile flux / forcing sums for the domain means
        wa.tiles.tile_sums_device(sw_out, SW12, d, sums[:16])          # native deterministic reduction
        if do_lw:
            wa.tiles.tile_sums_device(lw_out, LW12, d, sums[16:])
        if rehearse:                       # gloo reduces host tensors
            h = sums.cpu()
            wa.tiles.allreduce_sums(h)
            sums.copy_(h)
        else:
            wa.tiles.allreduce_sums(sums)

    def opt_call(outs=None):
        if not empty:
            wa.engine_optics.optical_averaging(idx=ms.idx, aer_op_opt=1, **d, **opt_in, **(outs or opt_out))

    def step(ev=None):
        if ev:
            ev[3].record()
        if do_opt:
            opt_call()
        if ev:
            ev[0].record()
        sw_call()
        if ev:
            ev[1].record()
        if do_lw:
            lw_call()
        if ev:
            ev[2].record()
        if world > 1:
            gather_sums()

    for _ in range(args.warmup):
        step()
    eng.sw_status()
    if do_lw:
        engl.lw_status()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        step(ev)
        events.append(ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.sw_status()
    if do_lw:
        engl.lw_status()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    sw_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))   # HIP events on the launch stream
    lw_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events])) if do_lw else 0.0
    opt_ms = float(np.mean([e[3].elapsed_time(e[0]) for e in events])) if do_opt else 0.0
    opt_part_ms = {}
    if do_opt and not empty:
        # the stage is two kernels (shortwave wavelengths, longwave band pairs): time each one on its own, outside
        # the timed region, by asking for one output group only
        for part, outs in (("sw", {n: t for n, t in opt_out.items() if not n.startswith("tauaerlw")}),
                           ("lw", {n: t for n, t in opt_out.items() if n.startswith("tauaerlw")})):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            opt_call(outs)
            e0.record()
            for _ in range(2):
                opt_call(outs)
            e1.record()
            torch.cuda.synchronize()
            opt_part_ms[part] = e0.elapsed_time(e1) / 2

    if rank == 0:
        value = ncol_job * args.steps / elapsed
        bpc, bpc_sw, bpc_lw = algorithmic_bytes_per_column(nk, mode)
        obpc = optics_bytes_per_column(nk, nbin) if do_opt else 0
        # dominant kernel by measured duration: the SW column kernel or one of the two optics kernels (the two LW
        # kernels are each shorter than the SW kernel; the events bracket them together)
        cand = [("sw_column_kernel", sw_ms, bpc_sw)]
        if do_opt:
            cand.append(("optics_kernel", opt_part_ms["sw"], optics_bytes_per_column(nk, nbin, part="sw")))
            cand.append(("optics_lw_kernel", opt_part_ms["lw"], optics_bytes_per_column(nk, nbin, part="lw")))
        dom_name, dom_ms, dom_bpc = max(cand, key=lambda c: c[1])
        achieved = ncol * dom_bpc / (dom_ms * 1e-3) / 1e9
        traffic, valu, traffic_source = None, {}, None
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        imported = {}   # kernel -> HBM bytes per step of the builder-run profile, scaled to this run's columns
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                for kn, kv in tj.get("kernels", {}).items():
                    if kv.get("hbm_bytes_per_launch") is not None and kv.get("columns_per_dispatch"):
                        imported[kn] = kv["hbm_bytes_per_launch"] * ncol / kv["columns_per_dispatch"]
                kd = tj.get("kernels", {}).get(dom_name, {})
                if kd.get("hbm_bytes_per_launch") is not None and kd.get("columns_per_dispatch"):
                    # PMC bytes per dispatch of the profiled run scaled to the columns of one step here (same kernel)
                    traffic = kd["hbm_bytes_per_launch"] * ncol / kd["columns_per_dispatch"]
                    traffic_source = {"file": tj.get("source"), "commit": tj.get("source_commit"),
                                      "how": "imported: PMC bytes per dispatch of a builder-run profile of this command "
                                             "(FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes), scaled by the column "
                                             "count; not measured by this run"}
                if kd.get("valu_issue_frac") is not None and kd.get("columns_per_dispatch"):   # what binds this kernel
                    valu = {"fp64_valu_issue_frac": kd["valu_issue_frac"],
                            "valu_wave_instructions_per_column": kd["valu_wave_instructions"] / kd["columns_per_dispatch"],
                            "source": tj.get("source")}
            except Exception:
                traffic = None
        tag = f" (BASELINE configs[{args.config}])" if is_config else ""
        cloudtxt = f"McICA clouds in {100 * args.clouds:.0f} % of the columns" if args.clouds > 0 else "cloud-free"
        if do_opt:
            workload = (f"{ni}x{nj}x{nk} synthetic domain, MOSAIC {nbin}-bin optical averaging + SW+LW aerosol+clean "
                        f"double-call, {'fp64' if prec == 'f64' else 'fp64 radiation, packed-fp32 Mie'}, {cloudtxt}{tag}")
        elif do_lw:
            workload = (f"{ni}x{nj}x{nk} synthetic domain, SW+LW aerosol+clean double-call, fp64, {cloudtxt}; aerosol "
                        f"optics supplied in the WRF-Chem compact form (no optical-averaging stage){tag}")
        else:
            workload = f"{ni}x{nj}x{nk} synthetic domain, RRTMG-SW aerosol+clean double-call only, fp64, {cloudtxt}{tag}"
        line = {
            "metric": "columns/sec, aerosol+clean RRTMG SW+LW double-call; flux RMSE vs reference",
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            **({"rehearsal": "ranks share one GPU, all-reduce over gloo: plumbing check, not a scaling measurement"}
               if rehearse else {}),
            "vs_baseline": None, "dtype": "f64" if prec == "f64" or not do_opt else "f64 (optical averaging f32)",
            "data": "synthetic",
            "config": {"workload": workload, "columns": ncol_job, "columns_rank0": ncol, "layers": nk,
                       "night_fraction": args.night,
                       "parallelism": (f"one domain in {world} j-slab(s), one per GPU" if strong else
                                       f"{world} domains, one per GPU") + "; RCCL all-reduce of 32 flux sums"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": dom_name, "kernel_ms": dom_ms,
                         "algorithmic_bytes_per_column": dom_bpc,
                         "sw_kernel_ms": sw_ms, "sw_algorithmic_bytes_per_column": bpc_sw,
                         "lw_kernel_ms": lw_ms, "lw_algorithmic_bytes_per_column": bpc_lw if do_lw else 0,
                         "optics_stage_ms": opt_ms, "optics_kernel_ms": opt_part_ms.get("sw", 0.0),
                         "optics_lw_kernel_ms": opt_part_ms.get("lw", 0.0), "optics_algorithmic_bytes_per_column": obpc,
                         "step_algorithmic_GBs": ncol * (bpc + obpc) / ((sw_ms + lw_ms + opt_ms) * 1e-3) / 1e9,
                         "valu": valu,
                         # every kernel of the step (the three largest are within a few per cent of each other, so which
                         # one is "dominant" changes from run to run): duration, algorithmic GB/s, fraction of the HBM peak
                         "kernels": [
                             {"kernel": nm, "ms": ms_, "algorithmic_bytes_per_column": b_,
                              "achieved": ncol * b_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else None,
                              "frac": ncol * b_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_ > 0 else None,
                              # HBM bytes the kernel(s) actually moved per step (imported like `traffic`) and the rate
                              # that is over this run's duration
                              "traffic": (sum(imported[q] for q in nm.split(" + ")) if all(q in imported for q in nm.split(" + "))
                                          else None),
                              "traffic_GBs": (sum(imported[q] for q in nm.split(" + ")) / (ms_ * 1e-3) / 1e9
                                              if ms_ > 0 and all(q in imported for q in nm.split(" + ")) else None)}
                             for nm, ms_, b_ in (cand + ([("lw_gas_kernel + lw_rt_fast_tbl_kernel", lw_ms, bpc_lw)] if do_lw else []))],
                         "note": "not HBM-bound at the algorithmic bytes (SURVEY 8d): the optical averaging is bound by fp64 "
                                 "issue, the radiation kernels by the memory path of their own scratch streams and table "
                                 "gathers (kernels[*].traffic); see roofline.fp64 and DESIGN.md 4.4"},
        }
        if world > 1:
            # what the one collective delivered: the column count and the domain-mean TOA upward fluxes (tiles.domain_means)
            hs_ = sums.cpu().numpy()
            line["allreduce"] = {"columns_reduced": float(hs_[13]), "columns_expected": float(ncol_job),
                                 "mean_swupt_Wm2": float(hs_[0] / hs_[13]) if hs_[13] > 0 else None,
                                 "mean_lwupt_Wm2": float(hs_[16] / hs_[29]) if do_lw and hs_[29] > 0 else None}
        if world == 1 and not args.no_cpu:
            try:
                fl = algorithmic_flops(wa, tab, tabl, st, ms, mode)
                if fl:
                    # the whole timed step: radiation + optical averaging, every operation of the CPU restatement
                    tf64 = value * fl["flops_per_column"] / 1e12
                    rad_s = (sw_ms + lw_ms) * 1e-3
                    tf64_rad = ncol * fl["flops_per_column_radiation"] / rad_s / 1e12 if rad_s > 0 else None
                    line["roofline"]["fp64"] = dict(
                        fl, achieved_TFLOPs=tf64, peak_TFLOPs=FP64_PEAK_TFLOPS, frac=tf64 / FP64_PEAK_TFLOPS,
                        radiation_only={"achieved_TFLOPs": tf64_rad, "frac": tf64_rad / FP64_PEAK_TFLOPS if tf64_rad else None,
                                        "over": "the two radiation calls' own kernel time (rank 0)"})
            except Exception as e:
                line["roofline"]["fp64"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu and args.clouds <= 0.0:
            try:
                line["cpu_baseline"] = cpu_baseline(wa, tab, tabl, st, ms, mode, args.cpu_sample,
                                                    args.cpu_sample_per_core, args.cpu_optics_sample)
            except Exception as e:   # never let the baseline leg kill the bench line
                line["cpu_baseline"] = {"value": None, "unit": "columns/s", "cores": 0, "kind": "reference",
                                        "sample": f"failed: {e!r}"}
            try:
                line["flux_rmse"] = flux_rmse(wa, tab, tabl, st, ms, sw_out, lw_out, max(1, min(st.nj, 500 // ni)), mode)
            except Exception as e:
                line["flux_rmse"] = {"error": repr(e)}
        if world == 1 and args.cloudy_pass and args.clouds <= 0.0 and not empty:
            # SURVEY 8d's full-domain atmosphere (McICA clouds in 20 % of the columns, 10 % night columns): the same
            # domain and kernels' cloudy instantiations, timed per stage; outside `value`
            try:
                import copy
                stc = copy.copy(st)
                stc.a2, stc.a3 = dict(st.a2), dict(st.a3)
                nightc = night_rows(ni, nj, nseed, 0.1)[j0:j1]
                stc.a2["xcoszen"] = np.where(nightc, -0.3, st.a2["xcoszen"])
                cflags = wa.synth_clouds.add_clouds(stc, seed=cseed, cloudy_fraction=0.2)
                cin = {n: dev(stc.a3[n]) for n in CLD3}
                cin.update({n: dev(stc.a2[n]) for n in CLD2})
                sw_in_c = dict(sw_in, xcoszen=dev(stc.a2["xcoszen"]))

                def cstep(ev=None):
                    if ev:
                        ev[0].record()
                    if do_opt:
                        opt_call()
                    if ev:
                        ev[1].record()
                    eng.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, **cflags, **d, **sw_in_c, **cin, **sw_out)
                    if ev:
                        ev[2].record()
                    if do_lw:
                        engl.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, **cflags, **d, **lw_in, **cin, **lw_out)
                    if ev:
                        ev[3].record()
                cstep()
                torch.cuda.synchronize()
                evs = []
                tc0 = time.perf_counter()
                for _ in range(2):
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    cstep(ev)
                    evs.append(ev)
                torch.cuda.synchronize()
                tc = (time.perf_counter() - tc0) / 2
                eng.sw_status()
                if do_lw:
                    engl.lw_status()
                line["cloudy"] = {
                    "what": "same domain, SURVEY 8d full-domain atmosphere: McICA clouds (1-3 decks) in 20 % of the columns, "
                            "10 % night columns; 2 steps after 1 warm-up, outside `value`",
                    "ms_per_step": 1e3 * tc, "columns_per_s": ncol / tc,
                    "optics_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in evs])),
                    "sw_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in evs])),
                    "lw_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in evs])),
                    "cloudy_columns": int((stc.a3["cldfra3d"].max(axis=1) > 0).sum()),
                    "night_columns": int(nightc.sum())}
                del cin, sw_in_c
            except Exception as e:
                line["cloudy"] = {"error": repr(e)}
        if world == 1 and args.host_path_rows > 0 and args.clouds <= 0.0:
            # the call a WRF build makes (mem = 0: host arrays through the Fortran shim's entry points), PCIe included
            try:
                rows = min(args.host_path_rows, st.nj)
                dh = dims_of(ni, st.nj, nk, dict(jte=rows))
                hin = {n: np.ascontiguousarray(st.a2[n]) for n in SW_IN2}
                hin.update({n: np.ascontiguousarray(st.a3[n]) for n in ST_IN3 + SW_AER})
                hout = {"rthratensw": np.zeros((st.nj, nk + 1, ni))}
                hout.update({n: np.zeros((st.nj, ni)) for n in SW_OUT2})
                lin = {n: np.ascontiguousarray(st.a2[n]) for n in LW_IN2}
                lin.update({n: hin[n] for n in ST_IN3})
                lin.update({n: np.ascontiguousarray(st.a3[n]) for n in LW_AER})
                lout = {"rthratenlw": np.zeros((st.nj, nk + 1, ni))}
                lout.update({n: np.zeros((st.nj, ni)) for n in LW_OUT2})

                def host_step():
                    eng.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **dh, **hin, **hout)
                    if do_lw:
                        engl.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **dh, **lin, **lout)
                host_step()
                reps = []
                for _ in range(5):   # the host side of this path shares the box's cores with other tenants: median of five
                    t1 = time.perf_counter()
                    host_step()
                    reps.append(time.perf_counter() - t1)
                hs = float(np.median(reps))
                # the same with the arrays pinned in place (RRTMG_HIP_REGISTER_HOST=1, csrc/tile_staging.h: no packing pass on
                # the host; an opt-in for hosts whose arrays live as long as the run, as WRF's do -- these do)
                os.environ["RRTMG_HIP_REGISTER_HOST"] = "1"
                try:
                    host_step()
                    reps_reg = []
                    for _ in range(5):
                        t1 = time.perf_counter()
                        host_step()
                        reps_reg.append(time.perf_counter() - t1)
                finally:
                    os.environ["RRTMG_HIP_REGISTER_HOST"] = "0"
                    # the registry remembers arrays by address: tell it before these are freed
                    wa.engine.unregister_host(*[v for d in (hin, hout, lin, lout) for v in d.values()])
                line["host_path"] = {"columns_per_s": rows * ni / hs, "columns": rows * ni, "seconds_per_call_pair": hs,
                                     "best_of_5_columns_per_s": rows * ni / min(reps),
                                     "registered_columns_per_s": rows * ni / float(np.median(reps_reg)),
                                     "what": "rrtmg_hip_swrad" + (" + rrtmg_hip_lwrad" if do_lw else "") +
                                             " with mem = 0 (host arrays: packed in j-chunks, staged over PCIe and "
                                             "run on the calling thread's stream, chunks overlapping -- the call the "
                                             "Fortran shim makes); radiation only, never `value`"}
                if do_opt:
                    # the optical-averaging entry module_optical_averaging_hip.F90 calls, host arrays as well
                    oh_in = dict(chem=np.ascontiguousarray(ms.chem), alt=np.ascontiguousarray(ms.alt),
                                 dz8w=np.ascontiguousarray(ms.dz8w))
                    oh_out = {n: np.zeros((st.nj, nk + 1, ni)) for n in opt_out}

                    def host_opt():
                        wa.engine_optics.optical_averaging(idx=ms.idx, aer_op_opt=1, **dh, **oh_in, **oh_out)
                    host_opt()
                    reps = []
                    for _ in range(3):
                        t1 = time.perf_counter()
                        host_opt()
                        reps.append(time.perf_counter() - t1)
                    ho = float(np.median(reps))
                    os.environ["RRTMG_HIP_REGISTER_HOST"] = "1"
                    try:
                        host_opt()
                        reps_reg = []
                        for _ in range(3):
                            t1 = time.perf_counter()
                            host_opt()
                            reps_reg.append(time.perf_counter() - t1)
                    finally:
                        os.environ["RRTMG_HIP_REGISTER_HOST"] = "0"
                        wa.engine.unregister_host(*oh_in.values(), *oh_out.values())
                    line["host_path"]["optics"] = {
                        "columns_per_s": rows * ni / ho, "seconds_per_call": ho,
                        "registered_columns_per_s": rows * ni / float(np.median(reps_reg)),
                        "what": "rrtmg_hip_optical_averaging with mem = 0 on the same rows: the section tracers of every "
                                "cell over PCIe in, 28 fields per cell out"}
                    line["host_path"]["whole_step_columns_per_s"] = rows * ni / (hs + ho)
            except Exception as e:
                line["host_path"] = {"error": repr(e)}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
