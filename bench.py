#!/usr/bin/env python3
"""Benchmark of the hot path: aerosol + clean-sky RRTMG SW+LW double call on MI355X.

A "step" is one pass of the HIP path -- MOSAIC 4-bin optical averaging, then
RRTMG_SWRAD + RRTMG_LWRAD, each with its aerosol and clean-sky call -- over one
batch of synthetic WRF columns resident in HBM.  Workload at N=1: 200x200x40
synthetic columns, fp64 = BASELINE.json configs[2] (`--mode full`, default).
`--mode swlw` skips the optics stage (aerosol optics supplied in the WRF-Chem
compact form, 12 SW + 16 LW arrays); `--mode sw` runs configs[1] (SW double
call only).  With --gpus N each rank owns its own 200x200 slab of
(i,j) columns (weak scaling, no data-path collective); the only exchange is an
all-reduce of the per-tile flux sums (32 doubles) over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
P_TOP = 5000.0                 # Pa, model top of the synthetic atmosphere


def algorithmic_bytes_per_column(nk, mode, real_bytes=8):
    """Compulsory boundary traffic with the WRF-Chem compact aerosol form (DESIGN.md 'Measurement').
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
    return (rd + 12 + 16) * nk * real_bytes


def cpu_optics_seconds_per_column(wa, st, ms, ncol_sample):
    """The optical-averaging stage has no reference arithmetic: timed from the C port, one thread."""
    from oracle import oracle_optics as oo
    from tests.helpers_optics import oracle_tables
    lib = oo._lib()
    lib.omp_set_num_threads(1)
    nj = max(1, min(st.nj, ncol_sample // st.ni))
    d = dict(ims=1, ime=st.ni, jms=1, jme=st.nj, kms=1, kme=st.nk + 1, its=1, ite=st.ni, jts=1, jte=nj, kts=1, kte=st.nk)
    t0 = time.perf_counter()
    oo.tile(oracle_tables(wa, ms.nbin), ms.chem, ms.alt, ms.dz8w, ms.idx, d)
    secs = time.perf_counter() - t0
    return secs / (nj * st.ni), nj * st.ni


def cpu_baseline(tab_sw, tab_lw, st, ncol_sample, mode):
    """Reference Fortran (oracle/_ref, built from /root/reference by oracle/build_ref.sh) timed on one host
    core over a bounded sample of the same workload; falls back to the C port if the binaries are absent."""
    from tools.make_golden import sw_columns_from_state, run_ref_sw
    from tools.make_golden_lw import lw_columns_from_state, run_ref_lw
    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_sw")) and \
        os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_lw"))
    cols = {k: v[:ncol_sample] for k, v in sw_columns_from_state(st).items()}
    if have_ref:
        _, so = run_ref_sw(tab_sw, cols, clean=1)
        secs_sw = float(so.split("seconds=")[1].split()[0])
        secs_lw = 0.0
        if mode != "sw":
            lcols = {k: v[:ncol_sample] for k, v in lw_columns_from_state(st).items()}
            _, lo = run_ref_lw(tab_lw, lcols, clean=1)
            secs_lw = float(lo.split("seconds=")[1].split()[0])
        secs = secs_sw + secs_lw
        return {"value": ncol_sample / secs, "unit": "columns/s", "cores": 1, "kind": "reference",
                "sample": f"{ncol_sample} columns of the bench workload through the reference rrtmg_sw"
                          f"{' + rrtmg_lw' if mode != 'sw' else ''} (flang -O2, fp64, 1 core): "
                          f"SW {secs_sw:.2f} s, LW {secs_lw:.2f} s solver time",
                "seconds_per_column": secs / ncol_sample}
    from oracle import oracle as orc
    o = orc.SwOracle(tab_sw)
    n = min(ncol_sample, 256)
    t0 = time.perf_counter()
    o.rrtmg_sw({k: v[:n] for k, v in cols.items()}, clean=1)
    secs = time.perf_counter() - t0
    return {"value": n / secs, "unit": "columns/s", "cores": 1, "kind": "port",
            "sample": f"{n} columns through the C restatement of rrtmg_sw only (gcc -O2), {secs:.2f} s"}


def flux_rmse(tab_sw, tab_lw, st, sw_out, lw_out, nrows, mode):
    """The second half of the metric: root-mean-square difference between the device's fluxes / heating rates and
    the CPU restatement of the reference (bit-exact with the reference solvers on the golden fixtures) over the
    first `nrows` rows of the tile.  Part of the CPU-baseline leg (the only place bench.py touches oracle/)."""
    from oracle import oracle as orc, oracle_lw
    from tests.helpers import run_oracle_swrad
    from tests.helpers_lw import run_oracle_lwrad
    tile = dict(jts=1, jte=nrows)
    res = {"columns": nrows * st.ni, "against": "oracle/ (C restatement, pinned bit for bit on the reference solvers)"}

    def rms(got, ref, names, rows3=False):
        d = []
        for n in names:
            g = got[n].cpu().numpy()
            d.append((g[:nrows] - ref[n][:nrows]).ravel() if not rows3 else (g[:nrows, :st.nk] - ref[n][:nrows, :st.nk]).ravel())
        d = np.concatenate(d)
        return float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())

    ref = run_oracle_swrad(orc.SwOracle(tab_sw), st, tile=tile, profiles=False)
    sw2 = [n for n in ref if n.startswith(("swup", "swdn")) and ref[n].ndim == 2]
    res["sw_flux_rmse_Wm2"], res["sw_flux_maxabs_Wm2"] = rms(sw_out, ref, sw2)
    res["sw_heating_rmse_K_per_day"] = rms(sw_out, ref, ["rthratensw"], rows3=True)[0] * 86400.0
    if mode != "sw":
        nlay = oracle_lw.nlayers(st.nk + 1, P_TOP)
        refl = run_oracle_lwrad(oracle_lw.LwOracle(tab_lw), st, nlay, tile=tile, profiles=False)
        lw2 = [n for n in refl if n.startswith(("lwup", "lwdn")) and refl[n].ndim == 2]
        res["lw_flux_rmse_Wm2"], res["lw_flux_maxabs_Wm2"] = rms(lw_out, refl, lw2)
        res["lw_heating_rmse_K_per_day"] = rms(lw_out, refl, ["rthratenlw"], rows3=True)[0] * 86400.0
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ni", type=int, default=200)
    ap.add_argument("--nj", type=int, default=200)
    ap.add_argument("--nk", type=int, default=40)
    ap.add_argument("--mode", choices=["full", "swlw", "sw"], default="full")
    ap.add_argument("--nbin", type=int, default=4, help="MOSAIC sections of the optics stage (4 or 8)")
    ap.add_argument("--optics-precision", choices=["f64", "mixed"], default="f64",
                    help="mixed: the Mie series of the optical averaging in packed fp32 (BASELINE configs[4])")
    ap.add_argument("--cpu-optics-sample", type=int, default=400)
    ap.add_argument("--night", type=float, default=0.0, help="fraction of night columns (0 for kernel timing)")
    ap.add_argument("--cpu-sample", type=int, default=6000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--clouds", type=float, default=0.0,
                    help="fraction of columns with 1-3 cloud decks (McICA path; 0 = cloud-free headline workload)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import wrfchem_arc_amd as wa
    from tests.helpers import AER, IN2, IN3, OUT2, dims_of
    from tests import helpers_lw as hl

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
    eng, engl = wa.engine, wa.engine_lw
    tmp = tempfile.mkdtemp(prefix="rrtmg_bench_")
    tab = os.path.join(tmp, "RRTMG_SW_DATA")
    tabl = os.path.join(tmp, "RRTMG_LW_DATA")
    wa.synth.write_sw_data(tab, seed=20260001, real_bytes=8)
    wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
    eng.rrtmg_swinit(tab, 8, device=local_rank)
    do_lw = args.mode != "sw"
    do_opt = args.mode == "full"
    if do_opt:
        wa.engine_optics.optics_init(args.nbin, device=local_rank)
        wa.engine_optics.optics_precision(args.optics_precision)
    if do_lw:
        engl.rrtmg_lwinit(P_TOP, tabl, kme=args.nk + 1, device=local_rank)

    # each rank: its own (i,j) slab of the domain
    st = wa.synth.make_wrf_state(args.ni, args.nj, args.nk, seed=20260002 + 1000 * rank, night_fraction=args.night)
    ncol = st.ncol
    d = dims_of(st)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    sw_in = {n: dev(st.a2[n]) for n in IN2}
    sw_in.update({n: dev(st.a3[n]) for n in IN3 + AER})
    sw_out = {"rthratensw": torch.zeros((st.nj, st.nk + 1, st.ni), dtype=torch.float64, device="cuda")}
    sw_out.update({n: torch.zeros((st.nj, st.ni), dtype=torch.float64, device="cuda") for n in OUT2})
    lw_in = {n: dev(st.a2[n]) for n in hl.IN2}
    lw_in.update({n: sw_in[n] for n in hl.IN3})            # the same state arrays, not copies
    lw_in.update({n: dev(st.a3[n]) for n in hl.AER})
    lw_out = {"rthratenlw": torch.zeros((st.nj, st.nk + 1, st.ni), dtype=torch.float64, device="cuda")}
    lw_out.update({n: torch.zeros((st.nj, st.ni), dtype=torch.float64, device="cuda") for n in hl.OUT2})
    ms = None
    if do_opt:
        # MOSAIC tracers -> the aerosol inputs of the two radiation calls (written in place every step)
        ms = wa.synth_mosaic.make_mosaic_state(st, nbin=args.nbin, seed=20260005 + 1000 * rank)
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
    sw12 = ["swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
            "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln"]
    lw12 = ["lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
            "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"]

    cld_flags, cld_in = dict(icloud=0), {}
    if args.clouds > 0.0:
        from tests.helpers import cloud_arrays
        cld_flags = wa.synth_clouds.add_clouds(st, seed=20260007 + 1000 * rank, cloudy_fraction=args.clouds)
        cld_in = {n: dev(v) for n, v in cloud_arrays(st, cld_flags).items()}

    def sw_call():
        eng.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, **cld_flags, **d, **sw_in, **cld_in, **sw_out)

    def lw_call():
        engl.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, **cld_flags, **d, **lw_in, **cld_in, **lw_out)

    def gather_sums():
        # the only exchange on the path: per-tile flux / forcing sums for the domain means
        wa.tiles.tile_sums_device(sw_out, sw12, d, sums[:16])          # native deterministic reduction
        if do_lw:
            wa.tiles.tile_sums_device(lw_out, lw12, d, sums[16:])
        wa.tiles.allreduce_sums(sums)

    def opt_call():
        wa.engine_optics.optical_averaging(idx=ms.idx, aer_op_opt=1, **d, **opt_in, **opt_out)

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
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    sw_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))   # HIP events on the launch stream
    lw_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events])) if do_lw else 0.0
    opt_ms = float(np.mean([e[3].elapsed_time(e[0]) for e in events])) if do_opt else 0.0
    opt_part_ms = {}
    if do_opt:
        # the stage is two kernels (shortwave wavelengths, longwave band pairs): time each one on its own, outside
        # the timed region, by asking for one output group only
        for part, outs in (("sw", {n: t for n, t in opt_out.items() if not n.startswith("tauaerlw")}),
                           ("lw", {n: t for n, t in opt_out.items() if n.startswith("tauaerlw")})):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            wa.engine_optics.optical_averaging(idx=ms.idx, aer_op_opt=1, **d, **opt_in, **outs)
            e0.record()
            for _ in range(3):
                wa.engine_optics.optical_averaging(idx=ms.idx, aer_op_opt=1, **d, **opt_in, **outs)
            e1.record()
            torch.cuda.synchronize()
            opt_part_ms[part] = e0.elapsed_time(e1) / 3

    if rank == 0:
        value = world * ncol * args.steps / elapsed
        bpc, bpc_sw, bpc_lw = algorithmic_bytes_per_column(args.nk, args.mode)
        # dominant kernel by measured duration: the SW column kernel or one of the two optics kernels (the two
        # LW kernels are shorter than the SW kernel; the events bracket them together)
        obpc = optics_bytes_per_column(args.nk, args.nbin) if do_opt else 0
        cand = [("sw_column_kernel", sw_ms, bpc_sw)]
        if do_opt:
            cand.append(("optics_kernel", opt_part_ms["sw"], optics_bytes_per_column(args.nk, args.nbin, part="sw")))
            cand.append(("optics_lw_kernel", opt_part_ms["lw"], optics_bytes_per_column(args.nk, args.nbin, part="lw")))
        dom_name, dom_ms, dom_bpc = max(cand, key=lambda c: c[1])
        achieved = ncol * dom_bpc / (dom_ms * 1e-3) / 1e9
        traffic = None
        valu = {}
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic = tj.get("kernels", {}).get(dom_name, {}).get("hbm_bytes_per_launch")
                kd = tj.get("kernels", {}).get(dom_name, {})
                if kd.get("valu_issue_frac") is not None:   # what actually binds this kernel
                    valu = {"fp64_valu_issue_frac": kd["valu_issue_frac"],
                            "valu_wave_instructions": kd["valu_wave_instructions"],
                            "source": tj.get("source")}
                if traffic is None and dom_name == "sw_column_kernel":
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        what = "SW+LW" if do_lw else "RRTMG-SW"
        if do_opt:
            workload = (f"{args.ni}x{args.nj}x{args.nk} synthetic columns per GPU, MOSAIC {args.nbin}-bin optical "
                        "averaging + SW+LW aerosol+clean double-call, fp64, "
                        + (f"McICA clouds in {100 * args.clouds:.0f} % of the columns" if args.clouds > 0 else "cloud-free")
                        + (" (BASELINE configs[2])" if (args.ni, args.nj, args.nk, args.nbin) == (200, 200, 40, 4) else
                           " (the domain and optics of BASELINE configs[3] on one GPU)"
                           if (args.ni, args.nj, args.nk, args.nbin) == (500, 500, 50, 8) else ""))
        elif do_lw:
            workload = (f"{args.ni}x{args.nj}x{args.nk} synthetic column batch per GPU, SW+LW aerosol+clean "
                        "double-call, fp64, cloud-free; aerosol optics supplied in the WRF-Chem compact form "
                        "(BASELINE configs[2] without the optical-averaging stage)")
        else:
            workload = (f"{args.ni}x{args.nj}x{args.nk} synthetic column batch per GPU, RRTMG-SW aerosol+clean "
                        "double-call only, fp64 (BASELINE configs[1])")
        line = {
            "metric": "columns/sec, aerosol+clean RRTMG SW+LW double-call; flux RMSE vs reference",
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.optics_precision == "f64" or not do_opt else "f64 (optical averaging f32)",
            "data": "synthetic",
            "config": {"workload": workload,
                       "columns_per_gpu": ncol, "layers": args.nk, "night_fraction": args.night,
                       "parallelism": f"(i,j) slabs over {world} GPU(s), RCCL all-reduce of 32 flux sums"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dom_name, "kernel_ms": dom_ms,
                         "algorithmic_bytes_per_column": dom_bpc,
                         "sw_kernel_ms": sw_ms, "sw_algorithmic_bytes_per_column": bpc_sw,
                         "lw_kernel_ms": lw_ms, "lw_algorithmic_bytes_per_column": bpc_lw if do_lw else 0,
                         "optics_stage_ms": opt_ms, "optics_kernel_ms": opt_part_ms.get("sw", 0.0),
                         "optics_lw_kernel_ms": opt_part_ms.get("lw", 0.0), "optics_algorithmic_bytes_per_column": obpc,
                         "step_algorithmic_GBs": ncol * (bpc + obpc) / ((sw_ms + lw_ms + opt_ms) * 1e-3) / 1e9,
                         "valu": valu,
                         "note": "fp64-VALU-bound path (SURVEY 8d); see DESIGN.md for the instruction-issue ceiling"},
        }
        if world == 1 and not args.no_cpu and args.clouds <= 0.0:
            try:
                if do_opt:   # the CPU leg runs the radiation on the optics the device produced
                    for n, t in list(sw_in.items()) + list(lw_in.items()):
                        if n.startswith(("tauaer", "waer", "gaer")):
                            st.a3[n] = t.cpu().numpy()
                cb = cpu_baseline(tab, tabl, st, min(args.cpu_sample, ncol), args.mode)
                if do_opt and cb.get("seconds_per_column"):
                    spc_opt, n_opt = cpu_optics_seconds_per_column(wa, st, ms, min(args.cpu_optics_sample, ncol))
                    cb["value"] = 1.0 / (cb["seconds_per_column"] + spc_opt)
                    cb["sample"] += (f"; optical averaging: {n_opt} columns through the C port of the engine's own "
                                     f"specification (no reference arithmetic exists), {spc_opt * n_opt:.2f} s, 1 thread")
                line["cpu_baseline"] = cb
                line["flux_rmse"] = flux_rmse(tab, tabl, st, sw_out, lw_out, max(1, min(st.nj, 400 // st.ni)), args.mode)
            except Exception as e:   # never let the baseline leg kill the bench line
                line["cpu_baseline"] = {"value": None, "unit": "columns/s", "cores": 0, "kind": "reference",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
