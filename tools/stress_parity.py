"""Randomised parity sweep, device against oracle: many small random configurations of the three entry points (layer
counts, tile shapes, clouds with and without radii, night fractions, profiles on / off, clean call on / off, both memory
modes, all mixing rules), each compared at the tolerance of the GPU tests.  A development aid for spare GPU time, not
part of the test suite (tests/ covers fixed cases of the same kinds):
    gpurun -- 'python tools/stress_parity.py --minutes 8 --seed 1 > gpurun_out/stress.log 2>&1'
Prints a line per case; the last line is "stress: N cases, M failures"."""
import argparse
import importlib
import os
import sys
import tempfile
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mixed", action="store_true", help="optics cases only, in the mixed-precision mode, at its test tolerances")
    args = ap.parse_args()
    wa = importlib.import_module("wrfchem-arc-interactions_amd")
    from oracle import oracle as orc
    from oracle import oracle_lw
    orc.build()
    from helpers import compare_sw, run_hip_swrad, run_oracle_swrad
    from helpers_lw import run_hip_lwrad, run_oracle_lwrad
    from helpers_optics import LEG_NAMES, LW_NAMES, SW_NAMES, run_hip_optics, run_oracle_optics, tile_mask

    tmp = tempfile.mkdtemp(prefix="stress_")
    sw_tab, lw_tab = os.path.join(tmp, "RRTMG_SW_DATA"), os.path.join(tmp, "RRTMG_LW_DATA")
    wa.synth.write_sw_data(sw_tab, seed=20260001, real_bytes=8)
    wa.synth.write_lw_data(lw_tab, seed=20260003, real_bytes=8)
    wa.engine.rrtmg_swinit(sw_tab, device=0)
    sw_orc = orc.SwOracle(sw_tab)
    lw_orc = oracle_lw.LwOracle(lw_tab)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + 60.0 * args.minutes
    ncase = nfail = 0
    while time.time() < t_end:
        ncase += 1
        kind = "optics" if args.mixed else ["sw", "lw", "optics", "routes"][int(rng.integers(0, 4))]
        ni, nj = int(rng.integers(1, 9)), int(rng.integers(1, 7))
        nk = int(rng.choice([12, 20, 28, 33, 40, 50, 57, 64, 80, 100]))
        seed = int(rng.integers(1, 2**31 - 1))
        st = wa.synth.make_wrf_state(ni, nj, nk, seed=seed, night_fraction=float(rng.choice([0.0, 0.1, 0.5])))
        tile = None
        if rng.uniform() < 0.3 and ni > 2 and nj > 2:
            its, jts = int(rng.integers(1, ni)), int(rng.integers(1, nj))
            tile = dict(its=its, ite=int(rng.integers(its, ni + 1)), jts=jts, jte=int(rng.integers(jts, nj + 1)))
        desc = f"{kind} {ni}x{nj}x{nk} seed {seed} tile {tile}"
        try:
            if kind == "routes":
                # the shortwave call's other input routes, cloud-free: six ECMWF aerosol types (aer_opt 1, SW:9313-9340),
                # band-resolved aerosol (aer_opt 2 / 3, SW:10982-10991), SSiB spectral albedos (SW:10692-10703), each
                # with or without the WRF-Chem arrays, both memory modes, sub-tiles
                import torch
                from helpers import AER, IN2, IN3, alloc_outputs, dims_of
                chem = int(rng.integers(0, 2))
                arrays = {n: st.a2[n] for n in IN2}
                arrays.update({n: st.a3[n] for n in IN3 + (AER if chem else [])})
                route = ["aerod", "aer3d", "ssib"][int(rng.integers(0, 3))]
                shp2, shp3 = st.a2["albedo"].shape, (st.nj, st.nk + 1, st.ni)
                if route == "aerod":
                    e = np.zeros((6,) + shp3)
                    z = np.arange(st.nk)[None, :, None] / st.nk
                    for ia, (amp, scale) in enumerate([(0.2, 0.08), (0.1, 0.06), (0.3, 0.3), (0.1, 0.04), (0.02, 0.6), (0.01, 0.8)]):
                        e[ia, :, :st.nk, :] = rng.uniform(0, amp, (st.nj, 1, st.ni)) * np.exp(-z / scale) / st.nk * 4
                    e[:, :, :, ::4] = 0.0
                    arrays["aerod"] = e
                elif route == "aer3d":
                    tau = np.zeros((14,) + shp3)
                    tau[:, :, :st.nk, :] = rng.uniform(0.0, 0.03, size=(14, st.nj, st.nk, st.ni))
                    tau[:, :, :, ::3] = 0.0
                    arrays.update(tauaer3d_sw=tau, ssaaer3d_sw=rng.uniform(0.7, 1.0, size=tau.shape),
                                  asyaer3d_sw=rng.uniform(0.5, 0.85, size=tau.shape))
                else:
                    arrays["xland"] = np.where(rng.uniform(size=shp2) < 0.6, 1.0, 2.0)
                    for nm, lo, hi in (("alswvisdir", 0.03, 0.25), ("alswvisdif", 0.03, 0.25), ("alswnirdir", 0.15, 0.55),
                                       ("alswnirdif", 0.15, 0.55)):
                        arrays[nm] = rng.uniform(lo, hi, shp2)
                device = bool(rng.integers(0, 2))
                profiles = bool(rng.integers(0, 2))
                desc = f"routes {route} {ni}x{nj}x{nk} seed {seed} chem {chem} tile {tile} {device=} {profiles=}"
                d = dims_of(st, tile)
                ref, out = alloc_outputs(st, profiles), alloc_outputs(st, profiles)
                orc.swrad_tile(sw_orc, d, arrays, 1370.0, chem, 1, ref)
                if device:
                    dev_in = {n: torch.from_numpy(np.ascontiguousarray(v)).cuda() for n, v in arrays.items()}
                    dev_out = {n: torch.from_numpy(v).cuda() for n, v in out.items()}
                    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=chem, clean_atm_diag=1, icloud=0, **d, **dev_in, **dev_out)
                    wa.engine.sw_status()
                    got = {n: v.cpu().numpy() for n, v in dev_out.items()}
                else:
                    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=chem, clean_atm_diag=1, icloud=0, **d,
                                          **{n: np.ascontiguousarray(v) for n, v in arrays.items()}, **out)
                    got = out
                compare_sw(got, ref, 1e-9, 1e-9, names=[n for n in got if not n.endswith("dif")])
                compare_sw(got, ref, 1e-8, 1e-8, names=[n for n in got if n.endswith("dif")])
            elif kind in ("sw", "lw"):
                cloud = None
                if rng.uniform() < 0.5:
                    cloud = wa.synth_clouds.add_clouds(st, seed=seed + 1, cloudy_fraction=float(rng.choice([0.3, 0.6, 1.0])),
                                                       with_radii=bool(rng.integers(0, 2)))
                if cloud is not None and rng.uniform() < 0.4:   # the hydrometeor variants of the wrapper (SW:10461-10492, 10883-10898)
                    variant = ["f_ice_phy", "fer_hires", "cammgmp", "warm_rain"][int(rng.integers(0, 4))]
                    if variant == "f_ice_phy":
                        cloud.update(f_ice_phy=1, absent=("qi3d",))
                        st.a3["qs3d"] = st.a3["qs3d"] + st.a3["qi3d"]
                    elif variant == "fer_hires":
                        cloud.update(mp_physics=5)
                    elif variant == "cammgmp":
                        cloud.update(is_cammgmp_used=1)
                        st.a3["lradius"] = rng.uniform(1.0, 70.0, st.a3["qc3d"].shape)
                        st.a3["iradius"] = rng.uniform(2.0, 160.0, st.a3["qc3d"].shape)
                    else:
                        cloud.update(warm_rain=True)
                    desc += f" {variant}"
                kw = dict(tile=tile, cloud=cloud, profiles=bool(rng.integers(0, 2)), clean=int(rng.integers(0, 2)),
                          aer=int(rng.integers(0, 2)))
                device = bool(rng.integers(0, 2))
                r4 = rng.uniform() < 0.25     # a stock (REAL*4) build's arrays (with clouds: single-precision McICA seeds)
                if r4:
                    for dd in (st.a2, st.a3):
                        for n in dd:
                            dd[n] = dd[n].astype(np.float32).astype(np.float64)
                rb = 4 if r4 else 8
                desc += f" cloud {cloud is not None} {kw['profiles']=} {kw['clean']=} {kw['aer']=} {device=} {r4=}"
                if kind == "sw":
                    got = run_hip_swrad(wa.engine, st, device=device, real_bytes=rb, **kw)
                    ref = run_oracle_swrad(sw_orc, st, seed_real4=int(r4), **kw)
                else:
                    wa.engine_lw.rrtmg_lwinit(5000.0, lw_tab, kme=nk + 1, device=0)
                    nlay = wa.engine_lw.lw_nlayers(nk + 1, 5000.0, real_bytes=rb)
                    got = run_hip_lwrad(wa.engine_lw, st, device=device, real_bytes=rb, **kw)
                    ref = run_oracle_lwrad(lw_orc, st, nlay, seed_real4=int(r4), **kw)
                # the diffuse surface fluxes are differences of two fluxes a hundred times their size: 1e-8
                if r4:
                    compare_sw(got, ref, 3e-7, 2e-4 if cloud is not None else 1e-6)
                else:
                    compare_sw(got, ref, 1e-9, 1e-9, names=[n for n in got if not n.endswith("dif")])
                    compare_sw(got, ref, 1e-8, 1e-8, names=[n for n in got if n.endswith("dif")])
            else:
                nbin = int(rng.choice([4, 8]))
                nk = min(nk, 20)
                st = wa.synth.make_wrf_state(ni, nj, nk, seed=seed)
                wa.engine_optics.optics_init(nbin, device=0)
                ms = wa.synth_mosaic.make_mosaic_state(st, nbin=nbin, seed=seed + 2)
                opt = int(rng.choice([1, 2, 3, 4] if args.mixed else [1, 2, 3, 4, 5]))
                leg = (not args.mixed) and rng.uniform() < 0.25
                device = bool(rng.integers(0, 2))
                if tile is not None:
                    tile = dict(tile, kts=int(rng.integers(1, 3)), kte=int(rng.integers(nk - 2, nk + 1)))
                desc = f"optics {ni}x{nj}x{nk} nbin {nbin} opt {opt} leg {leg} seed {seed} tile {tile} {device=}"
                names = SW_NAMES + LW_NAMES + (LEG_NAMES if leg else [])
                wa.engine_optics.optics_precision("mixed" if args.mixed else "f64")
                o4 = (not args.mixed) and rng.uniform() < 0.2
                if o4:
                    ms.chem = ms.chem.astype(np.float32).astype(np.float64)
                    ms.alt = ms.alt.astype(np.float32).astype(np.float64)
                    ms.dz8w = ms.dz8w.astype(np.float32).astype(np.float64)
                    desc += " r4"
                got = run_hip_optics(wa, st, ms, aer_op_opt=opt, names=names, tile=tile, device=device, real_bytes=4 if o4 else 8)
                ref = run_oracle_optics(wa, st, ms, mix={1: 0, 3: 0, 2: 1, 4: 1, 5: 2}[opt], legendre=leg, tile=tile)
                mask = tile_mask(st, ms.alt.shape, tile)
                for n in names:
                    g, r = (got[n][:, mask], ref[n][:, mask]) if n in LEG_NAMES else (got[n][mask], ref[n][mask])
                    tol = 1e-9 + (0 if n in LEG_NAMES else 1e-8 * np.abs(r))
                    if o4:
                        tol = 3e-7 * np.abs(r) + 1e-6 * (1.0 if n in LEG_NAMES else np.abs(r).max())
                    if args.mixed:   # the bounds of tests/test_optics_gpu.py::test_optics_mixed_precision
                        rel = 2e-4 if n.startswith("bscoef") else (1e-4 if "lw" in n else 2e-5)
                        tol = 1e-5 if n.startswith(("waer", "gaer")) else rel * np.abs(r) + 1e-9 * np.abs(r).max()
                    if not (np.isfinite(g).all() and (np.abs(g - r) <= tol + 1e-30).all()):
                        raise AssertionError(f"{n}: max diff {np.abs(g - r).max():.3e} (rel {np.max(np.abs(g - r) / (np.abs(r) + 1e-300)):.3e})")
            print(f"ok   {desc}", flush=True)
        except Exception as e:   # noqa: BLE001 -- every kind of failure is a finding
            nfail += 1
            print(f"FAIL {desc}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    print(f"stress: {ncase} cases, {nfail} failures", flush=True)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
