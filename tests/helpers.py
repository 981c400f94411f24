"""Shared test plumbing: run RRTMG_SWRAD through the HIP C-ABI and through the
CPU oracle on the same synthetic WRF state, and compare."""
import numpy as np

SENTINEL = -7777.0

OUT2 = ["coszr", "gsw", "swcf", "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
        "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln",
        "swvisdir", "swvisdif", "swnirdir", "swnirdif", "swddir", "swddni", "swddif"]
OUT3F = ["swupflx", "swupflxc", "swupflxcln", "swdnflx", "swdnflxc", "swdnflxcln"]
IN2 = ["xcoszen", "albedo", "tsk"]
IN3 = ["t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d"]
AER = ["tauaer300", "tauaer400", "tauaer600", "tauaer999", "gaer300", "gaer400", "gaer600", "gaer999",
       "waer300", "waer400", "waer600", "waer999"]


def dims_of(st, tile=None):
    d = dict(ims=1, ime=st.ni, jms=1, jme=st.nj, kms=1, kme=st.nk + 1,
             its=1, ite=st.ni, jts=1, jte=st.nj, kts=1, kte=st.nk)
    if tile:
        d.update(tile)
    return d


def alloc_outputs(st, profiles=True):
    out = {"rthratensw": np.full((st.nj, st.nk + 1, st.ni), SENTINEL)}
    for n in OUT2:
        out[n] = np.full((st.nj, st.ni), SENTINEL)
    if profiles:
        for n in OUT3F:
            out[n] = np.full((st.nj, st.nk + 3, st.ni), SENTINEL)
    return out


CLD3 = ["cldfra3d", "qc3d", "qr3d", "qi3d", "qs3d", "re_cloud", "re_ice", "re_snow", "qndrop3d"]
CLD2 = ["xland", "xice", "snow"]


def cloud_arrays(st, cloud):
    """Cloud arguments present for this call (cloud = flags returned by synth_clouds.add_clouds, or None)."""
    if not cloud or not cloud.get("icloud"):
        return {}
    need = ["cldfra3d", "qc3d", "qr3d", "qi3d", "qs3d"]
    need += ["re_cloud"] * bool(cloud.get("has_reqc")) + ["re_ice"] * bool(cloud.get("has_reqi"))
    need += ["re_snow"] * bool(cloud.get("has_reqs")) + ["qndrop3d"] * (cloud.get("progn") == 1)
    need += ["lradius", "iradius"] * bool(cloud.get("is_cammgmp_used"))
    arr = {n: st.a3[n] for n in need if n in st.a3 and n not in cloud.get("absent", ())}
    arr.update({n: st.a2[n] for n in CLD2})
    return arr


def run_oracle_swrad(orc_obj, st, tile=None, clean=1, aer=1, solcon=1370.0, profiles=True, cloud=None, seed_real4=0):
    from oracle import oracle as orc
    out = alloc_outputs(st, profiles)
    arrays = {n: st.a2[n] for n in IN2}
    arrays.update({n: st.a3[n] for n in IN3 + AER})
    arrays.update(cloud_arrays(st, cloud))
    cl = dict(cloud, seed_real4=seed_real4) if cloud else None
    orc.swrad_tile(orc_obj, dims_of(st, tile), arrays, solcon, aer, clean, out, cloud=cl)
    return out


def run_hip_swrad(eng, st, device=True, real_bytes=8, tile=None, clean=1, aer=1, solcon=1370.0, profiles=True,
                  drop=(), cloud=None):
    """Call the product through the C-ABI.  device=True: torch CUDA tensors
    (mem=1); device=False: numpy host arrays staged by the library (mem=0)."""
    dt = np.float64 if real_bytes == 8 else np.float32
    out = alloc_outputs(st, profiles)
    host = {n: np.ascontiguousarray(st.a2[n], dtype=dt) for n in IN2}
    host.update({n: np.ascontiguousarray(st.a3[n], dtype=dt) for n in IN3})
    if aer:
        host.update({n: np.ascontiguousarray(st.a3[n], dtype=dt) for n in AER})
    host.update({n: np.ascontiguousarray(v, dtype=dt) for n, v in cloud_arrays(st, cloud).items()})
    flags = {k: v for k, v in (cloud or {}).items() if k != "absent"}
    flags.setdefault("icloud", 0)
    outs = {n: np.ascontiguousarray(v, dtype=dt) for n, v in out.items() if n not in drop}
    d = dims_of(st, tile)
    if device:
        import torch
        dev_in = {n: torch.from_numpy(v).cuda() for n, v in host.items()}
        dev_out = {n: torch.from_numpy(v).cuda() for n, v in outs.items()}
        eng.rrtmg_swrad(solcon=solcon, aer_ra_feedback=aer, clean_atm_diag=clean, **flags, **d, **dev_in, **dev_out)
        eng.sw_status()
        torch.cuda.synchronize()
        return {n: v.cpu().numpy().astype(np.float64) for n, v in dev_out.items()}
    eng.rrtmg_swrad(solcon=solcon, aer_ra_feedback=aer, clean_atm_diag=clean, **flags, **d, **host, **outs)
    return {n: v.astype(np.float64) for n, v in outs.items()}


def compare_sw(got, ref, rtol, atol, names=None):
    """Worst |got-ref|/(atol+rtol*|ref|) over all fields must be <= 1; returns worst relative error."""
    worst_rel = 0.0
    for n in (names or got.keys()):
        g, r = got[n], ref[n]
        assert g.shape == r.shape, n
        err = np.abs(g - r)
        tol = atol + rtol * np.abs(r)
        bad = err > tol
        if bad.any():
            idx = np.unravel_index(np.argmax(err / tol), err.shape)
            raise AssertionError(f"{n}: {bad.sum()} values out of tolerance; worst at {idx}: "
                                 f"got {g[idx]!r} ref {r[idx]!r}")
        m = np.abs(r) > 1e-6
        if m.any():
            worst_rel = max(worst_rel, float((err[m] / np.abs(r[m])).max()))
    return worst_rel
