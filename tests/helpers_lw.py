"""Longwave test plumbing: RRTMG_LWRAD through the HIP C-ABI and through the oracle."""
import numpy as np

try:
    from helpers import SENTINEL, cloud_arrays, dims_of
except ImportError:  # imported as tests.helpers_lw (bench.py)
    from tests.helpers import SENTINEL, cloud_arrays, dims_of

OUT2 = ["glw", "olr", "lwcf", "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
        "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"]
OUT3F = ["lwupflx", "lwupflxc", "lwupflxcln", "lwdnflx", "lwdnflxc", "lwdnflxcln"]
IN2 = ["emiss", "tsk"]
IN3 = ["t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d"]
AER = [f"tauaerlw{b}" for b in range(1, 17)]


def alloc_outputs(st, profiles=True):
    out = {"rthratenlw": np.full((st.nj, st.nk + 1, st.ni), SENTINEL)}
    for n in OUT2:
        out[n] = np.full((st.nj, st.ni), SENTINEL)
    if profiles:
        for n in OUT3F:
            out[n] = np.full((st.nj, st.nk + 3, st.ni), SENTINEL)
    return out


def run_oracle_lwrad(orc_obj, st, nlay, tile=None, clean=1, aer=1, profiles=True, cloud=None, seed_real4=0):
    from oracle import oracle_lw
    out = alloc_outputs(st, profiles)
    arrays = {n: st.a2[n] for n in IN2}
    arrays.update({n: st.a3[n] for n in IN3 + AER})
    arrays.update(cloud_arrays(st, cloud))
    cl = dict(cloud, seed_real4=seed_real4) if cloud else None
    oracle_lw.lwrad_tile(orc_obj, dims_of(st, tile), nlay, arrays, aer, clean, out, cloud=cl)
    return out


def run_hip_lwrad(eng, st, device=True, real_bytes=8, tile=None, clean=1, aer=1, profiles=True, drop=(), cloud=None):
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
        eng.rrtmg_lwrad(aer_ra_feedback=aer, clean_atm_diag=clean, **flags, **d, **dev_in, **dev_out)
        eng.lw_status()
        torch.cuda.synchronize()
        return {n: v.cpu().numpy().astype(np.float64) for n, v in dev_out.items()}
    eng.rrtmg_lwrad(aer_ra_feedback=aer, clean_atm_diag=clean, **flags, **d, **host, **outs)
    return {n: v.astype(np.float64) for n, v in outs.items()}
