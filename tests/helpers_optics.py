"""Optical-averaging test plumbing: the HIP stage through the C ABI and the oracle on the same MOSAIC tile."""
import numpy as np

try:
    from helpers import SENTINEL, dims_of
except ImportError:  # imported as tests.helpers_optics (bench.py)
    from tests.helpers import SENTINEL, dims_of

SW_NAMES = [f"{s}{w}" for s in ("tauaer", "waer", "gaer", "extaer", "bscoef") for w in range(1, 5)]
LW_NAMES = [f"{s}{b}" for s in ("tauaerlw", "extaerlw") for b in range(1, 17)]


def optics_dims(st, tile=None):
    """3-D dims with kte = the model layers (the optics fill kts:kte only)."""
    return dims_of(st, tile)


def oracle_tables(wa, nbin):
    from oracle import oracle_optics as oo
    wl, ref, den = wa.engine_optics.table_info()
    return oo.make_tables(nbin, wa.engine_optics.MOSAIC_DLO_UM, wa.engine_optics.MOSAIC_DHI_UM, wl, ref, den)


def run_oracle_optics(wa, st, ms, tile=None, mix=0, legendre=False):
    from oracle import oracle_optics as oo
    return oo.tile(oracle_tables(wa, ms.nbin), ms.chem, ms.alt, ms.dz8w, ms.idx, optics_dims(st, tile), mix=mix,
                   legendre=legendre)


LEG_NAMES = [f"l{k}aer" for k in range(2, 8)]   # (4, nj_m, nk_m, ni_m) == Fortran (i,k,j,4)


def run_hip_optics(wa, st, ms, device=True, real_bytes=8, tile=None, aer_op_opt=1, names=None, idx_cw=None):
    dt = np.float64 if real_bytes == 8 else np.float32
    names = SW_NAMES + LW_NAMES if names is None else names
    shape = ms.alt.shape
    host_in = dict(chem=np.ascontiguousarray(ms.chem, dtype=dt), alt=np.ascontiguousarray(ms.alt, dtype=dt),
                   dz8w=np.ascontiguousarray(ms.dz8w, dtype=dt))
    outs = {n: np.full((4,) + tuple(shape) if n in LEG_NAMES else shape, SENTINEL, dtype=dt) for n in names}
    d = optics_dims(st, tile)
    eo = wa.engine_optics
    if device:
        import torch
        dev_in = {n: torch.from_numpy(v).cuda() for n, v in host_in.items()}
        dev_out = {n: torch.from_numpy(v).cuda() for n, v in outs.items()}
        eo.optical_averaging(idx=ms.idx, idx_cw=idx_cw, aer_op_opt=aer_op_opt, **d, **dev_in, **dev_out)
        torch.cuda.synchronize()
        return {n: v.cpu().numpy().astype(np.float64) for n, v in dev_out.items()}
    eo.optical_averaging(idx=ms.idx, idx_cw=idx_cw, aer_op_opt=aer_op_opt, **d, **host_in, **outs)
    return {n: v.astype(np.float64) for n, v in outs.items()}


def tile_mask(st, shape, tile=None):
    """True inside the written region its:ite, kts:kte, jts:jte of a (nj, nk_mem, ni) array."""
    d = dims_of(st, tile)
    m = np.zeros(shape, dtype=bool)
    m[d["jts"] - d["jms"]: d["jte"] - d["jms"] + 1, d["kts"] - d["kms"]: d["kte"] - d["kms"] + 1,
      d["its"] - d["ims"]: d["ite"] - d["ims"] + 1] = True
    return m
