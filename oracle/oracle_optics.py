"""ctypes access to the optical-averaging half of the CPU restatement -- TEST INFRASTRUCTURE.

Parity unpinned: the reference repository holds no arithmetic for this stage (see optics_oracle.c).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .oracle import lib as _baselib, dp

NSPEC, NSLOT, NWAVE, NSW, NLW = 9, 10, 20, 4, 16
SW_FIELDS = ["tau", "omega", "g", "ext", "bsc"]


class Tables(C.Structure):
    _fields_ = [("refr", C.c_double * NWAVE * NSPEC), ("refi", C.c_double * NWAVE * NSPEC),
                ("dens", C.c_double * NSPEC), ("wavelength_um", C.c_double * NWAVE),
                ("dlo_um", C.c_double * 8), ("dhi_um", C.c_double * 8), ("nbin", C.c_int)]


def _lib():
    l = _baselib()
    if not getattr(l, "_optics_ready", False):
        l.optics_oracle_mie.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        if hasattr(l, "optics_oracle_mie_coated"):   # (not in the operation-counting build, oracle/count)
            l.optics_oracle_mie_coated.argtypes = [C.c_double] * 6 + [dp]
            l.optics_oracle_last_moments.argtypes = [dp]
            l.optics_oracle_tile_legendre.argtypes = ([C.c_void_p, C.c_int] + [C.c_int] * 9
                                                      + [C.c_int, C.POINTER(C.c_int), dp, dp, dp, dp, dp, dp])
        l.optics_oracle_cell.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_double, C.c_double, dp, dp]
        l.optics_oracle_tile.argtypes = [C.c_void_p, C.c_int] + [C.c_int] * 9 + [C.c_int, C.POINTER(C.c_int), dp, dp, dp, dp, dp]
        l._optics_ready = True
    return l


def mie(x, m):
    """(Qext, Qsca, g, Qback) of a homogeneous sphere; m = n + ik with k >= 0."""
    out = np.zeros(4)
    _lib().optics_oracle_mie(float(x), float(np.real(m)), float(np.imag(m)), out.ctypes.data_as(dp))
    return tuple(out)


def mie_coated(x, y, m1, m2):
    """(Qext, Qsca, g, Qback), referred to the outer radius, of a sphere of size parameter y and shell index m2 around
    a core of size parameter x <= y and index m1."""
    out = np.zeros(4)
    _lib().optics_oracle_mie_coated(float(x), float(y), float(np.real(m1)), float(np.imag(m1)), float(np.real(m2)),
                                    float(np.imag(m2)), out.ctypes.data_as(dp))
    return tuple(out)


def moments(x, m, core=None):
    """Legendre moments chi_0..chi_7 of the normalised phase function of a homogeneous sphere (x, m), or of a coated
    one with core = (x_core, m_core) inside it."""
    if core is None:
        mie(x, m)
    else:
        mie_coated(core[0], x, core[1], m)
    out = np.zeros(8)
    _lib().optics_oracle_last_moments(out.ctypes.data_as(dp))
    return out


def make_tables(nbin, dlo_um, dhi_um, wavelength_um, refindex, density):
    """refindex [9][20][2], density [9], wavelength_um [20] (e.g. from engine_optics.table_info())."""
    t = Tables()
    ref = np.asarray(refindex, dtype=np.float64).reshape(NSPEC, NWAVE, 2)
    for s in range(NSPEC):
        for w in range(NWAVE):
            t.refr[s][w] = ref[s, w, 0]
            t.refi[s][w] = ref[s, w, 1]
        t.dens[s] = float(density[s])
    for w in range(NWAVE):
        t.wavelength_um[w] = float(wavelength_um[w])
    lo = float(dlo_um)
    for b in range(nbin):
        hi = float(dhi_um) if b == nbin - 1 else float(dlo_um) * (float(dhi_um) / float(dlo_um)) ** ((b + 1) / nbin)
        t.dlo_um[b], t.dhi_um[b] = lo, hi
        lo = hi
    t.nbin = nbin
    return t


def cell(tables, mass, num, alt, dz, mix=0):
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    num = np.ascontiguousarray(num, dtype=np.float64)
    osw, olw = np.zeros((NSW, 5)), np.zeros((NLW, 2))
    _lib().optics_oracle_cell(C.addressof(tables), int(mix), mass.ctypes.data_as(dp), num.ctypes.data_as(dp),
                              float(alt), float(dz), osw.ctypes.data_as(dp), olw.ctypes.data_as(dp))
    return osw, olw


def tile(tables, chem, alt, dz8w, idx, dims, mix=0, legendre=False):
    """chem (num_chem, nj_m, nk_m, ni_m) C-order == Fortran (i,k,j,species); alt, dz8w (nj_m, nk_m, ni_m); idx
    [nbin][10] 1-based (0 = absent); dims = dict(ims..kte).  Returns dict of 3-D arrays named like the registry."""
    chem = np.ascontiguousarray(chem, dtype=np.float64)
    alt = np.ascontiguousarray(alt, dtype=np.float64)
    dz8w = np.ascontiguousarray(dz8w, dtype=np.float64)
    num_chem, nj_m, nk_m, ni_m = chem.shape
    d = dims
    assert (nj_m, nk_m, ni_m) == (d["jme"] - d["jms"] + 1, d["kme"] - d["kms"] + 1, d["ime"] - d["ims"] + 1)
    idx0 = np.full((8, NSLOT), -1, dtype=np.int32)
    idx = np.asarray(idx, dtype=np.int32)
    idx0[: idx.shape[0], :] = idx - 1
    sw5 = np.zeros((NSW, 5, nj_m, nk_m, ni_m))
    lw2 = np.zeros((NLW, 2, nj_m, nk_m, ni_m))
    args = [C.addressof(tables), int(mix), ni_m, nk_m, nj_m,
            d["its"] - d["ims"], d["kts"] - d["kms"], d["jts"] - d["jms"],
            d["ite"] - d["its"] + 1, d["kte"] - d["kts"] + 1, d["jte"] - d["jts"] + 1,
            num_chem, idx0.ctypes.data_as(C.POINTER(C.c_int)), chem.ctypes.data_as(dp),
            alt.ctypes.data_as(dp), dz8w.ctypes.data_as(dp), sw5.ctypes.data_as(dp), lw2.ctypes.data_as(dp)]
    out = {}
    if legendre:   # l2aer..l7aer as (4, nj_m, nk_m, ni_m) == Fortran (i,k,j,4)
        leg = np.zeros((NSW, 6, nj_m, nk_m, ni_m))
        _lib().optics_oracle_tile_legendre(*args, leg.ctypes.data_as(dp))
        for k in range(6):
            out[f"l{k + 2}aer"] = np.ascontiguousarray(leg[:, k])
    else:
        _lib().optics_oracle_tile(*args)
    for w in range(NSW):
        out[f"tauaer{w + 1}"] = sw5[w, 0]
        out[f"waer{w + 1}"] = sw5[w, 1]
        out[f"gaer{w + 1}"] = sw5[w, 2]
        out[f"extaer{w + 1}"] = sw5[w, 3]
        out[f"bscoef{w + 1}"] = sw5[w, 4]
    for b in range(NLW):
        out[f"tauaerlw{b + 1}"] = lw2[b, 0]
        out[f"extaerlw{b + 1}"] = lw2[b, 1]
    return out


def modal_to_sections(tables, chem, idx_mode, sigma, dims):
    """Modal tracers -> sectional tracers (modal_sections.hip restated).  chem (num_chem, nj_m, nk_m, ni_m);
    idx_mode [nmode][10] 1-based (0 = absent); sigma [nmode].  Returns chem_sec (nbin*10, nj_m, nk_m, ni_m), filled
    inside the tile its:ite, kts:kte, jts:jte and zero elsewhere."""
    chem = np.ascontiguousarray(chem, dtype=np.float64)
    idx_mode = np.asarray(idx_mode, dtype=np.int64)
    nmode = idx_mode.shape[0]
    nbin = tables.nbin
    _, nj_m, nk_m, ni_m = chem.shape
    d = dims
    l = _lib()
    l.optics_oracle_modal_cell.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp]
    sec = np.zeros((nbin * NSLOT, nj_m, nk_m, ni_m))
    sig = np.ascontiguousarray(sigma, dtype=np.float64)
    mass, num = np.zeros((nmode, NSPEC)), np.zeros(nmode)
    sm, sn = np.zeros((nbin, NSPEC)), np.zeros(nbin)
    for j in range(d["jts"] - d["jms"], d["jte"] - d["jms"] + 1):
        for k in range(d["kts"] - d["kms"], d["kte"] - d["kms"] + 1):
            for i in range(d["its"] - d["ims"], d["ite"] - d["ims"] + 1):
                for m in range(nmode):
                    for s in range(NSPEC):
                        mass[m, s] = chem[idx_mode[m, s] - 1, j, k, i] if idx_mode[m, s] > 0 else 0.0
                    num[m] = chem[idx_mode[m, NSPEC] - 1, j, k, i] if idx_mode[m, NSPEC] > 0 else 0.0
                l.optics_oracle_modal_cell(C.addressof(tables), nmode, mass.ctypes.data_as(dp), num.ctypes.data_as(dp),
                                           sig.ctypes.data_as(dp), sm.ctypes.data_as(dp), sn.ctypes.data_as(dp))
                for b in range(nbin):
                    sec[b * NSLOT:b * NSLOT + NSPEC, j, k, i] = sm[b]
                    sec[b * NSLOT + NSPEC, j, k, i] = sn[b]
    return sec
