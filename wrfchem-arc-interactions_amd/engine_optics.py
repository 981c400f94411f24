"""Host-side mirror of the aerosol optical-averaging stage, forwarding to librrtmg_hip.so.

WRF-Chem calls this stage ``optical_averaging`` (upstream chem/module_optical_averaging.F, driven from
chem/optical_driver.F); the reference repository holds only its interface: the MOSAIC tracers it reads
(Registry/registry.chem:2325-2447) and the arrays it fills (registry.chem:1332-1387), handed to the radiation
driver at dyn_em/module_first_rk_step_part1.F:374-385.  Keyword names below are those registry names.
See engine.py for the memory conventions (numpy = host pointers, torch CUDA tensors = HBM pointers).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import ERRLEN, _check, _ptr, load_library

SPECIES = ["so4", "no3", "cl", "nh4", "na", "oin", "oc", "bc", "water", "num"]   # slots of idx[bin][.]
MAXBIN, NSLOT, NSW, NLW, NWAVE, NSPEC = 8, 10, 4, 16, 20, 9
MOSAIC_DLO_UM, MOSAIC_DHI_UM = 0.0390625, 10.0

_SW_OUT = ["tauaer", "waer", "gaer", "extaer", "bscoef"]
_LW_OUT = ["tauaerlw", "extaerlw"]


class OpticsArgs(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("ims", "ime", "jms", "jme", "kms", "kme",
                                         "its", "ite", "jts", "jte", "kts", "kte", "real_bytes", "mem")]
                + [("stream", C.c_void_p), ("aer_op_opt", C.c_int), ("num_chem", C.c_int),
                   ("idx", (C.c_int * NSLOT) * MAXBIN),
                   ("chem", C.c_void_p), ("alt", C.c_void_p), ("dz8w", C.c_void_p),
                   ("tauaer", C.c_void_p * 4), ("waer", C.c_void_p * 4), ("gaer", C.c_void_p * 4),
                   ("extaer", C.c_void_p * 4), ("bscoef", C.c_void_p * 4),
                   ("tauaerlw", C.c_void_p * 16), ("extaerlw", C.c_void_p * 16),
                   ("idx_cw", (C.c_int * NSLOT) * MAXBIN), ("laer", C.c_void_p * 6)])


EXPORTS_OPTICS = ["rrtmg_hip_optics_init", "rrtmg_hip_optical_averaging", "rrtmg_hip_optics_finalize",
                  "rrtmg_hip_optics_table_info", "rrtmg_hip_modal_to_sections", "rrtmg_hip_modal_section_idx",
                  "rrtmg_hip_optics_precision"]

MAXMODE = 4
MADE_SIGMA = (1.7, 2.0, 2.2)    # Aitken, accumulation, coarse mode of MADE/SORGAM


class ModalArgs(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("ims", "ime", "jms", "jme", "kms", "kme",
                                         "its", "ite", "jts", "jte", "kts", "kte", "real_bytes", "mem")]
                + [("stream", C.c_void_p), ("nmode", C.c_int), ("num_chem", C.c_int),
                   ("sigma", C.c_double * MAXMODE), ("idx", (C.c_int * NSLOT) * MAXMODE),
                   ("chem", C.c_void_p), ("chem_sec", C.c_void_p)])

_ready = False


def _lib():
    global _ready
    lib = load_library()
    if not _ready:
        lib.rrtmg_hip_optics_init.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
        lib.rrtmg_hip_optical_averaging.argtypes = [C.POINTER(OpticsArgs), C.c_char_p]
        lib.rrtmg_hip_optics_table_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _ready = True
    return lib


def optics_init(nbin=4, dlo_um=MOSAIC_DLO_UM, dhi_um=MOSAIC_DHI_UM, refindex=None, density=None, device=0):
    """Sections and material table.  refindex [9][20][2] / density [9] replace the built-in table."""
    ref = None if refindex is None else np.ascontiguousarray(refindex, dtype=np.float64).reshape(NSPEC, NWAVE, 2)
    den = None if density is None else np.ascontiguousarray(density, dtype=np.float64).reshape(NSPEC)
    err = C.create_string_buffer(ERRLEN)
    _check(_lib().rrtmg_hip_optics_init(int(nbin), float(dlo_um), float(dhi_um),
                                        None if ref is None else ref.ctypes.data,
                                        None if den is None else den.ctypes.data, int(device), err), err)


def table_info():
    """(wavelength_um[20], refindex[9,20,2], density[9]) the engine uses (built-in table before init)."""
    wl, ref, den = np.zeros(NWAVE), np.zeros((NSPEC, NWAVE, 2)), np.zeros(NSPEC)
    _lib().rrtmg_hip_optics_table_info(wl.ctypes.data, ref.ctypes.data, den.ctypes.data)
    return wl, ref, den


def optical_averaging(*, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte, chem, alt, dz8w, idx,
                      aer_op_opt=1, stream=None, idx_cw=None, **outputs):
    """chem: (num_chem, nj_m, nk_m, ni_m) C-order == Fortran chem(i,k,j,species); idx[bin][slot] 1-based species
    numbers (0 = not carried) in the SPECIES order; idx_cw: the same for the cloud-borne counterparts ("_cw" tracers,
    added to the interstitial ones; None = none); l2aer..l7aer: (4, nj_m, nk_m, ni_m) == Fortran (i,k,j,4), Legendre
    moments of the phase function (all six or none); outputs by registry name: tauaer1..4, waer1..4, gaer1..4,
    extaer1..4, bscoef1..4, tauaerlw1..16, extaerlw1..16 (each group complete or absent)."""
    lib = _lib()
    a = OpticsArgs()
    for n, v in dict(ims=ims, ime=ime, jms=jms, jme=jme, kms=kms, kme=kme, its=its, ite=ite, jts=jts, jte=jte,
                     kts=kts, kte=kte).items():
        setattr(a, n, int(v))
    a.aer_op_opt = int(aer_op_opt)
    idx = np.asarray(idx, dtype=np.int64)
    if idx.ndim != 2 or idx.shape[1] != NSLOT or idx.shape[0] > MAXBIN:
        raise ValueError("idx must be [nbin <= 8][10]")
    for b in range(idx.shape[0]):
        for s in range(NSLOT):
            a.idx[b][s] = int(idx[b, s])
    if idx_cw is not None:
        idx_cw = np.asarray(idx_cw, dtype=np.int64)
        if idx_cw.shape != idx.shape:
            raise ValueError("idx_cw must have the shape of idx")
        for b in range(idx_cw.shape[0]):
            for s in range(NSLOT):
                a.idx_cw[b][s] = int(idx_cw[b, s])
    a.num_chem = int(chem.shape[0])
    dev = size = None

    def put(v):
        nonlocal dev, size
        p, is_dev, isz = _ptr(v)
        if p is None:
            return None
        if dev is None:
            dev, size = is_dev, isz
        elif dev != is_dev or size != isz:
            raise ValueError("all arrays must live in the same memory space and have the same REAL size")
        return p

    a.chem, a.alt, a.dz8w = put(chem), put(alt), put(dz8w)
    for n, v in outputs.items():
        if len(n) == 5 and n[0] == "l" and n[1] in "234567" and n[2:] == "aer":
            a.laer[int(n[1]) - 2] = put(v)
            continue
        for stem in _LW_OUT + _SW_OUT:   # 'tauaerlw' before 'tauaer'
            if n.startswith(stem) and n[len(stem):].isdigit():
                i = int(n[len(stem):]) - 1
                arr = getattr(a, stem)
                if not 0 <= i < len(arr):
                    raise TypeError(f"unknown optical_averaging argument: {n}")
                arr[i] = put(v)
                break
        else:
            raise TypeError(f"unknown optical_averaging argument: {n}")
    a.real_bytes = size
    a.mem = 1 if dev else 0
    if dev:
        import torch
        a.stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    err = C.create_string_buffer(ERRLEN)
    _check(lib.rrtmg_hip_optical_averaging(C.byref(a), err), err)


def optics_finalize():
    _lib().rrtmg_hip_optics_finalize()


def modal_section_idx(nbin):
    """idx table ([nbin][10], 1-based) of the sectional array modal_to_sections fills."""
    buf = (C.c_int * (MAXBIN * NSLOT))()
    lib = _lib()
    lib.rrtmg_hip_modal_section_idx.argtypes = [C.c_int, C.POINTER(C.c_int)]
    if lib.rrtmg_hip_modal_section_idx(int(nbin), buf) != 0:
        raise ValueError("nbin must be 1..8")
    return np.array(buf, dtype=np.int64).reshape(MAXBIN, NSLOT)[:nbin]


def modal_to_sections(*, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte, chem, idx, sigma, chem_sec,
                      stream=None):
    """Modal aerosol (MADE/SORGAM type) -> the sectional tracers optical_averaging takes.  chem: (num_chem, nj_m, nk_m,
    ni_m); idx[mode][slot] 1-based species numbers in SPECIES order (0 = not carried); sigma[mode]; chem_sec:
    (nbin*10, nj_m, nk_m, ni_m), written inside the tile (use modal_section_idx(nbin) as its idx table)."""
    lib = _lib()
    lib.rrtmg_hip_modal_to_sections.argtypes = [C.POINTER(ModalArgs), C.c_char_p]
    a = ModalArgs()
    for n, v in dict(ims=ims, ime=ime, jms=jms, jme=jme, kms=kms, kme=kme, its=its, ite=ite, jts=jts, jte=jte,
                     kts=kts, kte=kte).items():
        setattr(a, n, int(v))
    idx = np.asarray(idx, dtype=np.int64)
    if idx.ndim != 2 or idx.shape[1] != NSLOT or idx.shape[0] > MAXMODE or len(sigma) != idx.shape[0]:
        raise ValueError("idx must be [nmode <= 4][10] with one sigma per mode")
    a.nmode = idx.shape[0]
    for m in range(idx.shape[0]):
        a.sigma[m] = float(sigma[m])
        for s in range(NSLOT):
            a.idx[m][s] = int(idx[m, s])
    a.num_chem = int(chem.shape[0])
    pc, dev_c, sz_c = _ptr(chem)
    ps, dev_s, sz_s = _ptr(chem_sec)
    if dev_c != dev_s or sz_c != sz_s:
        raise ValueError("chem and chem_sec must live in the same memory space and have the same REAL size")
    a.chem, a.chem_sec = pc, ps
    a.real_bytes, a.mem = int(sz_c), 1 if dev_c else 0
    if dev_c:
        import torch
        a.stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    err = C.create_string_buffer(ERRLEN)
    _check(lib.rrtmg_hip_modal_to_sections(C.byref(a), err), err)


def optics_precision(mode):
    """"f64" / 0: fp64 throughout (default); "mixed" / 1: the coefficients and sums of the Mie series in packed fp32,
    its recurrences in fp64 (BASELINE configs[4])."""
    m = {"f64": 0, "mixed": 1}.get(mode, mode)
    lib = _lib()
    lib.rrtmg_hip_optics_precision.argtypes = [C.c_int, C.c_char_p]
    err = C.create_string_buffer(ERRLEN)
    _check(lib.rrtmg_hip_optics_precision(int(m), err), err)
