"""Host-side mirror of the reference's longwave operator interface:
``rrtmg_lwinit`` (module_ra_rrtmg_lw.F:12845) and ``RRTMG_LWRAD`` (:11451),
forwarding to librrtmg_hip.so.  See engine.py for the conventions."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import CP, ERRLEN, WrfFatal, _CAMRAD, _CLD2, _CLD3, _CLDFLAGS, _check, _ptr, load_library

_IN2 = ["emiss", "tsk"]
_IN3 = ["t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d", "o33d"]
_AER = [f"tauaerlw{b}" for b in range(1, 17)]
_OUT = ["rthratenlw", "glw", "olr", "lwcf",
        "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
        "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln",
        "lwupflx", "lwupflxc", "lwupflxcln", "lwdnflx", "lwdnflxc", "lwdnflxcln"]


class LwradArgs(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("ims", "ime", "jms", "jme", "kms", "kme",
                                         "its", "ite", "jts", "jte", "kts", "kte", "real_bytes", "mem")]
                + [("stream", C.c_void_p)]
                + [(n, C.c_int) for n in ("aer_ra_feedback", "clean_atm_diag", "icloud", "o3input")]
                + [(n, C.c_void_p) for n in _IN2 + _IN3]
                + [("tauaerlw", C.c_void_p * 16)]
                + [(n, C.c_void_p) for n in _OUT]
                + [(n, C.c_void_p) for n in _CLD3 + _CLD2]
                + [(n, C.c_int) for n in _CLDFLAGS]
                + [("g", C.c_double)]
                + [(n, C.c_void_p) for n in _CAMRAD])


EXPORTS_LW = ["rrtmg_hip_lw_init", "rrtmg_hip_lw_nlayers", "rrtmg_hip_lw_nlayers_real", "rrtmg_hip_lwrad", "rrtmg_hip_lw_status",
              "rrtmg_hip_lw_finalize", "rrtmg_hip_lw_reduce_tables", "rrtmg_hip_lw_table_info"]

_ready = False


def _lib():
    global _ready
    lib = load_library()
    if not _ready:
        lib.rrtmg_hip_lw_init.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_char_p]
        lib.rrtmg_hip_lw_nlayers.argtypes = [C.c_int, C.c_double]
        lib.rrtmg_hip_lw_nlayers_real.argtypes = [C.c_int, C.c_double, C.c_int]
        lib.rrtmg_hip_lwrad.argtypes = [C.POINTER(LwradArgs), C.c_char_p]
        lib.rrtmg_hip_lw_status.argtypes = [C.c_char_p]
        lib.rrtmg_hip_lw_reduce_tables.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_void_p, C.c_longlong,
                                                   C.POINTER(C.c_longlong), C.c_void_p, C.c_void_p, C.c_char_p]
        lib.rrtmg_hip_lw_table_info.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p]
        _ready = True
    return lib


def lw_nlayers(kme, p_top, real_bytes=8):
    """NLAYERS of rrtmg_lwinit (LW:12861) with the quotient evaluated in the caller's REAL (real_bytes 4 or 8), as
    rrtmg_hip_lwrad forms it for the arrays it is handed."""
    return _lib().rrtmg_hip_lw_nlayers_real(int(kme), float(p_top), int(real_bytes))


def rrtmg_lwinit(p_top, data_path: str, kme: int, data_real_bytes: int = 8, cpdair: float = CP, device: int = 0):
    """rrtmg_lwinit(p_top, allowed_to_read, ...) (LW:12845-12874): sets NLAYERS from p_top, reads RRTMG_LW_DATA."""
    err = C.create_string_buffer(ERRLEN)
    _check(_lib().rrtmg_hip_lw_init(data_path.encode(), data_real_bytes, cpdair, float(p_top), int(kme), device, err), err)


def rrtmg_lwrad(*, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte,
                aer_ra_feedback=0, clean_atm_diag=0, icloud=0, o3input=0, stream=None,
                has_reqc=0, has_reqi=0, has_reqs=0, warm_rain=False, progn=0, is_cammgmp_used=False, f_ice_phy=False,
                mp_physics=0, g=9.81, **arrays):
    """RRTMG_LWRAD (LW:11451-12700); keyword names are the reference's dummy argument names in lower case."""
    lib = _lib()
    a = LwradArgs()
    for n, v in dict(ims=ims, ime=ime, jms=jms, jme=jme, kms=kms, kme=kme, its=its, ite=ite, jts=jts, jte=jte,
                     kts=kts, kte=kte).items():
        setattr(a, n, int(v))
    a.aer_ra_feedback = int(aer_ra_feedback)
    a.clean_atm_diag = int(clean_atm_diag)
    a.icloud = int(icloud)
    a.o3input = int(o3input)
    for n, v in dict(has_reqc=has_reqc, has_reqi=has_reqi, has_reqs=has_reqs, warm_rain=warm_rain, progn=progn,
                     is_cammgmp_used=is_cammgmp_used, f_ice_phy=f_ice_phy, mp_physics=mp_physics).items():
        setattr(a, n, int(v))
    a.g = float(g)
    known = set(_IN2 + _IN3 + _AER + _OUT + _CLD3 + _CLD2 + _CAMRAD)
    unknown = set(arrays) - known
    if unknown:
        raise TypeError(f"unknown RRTMG_LWRAD arguments: {sorted(unknown)}")
    dev = size = None
    for n, v in arrays.items():
        p, is_dev, isz = _ptr(v)
        if p is None:
            continue
        if dev is None:
            dev, size = is_dev, isz
        elif dev != is_dev or size != isz:
            raise ValueError("all arrays must live in the same memory space and have the same REAL size")
        if n.startswith("tauaerlw"):
            a.tauaerlw[int(n[8:]) - 1] = p
        else:
            setattr(a, n, p)
    a.real_bytes = size
    a.mem = 1 if dev else 0
    if dev:
        import torch
        a.stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    err = C.create_string_buffer(ERRLEN)
    _check(lib.rrtmg_hip_lwrad(C.byref(a), err), err)


def lw_status():
    err = C.create_string_buffer(ERRLEN)
    _check(_lib().rrtmg_hip_lw_status(err), err)


def lw_reduce_tables(data_path, real_bytes=8, cpdair=CP):
    lib = _lib()
    err = C.create_string_buffer(ERRLEN)
    n = C.c_longlong(0)
    _check(lib.rrtmg_hip_lw_reduce_tables(data_path.encode(), real_bytes, cpdair, None, 0, C.byref(n), None, None, err), err)
    k, e, t = np.zeros(n.value), np.zeros(10001), np.zeros(10001)
    _check(lib.rrtmg_hip_lw_reduce_tables(data_path.encode(), real_bytes, cpdair, k.ctypes.data, n.value, C.byref(n),
                                          e.ctypes.data, t.ctypes.data, err), err)
    return k, e, t


def lw_table_info(data_path, g, real_bytes=8):
    lib = _lib()
    err = C.create_string_buffer(ERRLEN)
    f = np.zeros(24, dtype=np.int32)
    d = np.zeros(8)
    _check(lib.rrtmg_hip_lw_table_info(data_path.encode(), real_bytes, g, f.ctypes.data, d.ctypes.data, err), err)
    names = ["band", "ig", "ng", "off_self", "off_for", "off_major_lo", "off_major_up", "off_frac_lo", "off_frac_up",
             "cfg_lo", "cfg_up", "mlo0", "mlo1", "mlo2", "mup0", "mup1", "nlo0", "nlo1", "nlo2", "nup0", "nup1", "stride"]
    out = dict(zip(names, f.tolist()))
    out.update(dict(zip(["frac_lo", "frac_up", "halo_lo0", "halo_lo1", "halo_up0", "halo_up1", "postscale_up",
                         "delwave"], d.tolist())))
    return out
