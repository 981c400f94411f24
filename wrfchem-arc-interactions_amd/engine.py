"""Host-side mirror of the reference's operator interface for the hot path.

``rrtmg_swinit`` / ``rrtmg_swrad`` keep the names, argument names and error
behaviour of the Fortran module procedures they stand for
(module_ra_rrtmg_sw.F:11211 ``rrtmg_swinit``, :9901 ``RRTMG_SWRAD``) and
forward to the C ABI in ``librrtmg_hip.so`` (include/rrtmg_hip.h).  Arrays may
be numpy arrays (host memory, staged by the library) or torch CUDA tensors
(HBM-resident, enqueued on torch's current stream).  There is no CPU path: if
the HIP library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librrtmg_hip.so")
ERRLEN = 256
CP = 7.0 * 287.0 / 2.0   # WRF module_model_constants: cp = 7.*r_d/2.


class WrfFatal(RuntimeError):
    """What the Fortran shim turns into ``wrf_error_fatal(msg)``."""


_IN2 = ["xcoszen", "albedo", "tsk"]
_IN3 = ["t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d", "o33d",
        "tauaer300", "tauaer400", "tauaer600", "tauaer999",
        "gaer300", "gaer400", "gaer600", "gaer999",
        "waer300", "waer400", "waer600", "waer999"]
_OUT3 = ["rthratensw"]
_OUT2 = ["coszr", "gsw", "swcf",
         "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
         "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln",
         "swvisdir", "swvisdif", "swnirdir", "swnirdif", "swddir", "swddni", "swddif"]
_OUT3F = ["swupflx", "swupflxc", "swupflxcln", "swdnflx", "swdnflxc", "swdnflxcln"]


# cloud arguments of RRTMG_SWRAD / RRTMG_LWRAD used when icloud /= 0 (SW:9901-9945)
_CLD3 = ["cldfra3d", "qc3d", "qr3d", "qi3d", "qs3d", "re_cloud", "re_ice", "re_snow", "qndrop3d"]
_CLD2 = ["xland", "xice", "snow"]
_CLDFLAGS = ["has_reqc", "has_reqi", "has_reqs", "warm_rain", "progn", "is_cammgmp_used", "f_ice_phy", "mp_physics"]


_AER4 = ["tauaer3d_sw", "ssaaer3d_sw", "asyaer3d_sw"]   # (14, nj, nk_mem, ni) == Fortran (i,k,j,band)
_AEROD = ["aerod"]                                      # (6, nj, nk_mem, ni) == Fortran (i,k,j,type), aer_opt = 1
_SSIB = ["alswvisdir", "alswvisdif", "alswnirdir", "alswnirdif"]   # 2-D spectral albedos of sf_surface_physics = 8
_CAMRAD = ["lradius", "iradius"]                        # 3-D effective radii (um) of the CAM-MG microphysics


class SwradArgs(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("ims", "ime", "jms", "jme", "kms", "kme",
                                         "its", "ite", "jts", "jte", "kts", "kte", "real_bytes", "mem")]
                + [("stream", C.c_void_p), ("solcon", C.c_double)]
                + [(n, C.c_int) for n in ("aer_ra_feedback", "clean_atm_diag", "icloud", "o3input")]
                + [(n, C.c_void_p) for n in _IN2 + _IN3 + _OUT3 + _OUT2 + _OUT3F]
                + [(n, C.c_void_p) for n in _CLD3 + _CLD2]
                + [(n, C.c_int) for n in _CLDFLAGS]
                + [("g", C.c_double)]
                + [(n, C.c_void_p) for n in _AER4 + _AEROD + _SSIB + _CAMRAD])


_lib = None


def load_library(path: str | None = None):
    """Load librrtmg_hip.so; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise ImportError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the engine has no CPU fallback)")
        try:
            # torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1; load it first so that
            # the engine binds to the SAME HIP runtime instance (two runtimes in one process cannot
            # both own the device, and torch tensors must be visible to our stream).
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(p)
        lib.rrtmg_hip_version.restype = C.c_char_p
        lib.rrtmg_hip_sw_init.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_char_p]
        lib.rrtmg_hip_swrad.argtypes = [C.POINTER(SwradArgs), C.c_char_p]
        lib.rrtmg_hip_sw_status.argtypes = [C.c_char_p]
        lib.rrtmg_hip_swrad_staging_bytes.argtypes = [C.POINTER(SwradArgs)]
        lib.rrtmg_hip_swrad_staging_bytes.restype = C.c_longlong
        lib.rrtmg_hip_sw_reduce_tables.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_void_p, C.c_longlong,
                                                   C.POINTER(C.c_longlong), C.c_void_p, C.c_char_p]
        lib.rrtmg_hip_sw_table_info.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p]
        _lib = lib
    return _lib


EXPORTS = ["rrtmg_hip_sw_init", "rrtmg_hip_swrad", "rrtmg_hip_sw_status", "rrtmg_hip_swrad_staging_bytes",
           "rrtmg_hip_sw_tile_sums", "rrtmg_hip_finalize", "rrtmg_hip_unregister_host", "rrtmg_hip_pick_device",
           "rrtmg_hip_version",
           "rrtmg_hip_sw_reduce_tables", "rrtmg_hip_sw_table_info"]


def _check(rc, err):
    if rc != 0:
        raise WrfFatal(err.value.decode(errors="replace"))


def rrtmg_swinit(data_path: str, data_real_bytes: int = 8, cpdair: float = CP, device: int = 0):
    """rrtmg_swinit (SW:11211): read RRTMG_SW_DATA, reduce g-points, upload tables."""
    err = C.create_string_buffer(ERRLEN)
    _check(load_library().rrtmg_hip_sw_init(data_path.encode(), data_real_bytes, cpdair, device, err), err)


def _ptr(a):
    """(address, is_device, itemsize) of a numpy array or torch tensor."""
    if a is None:
        return None, None, None
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("arrays must be contiguous in WRF memory order")
        return a.ctypes.data, False, a.dtype.itemsize
    # torch tensor
    if not a.is_contiguous():
        raise ValueError("tensors must be contiguous in WRF memory order")
    return a.data_ptr(), a.is_cuda, a.element_size()


def rrtmg_swrad(*, ims, ime, jms, jme, kms, kme, its, ite, jts, jte, kts, kte,
                solcon, aer_ra_feedback=0, clean_atm_diag=0, icloud=0, o3input=0, stream=None,
                has_reqc=0, has_reqi=0, has_reqs=0, warm_rain=False, progn=0, is_cammgmp_used=False, f_ice_phy=False,
                mp_physics=0, g=9.81, **arrays):
    """RRTMG_SWRAD (SW:9901-11207).  Keyword names are the reference's dummy
    argument names in lower case; arrays absent from ``arrays`` are "not
    PRESENT".  3-D arrays have shape (nj, nk_mem, ni) == Fortran (ims:ime,kms:kme,jms:jme)."""
    lib = load_library()
    a = SwradArgs()
    for n, v in dict(ims=ims, ime=ime, jms=jms, jme=jme, kms=kms, kme=kme, its=its, ite=ite, jts=jts, jte=jte,
                     kts=kts, kte=kte).items():
        setattr(a, n, int(v))
    a.solcon = float(solcon)
    a.aer_ra_feedback = int(aer_ra_feedback)
    a.clean_atm_diag = int(clean_atm_diag)
    a.icloud = int(icloud)
    a.o3input = int(o3input)
    for n, v in dict(has_reqc=has_reqc, has_reqi=has_reqi, has_reqs=has_reqs, warm_rain=warm_rain, progn=progn,
                     is_cammgmp_used=is_cammgmp_used, f_ice_phy=f_ice_phy, mp_physics=mp_physics).items():
        setattr(a, n, int(v))
    a.g = float(g)
    known = set(_IN2 + _IN3 + _OUT3 + _OUT2 + _OUT3F + _CLD3 + _CLD2 + _AER4 + _AEROD + _SSIB + _CAMRAD)
    unknown = set(arrays) - known
    if unknown:
        raise TypeError(f"unknown RRTMG_SWRAD arguments: {sorted(unknown)}")
    dev = None
    size = None
    for n, v in arrays.items():
        p, is_dev, isz = _ptr(v)
        if p is None:
            continue
        if dev is None:
            dev, size = is_dev, isz
        elif dev != is_dev or size != isz:
            raise ValueError("all arrays must live in the same memory space and have the same REAL size")
        setattr(a, n, p)
    a.real_bytes = size
    a.mem = 1 if dev else 0
    if dev:
        import torch
        a.stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    err = C.create_string_buffer(ERRLEN)
    _check(lib.rrtmg_hip_swrad(C.byref(a), err), err)


def unregister_host(*arrays):
    """Drop numpy arrays from the RRTMG_HIP_REGISTER_HOST registry before they are freed (rrtmg_hip_unregister_host)."""
    lib = load_library()
    lib.rrtmg_hip_unregister_host.argtypes = [C.c_void_p]
    for a in arrays:
        if isinstance(a, np.ndarray):
            lib.rrtmg_hip_unregister_host(C.c_void_p(a.ctypes.data))


def sw_status():
    err = C.create_string_buffer(ERRLEN)
    _check(load_library().rrtmg_hip_sw_status(err), err)


def sw_reduce_tables(data_path, real_bytes=8, cpdair=CP):
    """Host-only: packed reduced tables + exp table (no GPU needed)."""
    lib = load_library()
    err = C.create_string_buffer(ERRLEN)
    n = C.c_longlong(0)
    _check(lib.rrtmg_hip_sw_reduce_tables(data_path.encode(), real_bytes, cpdair, None, 0, C.byref(n), None, err), err)
    k = np.zeros(n.value)
    e = np.zeros(10001)
    _check(lib.rrtmg_hip_sw_reduce_tables(data_path.encode(), real_bytes, cpdair, k.ctypes.data, n.value,
                                          C.byref(n), e.ctypes.data, err), err)
    return k, e


def sw_table_info(data_path, g, real_bytes=8):
    lib = load_library()
    err = C.create_string_buffer(ERRLEN)
    f = np.zeros(16, dtype=np.int32)
    d = np.zeros(10)
    _check(lib.rrtmg_hip_sw_table_info(data_path.encode(), real_bytes, g, f.ctypes.data, d.ctypes.data, err), err)
    names = ["band", "ig", "ng", "off_major_lo", "off_major_up", "off_self", "off_for", "off_sflux", "off_rayla",
             "nsf", "layreffr", "solar_kind", "uvband", "cfg_lo", "cfg_up"]
    out = dict(zip(names, f.tolist()))
    out.update(dict(zip(["strrat_lo", "strrat_up", "mfac_lo", "mfac_up", "minor_lo", "minor_up", "rayl_lo",
                         "rayl_up", "scalekur"], d.tolist())))
    return out
