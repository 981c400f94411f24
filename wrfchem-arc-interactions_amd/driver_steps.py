"""Host-side mirror of the driver-side neighbours of the radiation call, forwarding to
librrtmg_hip.so: the AC* energy buckets (module_radiation_driver.F:2308-2381) and the swint_opt = 1 pair
``update_swinterp_parameters`` / ``interp_sw_radiation`` (module_radiation_driver.F:2668-2795).  Function and
argument names are the reference's; arrays are 2-D (nj_m, ni_m) == Fortran (ims:ime, jms:jme), all numpy (host)
or all torch CUDA tensors (device, enqueued on the current stream)."""
from __future__ import annotations

import ctypes as C

from .engine import ERRLEN, _check, _ptr, load_library

EXPORTS = ["rrtmg_hip_accumulate_fluxes", "rrtmg_hip_update_swinterp_parameters", "rrtmg_hip_interp_sw_radiation"]

# the driver's order (DRV:2330-2337, 2365-2372)
LW_BUCKETS = ("lwupt", "lwuptc", "lwdnt", "lwdntc", "lwupb", "lwupbc", "lwdnb", "lwdnbc")
SW_BUCKETS = ("swupt", "swuptc", "swdnt", "swdntc", "swupb", "swupbc", "swdnb", "swdnbc")


class Tile2d(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("ims", "ime", "jms", "jme", "its", "ite", "jts", "jte", "real_bytes", "mem")] + \
               [("stream", C.c_void_p)]


class UpdateArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("coszen", "coszen_loc", "swddir", "swdown", "swddir_ref", "bb", "Bx",
                                          "swdown_ref", "gg", "Gx", "coszen_ref")]


class ApplyArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("coszen_ref", "coszen_loc", "swddir_ref", "bb", "Bx", "swdown_ref", "gg",
                                          "Gx", "albedo", "swdown", "swddir", "swddni", "swddif", "gsw")]


def _lib():
    lib = load_library()
    if not getattr(lib, "_driver_steps_ready", False):
        lib.rrtmg_hip_accumulate_fluxes.argtypes = [C.POINTER(Tile2d), C.c_double, C.POINTER(C.c_void_p),
                                                    C.POINTER(C.c_void_p), C.c_char_p]
        lib.rrtmg_hip_update_swinterp_parameters.argtypes = [C.POINTER(Tile2d), C.POINTER(UpdateArgs), C.c_char_p]
        lib.rrtmg_hip_interp_sw_radiation.argtypes = [C.POINTER(Tile2d), C.POINTER(ApplyArgs), C.c_char_p]
        lib._driver_steps_ready = True
    return lib


def _tile(dims, arrays, stream):
    """Tile2d from ims..jte and the memory space / REAL size the arrays agree on."""
    dev = size = None
    ptrs = {}
    for n, v in arrays.items():
        p, is_dev, isz = _ptr(v)
        ptrs[n] = p
        if p is None:
            continue
        if dev is None:
            dev, size = is_dev, isz
        elif dev != is_dev or size != isz:
            raise ValueError("all arrays must live in the same memory space and have the same REAL size")
    if dev is None:
        raise ValueError("no array passed")
    t = Tile2d()
    for n in ("ims", "ime", "jms", "jme", "its", "ite", "jts", "jte"):
        setattr(t, n, int(dims[n]))
    t.real_bytes, t.mem = int(size), 1 if dev else 0
    if dev:
        import torch
        t.stream = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    return t, ptrs


def accumulate_fluxes(dims, dtaccum, fluxes: dict, buckets: dict, stream=None):
    """ac<name> += <name> * dtaccum for every name present in both dicts (keys: LW_BUCKETS + SW_BUCKETS names;
    ``buckets`` is keyed by the flux name, e.g. buckets['swupt'] is ACSWUPT)."""
    names = [n for n in LW_BUCKETS + SW_BUCKETS if n in fluxes and n in buckets]
    if not names:
        return
    t, _ = _tile(dims, {**{"f_" + n: fluxes[n] for n in names}, **{"a_" + n: buckets[n] for n in names}}, stream)
    fl = (C.c_void_p * 16)()
    ac = (C.c_void_p * 16)()
    for i, n in enumerate(LW_BUCKETS + SW_BUCKETS):
        if n in names:
            fl[i], ac[i] = _ptr(fluxes[n])[0], _ptr(buckets[n])[0]
    err = C.create_string_buffer(ERRLEN)
    _check(_lib().rrtmg_hip_accumulate_fluxes(C.byref(t), float(dtaccum), fl, ac, err), err)


def update_swinterp_parameters(dims, *, coszen, coszen_loc, swddir, swdown, swddir_ref, bb, Bx, swdown_ref, gg, Gx,
                               coszen_ref, stream=None):
    arrays = dict(coszen=coszen, coszen_loc=coszen_loc, swddir=swddir, swdown=swdown, swddir_ref=swddir_ref, bb=bb,
                  Bx=Bx, swdown_ref=swdown_ref, gg=gg, Gx=Gx, coszen_ref=coszen_ref)
    t, ptrs = _tile(dims, arrays, stream)
    a = UpdateArgs(**ptrs)
    err = C.create_string_buffer(ERRLEN)
    _check(_lib().rrtmg_hip_update_swinterp_parameters(C.byref(t), C.byref(a), err), err)


def interp_sw_radiation(dims, *, coszen_ref, coszen_loc, swddir_ref, bb, Bx, swdown_ref, gg, Gx, albedo, swdown, swddir,
                        swddni, swddif, gsw, stream=None):
    arrays = dict(coszen_ref=coszen_ref, coszen_loc=coszen_loc, swddir_ref=swddir_ref, bb=bb, Bx=Bx,
                  swdown_ref=swdown_ref, gg=gg, Gx=Gx, albedo=albedo, swdown=swdown, swddir=swddir, swddni=swddni,
                  swddif=swddif, gsw=gsw)
    t, ptrs = _tile(dims, arrays, stream)
    a = ApplyArgs(**ptrs)
    err = C.create_string_buffer(ERRLEN)
    _check(_lib().rrtmg_hip_interp_sw_radiation(C.byref(t), C.byref(a), err), err)
