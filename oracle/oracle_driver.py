"""ctypes access to the driver-side neighbours of the CPU restatement (driver_oracle.c) -- TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .oracle import lib as _baselib, dp


def _lib():
    l = _baselib()
    if not getattr(l, "_driver_ready", False):
        l.driver_oracle_accumulate.argtypes = [C.c_int, C.c_double, dp, dp]
        l.driver_oracle_swinterp_update.argtypes = [C.c_int, C.c_int] + [dp] * 11
        l.driver_oracle_swinterp_apply.argtypes = [C.c_int, C.c_int] + [dp] * 14
        l._driver_ready = True
    return l


def _p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(dp)


UPDATE_IN = ("coszen", "coszen_loc", "swddir", "swdown")
UPDATE_IO = ("swddir_ref", "bb", "Bx", "swdown_ref", "gg", "Gx", "coszen_ref")
APPLY_IN = ("coszen_ref", "coszen_loc", "swddir_ref", "bb", "Bx", "swdown_ref", "gg", "Gx", "albedo")
APPLY_OUT = ("swdown", "swddir", "swddni", "swddif", "gsw")


def accumulate(flux, acc, dt):
    """acc += flux * dt (in place), module_radiation_driver.F:2330-2337."""
    _lib().driver_oracle_accumulate(flux.size, float(dt), _p(flux), _p(acc))


def swinterp_update(inp, state, real4=False):
    """update_swinterp_parameters on flat fp64 arrays; `state` (UPDATE_IO) is updated in place."""
    n = inp["coszen"].size
    _lib().driver_oracle_swinterp_update(n, int(real4), *[_p(inp[k]) for k in UPDATE_IN], *[_p(state[k]) for k in UPDATE_IO])


def swinterp_apply(state, coszen_loc, albedo, out, real4=False):
    """interp_sw_radiation; `out` (APPLY_OUT) is overwritten."""
    n = coszen_loc.size
    src = dict(state, coszen_loc=coszen_loc, albedo=albedo)
    _lib().driver_oracle_swinterp_apply(n, int(real4), *[_p(src[k]) for k in APPLY_IN], *[_p(out[k]) for k in APPLY_OUT])
