"""ctypes access to the longwave half of the CPU restatement -- TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .oracle import lib as _baselib, dp, _p, CLOUD2, CLOUD3

LW_COL_OUT = ["uflx", "dflx", "hr", "uflxc", "dflxc", "hrc", "uflxcln", "dflxcln"]
LW_COL_IN = ["play", "plev", "tlay", "tlev", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr",
             "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr", "emis", "tauaer"]


class LwColumnIn(C.Structure):
    _fields_ = [("nlay", C.c_int), ("play", dp), ("plev", dp), ("tlay", dp), ("tlev", dp), ("tsfc", C.c_double),
                ("h2ovmr", dp), ("o3vmr", dp), ("co2vmr", dp), ("ch4vmr", dp), ("n2ovmr", dp), ("o2vmr", dp),
                ("cfc11vmr", dp), ("cfc12vmr", dp), ("cfc22vmr", dp), ("ccl4vmr", dp), ("emis", dp),
                ("tauaer", dp), ("clean_atm_diag", C.c_int),
                ("cldfrac", dp), ("clwp", dp), ("ciwp", dp), ("cswp", dp), ("rel", dp), ("rei", dp), ("res", dp),
                ("inflglw", C.c_int), ("iceflglw", C.c_int), ("liqflglw", C.c_int), ("seed_real4", C.c_int),
                ("cldfmc_out", dp)]


class LwColumnOut(C.Structure):
    _fields_ = [(n, dp) for n in LW_COL_OUT]


class LwTileIn(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("ims", "ime", "jms", "jme", "kms", "kme",
                                         "its", "ite", "jts", "jte", "kts", "kte", "nlayers")]
                + [(n, dp) for n in ("emiss", "tsk", "t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d")]
                + [("tauaerlw", dp * 16), ("aer_ra_feedback", C.c_int), ("clean_atm_diag", C.c_int)]
                + [("icloud", C.c_int)]
                + [(n, dp) for n in CLOUD3 + CLOUD2]
                + [(n, C.c_int) for n in ("has_reqc", "has_reqi", "has_reqs", "warm_rain", "progn")]
                + [("g", C.c_double), ("seed_real4", C.c_int)]
                + [("f_ice_phy", C.c_int), ("fer_hires", C.c_int), ("iradius", dp), ("lradius", dp)])


LW_TILE_OUT = ["rthratenlw", "glw", "olr", "lwcf",
               "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
               "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln",
               "lwupflx", "lwupflxc", "lwupflxcln", "lwdnflx", "lwdnflxc", "lwdnflxcln"]


class LwTileOut(C.Structure):
    _fields_ = [(n, dp) for n in LW_TILE_OUT]


def _lib():
    l = _baselib()
    if not getattr(l, "_lw_ready", False):
        l.lw_oracle_load.restype = C.c_void_p
        l.lw_oracle_load.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_char_p, C.c_int]
        l.lw_oracle_free.argtypes = [C.c_void_p]
        l.lw_oracle_table.restype = dp
        l.lw_oracle_table.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        l.lw_oracle_rrtmg_lw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.lw_oracle_lwrad_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        l.lw_oracle_nlayers.argtypes = [C.c_int, C.c_double]
        l._lw_ready = True
    return l


def nlayers(kme, p_top_pa):
    return _lib().lw_oracle_nlayers(int(kme), float(p_top_pa))


class LwOracle:
    CP = 7.0 * 287.0 / 2.0

    def __init__(self, table_path, real_bytes=8, cpdair=None):
        err = C.create_string_buffer(256)
        self.h = _lib().lw_oracle_load(table_path.encode(), real_bytes, cpdair or self.CP, err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def table(self, band, name):
        n = C.c_int(0)
        p = _lib().lw_oracle_table(self.h, band, name.encode(), C.byref(n))
        if not p or n.value == 0:
            return None
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def rrtmg_lw(self, cols: dict, clean=1) -> dict:
        ncol, nlay = cols["play"].shape
        out = {n: np.zeros((ncol, nlay if n in ("hr", "hrc") else nlay + 1)) for n in LW_COL_OUT}
        for ic in range(ncol):
            keep = {k: np.ascontiguousarray(cols[k][ic], dtype=np.float64) for k in LW_COL_IN}
            ci = LwColumnIn(nlay, _p(keep["play"]), _p(keep["plev"]), _p(keep["tlay"]), _p(keep["tlev"]),
                            float(cols["tsfc"][ic]), *[_p(keep[k]) for k in LW_COL_IN[4:]], clean)
            if "cldfrac" in cols:   # in-cloud state for the McICA path (see rrtmg_oracle.h)
                ck = {k: np.ascontiguousarray(cols[k][ic], dtype=np.float64)
                      for k in ("cldfrac", "clwp", "ciwp", "cswp", "rel", "rei", "res")}
                keep.update(ck)
                for k in ck:
                    setattr(ci, k, _p(ck[k]))
                fl = cols["cldflags"][ic] if np.ndim(cols["cldflags"]) > 1 else cols["cldflags"]
                ci.inflglw, ci.iceflglw, ci.liqflglw = int(fl[0]), int(fl[1]), int(fl[2])
                ci.seed_real4 = int(cols.get("seed_real4", 0))
                keep["mask"] = np.zeros((140, nlay))
                ci.cldfmc_out = _p(keep["mask"])
                out.setdefault("cldfmc", np.zeros((ncol, 140, nlay)))
            rows = [np.zeros(out[n].shape[1]) for n in LW_COL_OUT]
            co = LwColumnOut(*[_p(r) for r in rows])
            rc = _lib().lw_oracle_rrtmg_lw(self.h, C.byref(ci), C.byref(co))
            if rc:
                raise RuntimeError(f"lw_oracle_rrtmg_lw rc={rc}")
            for n, r in zip(LW_COL_OUT, rows):
                out[n][ic] = r
            if "cldfrac" in cols:
                out["cldfmc"][ic] = keep["mask"]
        return out


def lwrad_tile(orc: LwOracle, dims: dict, nlay: int, arrays: dict, aer_ra_feedback, clean_atm_diag, outputs: dict,
               cloud: dict = None):
    ti = LwTileIn()
    for n in ("ims", "ime", "jms", "jme", "kms", "kme", "its", "ite", "jts", "jte", "kts", "kte"):
        setattr(ti, n, int(dims[n]))
    ti.nlayers = int(nlay)
    keep = []
    for n in ("emiss", "tsk", "t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d"):
        a = np.ascontiguousarray(arrays[n], dtype=np.float64)
        keep.append(a)
        setattr(ti, n, _p(a))
    for b in range(16):
        a = np.ascontiguousarray(arrays[f"tauaerlw{b + 1}"], dtype=np.float64)
        keep.append(a)
        ti.tauaerlw[b] = _p(a)
    ti.aer_ra_feedback = int(aer_ra_feedback)
    ti.clean_atm_diag = int(clean_atm_diag)
    ti.g = 9.81
    if cloud:
        for n in ("icloud", "has_reqc", "has_reqi", "has_reqs", "warm_rain", "progn", "seed_real4", "f_ice_phy"):
            setattr(ti, n, int(cloud.get(n, 0)))
        ti.fer_hires = int(cloud.get("mp_physics", 0) in (5, 15))
        ti.g = float(cloud.get("g", 9.81))
        for n in CLOUD3 + CLOUD2 + ["iradius", "lradius"]:
            if arrays.get(n) is not None:
                a = np.ascontiguousarray(arrays[n], dtype=np.float64)
                keep.append(a)
                setattr(ti, n, _p(a))
    to = LwTileOut()
    for n in LW_TILE_OUT:
        if outputs.get(n) is not None:
            assert outputs[n].dtype == np.float64 and outputs[n].flags["C_CONTIGUOUS"]
            setattr(to, n, _p(outputs[n]))
    err = C.create_string_buffer(256)
    rc = _lib().lw_oracle_lwrad_tile(orc.h, C.byref(ti), C.byref(to), err, 256)
    if rc:
        raise RuntimeError(err.value.decode() or f"lw_oracle_lwrad_tile rc={rc}")
