"""ctypes access to the CPU restatement (oracle/liboracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBNAME = "liboracle.so"    # oracle/flopcount.py swaps in the operation-counting build for a moment
dp = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, _LIBNAME)
        if not os.path.exists(path):
            if _LIBNAME == "liboracle.so":
                build()
            else:
                subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "count")])
        _LIB = C.CDLL(path)
        _LIB.sw_oracle_load.restype = C.c_void_p
        _LIB.sw_oracle_load.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_char_p, C.c_int]
        _LIB.sw_oracle_free.argtypes = [C.c_void_p]
        _LIB.sw_oracle_table.restype = dp
        _LIB.sw_oracle_table.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        _LIB.sw_oracle_rrtmg_sw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.sw_oracle_swrad_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    return _LIB


class SwColumnIn(C.Structure):
    _fields_ = [("nlay", C.c_int),
                ("play", dp), ("plev", dp), ("tlay", dp), ("tlev", dp), ("tsfc", C.c_double),
                ("h2ovmr", dp), ("o3vmr", dp), ("co2vmr", dp), ("ch4vmr", dp), ("n2ovmr", dp), ("o2vmr", dp),
                ("asdir", C.c_double), ("asdif", C.c_double), ("aldir", C.c_double), ("aldif", C.c_double),
                ("coszen", C.c_double), ("adjes", C.c_double), ("scon", C.c_double), ("dyofyr", C.c_int),
                ("tauaer", dp), ("ssaaer", dp), ("asmaer", dp), ("clean_atm_diag", C.c_int),
                ("cldfrac", dp), ("clwp", dp), ("ciwp", dp), ("cswp", dp), ("rel", dp), ("rei", dp), ("res", dp),
                ("inflgsw", C.c_int), ("iceflgsw", C.c_int), ("liqflgsw", C.c_int), ("seed_real4", C.c_int),
                ("cldfmc_out", dp)]


SW_COL_OUT = ["swuflx", "swdflx", "swhr", "swuflxc", "swdflxc", "swhrc", "swuflxcln", "swdflxcln",
              "sibvisdir", "sibvisdif", "sibnirdir", "sibnirdif", "swdkdir", "swdkdif"]


class SwColumnOut(C.Structure):
    _fields_ = [(n, dp) for n in SW_COL_OUT]


def _p(a):
    return a.ctypes.data_as(dp)


class SwOracle:
    CP = 7.0 * 287.0 / 2.0

    def __init__(self, table_path, real_bytes=8, cpdair=None):
        err = C.create_string_buffer(256)
        self.h = lib().sw_oracle_load(table_path.encode(), real_bytes, cpdair or self.CP, err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def table(self, band, name):
        n = C.c_int(0)
        p = lib().sw_oracle_table(self.h, band, name.encode(), C.byref(n))
        if not p or n.value == 0:
            return None
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def rrtmg_sw(self, cols: dict, clean=1) -> dict:
        """cols: dict of arrays with leading dim ncol (see tools/make_golden.py)."""
        ncol, nlay = cols["play"].shape
        out = {n: np.zeros((ncol, nlay if n in ("swhr", "swhrc") else nlay + 1)) for n in SW_COL_OUT}
        for ic in range(ncol):
            keep = {k: np.ascontiguousarray(cols[k][ic], dtype=np.float64)
                    for k in ("play", "plev", "tlay", "tlev", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr",
                              "o2vmr", "tauaer", "ssaaer", "asmaer")}
            s = cols["scal"][ic]
            ci = SwColumnIn(nlay, _p(keep["play"]), _p(keep["plev"]), _p(keep["tlay"]), _p(keep["tlev"]),
                            float(cols["tsfc"][ic]), _p(keep["h2ovmr"]), _p(keep["o3vmr"]), _p(keep["co2vmr"]),
                            _p(keep["ch4vmr"]), _p(keep["n2ovmr"]), _p(keep["o2vmr"]),
                            s[0], s[1], s[2], s[3], s[4], s[5], s[6], 0,
                            _p(keep["tauaer"]), _p(keep["ssaaer"]), _p(keep["asmaer"]), clean)
            if "cldfrac" in cols:   # in-cloud state for the McICA path (see rrtmg_oracle.h)
                ck = {k: np.ascontiguousarray(cols[k][ic], dtype=np.float64)
                      for k in ("cldfrac", "clwp", "ciwp", "cswp", "rel", "rei", "res")}
                keep.update(ck)
                for k in ck:
                    setattr(ci, k, _p(ck[k]))
                fl = cols["cldflags"][ic] if np.ndim(cols["cldflags"]) > 1 else cols["cldflags"]
                ci.inflgsw, ci.iceflgsw, ci.liqflgsw = int(fl[0]), int(fl[1]), int(fl[2])
                ci.seed_real4 = int(cols.get("seed_real4", 0))
                mask = np.zeros((112, nlay))
                keep["mask"] = mask
                ci.cldfmc_out = _p(mask)
                out.setdefault("cldfmc", np.zeros((ncol, 112, nlay)))
            rows = [np.zeros(out[n].shape[1]) for n in SW_COL_OUT]
            co = SwColumnOut(*[_p(r) for r in rows])
            rc = lib().sw_oracle_rrtmg_sw(self.h, C.byref(ci), C.byref(co))
            if rc:
                raise RuntimeError(f"sw_oracle_rrtmg_sw rc={rc}")
            for n, r in zip(SW_COL_OUT, rows):
                out[n][ic] = r
            if "cldfrac" in cols:
                out["cldfmc"][ic] = keep["mask"]
        return out

    def close(self):
        if self.h:
            lib().sw_oracle_free(self.h)
            self.h = None


CLOUD3 = ["cldfra3d", "qc3d", "qr3d", "qi3d", "qs3d", "re_cloud", "re_ice", "re_snow", "qndrop3d"]
CLOUD2 = ["xland", "xice", "snow"]


class SwTileIn(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("ims", "ime", "jms", "jme", "kms", "kme",
                                         "its", "ite", "jts", "jte", "kts", "kte")]
                + [(n, dp) for n in ("xcoszen", "albedo", "tsk", "t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d",
                                     "tauaer300", "tauaer400", "tauaer600", "tauaer999",
                                     "gaer300", "gaer400", "gaer600", "gaer999",
                                     "waer300", "waer400", "waer600", "waer999")]
                + [("solcon", C.c_double), ("aer_ra_feedback", C.c_int), ("clean_atm_diag", C.c_int)]
                + [("icloud", C.c_int)]
                + [(n, dp) for n in CLOUD3 + CLOUD2]
                + [(n, C.c_int) for n in ("has_reqc", "has_reqi", "has_reqs", "warm_rain", "progn")]
                + [("g", C.c_double), ("seed_real4", C.c_int)]
                + [(n, dp) for n in ("tauaer3d_sw", "ssaaer3d_sw", "asyaer3d_sw", "aerod",
                                     "alswvisdir", "alswvisdif", "alswnirdir", "alswnirdif")]
                + [("f_ice_phy", C.c_int), ("fer_hires", C.c_int), ("iradius", dp), ("lradius", dp)])


SW_TILE_OUT = ["rthratensw", "gsw", "swcf", "coszr",
               "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
               "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln",
               "swvisdir", "swvisdif", "swnirdir", "swnirdif", "swddir", "swddni", "swddif",
               "swupflx", "swupflxc", "swupflxcln", "swdnflx", "swdnflxc", "swdnflxcln"]


class SwTileOut(C.Structure):
    _fields_ = [(n, dp) for n in SW_TILE_OUT]


def ecaer_bands(ecaer):
    """iaer = 6 (SW:9313-9340): ecaer (6, nlay) -> tauaer, ssaaer, asmaer (14, nlay)."""
    e = np.ascontiguousarray(ecaer, dtype=np.float64)
    nlay = e.shape[1]
    out = [np.zeros((14, nlay)) for _ in range(3)]
    l = lib()
    l.sw_oracle_ecaer_bands.argtypes = [C.c_int, dp, dp, dp, dp]
    l.sw_oracle_ecaer_bands(nlay, _p(e), *[_p(o) for o in out])
    return out


def swrad_tile(orc: "SwOracle", dims: dict, arrays: dict, solcon, aer_ra_feedback, clean_atm_diag, outputs: dict,
               cloud: dict = None):
    """RRTMG_SWRAD restatement on a tile; arrays/outputs are float64 numpy in WRF order.  cloud: icloud,
    has_reqc/i/s, warm_rain, progn, g, seed_real4 (the cloud arrays themselves travel in `arrays`)."""
    ti = SwTileIn()
    for n in ("ims", "ime", "jms", "jme", "kms", "kme", "its", "ite", "jts", "jte", "kts", "kte"):
        setattr(ti, n, int(dims[n]))
    keep = []
    for n, _t in SwTileIn._fields_:
        if _t is dp and arrays.get(n) is not None:
            a = np.ascontiguousarray(arrays[n], dtype=np.float64)
            keep.append(a)
            setattr(ti, n, _p(a))
    ti.solcon = float(solcon)
    ti.aer_ra_feedback = int(aer_ra_feedback)
    ti.clean_atm_diag = int(clean_atm_diag)
    ti.g = 9.81
    if cloud:
        for n in ("icloud", "has_reqc", "has_reqi", "has_reqs", "warm_rain", "progn", "seed_real4", "f_ice_phy"):
            setattr(ti, n, int(cloud.get(n, 0)))
        ti.fer_hires = int(cloud.get("mp_physics", 0) in (5, 15))
        ti.g = float(cloud.get("g", 9.81))
    to = SwTileOut()
    for n in SW_TILE_OUT:
        if outputs.get(n) is not None:
            assert outputs[n].dtype == np.float64 and outputs[n].flags["C_CONTIGUOUS"]
            setattr(to, n, _p(outputs[n]))
    err = C.create_string_buffer(256)
    rc = lib().sw_oracle_swrad_tile(orc.h, C.byref(ti), C.byref(to), err, 256)
    if rc:
        raise RuntimeError(err.value.decode() or f"sw_oracle_swrad_tile rc={rc}")
