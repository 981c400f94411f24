"""Exact fp64 operation count of the CPU restatement per column -- TEST INFRASTRUCTURE (bench.py's roofline.fp64).

oracle/count/ builds the oracle's column solvers (RRTMG_SWRAD / RRTMG_LWRAD restatements, rrtmg_sw / rrtmg_lw and
everything below them) as C++ with a counting `double`; this module runs a tile of the bench workload through that
build and reads the counters.  SURVEY 8d: "estimate to be replaced by an exact count from the CPU restatement".
The optical-averaging stage is counted the same way (optics_oracle.c through the same counting type, its complex
arithmetic spelled out in real operations); the stage has no reference arithmetic, so its figure is that of the
engine's own specification as the oracle evaluates it (downward recurrence from the start order for every particle --
the device carries the logarithmic derivative upward where that is accurate and does less).  The two parts are
reported separately: "radiation" is the part the reference defines.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import oracle as orc

NAMES = ("add", "mul", "div", "sqrt", "transcendental")


def _with_count_lib(fn):
    saved = (orc._LIB, orc._LIBNAME)
    orc._LIB, orc._LIBNAME = None, "liboracle_count.so"
    try:
        lib = orc.lib()
        lib.oracle_flops_read.argtypes = [C.POINTER(C.c_ulonglong)]
        return fn(lib)
    finally:
        orc._LIB, orc._LIBNAME = saved


def _read(lib):
    buf = (C.c_ulonglong * 5)()
    lib.oracle_flops_read(buf)
    return np.array(list(buf), dtype=np.float64)


def per_column(wa, tab_sw, tab_lw, st, ms, mode, ncols=0):
    """Operation counts per column of one row of the state (aerosol + clean double call, cloud-free):
    {"flops_per_column": adds + muls + divs + sqrts + transcendentals, each counted once, "sw": {...}, "lw": {...}}."""
    from . import oracle_lw
    ni, nk = st.ni, st.nk
    d = dict(ims=1, ime=ni, jms=1, jme=st.nj, kms=1, kme=nk + 1, its=1, ite=ni, jts=1, jte=1, kts=1, kte=nk)
    sw2 = ["coszr", "gsw", "swcf", "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln", "swupb", "swupbc",
           "swupbcln", "swdnb", "swdnbc", "swdnbcln", "swvisdir", "swvisdif", "swnirdir", "swnirdif", "swddir", "swddni",
           "swddif"]
    lw2 = ["glw", "olr", "lwcf", "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln", "lwupb", "lwupbc",
           "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"]

    def run(lib):
        res = {}
        out = {"rthratensw": np.zeros((st.nj, nk + 1, ni))}
        out.update({n: np.zeros((st.nj, ni)) for n in sw2})
        arrays = {n: st.a2[n] for n in ("xcoszen", "albedo", "tsk")}
        arrays.update({n: st.a3[n] for n in ("t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d")})
        arrays.update({n: st.a3[n] for n in st.a3 if n.startswith(("tauaer", "waer", "gaer")) and "lw" not in n})
        o = orc.SwOracle(tab_sw)
        day = int((st.a2["xcoszen"][0] > 0).sum())
        lib.oracle_flops_reset()
        orc.swrad_tile(o, d, arrays, 1370.0, 1, 1, out)
        sw = _read(lib) / max(day, 1)          # per daylit column (night columns cost nothing)
        res["sw"] = dict(zip(NAMES, sw.tolist()))
        total = sw.sum()
        if mode != "sw":
            nlay = oracle_lw.nlayers(nk + 1, st.p_top)
            outl = {"rthratenlw": np.zeros((st.nj, nk + 1, ni))}
            outl.update({n: np.zeros((st.nj, ni)) for n in lw2})
            arrays = {n: st.a2[n] for n in ("emiss", "tsk")}
            arrays.update({n: st.a3[n] for n in ("t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d")})
            arrays.update({n: st.a3[n] for n in st.a3 if n.startswith("tauaerlw")})
            ol = oracle_lw.LwOracle(tab_lw)
            lib.oracle_flops_reset()
            oracle_lw.lwrad_tile(ol, d, nlay, arrays, 1, 1, outl)
            lw = _read(lib) / ni
            res["lw"] = dict(zip(NAMES, lw.tolist()))
            total += lw.sum()
        res["flops_per_column_radiation"] = float(total)
        counted = ("every add, multiply, divide, square root and transcendental call of the CPU restatement "
                   "(wrapper + rrtmg_sw" + (" + rrtmg_lw" if mode != "sw" else "") + ", aerosol and clean call), "
                   f"one row of {ni} columns")
        if ms is not None and mode == "full":
            from . import oracle_optics as oo
            wl, ref, den = wa.engine_optics.table_info()
            tabs = oo.make_tables(ms.nbin, wa.engine_optics.MOSAIC_DLO_UM, wa.engine_optics.MOSAIC_DHI_UM, wl, ref, den)
            nc = min(ni, ncols) if ncols else ni
            lib.oracle_flops_reset()
            oo.tile(tabs, ms.chem, ms.alt, ms.dz8w, ms.idx, dict(d, ite=nc))
            op = _read(lib) / nc
            res["optics"] = dict(zip(NAMES, op.tolist()))
            res["flops_per_column_optics"] = float(op.sum())
            total += op.sum()
            counted += (f"; optical averaging: the oracle's Mie code (the engine's own specification, no reference "
                        f"arithmetic exists) on {nc} columns, complex operations counted as their real operations")
        else:
            counted += "; no optical-averaging stage in this workload"
        res["flops_per_column"] = float(total)
        res["counted"] = counted
        return res
    return _with_count_lib(run)
