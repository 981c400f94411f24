#!/usr/bin/env python3
"""Times the mem = 0 (host pointer) entry points on a 500 x 40 x 50 tile under the current environment
(RRTMG_HIP_CHUNK_COLS, RRTMG_HIP_PACK_THREADS): development aid for csrc/tile_staging.h."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import wrfchem_arc_amd as wa  # noqa: E402

ni, nj, nk = 500, int(os.environ.get("PROBE_ROWS", "40")), 50
tmp = tempfile.mkdtemp()
tab, tabl = os.path.join(tmp, "SW"), os.path.join(tmp, "LW")
wa.synth.write_sw_data(tab, seed=20260001, real_bytes=8)
wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
wa.engine.rrtmg_swinit(tab, 8, device=0)
wa.engine_lw.rrtmg_lwinit(bench.P_TOP, tabl, kme=nk + 1, device=0)
st = wa.synth.make_wrf_rows(ni, nj, nk, 0, nj, seed=20260002)
d = bench.dims_of(ni, nj, nk)
hin = {n: np.ascontiguousarray(st.a2[n]) for n in bench.SW_IN2}
hin.update({n: np.ascontiguousarray(st.a3[n]) for n in bench.ST_IN3 + bench.SW_AER})
hout = {"rthratensw": np.zeros((nj, nk + 1, ni))}
hout.update({n: np.zeros((nj, ni)) for n in bench.SW_OUT2})
lin = {n: np.ascontiguousarray(st.a2[n]) for n in bench.LW_IN2}
lin.update({n: hin[n] for n in bench.ST_IN3})
lin.update({n: np.ascontiguousarray(st.a3[n]) for n in bench.LW_AER})
lout = {"rthratenlw": np.zeros((nj, nk + 1, ni))}
lout.update({n: np.zeros((nj, ni)) for n in bench.LW_OUT2})


def sw():
    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **hin, **hout)


def lw():
    wa.engine_lw.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **lin, **lout)


for f in (sw, lw):
    f()
res = {}
for name, f in (("sw", sw), ("lw", lw)):
    t = time.perf_counter()
    for _ in range(3):
        f()
    res[name] = (time.perf_counter() - t) / 3
print("chunk_cols=%s pack_threads=%s rows=%d: sw %.2f ms  lw %.2f ms  pair %.2f ms = %.2f M columns/s" % (
    os.environ.get("RRTMG_HIP_CHUNK_COLS", "default"), os.environ.get("RRTMG_HIP_PACK_THREADS", "default"), nj,
    1e3 * res["sw"], 1e3 * res["lw"], 1e3 * (res["sw"] + res["lw"]), ni * nj / (res["sw"] + res["lw"]) / 1e6))
