"""Patch decomposition must not change a single bit (WRF users compare runs across processor counts): the whole
tile in one call against the same tile worked off as four sub-tiles, for the optical averaging, RRTMG_SWRAD and
RRTMG_LWRAD with clouds."""
import numpy as np
import pytest

from helpers import run_hip_swrad
from helpers_lw import run_hip_lwrad
from helpers_optics import run_hip_optics

pytestmark = pytest.mark.gpu
P_TOP = 5000.0


def _quarters(st):
    ih, jh = st.ni // 2, st.nj // 2
    return [dict(its=1, ite=ih, jts=1, jte=jh), dict(its=ih + 1, ite=st.ni, jts=1, jte=jh),
            dict(its=1, ite=ih, jts=jh + 1, jte=st.nj), dict(its=ih + 1, ite=st.ni, jts=jh + 1, jte=st.nj)]


def _inside(shape, t):
    m = np.zeros(shape, bool)
    if len(shape) == 3:
        m[t["jts"] - 1:t["jte"], :, t["its"] - 1:t["ite"]] = True
    else:
        m[t["jts"] - 1:t["jte"], t["its"] - 1:t["ite"]] = True
    return m


def _check(whole, parts, tiles):
    for n, w in whole.items():
        for p, t in zip(parts, tiles):
            m = _inside(w.shape, t)
            np.testing.assert_array_equal(p[n][m], w[m], err_msg=n)


def test_decomposition_invariance(hip_lib, wa, sw_table_file, tmp_path):
    st = wa.synth.make_wrf_state(13, 9, 40, seed=20260900, night_fraction=0.15)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260901, cloudy_fraction=0.5)
    tiles = _quarters(st)
    # shortwave
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    whole = run_hip_swrad(wa.engine, st, cloud=cloud)
    _check(whole, [run_hip_swrad(wa.engine, st, cloud=cloud, tile=t) for t in tiles], tiles)
    # longwave
    tabl = str(tmp_path / "RRTMG_LW_DATA")
    wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
    wa.engine_lw.rrtmg_lwinit(P_TOP, tabl, kme=st.nk + 1, device=0)
    whole = run_hip_lwrad(wa.engine_lw, st, cloud=cloud)
    _check(whole, [run_hip_lwrad(wa.engine_lw, st, cloud=cloud, tile=t) for t in tiles], tiles)
    # optical averaging, both precisions
    wa.engine_optics.optics_init(8, device=0)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260902)
    for prec in ("f64", "mixed"):
        try:
            wa.engine_optics.optics_precision(prec)
            whole = run_hip_optics(wa, st, ms)
            parts = [run_hip_optics(wa, st, ms, tile=t) for t in tiles]
        finally:
            wa.engine_optics.optics_precision("f64")
        nk = st.nk
        for n, w in whole.items():
            for p, t in zip(parts, tiles):
                sl = (slice(t["jts"] - 1, t["jte"]), slice(0, nk), slice(t["its"] - 1, t["ite"]))
                np.testing.assert_array_equal(p[n][sl], w[sl], err_msg=f"{prec} {n}")
