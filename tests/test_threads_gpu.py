"""Host threads calling the engine at the same time (the OpenMP tile loop of module_radiation_driver.F:975-982
with numtiles > 1): four threads work off the four quarters of a patch on one stream, tiles of unequal size so the
per-stream scratch buffers are resized in between; every quarter must equal the single-call result bit for bit."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from helpers import run_hip_swrad
from helpers_lw import run_hip_lwrad
from test_decomposition_gpu import _check

pytestmark = pytest.mark.gpu
P_TOP = 5000.0


def _uneven(st):
    ia, ja = st.ni // 3, st.nj // 4
    return [dict(its=1, ite=ia, jts=1, jte=ja), dict(its=ia + 1, ite=st.ni, jts=1, jte=ja),
            dict(its=1, ite=ia, jts=ja + 1, jte=st.nj), dict(its=ia + 1, ite=st.ni, jts=ja + 1, jte=st.nj)]


def test_concurrent_tiles(hip_lib, wa, sw_table_file, tmp_path):
    st = wa.synth.make_wrf_state(17, 11, 40, seed=20261000, night_fraction=0.1)
    cloud = wa.synth_clouds.add_clouds(st, seed=20261001, cloudy_fraction=0.4)
    tiles = _uneven(st)
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    tabl = str(tmp_path / "RRTMG_LW_DATA")
    wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
    wa.engine_lw.rrtmg_lwinit(P_TOP, tabl, kme=st.nk + 1, device=0)
    whole_sw = run_hip_swrad(wa.engine, st, cloud=cloud, device=False)
    whole_lw = run_hip_lwrad(wa.engine_lw, st, cloud=cloud, device=False)

    def work(t):
        return (run_hip_swrad(wa.engine, st, cloud=cloud, tile=t, device=False),
                run_hip_lwrad(wa.engine_lw, st, cloud=cloud, tile=t, device=False))

    with ThreadPoolExecutor(max_workers=4) as pool:
        for _ in range(6):
            # small tiles first, so later calls grow the scratch buffers while others are in flight
            res = list(pool.map(work, tiles))
            _check(whole_sw, [r[0] for r in res], tiles)
            _check(whole_lw, [r[1] for r in res], tiles)


def test_concurrent_tiles_share_the_output_arrays(hip_lib, wa, sw_table_file, tmp_path):
    """What the OpenMP tile loop of the driver does: every thread passes the SAME patch arrays and its own tile.  A
    call may touch its tile only (a full-array write-back would overwrite the neighbours' results with stale data),
    and night columns / untouched fields keep what the arrays held."""
    st = wa.synth.make_wrf_state(19, 13, 40, seed=20261010, night_fraction=0.15)
    tiles = _uneven(st)
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    tabl = str(tmp_path / "RRTMG_LW_DATA")
    wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
    wa.engine_lw.rrtmg_lwinit(P_TOP, tabl, kme=st.nk + 1, device=0)
    whole_sw = run_hip_swrad(wa.engine, st, device=False, profiles=False)
    whole_lw = run_hip_lwrad(wa.engine_lw, st, device=False, profiles=False)
    from helpers import AER, IN2, IN3, alloc_outputs, dims_of
    import helpers_lw as hl
    for _ in range(4):
        sw_out = {n: np.ascontiguousarray(v) for n, v in alloc_outputs(st, profiles=False).items()}
        lw_out = {n: np.ascontiguousarray(v) for n, v in hl.alloc_outputs(st, profiles=False).items()}
        sw_in = {n: np.ascontiguousarray(st.a2[n]) for n in IN2}
        sw_in.update({n: np.ascontiguousarray(st.a3[n]) for n in IN3 + AER})
        lw_in = {n: np.ascontiguousarray(st.a2[n]) for n in hl.IN2}
        lw_in.update({n: sw_in[n] for n in hl.IN3})
        lw_in.update({n: np.ascontiguousarray(st.a3[n]) for n in hl.AER})

        def work(t):
            d = dims_of(st, t)
            wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **sw_in, **sw_out)
            wa.engine_lw.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **lw_in, **lw_out)

        with ThreadPoolExecutor(max_workers=4) as pool:
            list(pool.map(work, tiles))
        for n in sw_out:
            np.testing.assert_array_equal(sw_out[n], whole_sw[n], err_msg=n)
        for n in lw_out:
            np.testing.assert_array_equal(lw_out[n], whole_lw[n], err_msg=n)


def test_fatal_goes_to_the_thread_that_raised_it(hip_lib, wa, sw_table_file):
    """Four concurrent host-pointer calls, one of them on a tile with a negative column aerosol optical depth: exactly
    that call returns the reference's fatal (SW:11031-11033), the other three succeed -- each call has its own status
    word."""
    st = wa.synth.make_wrf_state(16, 12, 40, seed=20261020)
    tiles = _uneven(st)
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    bad = tiles[2]
    st.a3["tauaer400"][bad["jts"] - 1 + 1, :, bad["its"] - 1 + 1] = -0.5
    ok = run_hip_swrad(wa.engine, st, tile=tiles[0], device=False)

    def work(t):
        try:
            run_hip_swrad(wa.engine, st, tile=t, device=False)
            return None
        except wa.engine.WrfFatal as e:
            return str(e)

    for _ in range(5):
        with ThreadPoolExecutor(max_workers=4) as pool:
            res = list(pool.map(work, tiles))
        assert [r is not None for r in res] == [False, False, True, False], res
        assert "Negative total optical depth" in res[2]
        i_bad, j_bad = bad["its"] + 1, bad["jts"] + 1
        assert f"{i_bad:5d}{j_bad:5d}" in res[2]
