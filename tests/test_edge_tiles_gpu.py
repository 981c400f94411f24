"""Edge cases of the tile loops: an empty tile (ite < its: the reference's i/j loops do not execute, so the call
returns without writing) and an all-night tile (coszen <= 0 everywhere: fluxes zero, heating untouched,
module_ra_rrtmg_sw.F:10230-10240 and :11173-11197) -- the device against the oracle."""
import numpy as np
import pytest

from helpers import SENTINEL, compare_sw, run_hip_swrad, run_oracle_swrad
from helpers_lw import run_hip_lwrad
from helpers_optics import run_hip_optics

pytestmark = pytest.mark.gpu
P_TOP = 5000.0


def test_empty_tile_is_a_no_op(hip_lib, wa, sw_table_file, tmp_path):
    st = wa.synth.make_wrf_state(6, 5, 40, seed=20261010)
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    tabl = str(tmp_path / "RRTMG_LW_DATA")
    wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
    wa.engine_lw.rrtmg_lwinit(P_TOP, tabl, kme=st.nk + 1, device=0)
    wa.engine_optics.optics_init(4, device=0)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20261011)
    for tile in (dict(its=4, ite=3, jts=1, jte=5), dict(its=1, ite=6, jts=3, jte=2)):
        for device in (True, False):
            for got in (run_hip_swrad(wa.engine, st, tile=tile, device=device),
                        run_hip_lwrad(wa.engine_lw, st, tile=tile, device=device)):
                for n, v in got.items():
                    assert (v == SENTINEL).all(), n
        got = run_hip_optics(wa, st, ms, tile=tile)
        for n, v in got.items():
            assert (v == SENTINEL).all(), n


def test_all_night_tile(hip_lib, wa, sw_table_file, sw_oracle):
    st = wa.synth.make_wrf_state(5, 4, 40, seed=20261012, night_fraction=1.1)
    assert (st.a2["xcoszen"] <= 0).all()
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    got = run_hip_swrad(wa.engine, st)
    ref = run_oracle_swrad(sw_oracle, st)
    assert compare_sw(got, ref, 0.0, 0.0) == 0.0
    for n in ("swupt", "swdnb", "swdnbcln", "swddir"):
        assert (got[n] == 0).all(), n
    assert (got["rthratensw"] == SENTINEL).all() and (got["gsw"] == SENTINEL).all()
