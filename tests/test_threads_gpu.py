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
