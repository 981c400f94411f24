"""Child process of test_host_chunks_gpu.py.  The parent sets RRTMG_HIP_CHUNK_COLS small, so that every mem = 0 call
below is cut into many j-chunks (csrc/tile_staging.h: up to 16, each staged, computed and fetched on its own); each
result must equal the mem = 1 call on the same tile bit for bit.  Prints CHUNKS-OK when all of them do."""
import importlib
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [here, os.path.dirname(here)]


def same(a, b, what):
    for n in a:
        np.testing.assert_array_equal(a[n], b[n], err_msg=f"{what}: {n}")


def main(tmp):
    wa = importlib.import_module("wrfchem-arc-interactions_amd")
    from helpers import run_hip_swrad
    from helpers_lw import run_hip_lwrad
    from helpers_optics import run_hip_optics
    assert int(os.environ["RRTMG_HIP_CHUNK_COLS"]) < 64
    st = wa.synth.make_wrf_state(9, 23, 40, seed=20261100, night_fraction=0.1)   # 23 rows: 16 chunks of 1 or 2 rows
    cloud = wa.synth_clouds.add_clouds(st, seed=20261101, cloudy_fraction=0.5)
    sw_tab, lw_tab = os.path.join(tmp, "RRTMG_SW_DATA"), os.path.join(tmp, "RRTMG_LW_DATA")
    wa.synth.write_sw_data(sw_tab, seed=20260001, real_bytes=8)
    wa.synth.write_lw_data(lw_tab, seed=20260003, real_bytes=8)
    wa.engine.rrtmg_swinit(sw_tab, device=0)
    wa.engine_lw.rrtmg_lwinit(5000.0, lw_tab, kme=st.nk + 1, device=0)
    tile = dict(its=2, ite=8, jts=3, jte=21)
    for kw in (dict(), dict(cloud=cloud), dict(cloud=cloud, tile=tile, profiles=False), dict(tile=tile, clean=0)):
        same(run_hip_swrad(wa.engine, st, device=True, **kw), run_hip_swrad(wa.engine, st, device=False, **kw), f"sw {kw.keys()}")
        same(run_hip_lwrad(wa.engine_lw, st, device=True, **kw), run_hip_lwrad(wa.engine_lw, st, device=False, **kw), f"lw {kw.keys()}")
    wa.engine_optics.optics_init(4, device=0)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20261102)
    for kw in (dict(), dict(tile=dict(tile, kts=2, kte=38))):
        same(run_hip_optics(wa, st, ms, device=True, **kw), run_hip_optics(wa, st, ms, device=False, **kw), f"optics {kw.keys()}")
    print("CHUNKS-OK")


if __name__ == "__main__":
    main(sys.argv[1])
