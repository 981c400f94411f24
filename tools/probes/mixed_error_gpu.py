"""Worst differences of the mixed optics path against the fp64 oracle on the state of tests/test_optics_gpu.py's
mixed-precision test (run on the GPU box: python tools/probes/mixed_error_gpu.py)."""
import importlib
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests")]
wa = importlib.import_module("wrfchem-arc-interactions_amd")
from helpers_optics import LW_NAMES, SW_NAMES, run_hip_optics, run_oracle_optics, tile_mask   # noqa: E402

for nbin, seed in ((8, 20260370), (4, 20260380)):
    wa.engine_optics.optics_init(nbin, device=0)
    st = wa.synth.make_wrf_state(9, 6, 30, seed=seed)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=nbin, seed=seed + 1)
    wa.engine_optics.optics_precision("mixed")
    got = run_hip_optics(wa, st, ms)
    wa.engine_optics.optics_precision("f64")
    orc = run_oracle_optics(wa, st, ms)
    mask = tile_mask(st, ms.alt.shape)
    worst = {}
    for n in SW_NAMES + LW_NAMES:
        g, r = got[n][mask], orc[n][mask]
        if n.startswith(("waer", "gaer")):
            e = float(np.abs(g - r).max())
        else:
            big = r > 1e-10 * r.max()
            e = float((np.abs(g - r)[big] / r[big]).max())
        key = n.rstrip("0123456789")
        worst[key] = max(worst.get(key, 0.0), e)
    print(nbin, {k: f"{v:.1e}" for k, v in worst.items()})
