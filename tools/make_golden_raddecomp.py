#!/usr/bin/env python3
"""Generate tests/golden/raddecomp.npz by running the REFERENCE's forcing decomposition
(/root/reference/analysis_scripts/RadDecomp_analysis_package/RadDecomp_functions.py:118-266: calc_Delta_S, calc_Delta_L,
calc_SW_DIRECT, calc_SW_INDIRECT, calc_SW_SEMIDIRECT, calc_LW_INDIRECT, calc_LW_SEMIDIRECT) on seeded statistics tables.

Runs only in the build container (imports the reference module from where it lies; nothing of it is copied).  The
fixture holds data only: the statistics fed in (24 "hours" x 4 variables x 4 scenarios x {avg, SE, SE_corr}) and the
numbers the reference returned for error_type 'SE' and 'SE_corr'.
"""
import importlib.util
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/analysis_scripts/RadDecomp_analysis_package/RadDecomp_functions.py"
VARS = ["SWUPT", "LWUPT", "LWUPTC", "SWUPTCLN"]
SCEN = ["BASE", "BASE_nA", "ALT", "ALT_nA"]
COLS = ["avg", "SE", "SE_corr"]
FUNCS = ["calc_Delta_S", "calc_Delta_L", "calc_SW_DIRECT", "calc_SW_INDIRECT", "calc_SW_SEMIDIRECT",
         "calc_LW_INDIRECT", "calc_LW_SEMIDIRECT"]


def main():
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("raddecomp_reference", REF)
    radfn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(radfn)
    rng = np.random.default_rng(20260777)
    nh = 24
    # stats[scenario][variable] = (nh, 3): avg, SE, SE_corr
    stats = {}
    for s in SCEN:
        for v in VARS:
            avg = {"SWUPT": 110., "LWUPT": 235., "LWUPTC": 262., "SWUPTCLN": 104.}[v] + rng.normal(0., 6., nh)
            se = rng.uniform(0.05, 0.8, nh)
            stats[(s, v)] = np.stack([avg, se, se * rng.uniform(1.0, 3.5, nh)], axis=1)

    def frame(s, v):
        return pd.DataFrame(stats[(s, v)], columns=COLS, index=pd.Index(np.arange(nh), name="Hour"))

    # the dictionaries the reference's load_Files builds: variable and variable + '_nA'
    base = {v: frame("BASE", v) for v in VARS}
    base.update({v + "_nA": frame("BASE_nA", v) for v in VARS})
    alt = {v: frame("ALT", v) for v in VARS}
    alt.update({v + "_nA": frame("ALT_nA", v) for v in VARS})
    out = {"hours": np.arange(nh)}
    for (s, v), a in stats.items():
        out[f"in_{s}_{v}"] = a
    for et in ("SE", "SE_corr"):
        for fn in FUNCS:
            val, err = getattr(radfn, fn)(base, alt, et)
            out[f"ref_{fn[5:]}_{et}"] = np.stack([np.asarray(val, dtype=np.float64), np.asarray(err, dtype=np.float64)], axis=1)
    path = os.path.join(ROOT, "tests", "golden", "raddecomp.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith("ref_")})


if __name__ == "__main__":
    sys.exit(main())
