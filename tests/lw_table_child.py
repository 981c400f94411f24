"""Child process of test_lw_table_gpu.py.  RRTMG_HIP_LW_TABLE (read once per process) picks the longwave transfer kernel
of cloud-free calls without profile outputs: the default keeps the (1 - exp_tbl, tfn_tbl) pairs in LDS
(lw_rt_fast_tbl_kernel), "0" re-evaluates every entry from its index (lw_rt_fast_kernel).  The table is filled by the
arithmetic of the re-evaluation, so the two must give the same bits; the parent runs this script under both settings and
compares the files.  Prints LWTABLE-OK."""
import importlib
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [here, os.path.dirname(here)]


def main(tmp, out):
    wa = importlib.import_module("wrfchem-arc-interactions_amd")
    from helpers_lw import run_hip_lwrad
    lw_tab = os.path.join(tmp, "RRTMG_LW_DATA")
    if not os.path.exists(lw_tab):
        wa.synth.write_lw_data(lw_tab, seed=20260003, real_bytes=8)
    res = {}
    for tag, (ni, nj, nk, p_top) in (("k40", (11, 9, 40, 5000.0)), ("k60", (7, 5, 60, 1000.0))):
        st = wa.synth.make_wrf_state(ni, nj, nk, seed=20261300 + nk, night_fraction=0.0)
        wa.engine_lw.rrtmg_lwinit(p_top, lw_tab, kme=nk + 1, device=0)
        cloud = wa.synth_clouds.add_clouds(st, seed=20261400 + nk, cloudy_fraction=0.6)   # McICA: lw_rt_fast_cld(_tbl)_kernel
        for sub, kw in (("aer", dict(profiles=False)), ("noclean", dict(profiles=False, clean=0)),
                        ("tile", dict(profiles=False, tile=dict(its=2, ite=ni - 1, jts=2, jte=nj - 1))),
                        ("cloud", dict(profiles=False, cloud=cloud)), ("cloud_noaer", dict(profiles=False, cloud=cloud, aer=0))):
            for n, v in run_hip_lwrad(wa.engine_lw, st, device=True, **kw).items():
                res[f"{tag}_{sub}_{n}"] = np.asarray(v)
    np.savez(out, **res)
    print("LWTABLE-OK")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
