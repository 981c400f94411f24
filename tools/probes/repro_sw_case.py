"""Reproduce one case of tools/stress_parity.py (shortwave) and print the fields that differ most."""
import importlib, os, sys, tempfile
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests")]
wa = importlib.import_module("wrfchem-arc-interactions_amd")
from oracle import oracle as orc
orc.build()
from helpers import run_hip_swrad, run_oracle_swrad
ni, nj, nk, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
night = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
tmp = tempfile.mkdtemp()
tab = os.path.join(tmp, "SW")
wa.synth.write_sw_data(tab, seed=20260001, real_bytes=8)
wa.engine.rrtmg_swinit(tab, device=0)
o = orc.SwOracle(tab)
for nf in ([night] if len(sys.argv) > 5 else [0.0, 0.1, 0.5]):
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=seed, night_fraction=nf)
    got = run_hip_swrad(wa.engine, st, device=False, profiles=True, clean=0, aer=1)
    ref = run_oracle_swrad(o, st, profiles=True, clean=0, aer=1)
    worst = []
    for n in got:
        d = np.abs(got[n] - ref[n]) / (np.abs(ref[n]) + 1e-9)
        i = np.unravel_index(np.argmax(d), d.shape)
        worst.append((float(d[i]), n, i, float(got[n][i]), float(ref[n][i])))
    worst.sort(reverse=True)
    print("night", nf)
    for w in worst[:8]:
        print("  %.2e %-12s %s got %.15g ref %.15g" % w)
