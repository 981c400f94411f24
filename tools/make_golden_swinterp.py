#!/usr/bin/env python3
"""Golden vectors for the swint_opt = 1 routines of the radiation driver: runs the reference's
update_swinterp_parameters / interp_sw_radiation (oracle/_ref/ref_swinterp, compiled from
module_radiation_driver.F:2668-2795 by oracle/build_ref.sh, fp64) over a synthetic day and stores inputs and
reference outputs in tests/golden/swinterp_day.npz.

Runs in the build container only (needs /root/reference for the build); the fixture is committed."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_swinterp")
GOLD = os.path.join(ROOT, "tests", "golden")


def synth_day(n=96, nstep=8, seed=20260901):
    """nstep radiation steps of n points: sunrise to sunset with points that stay dark, points whose zenith angle
    does not change between two steps (ratio exactly 1), weak fluxes below the 1 W m-2 floor, and flux histories
    steep / flat enough to hit both clamps of the exponent."""
    rng = np.random.default_rng(seed)
    phase = rng.uniform(0.0, 1.0, n)
    amp = rng.uniform(0.2, 1.0, n)
    amp[:6] = 0.0                                       # polar night
    steps = []
    for s in range(nstep):
        t = (s + 0.5) / nstep
        cz = amp * np.sin(np.pi * np.clip(t + 0.15 * (phase - 0.5), 0, 1))
        cz[cz < 2e-4] = np.where(rng.uniform(size=(cz < 2e-4).sum()) < 0.5, 0.0, 5e-5)
        if s > 0:
            same = slice(6, 12)
            cz[same] = steps[-1]["coszen"][same]        # unchanged zenith angle
        czl = np.clip(cz + rng.normal(0, 0.01, n), 0, 1) * (cz > 0)
        czl[12:18] = cz[12:18]
        trans = rng.uniform(0.2, 0.8, n)
        glo = 1361.0 * cz ** rng.uniform(1.0, 1.6, n) * trans
        glo[18:24] *= 1e-3                              # below the max(1., flux) floor
        glo[24:30] *= rng.uniform(0.01, 100.0, 6)       # wild history: exponent clamps
        dirf = glo * rng.uniform(0.0, 0.9, n)
        cl2 = np.clip(cz + rng.normal(0, 0.03, n), 0, 1)
        cl2[30:34] = 0.0
        alb = rng.uniform(0.05, 0.8, n)
        steps.append(dict(coszen=cz, coszen_loc=czl, swddir=dirf, swdown=glo, coszen_loc2=cl2, albedo=alb))
    return steps


def main():
    steps = synth_day()
    n, nstep = steps[0]["coszen"].size, len(steps)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("ii", n, nstep))
            for st in steps:
                for k in ("coszen", "coszen_loc", "swddir", "swdown", "coszen_loc2", "albedo"):
                    f.write(np.ascontiguousarray(st[k], dtype=np.float64).tobytes())
        subprocess.run([REF, fin, fout], check=True)
        raw = np.fromfile(fout, dtype=np.float64).reshape(nstep, 12, n)
    names = ("bb", "Bx", "gg", "Gx", "coszen_ref", "swddir_ref", "swdown_ref", "o_swdown", "o_swddir", "o_swddni",
             "o_swddif", "o_gsw")
    out = {"n": n, "nstep": nstep}
    for k in steps[0]:
        out["in_" + k] = np.stack([st[k] for st in steps])
    for i, k in enumerate(names):
        out["ref_" + k] = raw[:, i, :]
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "swinterp_day.npz"), **out)
    print("wrote swinterp_day.npz:", n, "points x", nstep, "steps; clamped exponents:",
          int(np.sum((out["ref_bb"] == -0.5) | (out["ref_bb"] == 2.5))), "zero:", int(np.sum(out["ref_bb"] == 0)))


if __name__ == "__main__":
    main()
