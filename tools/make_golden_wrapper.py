#!/usr/bin/env python3
"""Golden vectors for three helper routines of the reference's LW wrapper module that need nothing from WRF:
relcalc / reicalc (effective radii, module_ra_rrtmg_lw.F:14403-14491 with retab :11424-11442) and INIRAD / O3DATA
(ozone climatology, :12704-12840), run through oracle/_ref/ref_wrapper (compiled by oracle/build_ref.sh, fp64).
Stores inputs and reference outputs in tests/golden/wrapper_helpers.npz.

Runs in the build container only (needs /root/reference for the build); the fixture is committed."""
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_wrapper")
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    rng = np.random.default_rng(20260921)
    n, nk, nprof, nlev = 40, 30, 12, 43
    t = rng.uniform(170.0, 310.0, (nk, n))                 # Fortran t(n,nk): n fastest
    t[0, :8] = np.array([179.0, 180.0, 273.16, 273.0, 274.0, 253.16, 120.0, 400.0])   # table ends, melting point
    landfrac = rng.choice([0.0, 1.0], n)
    landm = np.where(rng.uniform(size=n) < 0.5, landfrac, rng.uniform(0, 1, n))
    icefrac = np.where(rng.uniform(size=n) < 0.6, 0.0, rng.uniform(0, 1, n))
    snowh = np.where(rng.uniform(size=n) < 0.6, 0.0, rng.uniform(0, 0.3, n))
    # pressure levels (mb), surface to top, kts:kte+2 as RRTMG_LWRAD passes them to inirad
    plev = np.zeros((nprof, nlev))
    for ip in range(nprof):
        ps = rng.uniform(600.0, 1040.0)
        top = rng.choice([50.0, 10.0, 1.0, 0.1])
        sig = np.linspace(0, 1, nlev - 1) ** rng.uniform(0.8, 1.6)
        plev[ip, :-1] = ps - (ps - top) * sig
        plev[ip, -1] = 0.5 * plev[ip, -2] if ip % 2 else 1.0e-5
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<4i", n, nk, nprof, nlev))
            for a in (t, landfrac, landm, icefrac, snowh, plev):
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        subprocess.run([REF, fin, fout], check=True)
        raw = np.fromfile(fout, dtype="<f8")
    rel = raw[:n * nk].reshape(nk, n)
    rei = raw[n * nk:2 * n * nk].reshape(nk, n)
    o3 = raw[2 * n * nk:].reshape(nprof, nlev - 1)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "wrapper_helpers.npz"), t=t, landfrac=landfrac, landm=landm,
                        icefrac=icefrac, snowh=snowh, plev=plev, ref_rel=rel, ref_rei=rei, ref_o3=o3)
    print("wrapper_helpers: rel", rel.min(), rel.max(), "rei", rei.min(), rei.max(), "o3", o3.min(), o3.max())


if __name__ == "__main__":
    main()
