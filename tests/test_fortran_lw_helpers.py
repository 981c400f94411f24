"""CPU: the host-side helpers module_ra_rrtmg_lw exports to other radiation schemes (inirad / o3data, relcalc,
reicalc, retab -- the reference's module_ra_rrtmg_sw imports them, SW:9894), as provided by the replacement module,
against the reference routines' own outputs (tests/golden/wrapper_helpers.npz).  Bit-exact in an -r8 build."""
import os
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FDIR = os.path.join(HERE, "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"

pytestmark = pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")


def test_exported_helpers_match_reference(hip_lib, tmp_path):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, "lw_helpers", "FFLAGS=-O1 -fdefault-real-8"])
    g = np.load(os.path.join(HERE, "golden", "wrapper_helpers.npz"))
    nk, n = g["t"].shape
    nprof, nlev = g["plev"].shape
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<4i", n, nk, nprof, nlev))
        for k in ("t", "landfrac", "landm", "icefrac", "snowh", "plev"):
            f.write(np.ascontiguousarray(g[k], dtype="<f8").tobytes())
    r = subprocess.run([os.path.join(FDIR, "lw_helpers"), "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype="<f8")
    np.testing.assert_array_equal(raw[:n * nk].reshape(nk, n), g["ref_rel"])
    np.testing.assert_array_equal(raw[n * nk:2 * n * nk].reshape(nk, n), g["ref_rei"])
    o = 2 * n * nk
    np.testing.assert_array_equal(raw[o:o + nprof * (nlev - 1)].reshape(nprof, nlev - 1), g["ref_o3"])
    retab = raw[o + nprof * (nlev - 1):]
    assert retab.size == 95 and retab[0] == 5.92779 and retab[-1] == 250.639
