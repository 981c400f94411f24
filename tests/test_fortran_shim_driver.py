"""The ISO_C_BINDING shim of the driver-side neighbours (module_radiation_driver_hip): the reference's own routine
names and argument lists.  GPU: equals the oracle for a REAL*8 (-r8) and a default REAL*4 build."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle_driver as od

HERE = os.path.dirname(os.path.abspath(__file__))
FDIR = os.path.join(HERE, "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"

pytestmark = pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")


def test_driver_shim_builds(hip_lib):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, "plumbing_driver", "FFLAGS=-O1 -fdefault-real-8"])
    out = subprocess.run(["nm", "-u", os.path.join(FDIR, "plumbing_driver")], capture_output=True, text=True).stdout
    for sym in ("rrtmg_hip_update_swinterp_parameters", "rrtmg_hip_interp_sw_radiation", "rrtmg_hip_accumulate_fluxes"):
        assert sym in out


@pytest.mark.gpu
@pytest.mark.parametrize("fflags,dt,tol", [("-O1 -fdefault-real-8", "<f8", 1e-12), ("-O1", "<f4", 2e-6)])
def test_driver_shim_plumbing(hip_lib, tmp_path, fflags, dt, tol):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, "plumbing_driver", f"FFLAGS={fflags}"])
    exe = os.path.join(FDIR, "plumbing_driver")
    rng = np.random.default_rng(21)
    n = 50
    r4 = dt == "<f4"
    q = (lambda a: a.astype(np.float32).astype(np.float64)) if r4 else (lambda a: a)
    inp = {k: q(rng.uniform(lo, hi, n)) for k, (lo, hi) in dict(coszen=(0, 1), coszen_loc=(0, 1), swddir=(0, 600),
                                                                swdown=(0, 900), cl2=(0, 1), albedo=(0.05, 0.8)).items()}
    inp["coszen"][:3] = 0.0
    state = {k: q(rng.uniform(lo, hi, n)) for k, (lo, hi) in dict(swddir_ref=(0, 600), bb=(-0.5, 2.5), Bx=(-10, 900),
                                                                  swdown_ref=(0, 900), gg=(-0.5, 2.5), Gx=(-10, 900),
                                                                  coszen_ref=(0.05, 1)).items()}
    flux, ac = q(rng.uniform(0, 500, (2, n))), q(rng.uniform(0, 1e5, (2, n)))
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<i", n))
        for k in ("coszen", "coszen_loc", "swddir", "swdown", "cl2", "albedo"):
            f.write(inp[k].astype(dt).tobytes())
        for k in od.UPDATE_IO:
            f.write(state[k].astype(dt).tobytes())
        f.write(flux.astype(dt).tobytes()); f.write(ac.astype(dt).tobytes())
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=dt).astype(np.float64)
    got_state = dict(zip(od.UPDATE_IO, raw[:7 * n].reshape(7, n)))
    got_out = dict(zip(od.APPLY_OUT, raw[7 * n:12 * n].reshape(5, n)))
    got_ac = raw[12 * n:].reshape(2, n)
    st = {k: v.copy() for k, v in state.items()}
    od.swinterp_update({k: np.ascontiguousarray(inp[k]) for k in od.UPDATE_IN}, st, real4=r4)
    st = {k: q(v) for k, v in st.items()}
    out = {k: np.zeros(n) for k in od.APPLY_OUT}
    od.swinterp_apply(st, np.ascontiguousarray(inp["cl2"]), np.ascontiguousarray(inp["albedo"]), out, real4=r4)
    for k in od.UPDATE_IO:
        np.testing.assert_allclose(got_state[k], q(st[k]), rtol=tol, atol=tol, err_msg=k)
    for k in od.APPLY_OUT:
        np.testing.assert_allclose(got_out[k], q(out[k]), rtol=tol, atol=tol * 1e3 if r4 else tol, err_msg=k)
    np.testing.assert_allclose(got_ac, q(ac + flux * 18.0), rtol=tol, atol=0)
