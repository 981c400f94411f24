"""The optical-averaging ISO_C_BINDING shim (module_optical_averaging_hip), called with the chem array and the
species numbers the way an optical_driver would.  GPU: equals the oracle (parity unpinned, see optics_oracle.c)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers_optics import run_oracle_optics, tile_mask

HERE = os.path.dirname(os.path.abspath(__file__))
FDIR = os.path.join(HERE, "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"

pytestmark = pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")


def test_optics_shim_builds(hip_lib):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, "plumbing_optics", "FFLAGS=-O1 -fdefault-real-8"])
    out = subprocess.run(["nm", "-u", os.path.join(FDIR, "plumbing_optics")], capture_output=True, text=True).stdout
    assert "rrtmg_hip_optical_averaging" in out and "rrtmg_hip_optics_init" in out


@pytest.mark.gpu
@pytest.mark.parametrize("fflags,dt,rtol", [("-O1 -fdefault-real-8", "<f8", 1e-8), ("-O1", "<f4", 3e-7)])
def test_optics_shim_plumbing_4x4x40(hip_lib, wa, oracle_mod, tmp_path, fflags, dt, rtol):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, "plumbing_optics", f"FFLAGS={fflags}"])
    exe = os.path.join(FDIR, "plumbing_optics")
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260000)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260006)
    if dt == "<f4":
        ms.chem = ms.chem.astype(np.float32).astype(np.float64)
        ms.alt = ms.alt.astype(np.float32).astype(np.float64)
        ms.dz8w = ms.dz8w.astype(np.float32).astype(np.float64)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<6i", st.ni, st.nj, st.nk, ms.nbin, ms.chem.shape[0], 1))
        f.write(np.ascontiguousarray(ms.idx, dtype="<i4").tobytes())      # [bin][species] == Fortran idx(species, bin)
        f.write(np.ascontiguousarray(ms.chem, dtype=dt).tobytes())
        f.write(np.ascontiguousarray(ms.alt, dtype=dt).tobytes())
        f.write(np.ascontiguousarray(ms.dz8w, dtype=dt).tobytes())
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=dt).astype(np.float64)
    n3 = st.nj * (st.nk + 1) * st.ni
    legs = raw[28 * n3:].reshape(6, 4, st.nj, st.nk + 1, st.ni)      # Fortran leg(i,k,j,4,6): l2..l7 at four wavelengths
    raw = raw[:28 * n3].reshape(28, st.nj, st.nk + 1, st.ni)
    names = ([f"tauaer{w}" for w in range(1, 5)] + [f"gaer{w}" for w in range(1, 5)] + [f"waer{w}" for w in range(1, 5)]
             + [f"tauaerlw{b}" for b in range(1, 17)])
    ref = run_oracle_optics(wa, st, ms)
    mask = tile_mask(st, ms.alt.shape)
    for i, n in enumerate(names):
        g, e = raw[i], ref[n]
        assert np.all(np.abs(g[mask] - e[mask]) <= rtol * np.abs(e[mask]) + 1e-30), n
        assert np.all(g[~mask] == -7777.0), n
    # the optional tail of the argument list: idx_cw (all zero: nothing changes) and the six Legendre-moment arrays
    refl = run_oracle_optics(wa, st, ms, legendre=True)
    for k in range(6):
        g, e = legs[k], refl[f"l{k + 2}aer"]
        assert np.all(np.abs(g[:, mask] - e[:, mask]) <= max(rtol, 1e-9) * 10 + 1e-30), k
        assert np.all(g[:, ~mask] == -7777.0), k
