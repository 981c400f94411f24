"""The longwave ISO_C_BINDING shim (module_ra_rrtmg_lw replacement), called by keyword like
module_radiation_driver.F:1528-1577 (BASELINE config 0, 4x4x40).  GPU: equals the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import compare_sw
from helpers_lw import run_oracle_lwrad

HERE = os.path.dirname(os.path.abspath(__file__))
FDIR = os.path.join(HERE, "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"
P_TOP = 5000.0

pytestmark = pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")


def test_lw_shim_builds(hip_lib):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, "plumbing_lw", "FFLAGS=-O1 -fdefault-real-8"])
    out = subprocess.run(["nm", "-u", os.path.join(FDIR, "plumbing_lw")], capture_output=True, text=True).stdout
    assert "rrtmg_hip_lwrad" in out and "rrtmg_hip_lw_init" in out


@pytest.mark.gpu
@pytest.mark.parametrize("fflags,dt,rtol,atol", [("-O1 -fdefault-real-8", "<f8", 1e-9, 1e-9), ("-O1", "<f4", 3e-7, 2e-6)])
def test_lw_shim_plumbing_4x4x40(hip_lib, wa, oracle_mod, tmp_path, fflags, dt, rtol, atol):
    from oracle import oracle_lw
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, "plumbing_lw", f"FFLAGS={fflags}"])
    exe = os.path.join(FDIR, "plumbing_lw")
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260000)
    rb = 8 if dt == "<f8" else 4
    if rb == 4:
        for d in (st.a2, st.a3):
            for n in d:
                d[n] = d[n].astype(np.float32).astype(np.float64)
    tab = tmp_path / ("RRTMG_LW_DATA_DBL" if rb == 8 else "RRTMG_LW_DATA")
    wa.synth.write_lw_data(str(tab), seed=20260003, real_bytes=rb)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<4i", st.ni, st.nj, st.nk, 1))
        f.write(np.array([P_TOP], dtype=dt).tobytes())
        for n in ("emiss", "tsk"):
            f.write(np.ascontiguousarray(st.a2[n], dtype=dt).tobytes())
        for n in ("t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d"):
            f.write(np.ascontiguousarray(st.a3[n], dtype=dt).tobytes())
        for b in range(1, 17):
            f.write(np.ascontiguousarray(st.a3[f"tauaerlw{b}"], dtype=dt).tobytes())
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=dt).astype(np.float64)
    n3, n2 = st.nj * (st.nk + 1) * st.ni, st.nj * st.ni
    got = {"rthratenlw": raw[:n3].reshape(st.nj, st.nk + 1, st.ni)}
    o = n3
    for n in ("glw", "olr", "lwcf", "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
              "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"):
        got[n] = raw[o:o + n2].reshape(st.nj, st.ni)
        o += n2
    assert o == raw.size
    orc = oracle_lw.LwOracle(str(tab), real_bytes=rb)
    ref = run_oracle_lwrad(orc, st, oracle_lw.nlayers(st.nk + 1, P_TOP), profiles=False)
    compare_sw(got, ref, rtol, atol, names=list(got))
