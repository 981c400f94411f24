"""The Fortran shims with clouds: RRTMG_SWRAD / RRTMG_LWRAD called by keyword with ICLOUD = 1 and the hydrometeor
arrays + F_Q* flags, as module_radiation_driver.F:1944-2001 / 1528-1577 does.  GPU: equals the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import compare_sw, run_oracle_swrad
from helpers_lw import run_oracle_lwrad

HERE = os.path.dirname(os.path.abspath(__file__))
FDIR = os.path.join(HERE, "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"
P_TOP = 5000.0

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")]
CLD = ("cldfra3d", "qc3d", "qr3d", "qi3d", "qs3d")


def _build(target):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, target, "FFLAGS=-O1 -fdefault-real-8"])
    return os.path.join(FDIR, target)


def test_sw_shim_with_clouds(hip_lib, sw_oracle, wa, tmp_path):
    exe = _build("plumbing_sw")
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260600, night_fraction=0.2)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260601, cloudy_fraction=0.7)
    wa.synth.write_sw_data(str(tmp_path / "RRTMG_SW_DATA_DBL"), seed=20260001, real_bytes=8)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<4i", st.ni, st.nj, st.nk, 1 + 10 * 1))        # clean_atm_diag + 10*icloud
        for n in ("xcoszen", "albedo", "tsk"):
            f.write(np.ascontiguousarray(st.a2[n], dtype="<f8").tobytes())
        for n in ("t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d",
                  "tauaer300", "tauaer400", "tauaer600", "tauaer999", "gaer300", "gaer400", "gaer600", "gaer999",
                  "waer300", "waer400", "waer600", "waer999") + CLD:
            f.write(np.ascontiguousarray(st.a3[n], dtype="<f8").tobytes())
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype="<f8")
    n3, n2 = st.nj * (st.nk + 1) * st.ni, st.nj * st.ni
    got = {"rthratensw": raw[:n3].reshape(st.nj, st.nk + 1, st.ni)}
    o = n3
    for n in ("coszr", "gsw", "swcf", "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
              "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln", "swddir", "swddni", "swddif"):
        got[n] = raw[o:o + n2].reshape(st.nj, st.ni)
        o += n2
    ref = run_oracle_swrad(sw_oracle, st, profiles=False, cloud=cloud)
    compare_sw(got, ref, 1e-9, 1e-9, names=list(got))
    assert np.abs(got["swcf"]).max() > 1.0


def test_lw_shim_with_clouds(hip_lib, oracle_mod, wa, tmp_path):
    from oracle import oracle_lw
    exe = _build("plumbing_lw")
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260610)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260611, cloudy_fraction=0.7)
    tab = tmp_path / "RRTMG_LW_DATA_DBL"
    wa.synth.write_lw_data(str(tab), seed=20260003, real_bytes=8)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<4i", st.ni, st.nj, st.nk, 1 + 10 * 1))
        f.write(np.array([P_TOP], dtype="<f8").tobytes())
        for n in ("emiss", "tsk"):
            f.write(np.ascontiguousarray(st.a2[n], dtype="<f8").tobytes())
        for n in ("t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d"):
            f.write(np.ascontiguousarray(st.a3[n], dtype="<f8").tobytes())
        for b in range(1, 17):
            f.write(np.ascontiguousarray(st.a3[f"tauaerlw{b}"], dtype="<f8").tobytes())
        for n in CLD:
            f.write(np.ascontiguousarray(st.a3[n], dtype="<f8").tobytes())
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype="<f8")
    n3, n2 = st.nj * (st.nk + 1) * st.ni, st.nj * st.ni
    got = {"rthratenlw": raw[:n3].reshape(st.nj, st.nk + 1, st.ni)}
    o = n3
    for n in ("glw", "olr", "lwcf", "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
              "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"):
        got[n] = raw[o:o + n2].reshape(st.nj, st.ni)
        o += n2
    orc = oracle_lw.LwOracle(str(tab), real_bytes=8)
    ref = run_oracle_lwrad(orc, st, oracle_lw.nlayers(st.nk + 1, P_TOP), profiles=False, cloud=cloud)
    compare_sw(got, ref, 1e-9, 1e-9, names=list(got))
    assert got["lwcf"].max() > 1.0
