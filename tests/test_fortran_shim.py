"""The ISO_C_BINDING shim: module_ra_rrtmg_sw (replacement) is called by keyword
exactly like module_radiation_driver.F:1944-2001 does (BASELINE config 0, 4x4x40).
CPU: it compiles, links against librrtmg_hip.so and fails loudly without a GPU.
GPU: its outputs equal the oracle's for the same state."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import SENTINEL, compare_sw, run_oracle_swrad

HERE = os.path.dirname(os.path.abspath(__file__))
FDIR = os.path.join(HERE, "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"

pytestmark = pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not installed")


def build_driver(hip_lib, fflags):
    subprocess.check_call(["make", "-s", "-C", FDIR, "clean"])
    subprocess.check_call(["make", "-s", "-C", FDIR, f"FFLAGS={fflags}"])
    return os.path.join(FDIR, "plumbing_sw")


def write_state(path, st, clean, dt):
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", st.ni, st.nj, st.nk, clean))
        for n in ("xcoszen", "albedo", "tsk"):
            f.write(np.ascontiguousarray(st.a2[n], dtype=dt).tobytes())
        for n in ("t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d",
                  "tauaer300", "tauaer400", "tauaer600", "tauaer999", "gaer300", "gaer400", "gaer600", "gaer999",
                  "waer300", "waer400", "waer600", "waer999"):
            f.write(np.ascontiguousarray(st.a3[n], dtype=dt).tobytes())


def read_out(path, st, dt):
    raw = np.fromfile(path, dtype=dt).astype(np.float64)
    n3, n2 = st.nj * (st.nk + 1) * st.ni, st.nj * st.ni
    out = {"rthratensw": raw[:n3].reshape(st.nj, st.nk + 1, st.ni)}
    o = n3
    for n in ("coszr", "gsw", "swcf", "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
              "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln", "swddir", "swddni", "swddif"):
        out[n] = raw[o:o + n2].reshape(st.nj, st.ni)
        o += n2
    assert o == raw.size
    return out


def test_shim_builds_and_fails_loudly_without_gpu(hip_lib, wa, tmp_path):
    import torch
    exe = build_driver(hip_lib, "-O1 -fdefault-real-8")
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260000)
    write_state(tmp_path / "in.bin", st, 1, "<f8")
    wa.synth.write_sw_data(str(tmp_path / "RRTMG_SW_DATA_DBL"), seed=20260001, real_bytes=8)
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0                       # wrf_error_fatal path, never a silent CPU fallback
    assert "no HIP device" in r.stderr
    assert not (tmp_path / "out.bin").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("fflags,dt,rtol", [("-O1 -fdefault-real-8", "<f8", 1e-9), ("-O1", "<f4", 3e-7)])
def test_shim_plumbing_4x4x40(hip_lib, sw_oracle, wa, tmp_path, fflags, dt, rtol):
    exe = build_driver(hip_lib, fflags)
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260000, night_fraction=0.2)
    if dt == "<f4":
        for d in (st.a2, st.a3):
            for n in d:
                d[n] = d[n].astype(np.float32).astype(np.float64)
        wa.synth.write_sw_data(str(tmp_path / "RRTMG_SW_DATA"), seed=20260001, real_bytes=4)
    else:
        wa.synth.write_sw_data(str(tmp_path / "RRTMG_SW_DATA_DBL"), seed=20260001, real_bytes=8)
    write_state(tmp_path / "in.bin", st, 1, dt)
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_out(tmp_path / "out.bin", st, dt)
    if dt == "<f4":
        # the oracle must see the fp32-rounded tables the shim loaded
        from oracle import oracle as orc
        sw_oracle = orc.SwOracle(str(tmp_path / "RRTMG_SW_DATA"), real_bytes=4)
    ref = run_oracle_swrad(sw_oracle, st, profiles=False)
    names = list(got)
    compare_sw(got, ref, rtol, 1e-9 if dt == "<f8" else 2e-6, names=names)
    night = st.a2["xcoszen"] <= 0
    assert (got["gsw"][night] == SENTINEL).all()


@pytest.mark.gpu
def test_shim_ssib_albedos(hip_lib, sw_oracle, wa, tmp_path):
    """sf_surface_physics = 8 through the Fortran interface: the four spectral albedos where xland < 1.5."""
    from helpers import AER, IN2, IN3, alloc_outputs, dims_of
    from oracle import oracle as orc
    exe = build_driver(hip_lib, "-O1 -fdefault-real-8")
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260000, night_fraction=0.2)
    wa.synth.write_sw_data(str(tmp_path / "RRTMG_SW_DATA_DBL"), seed=20260001, real_bytes=8)
    write_state(tmp_path / "in.bin", st, 1001, "<f8")
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_out(tmp_path / "out.bin", st, "<f8")
    arrays = {n: st.a2[n] for n in IN2}
    arrays.update({n: st.a3[n] for n in IN3 + AER})
    alb = st.a2["albedo"]
    xland = np.ones_like(alb)
    xland[:, ::3] = 2.0                                   # Fortran xland(ims:ime:3, :) = 2.
    arrays.update(xland=xland, alswvisdir=0.5 * alb, alswvisdif=0.6 * alb, alswnirdir=1.4 * alb, alswnirdif=1.5 * alb)
    ref = alloc_outputs(st)
    orc.swrad_tile(sw_oracle, dims_of(st), arrays, 1370.0, 1, 1, ref)
    compare_sw(got, ref, 1e-9, 1e-9, names=[n for n in got if n in ref])


@pytest.mark.gpu
def test_shim_aer_opt_1(hip_lib, sw_oracle, wa, tmp_path):
    """aer_opt = 1 through the Fortran interface: aerod(ims:ime,kms:kme,jms:jme,1:6) replaces the other aerosol
    inputs (iaer = 6)."""
    from helpers import AER, IN2, IN3, alloc_outputs, dims_of
    from oracle import oracle as orc
    from test_sw_aerod_gpu import _aerod
    exe = build_driver(hip_lib, "-O1 -fdefault-real-8")
    st = wa.synth.make_wrf_state(4, 4, 40, seed=20260000, night_fraction=0.2)
    wa.synth.write_sw_data(str(tmp_path / "RRTMG_SW_DATA_DBL"), seed=20260001, real_bytes=8)
    aerod = _aerod(st, 20260731)
    write_state(tmp_path / "in.bin", st, 101, "<f8")
    with open(tmp_path / "in.bin", "ab") as f:
        f.write(np.ascontiguousarray(aerod, dtype="<f8").tobytes())
    r = subprocess.run([exe, "in.bin", "out.bin"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_out(tmp_path / "out.bin", st, "<f8")
    arrays = {n: st.a2[n] for n in IN2}
    arrays.update({n: st.a3[n] for n in IN3 + AER})   # passed with aer_ra_feedback = 1, ignored on this route
    arrays["aerod"] = aerod
    ref = alloc_outputs(st)
    orc.swrad_tile(sw_oracle, dims_of(st), arrays, 1370.0, 1, 1, ref)
    compare_sw(got, ref, 1e-9, 1e-9, names=[n for n in got if n in ref])
