"""GPU parity tests proper: RRTMG_SWRAD through the HIP C-ABI against the CPU
oracle on the same seeded synthetic WRF state.

Tolerance: the device evaluates the same fp64 formulas with FMA contraction
and a different (wave-reduction) summation order over the 112 g-points, and
three bands use a re-associated but algebraically identical gas-optics
expression, so results agree to rounding, not bitwise:
   |hip - oracle| <= 1e-9 * |oracle| + 1e-9   (W m-2, K s-1 scaled fields alike)
"""
import numpy as np
import pytest

from helpers import SENTINEL, compare_sw, run_hip_swrad, run_oracle_swrad

pytestmark = pytest.mark.gpu
RTOL = 1e-9
ATOL = 1e-9


@pytest.fixture(scope="module")
def eng(hip_lib, sw_table_file):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    hip_lib.rrtmg_swinit(sw_table_file, 8, device=0)
    return hip_lib


def _check(got, ref):
    worst = compare_sw(got, ref, RTOL, ATOL)
    assert worst < 1e-8
    return worst


@pytest.mark.parametrize("ni,nj,nk,seed", [(8, 6, 40, 1), (5, 3, 50, 2), (3, 2, 28, 3), (4, 2, 60, 4), (3, 2, 100, 5)])
def test_swrad_matches_oracle(eng, sw_oracle, wa, ni, nj, nk, seed):
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260100 + seed, night_fraction=0.15)
    got = run_hip_swrad(eng, st, device=True)
    ref = run_oracle_swrad(sw_oracle, st)
    _check(got, ref)
    night = st.a2["xcoszen"] <= 0
    if night.any():
        # night columns: fluxes zeroed, gsw and the heating tendency untouched (SW:11173-11197)
        assert (got["swupt"][night] == 0).all() and (got["swdnbcln"][night] == 0).all()
        assert (got["gsw"][night] == SENTINEL).all()
        assert (got["rthratensw"][:, :nk, :][np.broadcast_to(night[:, None, :], (nj, nk, ni))] == SENTINEL).all()
    # cloud-free: clear-sky == all-sky exactly, cloud forcing 0
    np.testing.assert_array_equal(got["swuptc"], got["swupt"])
    np.testing.assert_array_equal(got["swdnbc"], got["swdnb"])
    day = ~night
    assert (got["swcf"][day] == 0).all()
    # absorbing aerosol: the clean-sky surface gets more sun than the polluted one
    assert (got["swdnbcln"][day] > got["swdnb"][day]).all()


@pytest.mark.parametrize("ni,nj,nk,seed,clean,aer", [(8, 6, 40, 11, 1, 1), (5, 3, 50, 12, 1, 1), (3, 2, 27, 13, 1, 1),
                                                     (4, 2, 60, 14, 0, 1), (3, 2, 100, 15, 1, 1), (4, 2, 40, 16, 1, 0)])
def test_swrad_without_profiles_matches_oracle(eng, sw_oracle, wa, ni, nj, nk, seed, clean, aer):
    """What the patched radiation_driver actually calls (DRV:1944-2001 passes SWUPFLX.. only if it received them, and
    first_rk_step_part1 does not pass them): no layer-flux profiles.  The clean-sky call then needs TOA and surface
    fluxes only, which the kernel composes in the upward sweep -- same tolerance as the profile path."""
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260100 + seed, night_fraction=0.15)
    got = run_hip_swrad(eng, st, device=True, profiles=False, clean=clean, aer=aer)
    ref = run_oracle_swrad(sw_oracle, st, profiles=False, clean=clean, aer=aer)
    _check(got, ref)
    if not aer:   # no aerosol: the clean-sky call is the same call (NCLX:27-32 takes the one for the other)
        np.testing.assert_array_equal(got["swuptcln"], got["swupt"])
        np.testing.assert_array_equal(got["swdnbcln"], got["swdnb"])
    full = run_hip_swrad(eng, st, device=True, profiles=True, clean=clean, aer=aer)
    # the two kernel instantiations agree to rounding (not bitwise: the compiler contracts different FMAs)
    compare_sw(got, full, 1e-11, 1e-11, names=list(got))


def test_host_pointer_mode_equals_device_mode(eng, wa):
    st = wa.synth.make_wrf_state(6, 3, 40, seed=20260111, night_fraction=0.2)
    a = run_hip_swrad(eng, st, device=True)
    b = run_hip_swrad(eng, st, device=False)
    for n in a:
        np.testing.assert_array_equal(a[n], b[n], err_msg=n)


def test_scratch_cap_splits_the_tile_into_launches(eng, wa, monkeypatch):
    """RRTMG_HIP_SW_SCRATCH_MB bounds the layer scratch: a tile larger than the cap is worked off in several
    launches (here 8 columns each) and gives the same bits as a single launch."""
    st = wa.synth.make_wrf_state(7, 5, 40, seed=20260155, night_fraction=0.2)
    one = run_hip_swrad(eng, st, device=True)
    monkeypatch.setenv("RRTMG_HIP_SW_SCRATCH_MB", "1")
    many = run_hip_swrad(eng, st, device=True)
    for n in one:
        np.testing.assert_array_equal(one[n], many[n], err_msg=n)


def test_clean_diag_off_gives_zero_cln(eng, sw_oracle, wa):
    st = wa.synth.make_wrf_state(4, 2, 40, seed=20260112)
    got = run_hip_swrad(eng, st, clean=0)
    ref = run_oracle_swrad(sw_oracle, st, clean=0)
    _check(got, ref)
    for n in ("swuptcln", "swdntcln", "swupbcln", "swdnbcln"):
        assert (got[n] == 0).all()            # SW:9473-9478
    assert (got["swupflxcln"][:, :42, :] == 0).all()


def test_no_aerosol_feedback_clean_equals_allsky(eng, sw_oracle, wa):
    st = wa.synth.make_wrf_state(4, 2, 40, seed=20260113)
    got = run_hip_swrad(eng, st, aer=0)
    ref = run_oracle_swrad(sw_oracle, st, aer=0)
    _check(got, ref)
    # ztaua == 0 makes call #1 == call #2 (the NCL extractor relies on it, NCLX:27-32)
    np.testing.assert_array_equal(got["swuptcln"], got["swupt"])
    np.testing.assert_array_equal(got["swdnbcln"], got["swdnb"])


def test_tile_inside_larger_memory_window(eng, sw_oracle, wa):
    st = wa.synth.make_wrf_state(9, 7, 40, seed=20260114, night_fraction=0.1)
    tile = dict(its=3, ite=7, jts=2, jte=5)
    got = run_hip_swrad(eng, st, tile=tile)
    ref = run_oracle_swrad(sw_oracle, st, tile=tile)
    _check(got, ref)
    # nothing outside the tile is written
    mask = np.ones((st.nj, st.ni), bool)
    mask[1:5, 2:7] = False
    for n in ("gsw", "swupt", "swddir", "coszr"):
        assert (got[n][mask] == SENTINEL).all(), n


def test_optional_outputs_absent(eng, sw_oracle, wa):
    st = wa.synth.make_wrf_state(4, 2, 40, seed=20260115)
    drop = ("swupflx", "swupflxc", "swupflxcln", "swdnflx", "swdnflxc", "swdnflxcln",
            "swvisdir", "swvisdif", "swnirdir", "swnirdif")
    got = run_hip_swrad(eng, st, drop=drop)
    ref = run_oracle_swrad(sw_oracle, st)
    compare_sw(got, ref, RTOL, ATOL, names=[n for n in got])


def test_real4_state_arrays(eng, sw_oracle, wa):
    """Stock WRF passes default REAL*4 arrays: the device widens them to fp64.  Compared with the oracle
    fed the same (rounded) inputs; outputs are rounded to float32 on the way out."""
    st = wa.synth.make_wrf_state(4, 3, 40, seed=20260116)
    for d in (st.a2, st.a3):
        for n in d:
            d[n] = d[n].astype(np.float32).astype(np.float64)
    got = run_hip_swrad(eng, st, real_bytes=4)
    ref = run_oracle_swrad(sw_oracle, st)
    compare_sw(got, ref, 3e-7, 1e-6)


def test_large_aod_is_rescaled_and_negative_is_fatal(eng, sw_oracle, wa):
    st = wa.synth.make_wrf_state(3, 2, 40, seed=20260117)
    for nm in ("300", "400", "600", "999"):
        st.a3["tauaer" + nm][0, :, 0] *= 40.0          # column AOD >> 6 -> rescaled (SW:11034-11047)
    got = run_hip_swrad(eng, st)
    ref = run_oracle_swrad(sw_oracle, st)
    _check(got, ref)
    st.a3["tauaer400"][1, :, 1] = -0.5                 # negative column AOD is fatal (SW:11031-11033)
    with pytest.raises(eng.WrfFatal, match="Negative total optical depth"):
        run_hip_swrad(eng, st)
    with pytest.raises(RuntimeError, match="Negative total optical depth"):
        run_oracle_swrad(sw_oracle, st)


def test_full_size_properties_200x200x40(eng, wa):
    """BASELINE config[1] size: size-independent properties instead of the (slow) oracle."""
    st = wa.synth.make_wrf_state(200, 200, 40, seed=20260001 + 1, night_fraction=0.1)
    got = run_hip_swrad(eng, st, profiles=False)
    day = st.a2["xcoszen"] > 0
    assert np.isfinite(got["swupt"]).all()
    toa_in = got["swdnt"][day]
    mu0 = st.a2["xcoszen"][day]
    # incoming TOA flux = adjflux * sum(sfluxzen) * mu0: proportional to mu0 for a fixed solar source
    ratio = toa_in / mu0
    assert ratio.std() / ratio.mean() < 0.02
    # energy: 0 <= up <= down at TOA, surface net == gsw
    assert (got["swupt"][day] >= 0).all() and (got["swupt"][day] <= toa_in).all()
    np.testing.assert_allclose(got["gsw"][day], (got["swdnb"] - got["swupb"])[day], rtol=0, atol=1e-10)
    # column-permutation invariance: shuffling i leaves each column's result unchanged
    perm = np.random.default_rng(0).permutation(st.ni)
    st2 = wa.synth.make_wrf_state(200, 200, 40, seed=20260001 + 1, night_fraction=0.1)
    for d in (st2.a2, st2.a3):
        for n in d:
            d[n] = np.ascontiguousarray(d[n][..., perm])
    got2 = run_hip_swrad(eng, st2, profiles=False)
    for n in ("swupt", "swdnb", "swuptcln", "swddir"):
        np.testing.assert_array_equal(got2[n], got[n][..., perm], err_msg=n)


def test_conservative_scattering_threshold_both_sides(eng, sw_oracle, wa):
    """reftra_sw switches to its conservative-scattering solution where the recovered single-scattering albedo
    zwo = zw / denom reaches zwcrit = 0.9999995 (SW:2557-2560).  The device forms the same test on the same rounded
    quotient.  Planted case: nearly white aerosol (1 - omega from 1e-9 to 3e-6, scanned over columns and layers) thick
    enough to swamp the gases in the window g-points, so that layers land on BOTH sides of the threshold and close to
    it -- the oracle's census says how many and how close -- and parity holds across it."""
    import ctypes as C
    from oracle import oracle as orc
    ni, nj, nk = 16, 8, 40
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260177, night_fraction=0.0)
    rng = np.random.default_rng(20260178)
    shape = st.a3["waer400"].shape
    one_minus = 10.0 ** rng.uniform(-9.0, -5.5, size=shape)
    for nm in ("300", "400", "600", "999"):
        st.a3["waer" + nm] = 1.0 - one_minus
        st.a3["gaer" + nm] = rng.uniform(0.0, 0.85, size=shape)
        st.a3["tauaer" + nm] = np.full(shape, 5.9 / nk)          # column AOD just under the guard of 6 (SW:11024-11047)
    lib = orc.lib()
    lib.sw_oracle_reftra_census.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.c_int]
    lib.sw_oracle_reftra_census(None, None, None, None, 1)
    ref = run_oracle_swrad(sw_oracle, st, profiles=False)
    ncons, ngen, above, below = C.c_long(), C.c_long(), C.c_double(), C.c_double()
    lib.sw_oracle_reftra_census(C.byref(ncons), C.byref(ngen), C.byref(above), C.byref(below), 1)
    assert ncons.value > 1000 and ngen.value > 1000            # both solutions are really taken ...
    assert above.value < 1e-9 and below.value < 1e-9           # ... by layers within 1e-9 of the threshold
    got = run_hip_swrad(eng, st, device=True, profiles=False)
    _check(got, ref)
    got = run_hip_swrad(eng, st, device=True, profiles=True)
    _check(got, run_oracle_swrad(sw_oracle, st, profiles=True))
