"""GPU parity tests of the cloudy (McICA) shortwave path: RRTMG_SWRAD with icloud /= 0 through the HIP C-ABI against
the CPU oracle (whose column solver incl. the KISS generator is bit-exact against the reference Fortran,
tests/test_oracle_golden_cloud.py).  The sub-column cloud masks must be identical, so the flux tolerance is the
same 1e-9 as for the cloud-free path."""
import numpy as np
import pytest

from helpers import compare_sw, run_hip_swrad, run_oracle_swrad

pytestmark = pytest.mark.gpu
RTOL = 1e-9
ATOL = 1e-9


def _init(wa, sw_table_file):
    import torch
    assert torch.cuda.is_available()
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    return wa.engine


@pytest.mark.parametrize("ni,nj,nk,seed,radii", [(8, 6, 40, 1, False), (5, 4, 50, 2, True), (4, 3, 28, 3, False),
                                                 (3, 3, 60, 4, True), (4, 3, 100, 5, False)])
def test_cloudy_swrad_matches_oracle(hip_lib, wa, sw_table_file, sw_oracle, ni, nj, nk, seed, radii):
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260400 + seed, night_fraction=0.1)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260410 + seed, cloudy_fraction=0.6, with_radii=radii)
    got = run_hip_swrad(eng, st, cloud=cloud)
    ref = run_oracle_swrad(sw_oracle, st, cloud=cloud)
    worst = compare_sw(got, ref, RTOL, ATOL)
    assert worst < 1e-8
    day = st.a2["xcoszen"] > 0
    cloudy = (st.a3["cldfra3d"].sum(axis=1) > 0) & day
    clear = (st.a3["cldfra3d"].sum(axis=1) == 0) & day
    assert cloudy.any() and clear.any()
    # clouds change the total-sky TOA flux; cloud-free columns keep clear == total exactly and zero forcing
    assert np.abs(got["swupt"][cloudy] - got["swuptc"][cloudy]).max() > 1.0
    np.testing.assert_array_equal(got["swupt"][clear], got["swuptc"][clear])
    assert (got["swcf"][clear] == 0).all() and (got["swcf"][cloudy] != 0).any()


@pytest.mark.parametrize("ni,nj,nk,seed,radii", [(8, 6, 40, 11, False), (5, 4, 50, 12, True), (4, 3, 100, 15, False)])
def test_cloudy_swrad_without_profiles_matches_oracle(hip_lib, wa, sw_table_file, sw_oracle, ni, nj, nk, seed, radii):
    """The call the driver makes (no layer-flux profiles): total-sky clean and clear-sky fluxes composed in the
    upward sweeps instead of stored and swept down."""
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260400 + seed, night_fraction=0.1)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260410 + seed, cloudy_fraction=0.6, with_radii=radii)
    got = run_hip_swrad(eng, st, cloud=cloud, profiles=False)
    ref = run_oracle_swrad(sw_oracle, st, cloud=cloud, profiles=False)
    worst = compare_sw(got, ref, RTOL, ATOL)
    assert worst < 1e-8


@pytest.mark.parametrize("fraction", [0.0, 1.0])
def test_cloud_lists_with_one_list_empty(hip_lib, wa, sw_table_file, sw_oracle, fraction):
    """icloud /= 0 sends cloud-free and cloudy columns to their own launches (csrc/cloud_lists.h): a tile where every
    column is cloud-free, and one where every column holds cloud, leave one of the two lists empty."""
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(7, 5, 40, seed=20260440, night_fraction=0.1)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260441, cloudy_fraction=fraction)
    cloudy = st.a3["cldfra3d"].sum(axis=1) > 0
    assert cloudy.all() if fraction else not cloudy.any()
    for device in (True, False):
        got = run_hip_swrad(eng, st, cloud=cloud, device=device)
        ref = run_oracle_swrad(sw_oracle, st, cloud=cloud)
        assert compare_sw(got, ref, RTOL, ATOL) < 1e-8


def test_cloudy_tile_subset_host_mode_and_no_clean(hip_lib, wa, sw_table_file, sw_oracle):
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(9, 7, 40, seed=20260420)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260421, cloudy_fraction=0.7)
    tile = dict(its=2, ite=8, jts=3, jte=6)
    a = run_hip_swrad(eng, st, device=True, tile=tile, cloud=cloud, clean=0)
    b = run_hip_swrad(eng, st, device=False, tile=tile, cloud=cloud, clean=0)
    ref = run_oracle_swrad(sw_oracle, st, tile=tile, cloud=cloud, clean=0)
    compare_sw(a, ref, RTOL, ATOL)
    for n in a:
        np.testing.assert_array_equal(a[n], b[n], err_msg=n)


def test_cloudy_real4_state_uses_single_precision_seeds(hip_lib, wa, sw_table_file, sw_oracle):
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(6, 5, 40, seed=20260430)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260431, cloudy_fraction=0.7)
    for d in (st.a2, st.a3):
        for n in d:
            d[n] = d[n].astype(np.float32).astype(np.float64)
    got = run_hip_swrad(eng, st, real_bytes=4, cloud=cloud)
    ref = run_oracle_swrad(sw_oracle, st, cloud=cloud, seed_real4=1)
    compare_sw(got, ref, 3e-7, 2e-4)


def test_cloud_refusals_and_fatal_sizes(hip_lib, wa, sw_table_file):
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(3, 2, 30, seed=20260440)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260441, cloudy_fraction=1.0, with_radii=True)
    with pytest.raises(wa.engine.WrfFatal, match="lradius and iradius"):     # CAM-MG radii announced, not passed
        run_hip_swrad(eng, st, cloud=dict(cloud, is_cammgmp_used=True))
    with pytest.raises(wa.engine.WrfFatal, match="FER_MP_HIRES needs qi3d"):
        run_hip_swrad(eng, st, cloud=dict(cloud, mp_physics=5, absent=("qi3d",)))
    with pytest.raises(wa.engine.WrfFatal, match="cldfra3d"):
        run_hip_swrad(eng, st, cloud=dict(cloud, absent=("cldfra3d",)))
    # a microphysics ice radius beyond the Fu table is fatal in cldprmc_sw (SW:2168-2173)
    st.a3["re_ice"][...] = 200.0e-6
    st.a3["cldfra3d"][:, 5, :] = 1.0
    st.a3["qi3d"][:, 5, :] = 1.0e-4
    st.a2["xcoszen"][...] = 0.5
    with pytest.raises(wa.engine.WrfFatal, match="ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS"):
        run_hip_swrad(eng, st, cloud=cloud)


def test_cloudy_full_size_properties_200x200x40(hip_lib, wa, sw_table_file):
    """BASELINE size with cloud in a fifth of the columns: size-independent properties -- the clear-sky fluxes of a
    cloudy run are the total-sky fluxes of the same state run without cloud, cloud-free columns are
    untouched by icloud, and a permutation of the columns permutes the results (the McICA seeds depend on the column's
    own pressures only)."""
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(200, 200, 40, seed=20260430, night_fraction=0.1)
    base = run_hip_swrad(eng, st, profiles=False)                      # icloud = 0
    cloud = wa.synth_clouds.add_clouds(st, seed=20260431, cloudy_fraction=0.2)
    got = run_hip_swrad(eng, st, cloud=cloud, profiles=False)
    day = st.a2["xcoszen"] > 0
    cloudy = (st.a3["cldfra3d"].sum(axis=1) > 0) & day
    clear = (st.a3["cldfra3d"].sum(axis=1) == 0) & day
    assert cloudy.sum() > 4000 and clear.sum() > 20000
    for n in got:
        if got[n].ndim == 2:
            assert np.isfinite(got[n][day]).all(), n
    # clear-sky fluxes of the cloudy run = total-sky fluxes of the cloud-free run, and cloud-free columns are untouched
    # by icloud -- to rounding, not bitwise: icloud = 0 and icloud /= 0 are two kernel instantiations (FMA contraction
    # differs), and in cloudy columns the clear-sky fluxes are composed in the upward sweep
    for c, t in (("swuptc", "swupt"), ("swdnbc", "swdnb"), ("swupbc", "swupb")):
        np.testing.assert_allclose(got[c][day], base[t][day], rtol=1e-11, atol=1e-11, err_msg=c)
        np.testing.assert_array_equal(got[c][clear], got[t][clear], err_msg=c)      # a copy in cloud-free columns
    for n in ("swupt", "swdnb", "swuptcln", "gsw"):
        np.testing.assert_allclose(got[n][clear], base[n][clear], rtol=1e-11, atol=1e-11, err_msg=n)
    assert (got["swcf"][clear] == 0).all()
    # cloud reflects: more upward flux at the top and less sunlight at the surface in most cloudy columns (McICA
    # samples one cloud state per g-point, so single columns may go either way)
    assert (got["swupt"][cloudy] > got["swuptc"][cloudy]).mean() > 0.7
    assert (got["swdnb"][cloudy] <= got["swdnbc"][cloudy]).mean() > 0.9
    perm = np.random.default_rng(0).permutation(st.ni)
    for d in (st.a2, st.a3):
        for n in d:
            d[n] = np.ascontiguousarray(d[n][..., perm])
    got2 = run_hip_swrad(eng, st, cloud=cloud, profiles=False)
    for n in ("swupt", "swdnb", "swuptcln", "swcf"):
        np.testing.assert_array_equal(got2[n], got[n][..., perm], err_msg=n)


@pytest.mark.parametrize("variant", ["f_ice_phy", "fer_hires", "cammgmp"])
def test_cloud_microphysics_variants(hip_lib, wa, sw_table_file, sw_oracle, variant):
    """The hydrometeor variants of the wrapper (SW:10461-10492, 10883-10898): "MP option 5" (snow carried alone and
    split 10/90 into ice and snow), Ferrier-Aligo (qi3d = all frozen water) and CAM-MG (effective radii supplied).
    HIP against the oracle's restatement of the same lines."""
    eng = _init(wa, sw_table_file)
    st = wa.synth.make_wrf_state(7, 5, 40, seed=20260450, night_fraction=0.1)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260451, cloudy_fraction=0.7)
    rng = np.random.default_rng(20260452)
    if variant == "f_ice_phy":
        cloud.update(f_ice_phy=1, absent=("qi3d",))                  # F_QC, F_QS, no F_QI
        st.a3["qs3d"] = st.a3["qs3d"] + st.a3["qi3d"]
    elif variant == "fer_hires":
        cloud.update(mp_physics=5)
    else:
        cloud.update(is_cammgmp_used=1)
        st.a3["lradius"] = rng.uniform(1.0, 70.0, st.a3["qc3d"].shape)   # beyond both clamps
        st.a3["iradius"] = rng.uniform(2.0, 160.0, st.a3["qc3d"].shape)
    got = run_hip_swrad(eng, st, cloud=cloud)
    ref = run_oracle_swrad(sw_oracle, st, cloud=cloud)
    assert compare_sw(got, ref, RTOL, ATOL) < 1e-8
    plain = {k: v for k, v in cloud.items() if k not in ("f_ice_phy", "mp_physics", "is_cammgmp_used", "absent")}
    base = run_hip_swrad(eng, st, cloud=plain)
    day = st.a2["xcoszen"] > 0
    assert np.abs(got["swupt"] - base["swupt"])[day].max() > 1e-3     # the variant matters
    if variant in ("f_ice_phy", "fer_hires"):
        # both blocks read the cloud water again (qc1d = qc3d, SW:10467 / 10487), which undoes the "T < 273.15: all
        # condensate is ice" rule for supercooled liquid: the fixture has such cells and they must count
        cold = (st.a3["t3d"] < 273.15) & (st.a3["qc3d"] > 0) & (st.a3["cldfra3d"] > 0)
        assert cold.sum() > 5
        keep = st.a3["qc3d"].copy()
        st.a3["qc3d"] = np.where(cold, 0.0, keep)
        dry = run_hip_swrad(eng, st, cloud=cloud)
        st.a3["qc3d"] = keep
        assert np.abs(got["swupt"] - dry["swupt"])[day].max() > 1e-3
