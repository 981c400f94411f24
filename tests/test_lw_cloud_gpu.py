"""GPU parity tests of the cloudy (McICA) longwave path: RRTMG_LWRAD with icloud /= 0 through the HIP C-ABI against
the CPU oracle (column solver incl. the KISS generator bit-exact against the reference Fortran,
tests/test_oracle_golden_cloud.py).  Same tolerance as the cloud-free path."""
import numpy as np
import pytest

from helpers import compare_sw
from helpers_lw import run_hip_lwrad, run_oracle_lwrad

pytestmark = pytest.mark.gpu
RTOL = 1e-9
ATOL = 1e-9
P_TOP = 5000.0


@pytest.fixture(scope="module")
def lw_table_file(tmp_path_factory, wa):
    p = tmp_path_factory.mktemp("tables_lw_cld") / "RRTMG_LW_DATA"
    wa.synth.write_lw_data(str(p), seed=20260003, real_bytes=8)
    return str(p)


@pytest.fixture(scope="module")
def lw_oracle(oracle_mod, lw_table_file):
    from oracle import oracle_lw
    return oracle_lw.LwOracle(lw_table_file)


def _init(wa, lw_table_file, nk):
    import torch
    assert torch.cuda.is_available()
    eng = wa.engine_lw
    eng.rrtmg_lwinit(P_TOP, lw_table_file, kme=nk + 1, device=0)
    return eng, eng.lw_nlayers(nk + 1, P_TOP)


@pytest.mark.parametrize("ni,nj,nk,seed,radii", [(8, 6, 40, 1, False), (5, 4, 50, 2, True), (4, 3, 28, 3, False),
                                                 (3, 3, 60, 4, True), (4, 3, 100, 5, True)])
def test_cloudy_lwrad_matches_oracle(hip_lib, wa, lw_table_file, lw_oracle, ni, nj, nk, seed, radii):
    eng, nlay = _init(wa, lw_table_file, nk)
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260500 + seed)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260510 + seed, cloudy_fraction=0.6, with_radii=radii)
    got = run_hip_lwrad(eng, st, cloud=cloud)
    ref = run_oracle_lwrad(lw_oracle, st, nlay, cloud=cloud)
    worst = compare_sw(got, ref, RTOL, ATOL)
    assert worst < 1e-8
    cloudy = st.a3["cldfra3d"].sum(axis=1) > 0
    assert cloudy.any() and (~cloudy).any()
    # clouds trap longwave: OLR below its clear-sky value, positive cloud forcing; cloud-free columns untouched
    # (the 100-layer case reaches into the synthetic stratosphere, where a thin cloud sits in warmer air than the
    # layers below it and the forcing may turn slightly negative)
    assert (got["lwcf"][cloudy] >= (0 if nk <= 64 else -1.0)).all() and got["lwcf"][cloudy].max() > 1.0
    np.testing.assert_array_equal(got["lwupt"][~cloudy], got["lwuptc"][~cloudy])
    assert (got["lwcf"][~cloudy] == 0).all()
    assert (got["olr"] == got["lwupt"]).all() and (got["glw"] == got["lwdnb"]).all()


@pytest.mark.parametrize("ni,nj,nk,seed,radii", [(8, 6, 40, 11, False), (5, 4, 50, 12, True), (4, 3, 100, 15, True)])
def test_cloudy_lwrad_without_profiles_matches_oracle(hip_lib, wa, lw_table_file, lw_oracle, ni, nj, nk, seed, radii):
    """The call the driver makes (no layer-flux profiles): clean-sky and clear-sky fluxes at the surface and the top
    from the downward sweep alone."""
    eng, nlay = _init(wa, lw_table_file, nk)
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260500 + seed)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260510 + seed, cloudy_fraction=0.6, with_radii=radii)
    got = run_hip_lwrad(eng, st, cloud=cloud, profiles=False)
    ref = run_oracle_lwrad(lw_oracle, st, nlay, cloud=cloud, profiles=False)
    assert compare_sw(got, ref, RTOL, ATOL) < 1e-8
    cloudy = st.a3["cldfra3d"].sum(axis=1) > 0
    np.testing.assert_array_equal(got["lwupt"][~cloudy], got["lwuptc"][~cloudy])
    assert (got["lwcf"][~cloudy] == 0).all()


@pytest.mark.parametrize("fraction", [0.0, 1.0])
def test_lw_cloud_lists_with_one_list_empty(hip_lib, wa, lw_table_file, lw_oracle, fraction):
    """icloud /= 0 sends cloud-free and cloudy columns to their own launches (csrc/cloud_lists.h): a tile where every
    column is cloud-free, and one where every column holds cloud, leave one of the two lists empty."""
    eng, nlay = _init(wa, lw_table_file, 40)
    st = wa.synth.make_wrf_state(7, 5, 40, seed=20260540)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260541, cloudy_fraction=fraction)
    cloudy = st.a3["cldfra3d"].sum(axis=1) > 0
    assert cloudy.all() if fraction else not cloudy.any()
    for device in (True, False):
        got = run_hip_lwrad(eng, st, cloud=cloud, device=device)
        ref = run_oracle_lwrad(lw_oracle, st, nlay, cloud=cloud)
        assert compare_sw(got, ref, RTOL, ATOL) < 1e-8


def test_cloudy_lw_tile_subset_host_mode_and_no_clean(hip_lib, wa, lw_table_file, lw_oracle):
    eng, nlay = _init(wa, lw_table_file, 40)
    st = wa.synth.make_wrf_state(9, 7, 40, seed=20260520)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260521, cloudy_fraction=0.7)
    tile = dict(its=2, ite=8, jts=3, jte=6)
    a = run_hip_lwrad(eng, st, device=True, tile=tile, cloud=cloud, clean=0)
    b = run_hip_lwrad(eng, st, device=False, tile=tile, cloud=cloud, clean=0)
    ref = run_oracle_lwrad(lw_oracle, st, nlay, tile=tile, cloud=cloud, clean=0)
    compare_sw(a, ref, RTOL, ATOL)
    for n in a:
        np.testing.assert_array_equal(a[n], b[n], err_msg=n)


def test_cloudy_lw_real4_state_uses_single_precision_seeds(hip_lib, wa, lw_table_file, lw_oracle):
    eng, nlay = _init(wa, lw_table_file, 40)
    st = wa.synth.make_wrf_state(6, 5, 40, seed=20260530)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260531, cloudy_fraction=0.7)
    for d in (st.a2, st.a3):
        for n in d:
            d[n] = d[n].astype(np.float32).astype(np.float64)
    got = run_hip_lwrad(eng, st, real_bytes=4, cloud=cloud)
    ref = run_oracle_lwrad(lw_oracle, st, nlay, cloud=cloud, seed_real4=1)
    compare_sw(got, ref, 3e-7, 2e-4)


def test_lw_cloud_refusals_and_fatal_sizes(hip_lib, wa, lw_table_file):
    eng, nlay = _init(wa, lw_table_file, 30)
    st = wa.synth.make_wrf_state(3, 2, 30, seed=20260540)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260541, cloudy_fraction=1.0, with_radii=True)
    with pytest.raises(wa.engine.WrfFatal, match="lradius and iradius"):
        run_hip_lwrad(eng, st, cloud=dict(cloud, is_cammgmp_used=True))
    with pytest.raises(wa.engine.WrfFatal, match="cldfra3d"):
        run_hip_lwrad(eng, st, cloud=dict(cloud, absent=("cldfra3d",)))
    st.a3["re_ice"][...] = 200.0e-6
    st.a3["cldfra3d"][:, 5, :] = 1.0
    st.a3["qi3d"][:, 5, :] = 1.0e-4
    with pytest.raises(wa.engine.WrfFatal, match="ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS"):
        run_hip_lwrad(eng, st, cloud=cloud)


def test_cloudy_lw_full_size_properties_200x200x40(hip_lib, wa, lw_table_file):
    """BASELINE size with cloud in a fifth of the columns: the clear-sky fluxes of a cloudy run are the total-sky
    fluxes of the same state run without cloud, cloud-free columns are untouched by icloud (bitwise), cloud reduces
    the outgoing longwave, and a permutation of the columns permutes the results."""
    eng, _ = _init(wa, lw_table_file, 40)
    st = wa.synth.make_wrf_state(200, 200, 40, seed=20260440)
    base = run_hip_lwrad(eng, st, profiles=False)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260441, cloudy_fraction=0.2)
    got = run_hip_lwrad(eng, st, cloud=cloud, profiles=False)
    cloudy = st.a3["cldfra3d"].sum(axis=1) > 0
    clear = ~cloudy
    assert cloudy.sum() > 4000
    for n in got:
        if got[n].ndim == 2:
            assert np.isfinite(got[n]).all(), n
    # to rounding, not bitwise: in a layer where some sub-column is cloudy rtrnmc evaluates the clear stream through
    # the cloudy-layer regime arithmetic (LW:3216-3282), which rounds differently from the cloud-free branch
    for c, t in (("lwuptc", "lwupt"), ("lwdnbc", "lwdnb"), ("lwupbc", "lwupb")):
        np.testing.assert_allclose(got[c], base[t], rtol=1e-13, atol=0, err_msg=c)
        np.testing.assert_array_equal(got[c][clear], base[t][clear], err_msg=c)
    for n in ("lwupt", "lwdnb", "lwuptcln", "glw", "olr"):
        np.testing.assert_array_equal(got[n][clear], base[n][clear], err_msg=n)
    assert (got["lwcf"][clear] == 0).all()
    # cloud tops are colder than the surface, cloud bases warmer than the clear sky above (in most columns: McICA
    # samples one cloud state per g-point)
    assert (got["lwupt"][cloudy] <= got["lwuptc"][cloudy]).mean() > 0.9
    assert (got["lwdnb"][cloudy] >= got["lwdnbc"][cloudy]).mean() > 0.9
    perm = np.random.default_rng(0).permutation(st.ni)
    for d in (st.a2, st.a3):
        for n in d:
            d[n] = np.ascontiguousarray(d[n][..., perm])
    got2 = run_hip_lwrad(eng, st, cloud=cloud, profiles=False)
    for n in ("lwupt", "lwdnb", "lwuptcln", "lwcf"):
        np.testing.assert_array_equal(got2[n], got[n][..., perm], err_msg=n)


@pytest.mark.parametrize("variant", ["f_ice_phy", "fer_hires", "cammgmp"])
def test_lw_cloud_microphysics_variants(hip_lib, wa, lw_table_file, lw_oracle, variant):
    """The hydrometeor variants of the wrapper (LW:12003-12034, 12473-12488), as in the shortwave test."""
    eng, nlay = _init(wa, lw_table_file, 40)
    st = wa.synth.make_wrf_state(7, 5, 40, seed=20260460)
    cloud = wa.synth_clouds.add_clouds(st, seed=20260461, cloudy_fraction=0.7)
    rng = np.random.default_rng(20260462)
    if variant == "f_ice_phy":
        cloud.update(f_ice_phy=1, absent=("qi3d",))
        st.a3["qs3d"] = st.a3["qs3d"] + st.a3["qi3d"]
    elif variant == "fer_hires":
        cloud.update(mp_physics=15)
    else:
        cloud.update(is_cammgmp_used=1)
        st.a3["lradius"] = rng.uniform(1.0, 70.0, st.a3["qc3d"].shape)
        st.a3["iradius"] = rng.uniform(2.0, 160.0, st.a3["qc3d"].shape)
    got = run_hip_lwrad(eng, st, cloud=cloud)
    ref = run_oracle_lwrad(lw_oracle, st, nlay, cloud=cloud)
    assert compare_sw(got, ref, RTOL, ATOL) < 1e-8
    plain = {k: v for k, v in cloud.items() if k not in ("f_ice_phy", "mp_physics", "is_cammgmp_used", "absent")}
    base = run_hip_lwrad(eng, st, cloud=plain)
    assert np.abs(got["lwupt"] - base["lwupt"]).max() > 1e-3
    if variant in ("f_ice_phy", "fer_hires"):
        # both blocks read the cloud water again (qc1d = qc3d, LW:12009 / 12029): supercooled liquid counts
        cold = (st.a3["t3d"] < 273.15) & (st.a3["qc3d"] > 0) & (st.a3["cldfra3d"] > 0)
        assert cold.sum() > 5
        keep = st.a3["qc3d"].copy()
        st.a3["qc3d"] = np.where(cold, 0.0, keep)
        dry = run_hip_lwrad(eng, st, cloud=cloud)
        st.a3["qc3d"] = keep
        assert np.abs(got["lwupt"] - dry["lwupt"]).max() > 1e-3
