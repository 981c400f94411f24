"""CPU: the cloudy (McICA) shortwave path of the oracle -- KISS sub-column generator, cloud optics and the
clear/total two-stream -- against fixtures produced by the REFERENCE Fortran itself (tools/make_golden_cloud.py:
mcica_subcol_sw + rrtmg_sw compiled from /root/reference).  Bit-exact, including the random cloud mask."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["sw_cloud_cols_k40_ice3", "sw_cloud_cols_k50_ice5"]


def _load(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    return g, {k[3:]: g[k] for k in g.files if k.startswith("in_")}


@pytest.mark.parametrize("name", NAMES)
def test_sw_cloudy_column_solver_matches_reference_bitwise(sw_oracle, oracle_mod, name):
    g, cols = _load(name)
    out = sw_oracle.rrtmg_sw(cols, clean=1)
    np.testing.assert_array_equal(out["cldfmc"], g["ref_cldfmc"], err_msg="McICA sub-column mask (KISS)")
    for n in oracle_mod.SW_COL_OUT:
        np.testing.assert_array_equal(out[n], g["ref_" + n], err_msg=n)


@pytest.mark.parametrize("name", NAMES)
def test_sw_cloud_fixture_properties(name):
    g, cols = _load(name)
    mask = g["ref_cldfmc"]                       # [col][g][layer]
    cldfrac = cols["cldfrac"]
    assert set(np.unique(mask)) <= {0.0, 1.0}
    # no cloud where the layer cloud fraction is zero; overcast layers cloudy in every sub-column
    assert (mask.sum(axis=1)[cldfrac == 0] == 0).all()
    assert (mask.sum(axis=1)[cldfrac == 1] == 112).all()
    # sub-column cloud cover approximates the layer cloud fraction (112 samples)
    part = (cldfrac > 0) & (cldfrac < 1)
    assert np.abs(mask.mean(axis=1)[part] - cldfrac[part]).max() < 0.25
    cloudy = mask.sum(axis=(1, 2)) > 0
    assert cloudy.any() and (~cloudy).any()
    # cloud-free columns: total == clear exactly; cloudy columns: clouds brighten the planet
    np.testing.assert_array_equal(g["ref_swuflx"][~cloudy], g["ref_swuflxc"][~cloudy])
    assert (g["ref_swuflx"][cloudy, -1] > g["ref_swuflxc"][cloudy, -1]).all()
    # the clean call is a total-sky call: with cloud it differs from the clear-sky aerosol flux
    assert np.abs(g["ref_swuflxcln"][cloudy, -1] - g["ref_swuflxc"][cloudy, -1]).max() > 1.0


LW_NAMES = ["lw_cloud_cols_k40_ice3", "lw_cloud_cols_k50_ice5"]


@pytest.fixture(scope="module")
def lw_oracle(oracle_mod, tmp_path_factory, wa):
    from oracle import oracle_lw
    p = tmp_path_factory.mktemp("tables_lw_cloud") / "RRTMG_LW_DATA"
    wa.synth.write_lw_data(str(p), seed=20260003, real_bytes=8)
    return oracle_lw.LwOracle(str(p))


@pytest.mark.parametrize("name", LW_NAMES)
def test_lw_cloudy_column_solver_matches_reference_bitwise(lw_oracle, name):
    """mcica_subcol_lw (permuteseed 150) + cldprmc + the cloudy branches of rtrnmc, both calls."""
    from oracle import oracle_lw
    g, cols = _load(name)
    out = lw_oracle.rrtmg_lw(cols, clean=1)
    np.testing.assert_array_equal(out["cldfmc"], g["ref_cldfmc"], err_msg="McICA sub-column mask (KISS)")
    for n in oracle_lw.LW_COL_OUT:
        np.testing.assert_array_equal(out[n], g["ref_" + n], err_msg=n)


@pytest.mark.parametrize("name", LW_NAMES)
def test_lw_cloud_fixture_properties(name):
    g, cols = _load(name)
    mask = g["ref_cldfmc"]
    cloudy = mask.sum(axis=(1, 2)) > 0
    assert cloudy.any() and (~cloudy).any()
    nk = cols["play"].shape[1] - 13
    assert mask[:, :, nk:].sum() == 0                     # no cloud in the buffer layers
    # cloud-free columns: clear == total exactly; clouds lower the outgoing longwave radiation
    np.testing.assert_array_equal(g["ref_uflx"][~cloudy], g["ref_uflxc"][~cloudy])
    assert (g["ref_uflx"][cloudy, -1] <= g["ref_uflxc"][cloudy, -1]).all()
    assert (g["ref_uflxc"][cloudy, -1] - g["ref_uflx"][cloudy, -1]).max() > 5.0
    # the two McICA streams differ: shortwave uses permuteseed 1, longwave 150
    assert mask.shape[1] == 140


def test_kiss_seed_single_precision_variant_differs(sw_oracle):
    """seed_real4 reproduces a stock REAL*4 build's seeds (fractional Pa in single precision): another mask."""
    g, cols = _load(NAMES[0])
    a = sw_oracle.rrtmg_sw(cols, clean=1)["cldfmc"]
    cols["seed_real4"] = 1
    b = sw_oracle.rrtmg_sw(cols, clean=1)["cldfmc"]
    assert a.sum() > 0 and not np.array_equal(a, b)
    part = (cols["cldfrac"] > 0) & (cols["cldfrac"] < 1)
    assert np.abs(b.mean(axis=1)[part] - cols["cldfrac"][part]).max() < 0.25
