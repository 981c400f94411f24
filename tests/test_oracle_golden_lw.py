"""CPU: the longwave oracle against fixtures produced by the REFERENCE Fortran rrtmg_lw
(tools/make_golden_lw.py, oracle/_ref/ref_lw).  Bit-exact."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lw_table_file(tmp_path_factory, wa):
    p = tmp_path_factory.mktemp("tables_lw") / "RRTMG_LW_DATA"
    wa.synth.write_lw_data(str(p), seed=20260003, real_bytes=8)
    return str(p)


@pytest.fixture(scope="module")
def lw_oracle(oracle_mod, lw_table_file):
    from oracle import oracle_lw
    return oracle_lw.LwOracle(lw_table_file)


@pytest.mark.parametrize("name", ["lw_cols_k40", "lw_cols_k50_variant"])
def test_lw_column_solver_matches_reference_bitwise(lw_oracle, name):
    from oracle import oracle_lw
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cols = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    out = lw_oracle.rrtmg_lw(cols, clean=1)
    for n in oracle_lw.LW_COL_OUT:
        np.testing.assert_array_equal(out[n], g["ref_" + n], err_msg=n)


def test_lw_reference_fixture_physics():
    g = np.load(os.path.join(GOLD, "lw_cols_k50_variant.npz"))
    # cloud-free: clear == all-sky (LW:3305-3311); aerosol-free columns: clean == all-sky
    np.testing.assert_array_equal(g["ref_uflx"], g["ref_uflxc"])
    np.testing.assert_array_equal(g["ref_uflx"][::3], g["ref_uflxcln"][::3])
    g = np.load(os.path.join(GOLD, "lw_cols_k40.npz"))
    # LW-absorbing aerosol lowers the OLR relative to the clean call
    assert (g["ref_uflxcln"][:, -1] > g["ref_uflx"][:, -1]).all()


def test_lw_clean_diag_off(lw_oracle):
    g = np.load(os.path.join(GOLD, "lw_cols_k40.npz"))
    cols = {k[3:]: g[k][:3] for k in g.files if k.startswith("in_")}
    out = lw_oracle.rrtmg_lw(cols, clean=0)
    assert (out["uflxcln"] == 0).all() and (out["dflxcln"] == 0).all()     # LW:11028-11033
    np.testing.assert_array_equal(out["uflx"], g["ref_uflx"][:3])
