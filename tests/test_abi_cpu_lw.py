"""CPU: the longwave host init of the product (loader + 256->140 g-point reduction + packing)
reproduces the oracle's (== the reference's) reduced tables bit for bit."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def lw_table_file(tmp_path_factory, wa):
    p = tmp_path_factory.mktemp("tables_lw") / "RRTMG_LW_DATA"
    wa.synth.write_lw_data(str(p), seed=20260003, real_bytes=8)
    return str(p)


def test_lw_packed_tables_equal_oracle_tables(hip_lib, wa, oracle_mod, lw_table_file):
    from oracle import oracle_lw
    eng = wa.engine_lw
    orc = oracle_lw.LwOracle(lw_table_file)
    K, exp_tbl, tfn_tbl = eng.lw_reduce_tables(lw_table_file)
    np.testing.assert_array_equal(exp_tbl, orc.table(1, "exp_tbl"))
    np.testing.assert_array_equal(tfn_tbl, orc.table(1, "tfn_tbl"))
    ngc = [10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2]
    g0 = 0
    checked = 0
    for ib, ng in enumerate(ngc):
        band = ib + 1
        for ig in range(ng):
            info = eng.lw_table_info(lw_table_file, g0 + ig)
            assert (info["band"], info["ig"], info["ng"]) == (ib, ig, ng)

            def col(off, nrow):
                return K[off + np.arange(nrow) * info["stride"]]

            def same(name, off):
                nonlocal checked
                t = orc.table(band, name)
                if t is None:
                    return
                nrow = t.size // ng
                np.testing.assert_array_equal(col(off, nrow), t.reshape(ng, nrow)[ig], err_msg=f"{band} {name}")
                checked += 1

            same("absa", info["off_major_lo"])
            same("selfref", info["off_self"])
            same("forref", info["off_for"])
            if orc.table(band, "absb") is not None:
                same("absb", info["off_major_up"])
            t = orc.table(band, "fracrefa")
            nfa = t.size // ng
            if nfa > 1:
                np.testing.assert_array_equal(col(info["off_frac_lo"], nfa), t.reshape(nfa, ng)[:, ig])
            else:
                assert info["frac_lo"] == t[ig]
            t = orc.table(band, "fracrefb")
            if t is not None:
                nfb = t.size // ng
                if nfb > 1:
                    np.testing.assert_array_equal(col(info["off_frac_up"], nfb), t.reshape(nfb, ng)[:, ig])
                elif band != 6:
                    assert info["frac_up"] == t[ig]
        g0 += ng
    assert g0 == 140 and checked > 500
    # minor tables and halocarbon coefficients of a few bands (slot order documented in lw_tables.hip)
    g3 = 22   # first g of band 3
    i3 = eng.lw_table_info(lw_table_file, g3)
    t = orc.table(3, "kam0")
    np.testing.assert_array_equal(K[i3["mlo0"] + np.arange(9 * 19) * i3["stride"]], t.reshape(16, 9 * 19)[0])
    t = orc.table(3, "kbm0")
    np.testing.assert_array_equal(K[i3["mup0"] + np.arange(5 * 19) * i3["stride"]], t.reshape(16, 5 * 19)[0])
    g8 = 88
    i8 = eng.lw_table_info(lw_table_file, g8)
    for slot, name in (("mlo0", "kam0"), ("mlo1", "kam2"), ("mlo2", "kam1"), ("mup0", "kbm0"), ("mup1", "kbm1")):
        t = orc.table(8, name)
        np.testing.assert_array_equal(K[i8[slot] + np.arange(19) * i8["stride"]], t.reshape(8, 19)[0], err_msg=name)
    assert i8["halo_lo0"] == orc.table(8, "gv0")[0] and i8["halo_lo1"] == orc.table(8, "gv1")[0]


def test_lw_nlayers_rule(hip_lib, wa):
    eng = wa.engine_lw
    # NLAYERS = kme + nint(p_top*0.01/4) - 1 (LW:12861)
    assert eng.lw_nlayers(41, 5000.0) == 53
    assert eng.lw_nlayers(51, 5000.0) == 63
    assert eng.lw_nlayers(41, 1000.0) == 43        # nint(2.5) = 3


def test_lw_missing_file_message(hip_lib, wa):
    with pytest.raises(wa.engine.WrfFatal, match="RRTMG_LW_DATA"):
        wa.engine_lw.lw_reduce_tables("/nonexistent/RRTMG_LW_DATA")
