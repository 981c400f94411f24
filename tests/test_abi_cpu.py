"""CPU: the C-ABI library loads, exports every symbol include/rrtmg_hip.h
declares, and its host-side init (file loader + g-point reduction + packing)
reproduces the oracle's (== the reference's) reduced tables bit for bit.
No compute entry point is called here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "rrtmg_hip.h")).read()
    declared = set(re.findall(r"\b(rrtmg_hip_[a-z_0-9]+)\s*\(", hdr))
    assert {"rrtmg_hip_sw_init", "rrtmg_hip_swrad", "rrtmg_hip_finalize"} <= declared
    lib = hip_lib.load_library()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert set(hip_lib.EXPORTS) <= declared
    assert b"gfx950" in lib.rrtmg_hip_version()


def test_compute_entry_fails_loudly_without_init(hip_lib):
    # before init (and without a GPU) the product must raise, never fall back to a CPU path
    a = np.zeros((2, 4, 2))
    b = np.zeros((2, 2))
    with pytest.raises(hip_lib.WrfFatal):
        hip_lib.rrtmg_swrad(ims=1, ime=2, jms=1, jme=2, kms=1, kme=4, its=1, ite=2, jts=1, jte=2, kts=1, kte=3,
                            solcon=1370.0, xcoszen=b, albedo=b, tsk=b, t3d=a, t8w=a, p3d=a, p8w=a, pi3d=a, qv3d=a)


def test_missing_table_file_reports_reference_message(hip_lib):
    with pytest.raises(hip_lib.WrfFatal, match="RRTMG_SW_DATA"):
        hip_lib.sw_reduce_tables("/nonexistent/RRTMG_SW_DATA")


def test_packed_tables_equal_oracle_tables(hip_lib, sw_oracle, sw_table_file):
    K, exp_tbl = hip_lib.sw_reduce_tables(sw_table_file)
    np.testing.assert_array_equal(exp_tbl, sw_oracle.table(16, "exp_tbl"))
    ngc = [6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12]
    nspa = [9, 9, 9, 9, 1, 9, 9, 1, 9, 1, 0, 1, 9, 1]
    nspb = [1, 5, 1, 1, 1, 5, 1, 0, 1, 0, 0, 1, 5, 1]
    g0 = 0
    for ib, ng in enumerate(ngc):
        band = 16 + ib
        for ig in range(ng):
            info = hip_lib.sw_table_info(sw_table_file, g0 + ig)
            assert (info["band"], info["ig"], info["ng"]) == (ib, ig, ng)

            def col(off, nrow):
                return K[off + np.arange(nrow) * ng]

            for name, key, nrow in (("absa", "off_major_lo", nspa[ib] * 65), ("absb", "off_major_up", nspb[ib] * 235),
                                    ("selfref", "off_self", 10)):
                t = sw_oracle.table(band, name)
                if t is None:
                    continue
                # oracle/reference layout [ig][row]; product layout [row][ig]
                np.testing.assert_array_equal(col(info[key], nrow), t.reshape(ng, nrow)[ig], err_msg=f"{band} {name}")
            t = sw_oracle.table(band, "forref")
            if t is not None:
                nfor = t.size // ng
                np.testing.assert_array_equal(col(info["off_for"], nfor), t.reshape(ng, nfor)[ig])
            t = sw_oracle.table(band, "sfluxref")
            np.testing.assert_array_equal(col(info["off_sflux"], info["nsf"]), t.reshape(info["nsf"], ng)[:, ig])
            t = sw_oracle.table(band, "rayla")
            if t is not None:
                np.testing.assert_array_equal(col(info["off_rayla"], 9), t.reshape(9, ng)[:, ig])
            for name, key in (("rayl", "rayl_lo"), ("raylb", "rayl_up"), ("abso3a", "minor_lo"), ("abso3b", "minor_up"),
                              ("absch4", "minor_lo"), ("absco2", "minor_lo"), ("absh2o", "minor_up")):
                t = sw_oracle.table(band, name)
                if t is not None and not (band == 24 and name == "rayl"):
                    assert info[key] == t[ig], (band, name)
        g0 += ng
    assert g0 == 112


def test_real4_table_file_is_read(hip_lib, wa, tmp_path):
    p = tmp_path / "RRTMG_SW_DATA"
    wa.synth.write_sw_data(str(p), seed=20260001, real_bytes=4)
    K4, _ = hip_lib.sw_reduce_tables(str(p), real_bytes=4)
    p8 = tmp_path / "RRTMG_SW_DATA_DBL"
    wa.synth.write_sw_data(str(p8), seed=20260001, real_bytes=8)
    K8, _ = hip_lib.sw_reduce_tables(str(p8), real_bytes=8)
    assert K4.shape == K8.shape
    np.testing.assert_allclose(K4, K8, rtol=2e-7)
    with pytest.raises(hip_lib.WrfFatal):
        hip_lib.sw_reduce_tables(str(p), real_bytes=8)   # wrong word size -> record length mismatch
