"""The operation-counting build of the oracle (oracle/count: the column solvers and the optical averaging compiled as
C++ with a counting double) that bench.py's roofline.fp64 block uses: builds, runs a small row, and counts what the
bench line states: about 3 MFLOP per column for the two radiation calls at 50 layers and several times
that for the optical averaging."""
import numpy as np


def test_counting_build_counts_radiation_and_optics(wa, tmp_path, oracle_mod):
    from oracle import flopcount
    ni, nk, nbin = 6, 50, 8
    st, ms = wa.synth_mosaic.make_mosaic_rows(ni, 1, nk, 0, 1, nbin=nbin, seed=20260005, wrf_seed=20260002)
    sw_tab, lw_tab = str(tmp_path / "RRTMG_SW_DATA"), str(tmp_path / "RRTMG_LW_DATA")
    wa.synth.write_sw_data(sw_tab, seed=20260001, real_bytes=8)
    wa.synth.write_lw_data(lw_tab, seed=20260003, real_bytes=8)
    # the aerosol arrays the radiation calls read: any plausible values (the count does not depend on them much)
    for nm in ("300", "400", "600", "999"):
        st.a3["tauaer" + nm] = np.full_like(st.a3["t3d"], 0.01)
        st.a3["waer" + nm] = np.full_like(st.a3["t3d"], 0.9)
        st.a3["gaer" + nm] = np.full_like(st.a3["t3d"], 0.7)
    for b in range(1, 17):
        st.a3[f"tauaerlw{b}"] = np.full_like(st.a3["t3d"], 0.001)
    c = flopcount.per_column(wa, sw_tab, lw_tab, st, ms, "full")
    assert "error" not in c
    rad, opt = c["flops_per_column_radiation"], c["flops_per_column_optics"]
    assert 2.0e6 < rad < 5.0e6, rad
    assert 5.0e6 < opt < 5.0e7, opt
    assert abs(c["flops_per_column"] - (rad + opt)) < 1.0
