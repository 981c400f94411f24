"""Domain statistics (misc_stats_library.ncl:405-458) and the forcing decomposition
(RadDecomp_functions.py:147-266): host arithmetic on the CPU, the device statistics kernel on the GPU."""
import numpy as np
import pytest


def test_domain_stats_host_definitions(wa):
    f = wa.forcing
    x = np.arange(1.0, 102.0)            # 101 values: percentile points fall on exact indices
    s = f.domain_stats_host(x[::-1])
    assert s["avg"] == 51.0 and s["min"] == 1.0 and s["max"] == 101.0 and s["N"] == 101
    assert s["median"] == 51.0 and s["x25"] == 26.0 and s["x75"] == 76.0 and s["x05"] == 6.0 and s["x95"] == 96.0
    assert s["stddev"] == pytest.approx(np.std(x, ddof=1)) and s["SE"] == pytest.approx(s["stddev"] / np.sqrt(101))
    assert s["morans_i"] == -9999.0
    # sorted[round(p/100 (N-1))] with halves rounded away from zero (NCL round(.,3)): N = 11, 25 % -> index 2.5 -> 3
    assert f.domain_stats_host(np.arange(11.0))["x25"] == 3.0
    assert f.domain_stats_host([])["N"] == 0.0


def test_morans_i_host_known_answers(wa):
    """Moran's I as calc_morans_i_2D forms it (neighbour + manhattan weights): a checkerboard is perfectly
    anti-correlated with its four neighbours, smooth fields are positively correlated, white noise is near zero;
    the weight sum counts every ordered pair of 4-neighbours."""
    f = wa.forcing
    jj, ii = np.mgrid[0:12, 0:16]
    n = 12 * 16
    board = ((ii + jj) % 2).astype(float)
    # I = sum_pairs d_a d_b / (W sd^2) with sd the sample (N-1) deviation: -(N-1)/N for the checkerboard
    assert f.morans_i_host(board) == pytest.approx(-(n - 1) / n, rel=1e-12)
    assert f.morans_i_host(ii + 0.5 * jj) > 0.9
    rng = np.random.default_rng(2)
    assert abs(f.morans_i_host(rng.normal(size=(60, 70)))) < 0.05
    assert f.morans_i_host(np.ones((5, 5))) == 0.0
    s = f.domain_stats_host(board, field2d=board)
    assert s["morans_i"] == pytest.approx(-(n - 1) / n) and s["SE_corr"] == pytest.approx(s["SE"] * s["morans_i"])


def test_decomposition_arithmetic(wa):
    f = wa.forcing
    rng = np.random.default_rng(5)
    mk = lambda: {n: {"avg": float(rng.normal(100, 20)), "SE": float(rng.uniform(0.1, 1.0))}
                  for n in ("SWUPT", "SWUPTCLN", "LWUPT", "LWUPTC")}
    base, base_na, alt, alt_na = mk(), mk(), mk(), mk()
    d = f.decompose(base, base_na, alt, alt_na, "SE")
    # Delta S (total SW change, ALT - BASE at TOA upward) = direct + indirect + semi-direct (RadDecomp_functions.py:181-190)
    delta_s = alt["SWUPT"]["avg"] - base["SWUPT"]["avg"]
    # the identity holds when the no-aerosol clean flux equals the no-aerosol total flux (SWUPTCLN_nA == SWUPT_nA)
    base_na["SWUPTCLN"]["avg"] = base_na["SWUPT"]["avg"]
    alt_na["SWUPTCLN"]["avg"] = alt_na["SWUPT"]["avg"]
    d = f.decompose(base, base_na, alt, alt_na, "SE")
    assert d["SW_DIRECT"][0] + d["SW_INDIRECT"][0] + d["SW_SEMIDIRECT"][0] == pytest.approx(delta_s)
    assert d["SW_DIRECT"][1] == pytest.approx(np.sqrt(base["SWUPTCLN"]["SE"] ** 2 + base["SWUPT"]["SE"] ** 2 +
                                                      alt["SWUPTCLN"]["SE"] ** 2 + alt["SWUPT"]["SE"] ** 2))
    lw_cre = lambda s: s["LWUPTC"]["avg"] - s["LWUPT"]["avg"]
    # with the LW indirect effect as the reference DOCUMENTS it (the default reproduces the reference's code, in which
    # it is identically zero: see test_decomposition_reproduces_the_reference)
    d = f.decompose(base, base_na, alt, alt_na, "SE", lw_indirect_as_documented=True)
    assert d["LW_INDIRECT"][0] + d["LW_SEMIDIRECT"][0] == pytest.approx(lw_cre(base) - lw_cre(alt))


def test_decomposition_reproduces_the_reference(wa):
    """forcing.decompose against the numbers the reference's own RadDecomp_functions.py returned for seeded
    statistics tables (tests/golden/raddecomp.npz, written by tools/make_golden_raddecomp.py in the build container):
    bit for bit, for both error types, including LW_INDIRECT as coded (zero, LWUPTC_nA read for every operand)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "raddecomp.npz"))
    f = wa.forcing
    cols = {"avg": 0, "SE": 1, "SE_corr": 2}
    for et in ("SE", "SE_corr"):
        for h in range(len(g["hours"])):
            scen = {s: {v: {c: float(g[f"in_{s}_{v}"][h, i]) for c, i in cols.items()}
                        for v in ("SWUPT", "LWUPT", "LWUPTC", "SWUPTCLN")} for s in ("BASE", "BASE_nA", "ALT", "ALT_nA")}
            d = f.decompose(scen["BASE"], scen["BASE_nA"], scen["ALT"], scen["ALT_nA"], et)
            for name in ("Delta_S", "Delta_L", "SW_DIRECT", "SW_INDIRECT", "SW_SEMIDIRECT", "LW_INDIRECT", "LW_SEMIDIRECT"):
                ref = g[f"ref_{name}_{et}"][h]
                assert d[name][0] == ref[0] and d[name][1] == ref[1], (name, et, h, d[name], ref)
            assert d["LW_INDIRECT"][0] == 0.0


@pytest.mark.gpu
def test_domain_stats_device_matches_host(hip_lib, wa):
    import torch
    f = wa.forcing
    rng = np.random.default_rng(6)
    nj, ni = 37, 53
    a = rng.normal(250.0, 40.0, size=(nj, ni))
    dims = dict(ims=1, ime=ni, jms=1, jme=nj, its=4, ite=50, jts=3, jte=30)
    t = torch.from_numpy(a).cuda()
    got = f.domain_stats(t, dims)
    ref = f.domain_stats_host(a[2:30, 3:50], field2d=a[2:30, 3:50])
    for n in f.STAT_NAMES:
        # Moran's I is a sum of cancelling products added in a different order: looser
        assert got[n] == pytest.approx(ref[n], rel=1e-9 if n in ("morans_i", "SE_corr") else 1e-12, abs=1e-13), n
    mask = (rng.uniform(size=(nj, ni)) < 0.4).astype(np.uint8)
    got = f.domain_stats(t, dims, mask=torch.from_numpy(mask).cuda())
    ref = f.domain_stats_host(a[2:30, 3:50][mask[2:30, 3:50] != 0], field2d=a[2:30, 3:50], mask=mask[2:30, 3:50])
    for n in f.STAT_NAMES:
        # Moran's I is a sum of cancelling products added in a different order: looser
        assert got[n] == pytest.approx(ref[n], rel=1e-9 if n in ("morans_i", "SE_corr") else 1e-12, abs=1e-13), n
    # a smooth field: strongly autocorrelated
    sm = np.add.outer(np.sin(np.arange(nj) / 6.0), np.cos(np.arange(ni) / 9.0))
    got = f.domain_stats(torch.from_numpy(sm).cuda(), dims)
    assert got["morans_i"] == pytest.approx(f.morans_i_host(sm[2:30, 3:50]), rel=1e-11) and got["morans_i"] > 0.9
    t32 = torch.from_numpy(a.astype(np.float32)).cuda()
    got = f.domain_stats(t32, dims)
    ref = f.domain_stats_host(a.astype(np.float32).astype(np.float64)[2:30, 3:50])
    assert got["median"] == ref["median"] and got["avg"] == pytest.approx(ref["avg"], rel=1e-12)
    empty = f.domain_stats(t, dims, mask=torch.zeros((nj, ni), dtype=torch.uint8, device="cuda"))
    assert empty["N"] == 0.0 and empty["avg"] == 0.0
