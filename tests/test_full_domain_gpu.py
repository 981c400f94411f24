"""GPU tests at the sizes BASELINE.json quotes: the whole chain optical averaging -> RRTMG_SWRAD -> RRTMG_LWRAD
(aerosol + clean double call) on

* configs[3]: 500x500x50, MOSAIC 8-bin, fp64 -- through the scratch batching of both radiation calls, and
* one rank's share of configs[4]: 1000x125x60 (1000x1000x60 over 8 GPUs), mixed-precision optics,

checked by size-independent properties (energy bounds, insolation ~ mu0, GSW = SWDNB - SWUPB, clean >= polluted
surface flux, OLR = LWUPT, a different batching gives the same bits) and, on the first row of the domain, against the
oracle run independently from the tracers on (its own Mie code, its own radiation).  A third test bounds the flux and
forcing error of the mixed-precision optics path against the fp64 path at small size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P_TOP = 5000.0

SW_IN2 = ["xcoszen", "albedo", "tsk"]
ST_IN3 = ["t3d", "t8w", "p3d", "p8w", "pi3d", "qv3d"]
SW_AER = ["tauaer300", "tauaer400", "tauaer600", "tauaer999", "gaer300", "gaer400", "gaer600", "gaer999",
          "waer300", "waer400", "waer600", "waer999"]
SW_OUT2 = ["coszr", "gsw", "swcf", "swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
           "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln",
           "swvisdir", "swvisdif", "swnirdir", "swnirdif", "swddir", "swddni", "swddif"]
LW_IN2 = ["emiss", "tsk"]
LW_AER = [f"tauaerlw{b}" for b in range(1, 17)]
LW_OUT2 = ["glw", "olr", "lwcf", "lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
           "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"]


def _dims(ni, nj, nk, tile=None):
    d = dict(ims=1, ime=ni, jms=1, jme=nj, kms=1, kme=nk + 1, its=1, ite=ni, jts=1, jte=nj, kts=1, kte=nk)
    if tile:
        d.update(tile)
    return d


class Chain:
    """Device-resident state of one slab and the three calls of a step (as bench.py makes them)."""

    def __init__(self, wa, sw_tab, lw_tab, st, ms, precision="f64"):
        import torch
        self.wa, self.st, self.ms, self.torch = wa, st, ms, torch
        ni, nj, nk = st.ni, st.nj, st.nk
        wa.engine.rrtmg_swinit(sw_tab, 8, device=0)
        wa.engine_lw.rrtmg_lwinit(P_TOP, lw_tab, kme=nk + 1, device=0)
        wa.engine_optics.optics_init(ms.nbin, device=0)
        wa.engine_optics.optics_precision(precision)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        self.d = _dims(ni, nj, nk)
        self.sw_in = {n: dev(st.a2[n]) for n in SW_IN2}
        self.sw_in.update({n: dev(st.a3[n]) for n in ST_IN3})
        self.sw_in.update({n: torch.zeros((nj, nk + 1, ni), dtype=torch.float64, device="cuda") for n in SW_AER})
        self.lw_in = {n: dev(st.a2[n]) for n in LW_IN2}
        self.lw_in.update({n: self.sw_in[n] for n in ST_IN3})
        self.lw_in.update({n: torch.zeros((nj, nk + 1, ni), dtype=torch.float64, device="cuda") for n in LW_AER})
        self.opt_in = dict(chem=dev(ms.chem), alt=dev(ms.alt), dz8w=dev(ms.dz8w))
        self.opt_out = {}
        for w, nm in enumerate(("300", "400", "600", "999"), start=1):
            for stem in ("tauaer", "waer", "gaer"):
                self.opt_out[f"{stem}{w}"] = self.sw_in[stem + nm]
        for b in range(1, 17):
            self.opt_out[f"tauaerlw{b}"] = self.lw_in[f"tauaerlw{b}"]

    def run(self):
        torch, wa, st = self.torch, self.wa, self.st
        ni, nj, nk = st.ni, st.nj, st.nk
        sw_out = {"rthratensw": torch.full((nj, nk + 1, ni), -7777.0, dtype=torch.float64, device="cuda")}
        sw_out.update({n: torch.full((nj, ni), -7777.0, dtype=torch.float64, device="cuda") for n in SW_OUT2})
        lw_out = {"rthratenlw": torch.full((nj, nk + 1, ni), -7777.0, dtype=torch.float64, device="cuda")}
        lw_out.update({n: torch.full((nj, ni), -7777.0, dtype=torch.float64, device="cuda") for n in LW_OUT2})
        wa.engine_optics.optical_averaging(idx=self.ms.idx, aer_op_opt=1, **self.d, **self.opt_in, **self.opt_out)
        wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **self.d, **self.sw_in, **sw_out)
        wa.engine_lw.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **self.d, **self.lw_in, **lw_out)
        wa.engine.sw_status()
        wa.engine_lw.lw_status()
        torch.cuda.synchronize()
        out = {n: v.cpu().numpy() for n, v in sw_out.items()}
        out.update({n: v.cpu().numpy() for n, v in lw_out.items()})
        return out


def _properties(st, out):
    nk = st.nk
    day = st.a2["xcoszen"] > 0
    for n, v in out.items():
        if v.ndim == 2 and n not in ("gsw",):
            assert np.isfinite(v).all(), n
    assert np.isfinite(out["rthratensw"][:, :nk][np.broadcast_to(day[:, None, :], (st.nj, nk, st.ni))]).all()
    assert np.isfinite(out["rthratenlw"][:, :nk]).all()
    # shortwave: TOA insolation ~ mu0; 0 <= up <= down; surface net == GSW; clean sky lets more sun to the surface;
    # no cloud: clear == all-sky
    ratio = out["swdnt"][day] / st.a2["xcoszen"][day]
    assert ratio.std() / ratio.mean() < 0.02
    assert (out["swupt"][day] >= 0).all() and (out["swupt"][day] <= out["swdnt"][day]).all()
    np.testing.assert_allclose(out["gsw"][day], (out["swdnb"] - out["swupb"])[day], rtol=0, atol=1e-10)
    assert (out["swdnbcln"][day] >= out["swdnb"][day]).all()
    np.testing.assert_array_equal(out["swuptc"], out["swupt"])
    assert (out["swupt"][~day] == 0).all()
    # layer heating adds up to the absorbed flux: sum_k cp/g * dp * dT/dt == (net TOA - net surface) is checked by the
    # oracle rows; here the sign: the sun heats
    assert (out["rthratensw"][:, :nk][np.broadcast_to(day[:, None, :], (st.nj, nk, st.ni))] > 0).mean() > 0.999
    # longwave: OLR, GLW are the TOA-up / surface-down fluxes; nothing comes down at the top; the aerosol changes the
    # OLR by a few W m-2 either way (an absorbing layer may sit in warmer or colder air than what it hides) and adds
    # to the downward flux at the surface; the surface emits more than the sky returns
    np.testing.assert_array_equal(out["olr"], out["lwupt"])
    np.testing.assert_array_equal(out["glw"], out["lwdnb"])
    assert (out["lwdnt"] == 0).all() and (out["lwdntcln"] == 0).all()
    assert np.abs(out["lwuptcln"] - out["lwupt"]).max() < 25.0 and np.abs(out["lwuptcln"] - out["lwupt"]).max() > 0.01
    assert (out["lwdnb"] >= out["lwdnbcln"] - 1e-9).all()
    assert (out["lwupb"] > out["lwdnb"]).all()
    assert ((out["lwupt"] > 80) & (out["lwupt"] < 450)).all()


def _oracle_first_row(wa, sw_tab, lw_tab, st, ms):
    """The oracle on row 1 of the slab, independent of the device from the tracers on."""
    from oracle import oracle as orc, oracle_lw, oracle_optics as oo
    ni, nj, nk = st.ni, st.nj, st.nk
    tile = dict(jts=1, jte=1)
    d = _dims(ni, nj, nk, tile)
    wl, ref, den = wa.engine_optics.table_info()
    tabs = oo.make_tables(ms.nbin, wa.engine_optics.MOSAIC_DLO_UM, wa.engine_optics.MOSAIC_DHI_UM, wl, ref, den)
    opt = oo.tile(tabs, ms.chem, ms.alt, ms.dz8w, ms.idx, d)
    a3 = dict(st.a3)
    for w, nm in enumerate(("300", "400", "600", "999"), start=1):
        for stem in ("tauaer", "waer", "gaer"):
            a = opt[f"{stem}{w}"].copy()
            a[a == -7777.0] = 0.0
            a3[stem + nm] = a
    for b in range(1, 17):
        a = opt[f"tauaerlw{b}"].copy()
        a[a == -7777.0] = 0.0
        a3[f"tauaerlw{b}"] = a
    ref = {"rthratensw": np.full((nj, nk + 1, ni), -7777.0), "rthratenlw": np.full((nj, nk + 1, ni), -7777.0)}
    ref.update({n: np.full((nj, ni), -7777.0) for n in SW_OUT2 + LW_OUT2})
    arrays = {n: st.a2[n] for n in SW_IN2}
    arrays.update({n: a3[n] for n in ST_IN3 + SW_AER})
    orc.swrad_tile(orc.SwOracle(sw_tab), d, arrays, 1370.0, 1, 1, ref)
    arrays = {n: st.a2[n] for n in LW_IN2}
    arrays.update({n: a3[n] for n in ST_IN3 + LW_AER})
    oracle_lw.lwrad_tile(oracle_lw.LwOracle(lw_tab), d, oracle_lw.nlayers(nk + 1, P_TOP), arrays, 1, 1, ref)
    return ref


def _compare_row(out, ref, st, rtol, atol):
    day = st.a2["xcoszen"][0] > 0
    for n in SW_OUT2 + LW_OUT2:
        g, r = out[n][0], ref[n][0]
        if n == "gsw":
            g, r = g[day], r[day]
        assert np.all(np.abs(g - r) <= atol + rtol * np.abs(r)), (n, np.abs(g - r).max())
    nk = st.nk
    for n in ("rthratensw", "rthratenlw"):
        g, r = out[n][0, :nk], ref[n][0, :nk]
        if n == "rthratensw":
            g, r = g[:, day], r[:, day]
        assert np.all(np.abs(g - r) <= atol * 1e-3 + rtol * np.abs(r)), (n, np.abs(g - r).max())


@pytest.fixture(scope="module")
def lw_table_file(tmp_path_factory, wa):
    p = tmp_path_factory.mktemp("tables_full") / "RRTMG_LW_DATA"
    wa.synth.write_lw_data(str(p), seed=20260003, real_bytes=8)
    return str(p)


def test_config3_500x500x50_8bin(hip_lib, wa, sw_table_file, lw_table_file, monkeypatch):
    st, ms = wa.synth_mosaic.make_mosaic_rows(500, 500, 50, 0, 500, nbin=8, seed=20260005, wrf_seed=20260002)
    night = np.random.default_rng(1).uniform(size=(500, 500)) < 0.1
    st.a2["xcoszen"] = np.where(night, -0.3, st.a2["xcoszen"])
    ch = Chain(wa, sw_table_file, lw_table_file, st, ms)
    out = ch.run()          # default caps (a third / a sixth of the device memory): one launch per kernel
    _properties(st, out)
    _compare_row(out, _oracle_first_row(wa, sw_table_file, lw_table_file, st, ms), st, 1e-8, 1e-8)
    # a different batching (more, smaller launches) gives the same bits
    monkeypatch.setenv("RRTMG_HIP_SW_SCRATCH_MB", "6000")
    monkeypatch.setenv("RRTMG_HIP_LW_SCRATCH_MB", "3000")
    out2 = ch.run()
    for n in ("swupt", "swdnb", "swuptcln", "gsw", "rthratensw", "lwupt", "lwdnb", "lwuptcln", "rthratenlw"):
        np.testing.assert_array_equal(out2[n], out[n], err_msg=n)


def test_config4_share_1000x125x60_mixed(hip_lib, wa, sw_table_file, lw_table_file):
    """One rank's slab of BASELINE configs[4] (rows 0..124 of the 1000x1000x60 domain), optical averaging in mixed
    precision.  Properties as above; the first row against the fp64 oracle at the mixed path's tolerance (the optics
    differ by up to 2e-5 in shortwave optical depth, 1e-5 in omega and g, 1e-4 in longwave absorption:
    tests/test_optics_gpu.py)."""
    st, ms = wa.synth_mosaic.make_mosaic_rows(1000, 1000, 60, 0, 125, nbin=4, seed=20260005, wrf_seed=20260002)
    ch = Chain(wa, sw_table_file, lw_table_file, st, ms, precision="mixed")
    try:
        out = ch.run()
    finally:
        wa.engine_optics.optics_precision("f64")
    _properties(st, out)
    ref = _oracle_first_row(wa, sw_table_file, lw_table_file, st, ms)
    _compare_row(out, ref, st, 2e-5, 5e-3)


def test_mixed_optics_flux_and_forcing_error_is_bounded(hip_lib, wa, sw_table_file, lw_table_file):
    """What the packed-fp32 terms of the Mie series cost downstream: TOA / surface fluxes, heating rates and the clean-minus-aerosol
    forcing of the mixed path against the fp64 path on the same 8-bin state."""
    st, ms = wa.synth_mosaic.make_mosaic_rows(64, 32, 40, 0, 32, nbin=8, seed=20260015, wrf_seed=20260012)
    ch = Chain(wa, sw_table_file, lw_table_file, st, ms)
    a = ch.run()
    wa.engine_optics.optics_precision("mixed")
    try:
        b = ch.run()
    finally:
        wa.engine_optics.optics_precision("f64")
    worst = {}
    for n in ("swupt", "swdnb", "swupb", "lwupt", "lwdnb"):
        worst[n] = np.abs(a[n] - b[n]).max()
        assert worst[n] < 0.005, (n, worst[n])                      # W m-2 (fluxes are O(100-1000))
    f_sw = np.abs((a["swuptcln"] - a["swupt"]) - (b["swuptcln"] - b["swupt"])).max()
    f_lw = np.abs((a["lwuptcln"] - a["lwupt"]) - (b["lwuptcln"] - b["lwupt"])).max()
    assert f_sw < 0.005 and f_lw < 0.001, (f_sw, f_lw)              # forcing is O(1-50) W m-2
    assert 86400.0 * np.abs(a["rthratensw"] - b["rthratensw"])[:, :40].max() < 2e-3    # K/day
    assert 86400.0 * np.abs(a["rthratenlw"] - b["rthratenlw"])[:, :40].max() < 2e-3
    assert max(worst.values()) > 0.0                                # the two paths are different arithmetic
