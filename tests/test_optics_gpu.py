"""GPU parity tests: the optical-averaging stage through the HIP C-ABI against the CPU oracle (parity unpinned:
the oracle restates the engine's own specification, see oracle/optics_oracle.c).

Tolerance: |hip - oracle| <= 1e-8*|oracle| + 1e-30 in fp64.  The Mie series is evaluated in the same order on both
sides; differences come from FMA contraction, reciprocal-based division and the device's sin/cos/cbrt, amplified
by the cancellation in psi_n for very small size parameters (longwave bands, x ~ 2e-3)."""
import numpy as np
import pytest

from helpers import SENTINEL
from helpers_optics import LEG_NAMES, LW_NAMES, SW_NAMES, run_hip_optics, run_oracle_optics, tile_mask

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def _init(wa, nbin):
    import torch
    assert torch.cuda.is_available()
    wa.engine_optics.optics_init(nbin, device=0)


def _compare(got, ref, mask, rtol, atol=1e-30):
    worst = 0.0
    for n, g in got.items():
        r = ref[n]
        assert np.isfinite(g[mask]).all(), n
        err = np.abs(g[mask] - r[mask]) - (rtol * np.abs(r[mask]) + atol)
        assert (err <= 0).all(), (n, float(np.max(np.abs(g[mask] - r[mask]) / (np.abs(r[mask]) + 1e-300))))
        worst = max(worst, float(np.max(np.abs(g[mask] - r[mask]) / (np.abs(r[mask]) + 1e-300))))
        assert (g[~mask] == SENTINEL).all(), f"{n}: wrote outside the tile"
    return worst


MIX_OF = {1: 0, 3: 0, 2: 1, 4: 1, 5: 2}   # aer_op_opt (registry.chem:4014-4018) -> volume, Maxwell-Garnett, core + shell


@pytest.mark.parametrize("nbin,ni,nj,nk,opt", [(4, 9, 5, 40, 1), (8, 6, 4, 30, 1), (4, 5, 3, 20, 2), (8, 4, 3, 12, 4),
                                               (8, 5, 3, 14, 5), (4, 6, 2, 20, 5)])
def test_optics_matches_oracle(hip_lib, wa, nbin, ni, nj, nk, opt):
    _init(wa, nbin)
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20260300 + nbin + nk)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=nbin, seed=20260310 + nk)
    got = run_hip_optics(wa, st, ms, aer_op_opt=opt)
    ref = run_oracle_optics(wa, st, ms, mix=MIX_OF[opt])
    mask = tile_mask(st, ms.alt.shape)
    _compare(got, ref, mask, RTOL)
    t2 = got["tauaer2"][mask]
    assert (t2 >= 0).all() and t2.max() > 1e-4
    w2 = got["waer2"][mask][t2 > 0]
    assert (w2 > 0.2).all() and (w2 <= 1.0).all()


def test_optics_without_backscatter_outputs(hip_lib, wa):
    """What the radiation path asks for: tauaer / waer / gaer and tauaerlw only (bench.py, the Fortran driver shim).
    Without a bscoef output the shortwave kernel runs its instantiation without the backscatter sums of the series;
    the fields it does write must equal the full call's bit for bit and match the oracle, also for the strongly
    absorbing particles that take the downward recurrence."""
    _init(wa, 8)
    st = wa.synth.make_wrf_state(7, 4, 16, seed=20260370)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260371)
    names = ms.names
    for b in (6, 7, 8):   # soot-rich coarse sections in every second column: Im(m) x beyond the upward window
        bc = names.index(f"bc_a{b:02d}")
        for sp in ("so4", "oin", "oc"):
            i = names.index(f"{sp}_a{b:02d}")
            moved = 0.95 * ms.chem[i][:, :, ::2]
            ms.chem[bc][:, :, ::2] += moved * 1.7 / wa.synth_mosaic.DENSITY[sp]
            ms.chem[i][:, :, ::2] -= moved
    rad = [n for n in SW_NAMES if n.startswith(("tauaer", "waer", "gaer"))] + [n for n in LW_NAMES if n.startswith("tauaerlw")]
    full = run_hip_optics(wa, st, ms)
    part = run_hip_optics(wa, st, ms, names=rad)
    assert set(part) == set(rad)
    for n in rad:
        np.testing.assert_array_equal(part[n], full[n], err_msg=n)
    ref = run_oracle_optics(wa, st, ms)
    _compare(part, {n: ref[n] for n in rad}, tile_mask(st, ms.alt.shape), RTOL)


def test_optics_tile_subset_and_host_mode(hip_lib, wa):
    _init(wa, 4)
    st = wa.synth.make_wrf_state(8, 6, 16, seed=20260320)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260321)
    tile = dict(its=3, ite=7, jts=2, jte=5, kts=2, kte=15)
    a = run_hip_optics(wa, st, ms, device=True, tile=tile)
    b = run_hip_optics(wa, st, ms, device=False, tile=tile)
    ref = run_oracle_optics(wa, st, ms, tile=tile)
    mask = tile_mask(st, ms.alt.shape, tile)
    _compare(a, ref, mask, RTOL)
    for n in a:
        np.testing.assert_array_equal(a[n], b[n], err_msg=n)


def test_optics_giant_wet_particles(hip_lib, wa):
    """Coarse sections swollen with water: size parameters from ~100 through the segment limit of the shortwave
    kernel (12 checkpoints x 16 terms per downward pass, nstop > 192 takes a second pass) up to the cap of 225, in
    every cell of a few columns so that whole waves and mixed waves both occur."""
    _init(wa, 8)
    st = wa.synth.make_wrf_state(7, 5, 14, seed=20260350)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260351)
    names = ms.names
    rng = np.random.default_rng(20260352)
    for b in (7, 8):
        w = names.index(f"water_a{b:02d}")
        # wet / dry volume ratios of 3 .. 80: wet diameters of the 10 um section up to 40 um
        factor = 10.0 ** rng.uniform(0.5, 2.6, size=ms.chem[w].shape)
        factor[:, :, ::3] = 1.0                      # leave every third column as it was
        ms.chem[w] *= factor
    got = run_hip_optics(wa, st, ms)
    ref = run_oracle_optics(wa, st, ms)
    mask = tile_mask(st, ms.alt.shape)
    _compare(got, ref, mask, RTOL)
    assert got["tauaer1"][mask].max() > 1e-3


def test_optics_strong_absorbers_take_the_downward_path(hip_lib, wa):
    """The device carries the logarithmic derivative upward where x >= 0.3 and Im(m) x < 10 and falls back to the
    oracle's downward recurrence elsewhere (4 checkpoints per thread: series beyond 64 terms take more passes).  Soot
    in the coarse sections of every second column puts Im(m) x between 3 and 70 at the shortwave wavelengths and
    around the limit in the longwave, next to ordinary aerosol in the same waves; both Maxwell-Garnett and volume
    mixing."""
    _init(wa, 8)
    st = wa.synth.make_wrf_state(8, 5, 12, seed=20260360)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260361)
    names = ms.names
    dry = ["so4", "no3", "cl", "nh4", "na", "oin", "oc"]
    for b in (5, 6, 7, 8):
        bc = names.index(f"bc_a{b:02d}")
        for sp in dry:
            i = names.index(f"{sp}_a{b:02d}")
            moved = 0.97 * ms.chem[i][:, :, ::2]
            ms.chem[bc][:, :, ::2] += moved * 1.7 / wa.synth_mosaic.DENSITY[sp]   # same volume as soot
            ms.chem[i][:, :, ::2] -= moved
        w = names.index(f"water_a{b:02d}")
        ms.chem[w][:, :, ::2] *= 0.05
    mask = tile_mask(st, ms.alt.shape)
    for opt, mix in ((1, 0), (2, 1), (5, 2)):
        got = run_hip_optics(wa, st, ms, aer_op_opt=opt)
        ref = run_oracle_optics(wa, st, ms, mix=mix)
        _compare(got, ref, mask, RTOL)
        w1 = got["waer1"][mask]
        assert w1.min() < 0.6          # soot columns absorb


def test_optics_longwave_fallback_with_a_replaced_material_table(hip_lib, wa):
    """A material table handed over at init with a strongly absorbing "dust" (Im m = 2.5 in every longwave band) and
    coarse sections swollen to tens of micrometres: Im(m) x exceeds 10 for most coarse particles, so the longwave
    kernel leaves the upward recurrence and takes its downward fallback (two checkpoints per thread: series beyond
    32 terms run in segments), next to ordinary particles in the same waves."""
    wl, ref, den = wa.engine_optics.table_info()
    ref = ref.copy()
    oin = 5
    ref[oin, 4:, 0] = 2.2
    ref[oin, 4:, 1] = 2.5
    try:
        wa.engine_optics.optics_init(8, refindex=ref, density=den, device=0)
        st = wa.synth.make_wrf_state(6, 4, 10, seed=20260370)
        ms = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260371)
        names = ms.names
        rng = np.random.default_rng(20260372)
        for b in (6, 7, 8):
            for sp in ("so4", "no3", "cl", "nh4", "na", "oc", "bc"):
                i = names.index(f"{sp}_a{b:02d}")
                moved = 0.9 * ms.chem[i][:, :, ::2]
                ms.chem[names.index(f"oin_a{b:02d}")][:, :, ::2] += moved * 2.6 / wa.synth_mosaic.DENSITY[sp]
                ms.chem[i][:, :, ::2] -= moved
            w = names.index(f"water_a{b:02d}")
            ms.chem[w] *= 10.0 ** rng.uniform(0.0, 1.5, size=ms.chem[w].shape)
        got = run_hip_optics(wa, st, ms)
        ref_out = run_oracle_optics(wa, st, ms)      # the oracle reads the engine's table
        _compare(got, ref_out, tile_mask(st, ms.alt.shape), RTOL)
        assert got["tauaerlw16"][tile_mask(st, ms.alt.shape)].max() > 1e-3
    finally:
        wa.engine_optics.optics_init(8, device=0)    # the built-in table again


def test_optics_real4_state(hip_lib, wa):
    _init(wa, 4)
    st = wa.synth.make_wrf_state(6, 4, 24, seed=20260330)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260331)
    got = run_hip_optics(wa, st, ms, real_bytes=4)
    # oracle on the same rounded inputs
    ms32 = type(ms)(nbin=ms.nbin, chem=ms.chem.astype(np.float32).astype(np.float64),
                    alt=ms.alt.astype(np.float32).astype(np.float64),
                    dz8w=ms.dz8w.astype(np.float32).astype(np.float64), idx=ms.idx, names=ms.names)
    ref = run_oracle_optics(wa, st, ms32)
    mask = tile_mask(st, ms.alt.shape)
    _compare(got, ref, mask, 3e-7, atol=1e-37)


def test_optics_groups_and_refusals(hip_lib, wa):
    _init(wa, 4)
    st = wa.synth.make_wrf_state(3, 2, 8, seed=20260340)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260341)
    only_lw = run_hip_optics(wa, st, ms, names=[f"tauaerlw{b}" for b in range(1, 17)])
    full = run_hip_optics(wa, st, ms)
    for n in only_lw:
        np.testing.assert_array_equal(only_lw[n], full[n])
    with pytest.raises(wa.engine.WrfFatal, match="complete or not at all"):
        run_hip_optics(wa, st, ms, names=["tauaer1", "waer1", "gaer1"])
    with pytest.raises(wa.engine.WrfFatal, match="aer_op_opt must be 1..5"):
        run_hip_optics(wa, st, ms, aer_op_opt=6)


def test_optics_feed_the_radiation(hip_lib, wa, sw_table_file, sw_oracle):
    """End to end: MOSAIC tracers -> (tau, omega, g) -> RRTMG_SWRAD double call, HIP against oracle."""
    from helpers import compare_sw, run_hip_swrad, run_oracle_swrad
    _init(wa, 4)
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    st = wa.synth.make_wrf_state(6, 4, 40, seed=20260350)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260351)
    got = run_hip_optics(wa, st, ms)
    ref = run_oracle_optics(wa, st, ms)
    for src, dst in (("1", "300"), ("2", "400"), ("3", "600"), ("4", "999")):
        for stem in ("tauaer", "waer", "gaer"):
            a = got[stem + src].copy()
            a[a == SENTINEL] = 0.0
            st.a3[stem + dst] = a
    hip = run_hip_swrad(wa.engine, st)
    for src, dst in (("1", "300"), ("2", "400"), ("3", "600"), ("4", "999")):
        for stem in ("tauaer", "waer", "gaer"):
            st.a3[stem + dst] = ref[stem + src]
    orc = run_oracle_swrad(sw_oracle, st)
    assert compare_sw(hip, orc, 1e-8, 1e-8) < 1e-7
    forcing = hip["swuptcln"] - hip["swupt"]
    assert np.abs(forcing).max() > 0.1


def test_optics_full_size_properties_200x200x40(hip_lib, wa):
    """BASELINE configs[2] size: size-independent properties instead of the (slow) oracle -- physical bounds, exact
    linearity in the particle number (masses and numbers doubled: every diameter, size parameter and refractive index
    is unchanged, so tau doubles exactly and omega, g stay bitwise the same), and invariance under a permutation of
    the columns."""
    _init(wa, 4)
    st = wa.synth.make_wrf_state(200, 200, 40, seed=20260360)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260361)
    got = run_hip_optics(wa, st, ms)
    mask = tile_mask(st, ms.alt.shape)
    for n, g in got.items():
        v = g[mask]
        assert np.isfinite(v).all(), n
        assert (v >= 0).all(), n
        if n.startswith(("waer", "gaer")):
            assert (v <= 1.0).all(), n
    assert got["tauaer1"][mask].max() > 1e-4 and got["tauaerlw8"][mask].max() > 1e-6
    # shortwave extinction falls from 300 to 999 nm on average (sub-micron particles dominate), absorption does not vanish
    assert got["tauaer1"][mask].mean() > got["tauaer4"][mask].mean()
    ms2 = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260361)
    ms2.chem = ms2.chem * 2.0
    got2 = run_hip_optics(wa, st, ms2)
    for n in got:
        if n.startswith(("waer", "gaer")):
            np.testing.assert_array_equal(got2[n][mask], got[n][mask], err_msg=n)
        else:   # optical depths, extinction and backscatter coefficients
            np.testing.assert_array_equal(got2[n][mask], 2.0 * got[n][mask], err_msg=n)
    perm = np.random.default_rng(1).permutation(st.ni)
    ms3 = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260361)
    ms3.chem = np.ascontiguousarray(ms3.chem[..., perm])
    ms3.alt = np.ascontiguousarray(ms3.alt[..., perm])
    ms3.dz8w = np.ascontiguousarray(ms3.dz8w[..., perm])
    got3 = run_hip_optics(wa, st, ms3)
    for n in ("tauaer2", "waer2", "gaer2", "tauaerlw5", "tauaerlw6"):
        np.testing.assert_array_equal(got3[n], got[n][..., perm], err_msg=n)


@pytest.mark.parametrize("opt", [1, 2])
def test_optics_mixed_precision(hip_lib, wa, opt):
    """The "mixed" path (BASELINE configs[4]): psi_n and the upward D_n in fp64, chi_n, a_n, b_n and the sums as packed
    fp32 pairs (two wavelengths / two longwave bands of a particle), the small-argument ladder in packed fp32.  Against the
    fp64 oracle: shortwave optical depths and coefficients within 2e-5 (measured 6e-7), single-scattering albedo and
    asymmetry within 1e-5 absolute (4e-7), backscatter -- an alternating sum -- within 2e-4 (2e-5), longwave
    absorption -- a difference of two sums -- within 1e-4 (1e-5); switching back restores the fp64 bits."""
    _init(wa, 8)
    st = wa.synth.make_wrf_state(9, 6, 30, seed=20260370)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260371)
    ref64 = run_hip_optics(wa, st, ms, aer_op_opt=opt)
    try:
        wa.engine_optics.optics_precision("mixed")
        got = run_hip_optics(wa, st, ms, aer_op_opt=opt)
    finally:
        wa.engine_optics.optics_precision("f64")
    orc = run_oracle_optics(wa, st, ms, mix=MIX_OF[opt])
    mask = tile_mask(st, ms.alt.shape)
    worst = 0.0
    for n in SW_NAMES + LW_NAMES:
        g, r = got[n][mask], orc[n][mask]
        assert np.isfinite(g).all(), n
        assert (got[n][~mask] == SENTINEL).all(), n
        if n.startswith(("waer", "gaer")):
            assert np.abs(g - r).max() < 1e-5, (n, float(np.abs(g - r).max()))
            continue
        tol = 2e-4 if n.startswith("bscoef") else (1e-4 if "lw" in n else 2e-5)
        big = r > 1e-10 * r.max()
        rel = np.abs(g - r)[big] / r[big]
        worst = max(worst, float(rel.max()) / tol)
        assert rel.max() < tol, (n, float(rel.max()))
        assert (g[~big] <= 1e-9 * r.max() + 2 * r[~big]).all(), n
    assert not np.array_equal(got["tauaerlw8"], ref64["tauaerlw8"]) and not np.array_equal(got["tauaer2"], ref64["tauaer2"])
    again = run_hip_optics(wa, st, ms, aer_op_opt=opt)
    for n in SW_NAMES + LW_NAMES:
        np.testing.assert_array_equal(again[n], ref64[n], err_msg=n)          # switching back restores fp64


def test_shell_exact_core_and_shell(hip_lib, wa):
    """aer_op_opt = 5 (shell_exact): black carbon as a concentric core.  Without black carbon the particle is the
    homogeneous sphere of the volume-mixing rule; with black carbon the result differs from both mixing rules and absorbs; host arrays (mem = 0) give the device-resident result; the mixed precision mode does not touch it."""
    _init(wa, 8)
    st = wa.synth.make_wrf_state(6, 4, 12, seed=20260390)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260391)
    mask = tile_mask(st, ms.alt.shape)
    shell = run_hip_optics(wa, st, ms, aer_op_opt=5)
    vol, mg = run_hip_optics(wa, st, ms, aer_op_opt=1), run_hip_optics(wa, st, ms, aer_op_opt=2)
    for other in (vol, mg):
        d = np.abs(shell["waer2"][mask] - other["waer2"][mask])
        assert d.max() > 1e-4
    assert (shell["waer2"][mask] < 1.0).all() and (shell["tauaerlw8"][mask] > 0).all()
    host = run_hip_optics(wa, st, ms, aer_op_opt=5, device=False)
    for n in shell:
        np.testing.assert_array_equal(shell[n], host[n], err_msg=n)
    try:
        wa.engine_optics.optics_precision("mixed")
        again = run_hip_optics(wa, st, ms, aer_op_opt=5)
    finally:
        wa.engine_optics.optics_precision("f64")
    for n in shell:
        np.testing.assert_array_equal(shell[n], again[n], err_msg=n)
    # no black carbon anywhere: the shell is the whole particle
    ms0 = wa.synth_mosaic.make_mosaic_state(st, nbin=8, seed=20260391)
    for i, nm in enumerate(ms0.names):
        if nm.startswith("bc_"):
            ms0.chem[i][...] = 0.0
    a, b = run_hip_optics(wa, st, ms0, aer_op_opt=5), run_oracle_optics(wa, st, ms0, mix=0)
    _compare(a, b, mask, RTOL)


@pytest.mark.parametrize("real_bytes", [8, 4])
def test_cloud_borne_tracers_are_added_to_the_interstitial_ones(hip_lib, wa, real_bytes):
    """idx_cw (the "_cw" tracers of the aqueous packages, registry.chem:3935-3936): cloud-borne mass and number are
    added to the interstitial ones before the particle geometry is formed -- the call equals, bit for bit, the call
    on tracers summed beforehand in the arrays' own type; device-resident and host arrays alike; no idx_cw = before."""
    import copy
    _init(wa, 4)
    st = wa.synth.make_wrf_state(7, 9, 10, seed=20260395)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260396)
    dt = np.float64 if real_bytes == 8 else np.float32
    nsp = ms.chem.shape[0]
    rng = np.random.default_rng(20260397)
    idx = np.asarray(ms.idx)
    # cloud-borne copies of every dry species and of the number, appended behind the interstitial tracers
    idx_cw = np.zeros_like(idx)
    extra = []
    for b in range(idx.shape[0]):
        for s in range(idx.shape[1]):
            if s == 8 or idx[b, s] == 0:      # no cloud-borne water
                continue
            frac = rng.uniform(0.0, 0.6, size=ms.chem.shape[1:]) * (rng.uniform(size=ms.chem.shape[1:]) < 0.5)
            extra.append(ms.chem[idx[b, s] - 1] * frac)
            idx_cw[b, s] = nsp + len(extra)
    both = copy.copy(ms)
    both.chem = np.concatenate([ms.chem, np.stack(extra)], axis=0)
    summed = copy.copy(ms)
    chem = np.ascontiguousarray(both.chem, dtype=dt)
    tot = chem[:nsp].copy()
    for b in range(idx.shape[0]):
        for s in range(idx.shape[1]):
            if idx_cw[b, s]:
                tot[idx[b, s] - 1] = chem[idx[b, s] - 1] + chem[idx_cw[b, s] - 1]
    summed.chem = tot
    ref = run_hip_optics(wa, st, summed, real_bytes=real_bytes)
    plain = run_hip_optics(wa, st, ms, real_bytes=real_bytes)
    assert np.abs(ref["tauaer2"] - plain["tauaer2"]).max() > 0
    for device in (True, False):
        got = run_hip_optics(wa, st, both, real_bytes=real_bytes, idx_cw=idx_cw, device=device)
        for n in got:
            np.testing.assert_array_equal(got[n], ref[n], err_msg=f"{n} device={device}")
        none = run_hip_optics(wa, st, both, real_bytes=real_bytes, device=device)
        for n in none:
            np.testing.assert_array_equal(none[n], plain[n], err_msg=n)
    with pytest.raises(wa.engine.WrfFatal, match="cloud-borne species index"):
        bad = idx_cw.copy()
        bad[0, 0] = both.chem.shape[0] + 1
        run_hip_optics(wa, st, both, idx_cw=bad)


@pytest.mark.parametrize("opt", [1, 2, 5])
def test_legendre_moments_match_oracle(hip_lib, wa, opt):
    """l2aer..l7aer (registry.chem:1382-1387): scattering-weighted Legendre moments of the phase function at the four
    shortwave wavelengths, against the oracle's quadrature; device-resident and host arrays alike; asking for them does
    not change the other outputs; the group is all six or none."""
    _init(wa, 4)
    st = wa.synth.make_wrf_state(5, 4, 9, seed=20260398)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20260399)
    names = SW_NAMES + LW_NAMES + LEG_NAMES
    got = run_hip_optics(wa, st, ms, aer_op_opt=opt, names=names)
    ref = run_oracle_optics(wa, st, ms, mix=MIX_OF[opt], legendre=True)
    mask = tile_mask(st, ms.alt.shape)
    for n in LEG_NAMES:
        for w in range(4):
            g, r = got[n][w][mask], ref[n][w][mask]
            assert np.isfinite(g).all() and np.abs(g - r).max() <= 1e-9, (n, w, float(np.abs(g - r).max()))
            assert (got[n][w][~mask] == SENTINEL).all()
        assert np.abs(got[n][:, mask]).max() <= 1.0
    assert got["l2aer"][1][mask].max() > 0.2          # forward-peaked phase functions
    plain = run_hip_optics(wa, st, ms, aer_op_opt=opt)
    for n in plain:
        np.testing.assert_array_equal(plain[n], got[n], err_msg=n)
    host = run_hip_optics(wa, st, ms, aer_op_opt=opt, names=LEG_NAMES, device=False)
    for n in LEG_NAMES:
        np.testing.assert_array_equal(host[n], got[n], err_msg=n)
    with pytest.raises(wa.engine.WrfFatal, match="complete or not at all"):
        run_hip_optics(wa, st, ms, names=["l2aer", "l3aer"])
