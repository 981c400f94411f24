"""Modal (MADE/SORGAM-type) aerosol -> sections (modal_sections.hip, rrtmg_hip_modal_to_sections): the oracle's
restatement on the CPU (conservation, log-normal shares), the HIP kernel against it on the GPU, and the chain
modal tracers -> sections -> optical averaging end to end.  Parity unpinned: the reference holds no arithmetic for
this stage (oracle/optics_oracle.c)."""
import math

import numpy as np
import pytest

from helpers import SENTINEL
from helpers_optics import oracle_tables, optics_dims


def _cell(oo, tab, mass, num, sigma):
    import ctypes as C
    lib = oo._lib()
    dp = C.POINTER(C.c_double)
    lib.optics_oracle_modal_cell.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp]
    nmode = len(num)
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    num = np.ascontiguousarray(num, dtype=np.float64)
    sig = np.ascontiguousarray(sigma, dtype=np.float64)
    sm, sn = np.zeros((tab.nbin, 9)), np.zeros(tab.nbin)
    lib.optics_oracle_modal_cell(C.addressof(tab), nmode, mass.ctypes.data_as(dp), num.ctypes.data_as(dp),
                                 sig.ctypes.data_as(dp), sm.ctypes.data_as(dp), sn.ctypes.data_as(dp))
    return sm, sn


def test_modal_oracle_conserves_and_places_the_modes(wa, oracle_mod):
    from oracle import oracle_optics as oo
    tab = oracle_tables(wa, 8)
    rng = np.random.default_rng(3)
    mass = rng.uniform(0.1, 5.0, (3, 9))
    dens = np.array([tab.dens[s] for s in range(9)])
    sigma = (1.7, 2.0, 2.2)
    dg_um = (0.05, 0.25, 1.2)
    num = []
    for m in range(3):
        vol = (mass[m, :8] / dens[:8]).sum() * 1e-6     # ug / (g cm-3) = 1e-6 cm3
        ls = math.log(sigma[m])
        num.append(vol / (math.pi / 6 * (dg_um[m] * 1e-4) ** 3 * math.exp(4.5 * ls * ls)))
    sm, sn = _cell(oo, tab, mass, num, sigma)
    np.testing.assert_allclose(sm.sum(axis=0), mass.sum(axis=0), rtol=1e-13)       # mass of every species
    np.testing.assert_allclose(sn.sum(), sum(num), rtol=1e-13)                     # number
    # one mode at a time: the number peaks in the section that holds dgn, the mass in the one that holds dgv
    lo = np.array([tab.dlo_um[b] for b in range(8)]); hi = np.array([tab.dhi_um[b] for b in range(8)])
    for m in range(3):
        one_m, one_n = np.zeros((3, 9)), np.zeros(3)
        one_m[m], one_n[m] = mass[m], num[m]
        s1, n1 = _cell(oo, tab, one_m, one_n, sigma)
        dgv = dg_um[m] * math.exp(3 * math.log(sigma[m]) ** 2)
        assert lo[n1.argmax()] <= max(dg_um[m], lo[0]) <= hi[n1.argmax()] * (1 + 1e-12)
        assert lo[s1[:, 0].argmax()] <= min(dgv, hi[-1]) <= hi[s1[:, 0].argmax()]
        # shares against the error function directly
        cdf = lambda d, dg: 0.5 * (1 + math.erf(math.log(d / dg) / (math.sqrt(2) * math.log(sigma[m]))))
        b = 3
        np.testing.assert_allclose(n1[b] / num[m], cdf(hi[b], dg_um[m]) - cdf(lo[b], dg_um[m]), rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(s1[b, 0] / mass[m, 0], cdf(hi[b], dgv) - cdf(lo[b], dgv), rtol=1e-9, atol=1e-15)
    # an empty mode (no number) contributes nothing
    sm0, sn0 = _cell(oo, tab, mass, [num[0], 0.0, num[2]], sigma)
    np.testing.assert_allclose(sm0.sum(axis=0), mass[0] + mass[2], rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("nbin,dtype,tol", [(8, np.float64, 1e-11), (4, np.float64, 1e-11), (8, np.float32, 3e-6)])
def test_modal_to_sections_matches_oracle(hip_lib, wa, nbin, dtype, tol):
    import torch
    from oracle import oracle_optics as oo
    wa.engine_optics.optics_init(nbin, device=0)
    st = wa.synth.make_wrf_state(6, 4, 20, seed=20260801)
    ms = wa.synth_mosaic.make_made_state(st, seed=20260802)
    d = optics_dims(st, dict(its=2, ite=5, jts=1, jte=3))
    chem = ms.chem.astype(dtype)
    sec = np.full((nbin * 10,) + chem.shape[1:], SENTINEL, dtype=dtype)
    t_chem, t_sec = torch.from_numpy(chem).cuda(), torch.from_numpy(sec).cuda()
    wa.engine_optics.modal_to_sections(**d, chem=t_chem, idx=ms.idx, sigma=ms.sigma, chem_sec=t_sec)
    got = t_sec.cpu().numpy().astype(np.float64)
    ref = oo.modal_to_sections(oracle_tables(wa, nbin), chem.astype(np.float64), ms.idx, ms.sigma, d)
    inside = np.zeros(chem.shape[1:], bool)
    inside[d["jts"] - 1:d["jte"], d["kts"] - 1:d["kte"], d["its"] - 1:d["ite"]] = True
    for n in range(nbin * 10):
        np.testing.assert_allclose(got[n][inside], ref[n][inside], rtol=tol, atol=1e-300, err_msg=str(n))
        assert (got[n][~inside] == SENTINEL).all()
    # host-pointer mode gives the same bits
    sec_h = np.full_like(sec, SENTINEL)
    wa.engine_optics.modal_to_sections(**d, chem=chem, idx=ms.idx, sigma=ms.sigma, chem_sec=sec_h)
    np.testing.assert_array_equal(sec_h, t_sec.cpu().numpy())
    np.testing.assert_array_equal(wa.engine_optics.modal_section_idx(nbin), np.arange(1, nbin * 10 + 1).reshape(nbin, 10))


@pytest.mark.gpu
def test_modal_aerosol_end_to_end(hip_lib, wa):
    """MADE-type tracers -> sections -> tau / omega / g: the device chain against the oracle chain."""
    import torch
    from oracle import oracle_optics as oo
    nbin = 8
    wa.engine_optics.optics_init(nbin, device=0)
    st = wa.synth.make_wrf_state(5, 3, 20, seed=20260811)
    ms = wa.synth_mosaic.make_made_state(st, seed=20260812)
    d = optics_dims(st)
    t_chem = torch.from_numpy(ms.chem).cuda()
    t_sec = torch.zeros((nbin * 10,) + ms.chem.shape[1:], dtype=torch.float64, device="cuda")
    wa.engine_optics.modal_to_sections(**d, chem=t_chem, idx=ms.idx, sigma=ms.sigma, chem_sec=t_sec)
    idx = wa.engine_optics.modal_section_idx(nbin)
    names = [f"{s}{w}" for s in ("tauaer", "waer", "gaer") for w in range(1, 5)] + [f"tauaerlw{b}" for b in range(1, 17)]
    outs = {n: torch.zeros(ms.alt.shape, dtype=torch.float64, device="cuda") for n in names}
    wa.engine_optics.optical_averaging(idx=idx, aer_op_opt=1, **d, chem=t_sec, alt=torch.from_numpy(ms.alt).cuda(),
                                       dz8w=torch.from_numpy(ms.dz8w).cuda(), **outs)
    tab = oracle_tables(wa, nbin)
    sec_ref = oo.modal_to_sections(tab, ms.chem, ms.idx, ms.sigma, d)
    ref = oo.tile(tab, sec_ref, ms.alt, ms.dz8w, idx, d)
    nk = st.nk
    for n in names:
        g, r = outs[n].cpu().numpy()[:, :nk, :], ref[n][:, :nk, :]
        assert np.isfinite(g).all()
        np.testing.assert_allclose(g, r, rtol=1e-8, atol=1e-30, err_msg=n)
    assert outs["tauaer2"].max().item() > 1e-4
