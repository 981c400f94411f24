"""CPU tests of the optical-averaging oracle.  The reference repository holds no arithmetic for this stage
(parity unpinned), so the oracle's Mie core is pinned on published known answers and on an
independent formulation instead."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def oo(oracle_mod):
    from oracle import oracle_optics
    return oracle_optics


# (m, x, Qext) -- Wiscombe (1979) NCAR/TN-140 test cases as tabulated by Du (2004, Appl. Opt. 43, table 1);
# our convention m = n + ik
WISCOMBE = [
    (0.75, 0.099, 7.417859e-06), (0.75, 0.101, 8.033538e-06), (0.75, 10.0, 2.232265), (0.75, 1000.0, 1.997908),
    (1.33 + 1e-5j, 1.0, 9.395198e-02), (1.33 + 1e-5j, 100.0, 2.101321),
    (1.5 + 1j, 0.055, 0.101491), (1.5 + 1j, 0.056, 0.103347), (1.5 + 1j, 1.0, 2.336321), (1.5 + 1j, 100.0, 2.097502),
    (10 + 10j, 1.0, 2.532993), (10 + 10j, 100.0, 2.071124),
]


def test_mie_bohren_huffman_appendix_case(oo):
    # Bohren & Huffman (1983) appendix A sample run: radius 0.525 um, wavelength 0.6328 um, m = 1.55
    qext, qsca, g, qback = oo.mie(2.0 * np.pi * 0.525 / 0.6328, 1.55)
    assert qext == pytest.approx(3.10543, abs=5e-6)
    assert qsca == pytest.approx(3.10543, abs=5e-6)
    assert qback == pytest.approx(2.92534, abs=5e-6)


@pytest.mark.parametrize("m,x,qext_ref", WISCOMBE)
def test_mie_wiscombe_cases(oo, m, x, qext_ref):
    qext = oo.mie(x, m)[0]
    assert qext == pytest.approx(qext_ref, rel=5e-6)   # tabulated to 6-7 significant digits


def _mie_scipy(x, m, nmax):
    """a_n, b_n straight from spherical Bessel functions (no recurrences of our own)."""
    from scipy.special import spherical_jn, spherical_yn
    n = np.arange(1, nmax + 1)
    z = m * x
    jx, djx = spherical_jn(n, x), spherical_jn(n, x, derivative=True)
    yx, dyx = spherical_yn(n, x), spherical_yn(n, x, derivative=True)
    jz, djz = spherical_jn(n, z), spherical_jn(n, z, derivative=True)
    psi_x, dpsi_x = x * jx, jx + x * djx
    xi_x, dxi_x = x * (jx + 1j * yx), (jx + 1j * yx) + x * (djx + 1j * dyx)
    psi_z, dpsi_z = z * jz, jz + z * djz
    a = (m * psi_z * dpsi_x - psi_x * dpsi_z) / (m * psi_z * dxi_x - xi_x * dpsi_z)
    b = (psi_z * dpsi_x - m * psi_x * dpsi_z) / (psi_z * dxi_x - m * xi_x * dpsi_z)
    qext = 2.0 / x**2 * np.sum((2 * n + 1) * (a + b).real)
    qsca = 2.0 / x**2 * np.sum((2 * n + 1) * (abs(a) ** 2 + abs(b) ** 2))
    gs = np.sum(n[:-1] * (n[:-1] + 2) / (n[:-1] + 1) * (a[:-1] * a[1:].conj() + b[:-1] * b[1:].conj()).real)
    gs += np.sum((2 * n + 1) / (n * (n + 1)) * (a * b.conj()).real)
    return qext, qsca, 4.0 / x**2 * gs / qsca


@pytest.mark.parametrize("m,x", [(1.33 + 1e-5j, 100.0), (1.33 + 1e-5j, 10.0), (1.52 + 1e-9j, 3.3), (1.85 + 0.71j, 7.0),
                                 (1.45 + 0.05j, 25.0), (1.7 + 0.55j, 0.02), (2.9 + 0.86j, 0.4), (1.33, 0.005)])
def test_mie_against_scipy_formulation(oo, m, x):
    nmax = int(x + 4.0 * x ** (1.0 / 3.0) + 2.0)
    qext, qsca, g = _mie_scipy(x, complex(m), nmax)
    o = oo.mie(x, m)
    assert o[0] == pytest.approx(qext, rel=1e-9)
    assert o[1] == pytest.approx(qsca, rel=1e-9)
    assert o[2] == pytest.approx(g, rel=1e-7, abs=1e-9)   # g ~ x^2 for tiny x: both forms cancel


def test_mie_limits(oo):
    # Rayleigh limit: Qsca -> 8/3 x^4 |(m^2-1)/(m^2+2)|^2, Qabs -> 4x Im((m^2-1)/(m^2+2)), g -> 0
    m, x = 1.5 + 0.1j, 1e-3
    al = (m * m - 1) / (m * m + 2)
    qext, qsca, g, _ = oo.mie(x, m)
    assert qsca == pytest.approx(8.0 / 3.0 * x**4 * abs(al) ** 2, rel=1e-5)
    assert qext - qsca == pytest.approx(4.0 * x * al.imag, rel=1e-5)
    assert abs(g) < 1e-5
    # extinction paradox: Qext -> 2 for large absorbing spheres; non-absorbing sphere: Qext == Qsca
    assert oo.mie(220.0, 1.5 + 0.3j)[0] == pytest.approx(2.0, abs=0.08)
    qe, qs, _, _ = oo.mie(12.0, 1.45)
    assert qe == pytest.approx(qs, rel=1e-12)
    for x in (0.05, 0.7, 4.0, 31.0, 180.0):
        qe, qs, g, qb = oo.mie(x, 1.6 + 0.2j)
        assert 0.0 < qs < qe and -1.0 < g < 1.0 and qb >= 0.0


def _tables(oo, wa, nbin):
    wl, ref, den = wa.engine_optics.table_info()
    return oo.make_tables(nbin, 0.0390625, 10.0, wl, ref, den)


def test_cell_properties(oo, wa):
    tb = _tables(oo, wa, 4)
    rng = np.random.default_rng(7)
    mass = rng.uniform(0.0, 3.0, size=(4, 9))
    num = np.array([4e9, 6e8, 2e7, 1e5])
    sw, lw = oo.cell(tb, mass, num, alt=0.85, dz=120.0)
    assert np.all(sw[:, 0] > 0) and np.all((sw[:, 1] > 0) & (sw[:, 1] <= 1)) and np.all(np.abs(sw[:, 2]) < 1)
    assert np.all(lw[:, 0] >= 0)
    # tau is extensive: doubling mass and number doubles tau and leaves omega, g unchanged
    sw2, lw2 = oo.cell(tb, 2 * mass, 2 * num, alt=0.85, dz=120.0)
    np.testing.assert_allclose(sw2[:, 0], 2 * sw[:, 0], rtol=1e-12)
    np.testing.assert_allclose(sw2[:, 1:3], sw[:, 1:3], rtol=1e-12)
    np.testing.assert_allclose(lw2[:, 0], 2 * lw[:, 0], rtol=1e-12)
    # tau = ext * dz
    np.testing.assert_allclose(sw[:, 0], sw[:, 3] * 1e-3 * 120.0, rtol=1e-13)
    # no aerosol -> zeros
    sw0, lw0 = oo.cell(tb, np.zeros((4, 9)), np.zeros(4), alt=0.85, dz=120.0)
    assert not sw0.any() and not lw0.any()
    # black carbon lowers the single-scattering albedo; Maxwell-Garnett differs from volume mixing only with BC
    m_bc = mass.copy(); m_bc[:, 7] += 2.0
    assert np.all(oo.cell(tb, m_bc, num, 0.85, 120.0)[0][:, 1] < sw[:, 1])
    m_nobc = mass.copy(); m_nobc[:, 7] = 0.0
    a, _ = oo.cell(tb, m_nobc, num, 0.85, 120.0, mix=0)
    b, _ = oo.cell(tb, m_nobc, num, 0.85, 120.0, mix=1)
    np.testing.assert_allclose(a, b, rtol=1e-12)
    c, _ = oo.cell(tb, mass, num, 0.85, 120.0, mix=1)
    assert np.max(np.abs(c[:, 1] - sw[:, 1])) > 1e-5
    # number far outside the section is clipped: result depends on volume only
    hi, _ = oo.cell(tb, mass, num * 1e6, 0.85, 120.0)
    hi2, _ = oo.cell(tb, mass, num * 1e7, 0.85, 120.0)
    np.testing.assert_allclose(hi, hi2, rtol=1e-12)


def test_optics_abi_exports_and_refusal(wa):
    import ctypes as C
    lib = wa.engine.load_library()
    for name in wa.engine_optics.EXPORTS_OPTICS:
        assert hasattr(lib, name), name
    wl, ref, den = wa.engine_optics.table_info()
    assert wl[0] == 0.3 and wl[3] == 0.999 and wl[4] == pytest.approx(1e4 / 180.0)
    assert ref.shape == (9, 20, 2) and np.all(ref[..., 0] > 1.0) and np.all(ref[..., 1] >= 0) and np.all(den > 0)
    import torch
    if not torch.cuda.is_available():
        st = wa.synth.make_wrf_state(2, 2, 5, seed=3)
        ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4)
        dims = dict(ims=1, ime=2, jms=1, jme=2, kms=1, kme=6, its=1, ite=2, jts=1, jte=2, kts=1, kte=5)
        out = {f"tauaerlw{b}": np.zeros_like(ms.alt) for b in range(1, 17)}
        with pytest.raises(wa.engine.WrfFatal, match="rrtmg_hip_optics_init has not been called"):
            wa.engine_optics.optical_averaging(chem=ms.chem, alt=ms.alt, dz8w=ms.dz8w, idx=ms.idx, **dims, **out)
        with pytest.raises(wa.engine.WrfFatal, match="no HIP device"):
            wa.engine_optics.optics_init(4)


def test_upward_recurrence_window(oo):
    """The device carries the logarithmic derivative upward (D_0 = cot y) where x >= 0.3 and Im(m) x < 10
    (csrc/optics_kernel.hip: mie_upward_ok); the oracle specifies the downward recurrence throughout.  Inside the
    window the two agree far below the stage's tolerance of 1e-8; outside they do not (the window is needed)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes"))
    from mie_updown import mie
    rng = np.random.default_rng(20260420)
    n = 6000
    x = 10 ** rng.uniform(np.log10(0.3), np.log10(225.0), n)
    m = rng.uniform(1.1, 2.6, n) + 1j * 10 ** rng.uniform(-9, 0.0, n)
    inside = m.imag * x < 10.0
    with np.errstate(all="ignore"):
        dn, up = mie(x, m, False), mie(x, m, True)
    for a, b, tol in zip(dn, up, (1e-10, 1e-10, 1e-10, 1e-9)):
        rel = np.abs(a - b)[inside] / np.maximum(np.abs(a[inside]), 1e-300)
        assert rel.max() < tol
    # the numpy downward series is the oracle's series (spot check), so the statement is about the oracle
    for i in range(0, n, 997):
        q = oo.mie(float(x[i]), complex(m[i]))
        assert q[0] == pytest.approx(dn[0][i], rel=1e-11) and q[1] == pytest.approx(dn[1][i], rel=1e-11)
    # strongly absorbing large particles: upward fails, the device keeps the downward path there
    bad = m.imag * x > 40.0
    rel = np.abs(dn[1] - up[1])[bad] / np.abs(dn[1][bad])
    assert np.nanmax(np.where(np.isfinite(rel), rel, 1.0)) > 1e-6


# ---- coated sphere (aer_op_opt = 5, shell_exact) ----
def test_coated_sphere_bohren_huffman_appendix_b_case(oo):
    """The example printed with BHCOAT (Bohren & Huffman 1983, appendix B): core radius 0.171 um, n = 1.59 + 0.66i; total
    radius 6.265 um, shell n = 1.409 + 0.1747i; wavelength 3 um -> QEXT 2.32803, QSCA 1.14341, QBACK 2.85099e-2."""
    k = 2.0 * np.pi / 3.0
    qext, qsca, g, qback = oo.mie_coated(k * 0.171, k * 6.265, 1.59 + 0.66j, 1.409 + 0.1747j)
    assert abs(qext - 2.32803) < 6e-6 and abs(qsca - 1.14341) < 6e-6 and abs(qback - 2.85099e-2) < 6e-8
    assert 0.0 < g < 1.0


def _coated_scipy(x, y, m1, m2, nmax):
    """Aden & Kerker's coefficients (Bohren & Huffman eq. 8.2) straight from spherical Bessel functions."""
    from scipy.special import spherical_jn, spherical_yn
    n = np.arange(1, nmax + 1)

    def rb(z):   # psi, psi', chi, chi' with chi = -z y_n(z)
        j, dj, yy, dy = spherical_jn(n, z), spherical_jn(n, z, derivative=True), spherical_yn(n, z), spherical_yn(n, z, derivative=True)
        return z * j, j + z * dj, -z * yy, -(yy + z * dy)

    p1, dp1, _, _ = rb(m1 * x)
    p2, dp2, c2, dc2 = rb(m2 * x)
    py, dpy, cy, dcy = rb(m2 * y)
    pv, dpv, cv, dcv = rb(y + 0j)
    xi, dxi = pv - 1j * cv, dpv - 1j * dcv
    A = (m2 * p2 * dp1 - m1 * dp2 * p1) / (m2 * c2 * dp1 - m1 * dc2 * p1)
    B = (m2 * p1 * dp2 - m1 * p2 * dp1) / (m2 * dc2 * p1 - m1 * dp1 * c2)
    a = (pv * (dpy - A * dcy) - m2 * dpv * (py - A * cy)) / (xi * (dpy - A * dcy) - m2 * dxi * (py - A * cy))
    b = (m2 * pv * (dpy - B * dcy) - dpv * (py - B * cy)) / (m2 * xi * (dpy - B * dcy) - dxi * (py - B * cy))
    qext = 2.0 / y**2 * np.sum((2 * n + 1) * (a + b).real)
    qsca = 2.0 / y**2 * np.sum((2 * n + 1) * (abs(a) ** 2 + abs(b) ** 2))
    gs = np.sum(n[:-1] * (n[:-1] + 2) / (n[:-1] + 1) * (a[:-1] * a[1:].conj() + b[:-1] * b[1:].conj()).real)
    gs += np.sum((2 * n + 1) / (n * (n + 1)) * (a * b.conj()).real)
    return qext, qsca, 4.0 / y**2 * gs / qsca


@pytest.mark.parametrize("x,y,m1,m2", [(0.3, 0.5, 1.85 + 0.71j, 1.45 + 1e-3j), (1.0, 2.5, 1.85 + 0.71j, 1.33 + 1e-8j),
                                       (2.0, 8.0, 1.95 + 0.79j, 1.53 + 0.02j), (0.02, 0.05, 1.85 + 0.71j, 1.5 + 0.1j),
                                       (7.0, 8.0, 1.85 + 0.71j, 1.4 + 0.0j), (0.5, 12.0, 1.85 + 0.71j, 1.33 + 1e-6j)])
def test_coated_sphere_against_scipy_formulation(oo, x, y, m1, m2):
    nmax = int(y + 4.0 * y ** (1.0 / 3.0) + 2.0)
    ref = _coated_scipy(x, y, complex(m1), complex(m2), nmax)
    got = oo.mie_coated(x, y, m1, m2)
    for g, r in zip(got[:3], ref):
        assert abs(g - r) <= 2e-9 * abs(r), (got, ref)


def test_coated_sphere_limits(oo):
    """Equal indices, a vanishing core and a vanishing shell give the homogeneous sphere; an absorbing core lowers the
    single-scattering albedo of a weakly absorbing particle."""
    for x, y, m in [(0.5, 1.0, 1.5 + 0.01j), (3.0, 10.0, 1.33 + 1e-6j), (0.001, 0.002, 1.8 + 0.7j), (20.0, 100.0, 1.5 + 1e-3j)]:
        a, b = oo.mie_coated(x, y, m, m), oo.mie(y, m)
        assert max(abs(p - q) for p, q in zip(a, b)) <= 1e-12 * max(abs(q) for q in b)
    a, b = oo.mie_coated(1e-6, 5.0, 1.85 + 0.71j, 1.5 + 0.01j), oo.mie(5.0, 1.5 + 0.01j)
    assert max(abs(p - q) / abs(q) for p, q in zip(a, b)) < 1e-12
    a, b = oo.mie_coated(5.0 * (1 - 1e-12), 5.0, 1.85 + 0.71j, 1.5 + 0.01j), oo.mie(5.0, 1.85 + 0.71j)
    assert max(abs(p - q) / abs(q) for p, q in zip(a, b)) < 1e-10
    qe0, qs0, _, _ = oo.mie(2.0, 1.45 + 1e-6j)
    qe1, qs1, _, _ = oo.mie_coated(1.0, 2.0, 1.85 + 0.71j, 1.45 + 1e-6j)
    assert qs1 / qe1 < 0.9 < qs0 / qe0


# ---- Legendre moments of the phase function (l2aer..l7aer) ----
def test_legendre_moments_known_answers(oo):
    """chi_0 = 1; chi_1 -- from the quadrature of the intensity -- equals the asymmetry parameter of the series (an
    independent formula); the Rayleigh phase function 3/4 (1 + mu^2) has chi_2 = 1/10 and nothing else; a coated sphere
    likewise."""
    c = oo.moments(0.01, 1.5 + 0.0j)
    assert abs(c[0] - 1.0) < 1e-14 and abs(c[2] - 0.1) < 1e-5 and np.abs(c[3:]).max() < 1e-5
    for x, m in [(0.3, 1.45 + 1e-3j), (1.0, 1.5 + 0.01j), (10.0, 1.33 + 1e-6j), (50.0, 1.55 + 0.006j), (3.0, 1.85 + 0.71j),
                 (224.0, 1.4 + 1e-4j)]:
        g = oo.mie(x, m)[2]
        c = oo.moments(x, m)
        assert abs(c[0] - 1.0) < 1e-14 and abs(c[1] - g) < 1e-11, (x, m, c[1], g)
        assert (np.abs(c) <= 1.0 + 1e-12).all()
    g = oo.mie_coated(2.0, 5.0, 1.85 + 0.71j, 1.5 + 0.01j)[2]
    c = oo.moments(5.0, 1.5 + 0.01j, core=(2.0, 1.85 + 0.71j))
    assert abs(c[1] - g) < 1e-11


def test_legendre_moments_reconstruct_the_phase_function(oo):
    """p(mu) = sum_k (2k+1) chi_k P_k(mu): for a small particle (x = 0.8: moments beyond k = 7 below 1e-5) the eight
    moments rebuild the phase function of the amplitude functions, evaluated here with scipy's Bessel functions."""
    from numpy.polynomial import legendre as L
    x, m = 0.8, 1.5 + 0.02j
    c = oo.moments(x, m)
    nmax = int(x + 4.0 * x ** (1.0 / 3.0) + 2.0)
    from scipy.special import spherical_jn, spherical_yn
    n = np.arange(1, nmax + 1)
    z = m * x
    jx, djx, yx, dyx = spherical_jn(n, x), spherical_jn(n, x, derivative=True), spherical_yn(n, x), spherical_yn(n, x, derivative=True)
    jz, djz = spherical_jn(n, z), spherical_jn(n, z, derivative=True)
    psi_x, dpsi_x = x * jx, jx + x * djx
    xi_x, dxi_x = x * (jx + 1j * yx), (jx + 1j * yx) + x * (djx + 1j * dyx)
    psi_z, dpsi_z = z * jz, jz + z * djz
    a = (m * psi_z * dpsi_x - psi_x * dpsi_z) / (m * psi_z * dxi_x - xi_x * dpsi_z)
    b = (psi_z * dpsi_x - m * psi_x * dpsi_z) / (psi_z * dxi_x - m * xi_x * dpsi_z)
    mu = np.linspace(-1, 1, 41)
    pi0, pi1 = np.zeros_like(mu), np.ones_like(mu)
    s1 = np.zeros_like(mu, dtype=complex); s2 = np.zeros_like(mu, dtype=complex)
    for k in range(1, nmax + 1):
        tau = k * mu * pi1 - (k + 1) * pi0
        cn = (2 * k + 1) / (k * (k + 1))
        s1 += cn * (a[k - 1] * pi1 + b[k - 1] * tau)
        s2 += cn * (a[k - 1] * tau + b[k - 1] * pi1)
        pi0, pi1 = pi1, ((2 * k + 1) * mu * pi1 - (k + 1) * pi0) / k
    inten = 0.5 * (abs(s1) ** 2 + abs(s2) ** 2)
    qsca = 2.0 / x**2 * np.sum((2 * n + 1) * (abs(a) ** 2 + abs(b) ** 2))
    p_true = 4.0 * inten / (x**2 * qsca)                        # integral over mu = 2
    p_fit = L.legval(mu, [(2 * k + 1) * c[k] for k in range(8)])
    assert np.abs(p_fit - p_true).max() < 2e-5
