"""Study: logarithmic derivative D_n(mx) by upward recurrence (from D_0 = cot y) against the downward recurrence the
optics stage specifies -- how far apart are the Mie efficiencies, as a function of x, Re m, Im m?
(numpy, vectorised over cases; run on the build host)"""
import numpy as np


def mie(x, m, upward):
    x = np.asarray(x, float); m = np.asarray(m, complex)
    y = x * m
    xstop = x + 4. * np.cbrt(x) + 2.
    nstop = xstop.astype(int)
    base = np.maximum(xstop, np.abs(y))
    nmx = (base + np.where(base > 8., 8. * np.cbrt(base), 0.)).astype(int) + 16
    N = int(nmx.max()) + 2
    ncase = x.size
    D = np.zeros((N + 1, ncase), complex)
    if upward:
        D[0] = np.cos(y) / np.sin(y)
        for n in range(1, int(nstop.max()) + 1):
            D[n] = 1. / (n / y - D[n - 1]) - n / y
    else:
        d = np.zeros(ncase, complex)
        for n in range(N, 1, -1):
            live = n <= nmx
            d = np.where(live, n / y - 1. / (d + n / y), 0.)
            D[n - 1] = d
    psi0 = np.cos(x); psi1 = np.sin(x); chi0 = -np.sin(x); chi1 = np.cos(x)
    qs = np.zeros(ncase); qe = np.zeros(ncase); gs = np.zeros(ncase); xb = np.zeros(ncase, complex)
    an1 = np.zeros(ncase, complex); bn1 = np.zeros(ncase, complex)
    for n in range(1, int(nstop.max()) + 1):
        live = n <= nstop
        psi = (2. * n - 1.) / x * psi1 - psi0
        chi = (2. * n - 1.) / x * chi1 - chi0
        xi = psi - 1j * chi; xi1 = psi1 - 1j * chi1
        t = D[n] / m + n / x
        an = (t * psi - psi1) / (t * xi - xi1)
        t = m * D[n] + n / x
        bn = (t * psi - psi1) / (t * xi - xi1)
        an = np.where(live, an, 0.); bn = np.where(live, bn, 0.)
        qs += (2 * n + 1) * (abs(an) ** 2 + abs(bn) ** 2)
        qe += (2 * n + 1) * (an + bn).real
        gs += (2. * n + 1.) / (n * (n + 1.)) * (an * np.conj(bn)).real
        if n > 1:
            gs += (n - 1.) * (n + 1.) / n * (an1 * np.conj(an) + bn1 * np.conj(bn)).real
        xb += (-1.) ** n * (2 * n + 1) * (an - bn)
        an1 = np.where(live, an, an1); bn1 = np.where(live, bn, bn1)
        psi0, psi1, chi0, chi1 = psi1, np.where(live, psi, psi1), chi1, np.where(live, chi, chi1)
        psi0 = np.where(live, psi0, psi0)
    return 2 / x ** 2 * qe, 2 / x ** 2 * qs, 2 * gs / np.maximum(qs, 1e-300), abs(xb) ** 2 / x ** 2


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    n = 40000
    x = 10 ** rng.uniform(np.log10(2.), np.log10(225.), n)
    mre = rng.uniform(1.3, 2.0, n)
    mim = 10 ** rng.uniform(-9, np.log10(0.75), n)
    m = mre + 1j * mim
    with np.errstate(all="ignore"):
        dn = mie(x, m, False); up = mie(x, m, True)
    names = ["qext", "qsca", "g", "qback"]
    rel = np.zeros(n)
    for a, b, nm in zip(dn, up, names):
        r = np.abs(a - b) / np.maximum(np.abs(a), 1e-300)
        r = np.where(np.isfinite(r), r, 1.)
        if nm != "qback":
            rel = np.maximum(rel, r)
        print(nm, "max rel diff", r.max(), "99.9%", np.quantile(r, 0.999))
    bx = x * mim
    for lo, hi in [(0, .1), (.1, 1), (1, 3), (3, 10), (10, 30), (30, 200)]:
        s = (bx >= lo) & (bx < hi)
        for xl, xh in [(2, 5), (5, 8), (8, 20), (20, 225)]:
            t = s & (x >= xl) & (x < xh)
            if t.any():
                print(f"mim*x in [{lo},{hi}) x in [{xl},{xh}): n={t.sum()} max rel {rel[t].max():.2e}")
    s = bx < 10.
    for a, b, nm in zip(dn, up, names):
        r = np.abs(a - b)[s] / np.maximum(np.abs(a[s]), 1e-300)
        print("mim*x < 10:", nm, "max rel", r.max(), " max abs", np.abs(a - b)[s].max())
