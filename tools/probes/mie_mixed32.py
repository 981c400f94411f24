"""Study for the mixed-precision optics path: the Mie series with psi_n carried in fp64 and everything else in fp32
(D_n by the upward recurrence from D_0 = cot y, or by a short downward ladder for small arguments), against the fp64
downward specification.  numpy complex64 / float32 stand in for the packed fp32 arithmetic of the kernels (same
precision, not the same operation order).  Run on the build host: python tools/probes/mie_mixed32.py"""
import numpy as np

from mie_updown import mie

f32, c64 = np.float32, np.complex64


def mie32(x, m, mode, ladder_extra=0):
    """mode 'up': D_n upward in fp32; 'lad': D_n downward in fp32 from nstop + steps (contraction rule)"""
    x = np.asarray(x, float); m = np.asarray(m, complex)
    y = x * m
    nstop = (x + 4. * np.cbrt(x) + 2.).astype(int)
    nmax = int(nstop.max())
    ncase = x.size
    y32 = y.astype(c64); m32 = m.astype(c64); minv = (1 / m).astype(c64); yinv = (1 / y).astype(c64)
    xinv32 = (1 / x).astype(f32)
    D = np.zeros((nmax + 2, ncase), c64)
    if mode == "up":
        D[0] = (np.cos(y) / np.sin(y)).astype(c64)       # cot y in fp64, then rounded
        for n in range(1, nmax + 1):
            ny = f32(n) * yinv
            D[n] = c64(1) / (ny - D[n - 1]) - ny
    else:
        g = 2. * np.log((2 * nstop + 3) / np.abs(y))
        steps = (18.5 / g).astype(int) + 2 + ladder_extra
        nstart = np.maximum(nstop + steps, 9)
        d = np.zeros(ncase, c64)
        for n in range(int(nstart.max()), 1, -1):
            ny = f32(n) * yinv
            d = np.where(n <= nstart, ny - c64(1) / (d + ny), c64(0)).astype(c64)
            if n - 1 <= nmax:
                D[n - 1] = d
    psi0 = np.cos(x); psi1 = np.sin(x)                   # fp64
    chi0 = (-np.sin(x)).astype(f32); chi1 = np.cos(x).astype(f32)
    qs = np.zeros(ncase, f32); qe = np.zeros(ncase, f32); gs = np.zeros(ncase, f32)
    an1 = np.zeros(ncase, c64); bn1 = np.zeros(ncase, c64)
    for n in range(1, nmax + 1):
        live = n <= nstop
        psi = (2. * n - 1.) / x * psi1 - psi0
        chi = (f32(2 * n - 1) * xinv32 * chi1 - chi0).astype(f32)
        p = psi.astype(f32); p1 = psi1.astype(f32)
        xi = (p - 1j * chi).astype(c64); xi1 = (p1 - 1j * chi1).astype(c64)
        nx = (f32(n) * xinv32).astype(f32)
        t = (D[n] * minv + nx).astype(c64)
        an = ((t * p - p1) / (t * xi - xi1)).astype(c64)
        t = (m32 * D[n] + nx).astype(c64)
        bn = ((t * p - p1) / (t * xi - xi1)).astype(c64)
        an = np.where(live, an, c64(0)); bn = np.where(live, bn, c64(0))
        qs += (f32(2 * n + 1) * (np.abs(an) ** 2 + np.abs(bn) ** 2)).astype(f32)
        qe += (f32(2 * n + 1) * (an + bn).real).astype(f32)
        gs += (f32((2. * n + 1.) / (n * (n + 1.))) * (an * np.conj(bn)).real).astype(f32)
        if n > 1:
            gs += (f32((n - 1.) * (n + 1.) / n) * (an1 * np.conj(an) + bn1 * np.conj(bn)).real).astype(f32)
        an1 = np.where(live, an, an1); bn1 = np.where(live, bn, bn1)
        psi0, psi1 = psi1, np.where(live, psi, psi1)
        chi0, chi1 = chi1, np.where(live, chi, chi1)
    return 2 / x ** 2 * qe, 2 / x ** 2 * qs, 2 * gs / np.maximum(qs, 1e-30)


def report(tag, x, m, mode, **kw):
    with np.errstate(all="ignore"):
        ref = mie(x, m, False)
        got = mie32(x, m, mode, **kw)
    qe, qs, g = ref[0], ref[1], ref[2]
    e_ext = np.abs(got[0] - qe) / qe
    e_abs = np.abs((got[0] - got[1]) - (qe - qs)) / np.maximum(qe - qs, 1e-300)
    e_sca = np.abs(got[1] - qs) / qe             # against the extinction: what omega sees
    e_g = np.abs(got[2] - g)
    for nm, e in (("ext", e_ext), ("abs", e_abs), ("sca/ext", e_sca), ("g", e_g)):
        e = np.where(np.isfinite(e), e, 1.)
        print(f"{tag:34s} {nm:8s} max {e.max():.2e}  99.9% {np.quantile(e, 0.999):.2e}  median {np.median(e):.1e}")
    return np.maximum(np.where(np.isfinite(e_ext), e_ext, 1.), np.where(np.isfinite(e_g), e_g, 1.))


if __name__ == "__main__":
    rng = np.random.default_rng(3)
    n = 30000
    mre = rng.uniform(1.3, 2.0, n)
    mim = 10 ** rng.uniform(-9, np.log10(0.75), n)
    m = mre + 1j * mim
    for lo, hi in [(0.05, 0.3), (0.3, 1.), (1., 3.), (3., 10.), (10., 30.), (30., 100.), (100., 225.)]:
        x = 10 ** rng.uniform(np.log10(lo), np.log10(hi), n)
        e = report(f"up   x in [{lo},{hi})", x, m, "up")
        b = x * mim
        for bl, bh in [(0, .5), (.5, 1), (1, 2), (2, 4), (4, 10)]:
            s = (b >= bl) & (b < bh)
            if s.any():
                print(f"      Im(m) x in [{bl},{bh}): n={s.sum():6d} worst(ext, g) {e[s].max():.2e}")
    # the small-argument ladder (longwave): |y| <= 2, nstop <= 8
    x = 10 ** rng.uniform(np.log10(0.0016), np.log10(1.5), n)
    mre = rng.uniform(1.1, 2.6, n); mim = 10 ** rng.uniform(-4, np.log10(1.2), n)
    m = mre + 1j * mim
    s = (np.abs(x * m) <= 2.) & ((x + 4 * np.cbrt(x) + 2).astype(int) <= 8)
    report("ladder |y| <= 2", x[s], m[s], "lad")
    report("ladder |y| <= 2, one step less", x[s], m[s], "lad", ladder_extra=-1)


def mie32_polefree(x, m):
    """psi_n(y) of the complex argument upward in fp32 (scaled by exp(-Im y)), a_n, b_n in the pole-free form
    a_n = [m psi_n(y) psi_n'(x) - psi_n(x) psi_n'(y)] / [m psi_n(y) xi_n'(x) - xi_n(x) psi_n'(y)] (psi_n' = psi_(n-1) - n/z psi_n);
    psi_n(x) in fp64, chi_n(x) fp32."""
    x = np.asarray(x, float); m = np.asarray(m, complex)
    y = x * m
    nstop = (x + 4. * np.cbrt(x) + 2.).astype(int)
    nmax = int(nstop.max()); ncase = x.size
    m32 = m.astype(c64); yinv = (1 / y).astype(c64); xinv32 = (1 / x).astype(f32)
    sc = np.exp(-np.abs(y.imag))
    A0 = (np.cos(y) * sc).astype(c64); A1 = (np.sin(y) * sc).astype(c64)     # psi_(-1)(y), psi_0(y), scaled
    psi0 = np.cos(x); psi1 = np.sin(x)
    chi0 = (-np.sin(x)).astype(f32); chi1 = np.cos(x).astype(f32)
    qs = np.zeros(ncase, f32); qe = np.zeros(ncase, f32); gs = np.zeros(ncase, f32)
    an1 = np.zeros(ncase, c64); bn1 = np.zeros(ncase, c64)
    for n in range(1, nmax + 1):
        live = n <= nstop
        A = (f32(2 * n - 1) * yinv * A1 - A0).astype(c64)          # psi_n(y)
        B = (A1 - f32(n) * yinv * A).astype(c64)                    # psi_n'(y)
        psi = (2. * n - 1.) / x * psi1 - psi0
        chi = (f32(2 * n - 1) * xinv32 * chi1 - chi0).astype(f32)
        p = psi.astype(f32); p1 = psi1.astype(f32)
        nx = (f32(n) * xinv32).astype(f32)
        dp = (p1 - nx * p).astype(f32)                              # psi_n'(x)
        xi = (p - 1j * chi).astype(c64); dxi = ((p1 - 1j * chi1) - nx * xi).astype(c64)
        mA = (m32 * A).astype(c64); mB = (m32 * B).astype(c64)
        an = ((mA * dp - B * p) / (mA * dxi - B * xi)).astype(c64)
        bn = ((A * dp - mB * p) / (A * dxi - mB * xi)).astype(c64)
        an = np.where(live, an, c64(0)); bn = np.where(live, bn, c64(0))
        qs += (f32(2 * n + 1) * (np.abs(an) ** 2 + np.abs(bn) ** 2)).astype(f32)
        qe += (f32(2 * n + 1) * (an + bn).real).astype(f32)
        gs += (f32((2. * n + 1.) / (n * (n + 1.))) * (an * np.conj(bn)).real).astype(f32)
        if n > 1:
            gs += (f32((n - 1.) * (n + 1.) / n) * (an1 * np.conj(an) + bn1 * np.conj(bn)).real).astype(f32)
        an1 = np.where(live, an, an1); bn1 = np.where(live, bn, bn1)
        psi0, psi1 = psi1, np.where(live, psi, psi1)
        chi0, chi1 = chi1, np.where(live, chi, chi1)
        A0, A1 = A1, A
    return 2 / x ** 2 * qe, 2 / x ** 2 * qs, 2 * gs / np.maximum(qs, 1e-30)


def report_pf(tag, x, m):
    with np.errstate(all="ignore"):
        ref = mie(x, m, False)
        got = mie32_polefree(x, m)
    qe, qs, g = ref[0], ref[1], ref[2]
    for nm, e in (("ext", np.abs(got[0] - qe) / qe), ("sca/ext", np.abs(got[1] - qs) / qe), ("g", np.abs(got[2] - g))):
        e = np.where(np.isfinite(e), e, 1.)
        print(f"{tag:34s} {nm:8s} max {e.max():.2e}  99.9% {np.quantile(e, 0.999):.2e}  median {np.median(e):.1e}")


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    n = 30000
    mre = rng.uniform(1.3, 2.0, n)
    mim = 10 ** rng.uniform(-9, np.log10(0.75), n)
    m = mre + 1j * mim
    for lo, hi in [(0.3, 1.), (1., 3.), (3., 10.), (10., 30.), (30., 100.), (100., 225.)]:
        x = 10 ** rng.uniform(np.log10(lo), np.log10(hi), n)
        s = x * mim < 10.
        report_pf(f"pole-free x in [{lo},{hi})", x[s], m[s])
