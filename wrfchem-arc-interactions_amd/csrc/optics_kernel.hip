// Aerosol optical averaging for gfx950: MOSAIC sectional aerosol -> (tau, omega, g) at the four shortwave
// wavelengths and absorption optical depth in the 16 RRTMG-LW bands, the inputs of rows S2 / L2 of the
// radiation path (SW:10993-11023, LW:12582-12604).
//
// Physics (this engine's specification; the reference repository does not contain the arithmetic):
//   per grid cell and size bin
//     volume of each species = mass / density; dry diameter from dry volume and number, clipped to the bin's
//     bounds with the number adjusted so that volume is conserved; wet diameter = dry * (Vwet/Vdry)^(1/3);
//     refractive index of the internally mixed particle = volume average of the species' indices
//     (aer_op_opt 1/3) or Maxwell-Garnett with black-carbon inclusions in the volume-averaged rest (2/4);
//     homogeneous-sphere Mie theory (Bohren & Huffman 1983, appendix A: downward recurrence of the
//     logarithmic derivative, upward Riccati-Bessel recurrences) -> Qext, Qsca, g, Qback
//   per cell: ext = sum_b N_b pi r_b^2 Qext, omega = sum sca / sum ext, g = sum g sca / sum sca,
//     tau = ext * dz; longwave: absorption only (RRTMG-LW has no scattering term).
//
// Mapping: one thread = one (cell, wavelength); lanes = consecutive i of the (i,k,j) tile so that every tracer
// load is a unit-stride segment; the 20 wavelength blocks of a cell group are issued back-to-back on one XCD
// so that the tracer lines are fetched from HBM once and re-read from that XCD's L2.  Compute-bound (fp64
// VALU): ~10^5 flop per cell against 0.6 KB of tracer data.
#include <hip/hip_runtime.h>

#include "optics_common.h"
#include "optics_kernel.h"

namespace rrtmg {

namespace {

constexpr int OPT_THREADS = 128;

__constant__ OpticsTables c_opt;
// weights of the asymmetry sum, functions of the (wave-uniform) term number only:
// c_gw[0][n] = (2n+1)/(n(n+1)), c_gw[1][n] = (n-1)(n+1)/n
__constant__ double c_gw[2][OPT_NMIE + 2];

struct cplx { double re, im; };

__device__ __forceinline__ double frcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx crecip(cplx a)
{
    const double r = frcp(a.re * a.re + a.im * a.im);
    return {a.re * r, -a.im * r};
}
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) { return cmul(a, crecip(b)); }
__device__ __forceinline__ cplx csqrt_pos(cplx a)   // principal root, Im >= 0 for Im(a) >= 0
{
    const double mod = sqrt(a.re * a.re + a.im * a.im);
    const double re = sqrt(0.5 * (mod + a.re));
    const double im = sqrt(fmax(0.5 * (mod - a.re), 0.));
    return {re, a.im < 0. ? -im : im};
}

__device__ __forceinline__ double ldr(const void *p, size_t idx, int real4)
{
    return real4 ? (double)reinterpret_cast<const float *>(p)[idx] : reinterpret_cast<const double *>(p)[idx];
}
__device__ __forceinline__ void str(void *p, size_t idx, int real4, double v)
{
    if (real4) reinterpret_cast<float *>(p)[idx] = (float)v;
    else reinterpret_cast<double *>(p)[idx] = v;
}

// Mie efficiencies of a homogeneous sphere, size parameter x, refractive index m = n + ik (k >= 0).
//
// The a_n, b_n series runs upward in n, the logarithmic derivative D_n(mx) is only stable downward.  A per-lane
// array of D_n (4 KB) lives in scratch memory and made the first version of this kernel HBM-bound (85 GB of
// scratch traffic per 1.6 M cells).  Here the downward pass keeps only every OPT_MB-th D_n in LDS
// (ck[j][thread]); the upward pass rebuilds each block of OPT_MB values from its checkpoint in registers
// (statically indexed) and consumes them.  D_n is recomputed once; nothing goes to scratch.
constexpr int OPT_MB = 16;
constexpr int OPT_NCK = (OPT_NMIE + OPT_MB - 1) / OPT_MB;   // 16 checkpoints cover nstop <= 256

struct MieAcc {
    double psi0, psi1, chi0, chi1, qs, qe, gs, sgn;
    cplx an1, bn1, xb;
};

// FULL = false: only the extinction and scattering sums (longwave bands use absorption alone)
template <bool FULL>
__device__ __forceinline__ void mie_term(MieAcc &A, int n, cplx D, cplx m, cplx minv, double xinv)
{
    const double en = (double)n;
    const double psi = (2. * en - 1.) * xinv * A.psi1 - A.psi0;
    const double chi = (2. * en - 1.) * xinv * A.chi1 - A.chi0;
    const cplx xi = {psi, -chi}, xi1 = {A.psi1, -A.chi1};
    cplx t = cmul(D, minv);
    t.re += en * xinv;
    const cplx na = {t.re * psi - A.psi1, t.im * psi}, da = csub(cmul(t, xi), xi1);
    t = cmul(m, D);
    t.re += en * xinv;
    const cplx nb = {t.re * psi - A.psi1, t.im * psi}, db = csub(cmul(t, xi), xi1);
    // a_n = na/da, b_n = nb/db with one reciprocal: 1/|da|^2 = |db|^2 / (|da|^2 |db|^2)
    const double ma = da.re * da.re + da.im * da.im, mb = db.re * db.re + db.im * db.im;
    const double r = frcp(ma * mb);
    const double ra = r * mb, rb = r * ma;
    const cplx an = {(na.re * da.re + na.im * da.im) * ra, (na.im * da.re - na.re * da.im) * ra};
    const cplx bn = {(nb.re * db.re + nb.im * db.im) * rb, (nb.im * db.re - nb.re * db.im) * rb};
    const double w2 = 2. * en + 1.;
    A.qs += w2 * (an.re * an.re + an.im * an.im + bn.re * bn.re + bn.im * bn.im);
    A.qe += w2 * (an.re + bn.re);
    if (FULL) {
        A.gs += c_gw[0][n] * (an.re * bn.re + an.im * bn.im);
        if (n > 1)
            A.gs += c_gw[1][n] * (A.an1.re * an.re + A.an1.im * an.im + A.bn1.re * bn.re + A.bn1.im * bn.im);
        A.xb.re += A.sgn * w2 * (an.re - bn.re);
        A.xb.im += A.sgn * w2 * (an.im - bn.im);
        A.sgn = -A.sgn;
        A.an1 = an; A.bn1 = bn;
    }
    A.psi0 = A.psi1; A.psi1 = psi; A.chi0 = A.chi1; A.chi1 = chi;
}

template <bool FULL>
__device__ __forceinline__ void mie_sphere(double x, cplx m, double2 *ck /* LDS [OPT_NCK][OPT_THREADS], this thread's column */,
                           double &qext, double &qsca, double &gasy, double &qback)
{
    const cplx y = {x * m.re, x * m.im};
    const double ymod = sqrt(y.re * y.re + y.im * y.im);
    const double xstop = x + 4. * cbrt(x) + 2.;
    const int nstop = (int)xstop;
    const int nck = (nstop + OPT_MB - 1) / OPT_MB;   // checkpoints D_{MB j}, j = 1..nck (MB*nck <= nstop + 15 < nmx)
    // BHMIE starts at max(xstop,|mx|)+15; that leaves D_n wrong by 1e-4..1 once |mx| > 10, because the start
    // value only decays where n exceeds |mx| by several |mx|^(1/3).  With the extra 8|mx|^(1/3) terms D_n is
    // converged to the last bit.
    const double base = fmax(xstop, ymod);
    const int nmx = (int)(base + (base > 8. ? 8. * cbrt(base) : 0.)) + 16;
    const cplx yinv = crecip(y);
    cplx d = {0., 0.};
    const int ntop = nck * OPT_MB;
    for (int n = nmx; n >= 2; --n) {   // D_{n-1} = n/y - 1/(D_n + n/y)
        const cplx ny = {(double)n * yinv.re, (double)n * yinv.im};
        d = csub(ny, crecip(cadd(d, ny)));
        const int n1 = n - 1;
        if (n1 <= ntop && (n1 & (OPT_MB - 1)) == 0) ck[(size_t)(n1 / OPT_MB - 1) * OPT_THREADS] = make_double2(d.re, d.im);
        if (n1 <= OPT_MB) break;       // the lowest block is rebuilt from its checkpoint below
    }
    MieAcc A;
    sincos(x, &A.psi1, &A.psi0);
    A.chi0 = -A.psi1; A.chi1 = A.psi0;
    A.qs = 0.; A.qe = 0.; A.gs = 0.; A.sgn = -1.;
    A.an1 = {0., 0.}; A.bn1 = {0., 0.}; A.xb = {0., 0.};
    const double xinv = frcp(x);
    const cplx minv = crecip(m);
    for (int j = 0; j < nck; ++j) {
        cplx dd[OPT_MB];
        const double2 c = ck[(size_t)j * OPT_THREADS];
        dd[OPT_MB - 1] = {c.x, c.y};                 // D_{MB (j+1)}
#pragma unroll
        for (int i = OPT_MB - 1; i >= 1; --i) {      // dd[i] = D_n with n = MB j + i + 1
            const double en = (double)(j * OPT_MB + i + 1);
            const cplx ny = {en * yinv.re, en * yinv.im};
            dd[i - 1] = csub(ny, crecip(cadd(dd[i], ny)));
        }
#pragma unroll
        for (int i = 0; i < OPT_MB; ++i) {
            const int n = j * OPT_MB + i + 1;
            if (n <= nstop) mie_term<FULL>(A, n, dd[i], m, minv, xinv);
        }
    }
    const double x2inv = xinv * xinv;
    qsca = 2. * x2inv * A.qs;
    qext = 2. * x2inv * A.qe;
    gasy = FULL && A.qs > 0. ? 2. * A.gs * frcp(A.qs) : 0.;
    qback = FULL ? x2inv * (A.xb.re * A.xb.re + A.xb.im * A.xb.im) : 0.;
}

// Two longwave evaluations of one particle (neighbouring bands) side by side.  The recurrences of D_n are serial
// chains of ~10 dependent fp64 operations per step and the longwave series are short, so a single evaluation
// leaves the SIMD waiting on its own results; two independent chains in one instruction stream fill those gaps.
// Both start the downward recurrence at the larger of their two start orders (a higher start is only more
// accurate), use register blocks of OPT_MB2 terms and the two halves of the thread's checkpoint column
// (OPT_NCK/2 checkpoints each: nstop <= OPT_MB2*OPT_NCK/2 = 64, the caller checks).  Extinction and scattering only.
constexpr int OPT_MB2 = 8, OPT_NCK2 = OPT_NCK / 2;

struct MieSetup {
    cplx yinv, minv;
    double xinv;
    int nstop, nmx;
};
__device__ __forceinline__ MieSetup mie_setup(double x, cplx m)
{
    MieSetup s;
    const cplx y = {x * m.re, x * m.im};
    const double ymod = sqrt(y.re * y.re + y.im * y.im);
    const double xstop = x + 4. * cbrt(x) + 2.;
    s.nstop = (int)xstop;
    const double base = fmax(xstop, ymod);
    s.nmx = (int)(base + (base > 8. ? 8. * cbrt(base) : 0.)) + 16;   // as in mie_sphere
    s.yinv = crecip(y);
    s.minv = crecip(m);
    s.xinv = frcp(x);
    return s;
}

__device__ __forceinline__ void mie_pair(double x1, cplx m1, double x2, cplx m2, const MieSetup &s1, const MieSetup &s2,
                                         double2 *ck, double &qext1, double &qsca1, double &qext2, double &qsca2)
{
    const int nstopm = s1.nstop > s2.nstop ? s1.nstop : s2.nstop;
    const int nck = (nstopm + OPT_MB2 - 1) / OPT_MB2, ntop = nck * OPT_MB2;
    const int nmx = s1.nmx > s2.nmx ? s1.nmx : s2.nmx;
    cplx d1 = {0., 0.}, d2 = {0., 0.};
    for (int n = nmx; n >= 2; --n) {   // D_{n-1} = n/y - 1/(D_n + n/y), both particles
        const double en = (double)n;
        const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
        d1 = csub(ny1, crecip(cadd(d1, ny1)));
        d2 = csub(ny2, crecip(cadd(d2, ny2)));
        const int n1 = n - 1;
        if (n1 <= ntop && (n1 & (OPT_MB2 - 1)) == 0) {
            ck[(size_t)(n1 / OPT_MB2 - 1) * OPT_THREADS] = make_double2(d1.re, d1.im);
            ck[(size_t)(OPT_NCK2 + n1 / OPT_MB2 - 1) * OPT_THREADS] = make_double2(d2.re, d2.im);
        }
        if (n1 <= OPT_MB2) break;
    }
    MieAcc A1, A2;
    sincos(x1, &A1.psi1, &A1.psi0);
    sincos(x2, &A2.psi1, &A2.psi0);
    A1.chi0 = -A1.psi1; A1.chi1 = A1.psi0; A1.qs = 0.; A1.qe = 0.;
    A2.chi0 = -A2.psi1; A2.chi1 = A2.psi0; A2.qs = 0.; A2.qe = 0.;
    for (int j = 0; j < nck; ++j) {
        cplx dd1[OPT_MB2], dd2[OPT_MB2];
        const double2 c1 = ck[(size_t)j * OPT_THREADS], c2 = ck[(size_t)(OPT_NCK2 + j) * OPT_THREADS];
        dd1[OPT_MB2 - 1] = {c1.x, c1.y};
        dd2[OPT_MB2 - 1] = {c2.x, c2.y};
#pragma unroll
        for (int i = OPT_MB2 - 1; i >= 1; --i) {
            const double en = (double)(j * OPT_MB2 + i + 1);
            const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
            dd1[i - 1] = csub(ny1, crecip(cadd(dd1[i], ny1)));
            dd2[i - 1] = csub(ny2, crecip(cadd(dd2[i], ny2)));
        }
#pragma unroll
        for (int i = 0; i < OPT_MB2; ++i) {
            const int n = j * OPT_MB2 + i + 1;
            if (n <= s1.nstop) mie_term<false>(A1, n, dd1[i], m1, s1.minv, s1.xinv);
            if (n <= s2.nstop) mie_term<false>(A2, n, dd2[i], m2, s2.minv, s2.xinv);
        }
    }
    const double f1 = 2. * s1.xinv * s1.xinv, f2 = 2. * s2.xinv * s2.xinv;
    qsca1 = f1 * A1.qs; qext1 = f1 * A1.qe;
    qsca2 = f2 * A2.qs; qext2 = f2 * A2.qe;
}

// refractive index of the internal mixture at wavelength slot w (volume average, or Maxwell-Garnett with BC inclusions)
__device__ __forceinline__ cplx mix_index(const double *vol, double vwet, int mix, int w)
{
    cplx m;
    double sr = 0., si = 0.;
#pragma unroll
    for (int sp = 0; sp < OPT_NSPEC; ++sp) {
        if (mix == 1 && sp == OS_BC) continue;
        sr += vol[sp] * c_opt.refr[sp][w];
        si += vol[sp] * c_opt.refi[sp][w];
    }
    if (mix == 0) {
        const double vi = frcp(vwet);
        m = {sr * vi, si * vi};
    } else {
        const double vbc = vol[OS_BC], vmat = vwet - vbc;
        const cplx mbc = {c_opt.refr[OS_BC][w], c_opt.refi[OS_BC][w]};
        if (!(vmat > 1.e-30 * vwet)) {
            m = mbc;
        } else {
            const double vi = frcp(vmat);
            const cplx mm = {sr * vi, si * vi};
            if (!(vbc > 0.)) {
                m = mm;
            } else {   // Maxwell-Garnett: e = em (ei + 2em + 2f(ei-em)) / (ei + 2em - f(ei-em))
                const double f = vbc * frcp(vwet);
                const cplx em = cmul(mm, mm), ei = cmul(mbc, mbc);
                const cplx de = csub(ei, em);
                const cplx base = {ei.re + 2. * em.re, ei.im + 2. * em.im};
                const cplx nume = {base.re + 2. * f * de.re, base.im + 2. * f * de.im};
                const cplx deno = {base.re - f * de.re, base.im - f * de.im};
                m = csqrt_pos(cmul(em, cdiv(nume, deno)));
            }
        }
    }
    return m;
}

// LW = false: one shortwave wavelength per workgroup (4 per cell group); LW = true: one pair of longwave bands
// (8 per cell group).  Two kernels so that each path gets its own register allocation.
template <bool LW>
__device__ __forceinline__ void optics_body(const OpticsKernelArgs &a, double2 *s_ck)
{
    // the workgroups of one cell group run back-to-back on one XCD (blockIdx & 7)
    constexpr int NW = LW ? OPT_NLW / 2 : OPT_NSW;
    const int b = blockIdx.x;
    const int xcd = b & 7, s = b >> 3;
    const int gl = s / NW, slot = s - gl * NW;
    constexpr bool lw = LW;
    const int w = lw ? OPT_NSW + 2 * slot : slot, w2 = w + 1;   // w2: second band of a longwave pair
    double2 *ck = s_ck + threadIdx.x;
    const long cell = ((long)gl * 8 + xcd) * OPT_THREADS + threadIdx.x;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (cell >= ncell) return;
    // nothing requested at this wavelength (output group absent): no work
    if (!lw ? !(a.tauaer[w] || a.extaer[w] || a.bscoef[w])
            : !(a.tauaerlw[w - OPT_NSW] || a.extaerlw[w - OPT_NSW] || a.tauaerlw[w2 - OPT_NSW] || a.extaerlw[w2 - OPT_NSW]))
        return;
    const int it = (int)(cell % a.ni_t);
    const long r = cell / a.ni_t;
    const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
    const size_t q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const double alt = ldr(a.alt, q, r4), dz = ldr(a.dz8w, q, r4);
    const double lam = c_opt.wavelength_um[w], lam2 = c_opt.wavelength_um[lw ? w2 : w];
    const double pi = 3.14159265358979323846;

    double ext = 0., sca = 0., gsca = 0., bsc = 0.;   // 1/cm (bsc: 1/(cm sr))
    double ext2 = 0., sca2 = 0.;                      // second band of a longwave pair
    for (int ib = 0; ib < a.nbin; ++ib) {
        double vol[OPT_NSPEC];   // 1e-6 cm3 per kg of dry air
        double vdry = 0.;
#pragma unroll
        for (int sp = 0; sp < OPT_NSPEC; ++sp) {
            const int id = a.idx[ib][sp];
            const double mass = id >= 0 ? fmax(ldr(a.chem, (size_t)id * sspec + q, r4), 0.) : 0.;
            vol[sp] = mass * frcp(c_opt.dens[sp]);
            if (sp != OS_WATER) vdry += vol[sp];
        }
        const int idn = a.idx[ib][OS_NUM];
        double num = idn >= 0 ? fmax(ldr(a.chem, (size_t)idn * sspec + q, r4), 0.) : 0.;
        if (!(vdry > 1.e-25)) continue;
        const double vwet = vdry + vol[OS_WATER];
        // dry diameter (um) from dry volume and number; clipped to the section, number follows (volume is kept)
        const double vpart = vdry * 1.e-6 * frcp(fmax(num, 1.e-30));   // cm3 per particle
        double ddry = 1.e4 * cbrt(vpart * (6. / pi));
        const double dlo = c_opt.dlo_um[ib], dhi = c_opt.dhi_um[ib];
        if (ddry < dlo || ddry > dhi) {
            ddry = ddry < dlo ? dlo : dhi;
            const double dcm = ddry * 1.e-4;
            num = vdry * 1.e-6 * frcp((pi / 6.) * dcm * dcm * dcm);
        }
        const double dwet = ddry * cbrt(vwet * frcp(vdry));
        const cplx m = mix_index(vol, vwet, a.mix, w);
        double x = pi * dwet * frcp(lam);
        x = x > OPT_XMAX ? OPT_XMAX : x;
        double qe, qs, g = 0., qb = 0., qe2 = 0., qs2 = 0.;
        if (!lw) {
            mie_sphere<true>(x, m, ck, qe, qs, g, qb);
        } else {
            const cplx mb = mix_index(vol, vwet, a.mix, w2);
            double xb = pi * dwet * frcp(lam2);
            xb = xb > OPT_XMAX ? OPT_XMAX : xb;
            const MieSetup s1 = mie_setup(x, m), s2 = mie_setup(xb, mb);
            if (s1.nstop <= OPT_MB2 * OPT_NCK2 && s2.nstop <= OPT_MB2 * OPT_NCK2) {
                mie_pair(x, m, xb, mb, s1, s2, ck, qe, qs, qe2, qs2);
            } else {   // size parameters beyond any aerosol at these wavelengths: one after the other
                double g0, b0;
                mie_sphere<false>(x, m, ck, qe, qs, g0, b0);
                mie_sphere<false>(xb, mb, ck, qe2, qs2, g0, b0);
            }
        }
        const double dcm = dwet * 1.e-4;
        const double cross = (num * frcp(alt) * 1.e-6) * (0.25 * pi * dcm * dcm);   // #/cm3 * cm2
        ext += cross * qe;
        sca += cross * qs;
        gsca += cross * qs * g;
        bsc += cross * qb * (0.25 / pi);
        ext2 += cross * qe2;
        sca2 += cross * qs2;
    }
    if (!lw) {
        if (a.tauaer[w]) {
            str(a.tauaer[w], q, r4, ext * 100. * dz);
            str(a.waer[w], q, r4, ext > 0. ? sca * frcp(ext) : 0.);
            str(a.gaer[w], q, r4, sca > 0. ? gsca * frcp(sca) : 0.);
        }
        if (a.extaer[w]) str(a.extaer[w], q, r4, ext * 1.e5);
        if (a.bscoef[w]) str(a.bscoef[w], q, r4, bsc * 1.e5);
    } else {
        const int ib = w - OPT_NSW;
        const double ab = fmax(ext - sca, 0.), ab2 = fmax(ext2 - sca2, 0.);
        if (a.tauaerlw[ib]) str(a.tauaerlw[ib], q, r4, ab * 100. * dz);
        if (a.extaerlw[ib]) str(a.extaerlw[ib], q, r4, ab * 1.e5);
        if (a.tauaerlw[ib + 1]) str(a.tauaerlw[ib + 1], q, r4, ab2 * 100. * dz);
        if (a.extaerlw[ib + 1]) str(a.extaerlw[ib + 1], q, r4, ab2 * 1.e5);
    }
}

__global__ void __launch_bounds__(OPT_THREADS) optics_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK * OPT_THREADS];
    optics_body<false>(a, s_ck);
}

__global__ void __launch_bounds__(OPT_THREADS) optics_lw_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK * OPT_THREADS];
    optics_body<true>(a, s_ck);
}

}  // namespace

hipError_t optics_upload_tables(const OpticsTables &t)
{
    static double gw[2][OPT_NMIE + 2];
    for (int n = 1; n < OPT_NMIE + 2; ++n) {
        const double en = (double)n;
        gw[0][n] = (2. * en + 1.) / (en * (en + 1.));
        gw[1][n] = (en - 1.) * (en + 1.) / en;
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_gw), gw, sizeof gw);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_opt), &t, sizeof(OpticsTables));
}

hipError_t optics_launch(const OpticsKernelArgs &a, hipStream_t stream)
{
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (ncell <= 0) return hipSuccess;
    const long groups = (ncell + OPT_THREADS - 1) / OPT_THREADS;
    const long groups8 = (groups + 7) / 8;
    const long blocks = groups8 * 8 * OPT_NSW, blocks_lw = groups8 * 8 * (OPT_NLW / 2);
    if (blocks_lw > 0x7fffffffL) return hipErrorInvalidValue;
    // skip a kernel none of whose outputs were asked for
    bool any_sw = false, any_lw = false;
    for (int w = 0; w < OPT_NSW; ++w) any_sw = any_sw || a.tauaer[w] || a.extaer[w] || a.bscoef[w];
    for (int b = 0; b < OPT_NLW; ++b) any_lw = any_lw || a.tauaerlw[b] || a.extaerlw[b];
    if (any_sw) hipLaunchKernelGGL(optics_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, stream, a);
    if (any_lw) hipLaunchKernelGGL(optics_lw_kernel, dim3((unsigned)blocks_lw), dim3(OPT_THREADS), 0, stream, a);
    return hipGetLastError();
}

}  // namespace rrtmg
