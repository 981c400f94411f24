/*
 * oracle/optics_oracle.c -- CPU restatement of the aerosol optical-averaging stage.  TEST INFRASTRUCTURE ONLY:
 * imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path.
 *
 * PARITY UNPINNED.  The arithmetic of this stage (upstream WRF-Chem chem/module_optical_averaging.F) is not part
 * of /root/reference; only its interface is (tracers: Registry/registry.chem:2325-2447, 3823-3826; outputs:
 * registry.chem:1332-1387; consumers: phys/module_ra_rrtmg_sw.F:10993-11023, module_ra_rrtmg_lw.F:12582-12604).
 * This file restates the engine's own specification (DESIGN.md "Optical averaging") in plain C99 with
 * <complex.h>, and its Mie core is pinned against published known answers instead (tests/test_optics_oracle.py):
 * Bohren & Huffman (1983) appendix A test case, Wiscombe (1979) / Du (2004) efficiency tables, the Rayleigh and
 * geometric-optics limits, and an independent formulation with scipy's spherical Bessel functions.
 */
#ifdef ORACLE_COUNT_FLOPS
/* the operation-counting build (oracle/count, C++): complex numbers are pairs of the counting type, flop_count.hpp */
typedef oracle_cplx cplx_t;
#define CPLX(re, im) oracle_cplx((re), (im))
extern "C" {
void optics_oracle_mie(double x, double mre, double mim, double *out4);
struct optics_oracle_tables;
void optics_oracle_cell(const struct optics_oracle_tables *t, int mix, const double *mass, const double *numb, double alt,
                        double dz, double *out_sw, double *out_lw);
void optics_oracle_tile(const struct optics_oracle_tables *t, int mix, int ni_m, int nk_m, int nj_m, int i0, int k0, int j0,
                        int ni_t, int nk_t, int nj_t, int num_chem, const int *idx, const double *chem, const double *alt,
                        const double *dz8w, double *sw5, double *lw2);
}
#else
#include <complex.h>
typedef double complex cplx_t;
#define CPLX(re, im) ((re) + I * (im))
#endif
#include <math.h>
#include <stddef.h>

#define NSPEC 9
#define NSLOT 10
#define NWAVE 20
#define NSW 4
#define NMIE 256
#define XMAX 225.0
#define S_BC 7
#define S_WATER 8
#define S_NUM 9

/* Homogeneous sphere after Bohren & Huffman (1983), appendix A (BHMIE): D_n by downward recurrence, Riccati-
 * Bessel psi/chi by upward recurrence, nstop = x + 4 x^(1/3) + 2.  BHMIE starts the D_n recurrence at
 * max(xstop,|mx|)+15, which leaves errors up to 1e-4 .. 1 in D_n for |mx| > 10 (the start value decays only
 * where n exceeds |mx| by several |mx|^(1/3)); the start is therefore raised by 8 |mx|^(1/3), after which D_n
 * equals the converged value to the last bit in a 20 000-case sweep. */
#ifndef ORACLE_COUNT_FLOPS
/* the coefficients of the last evaluation on this thread, for the Legendre moments (optics_oracle_last_moments) */
static _Thread_local cplx_t last_an[NMIE + 8], last_bn[NMIE + 8];
static _Thread_local int last_nstop;
static _Thread_local double last_x;
#define KEEP_COEFFS(n, a, b) do { if ((n) < NMIE + 8) { last_an[n] = (a); last_bn[n] = (b); } } while (0)
#define KEEP_END(ns, xx) do { last_nstop = (ns) < NMIE + 7 ? (ns) : NMIE + 7; last_x = (xx); } while (0)
#else
#define KEEP_COEFFS(n, a, b) do { } while (0)
#define KEEP_END(ns, xx) do { } while (0)
#endif
void optics_oracle_mie(double x, double mre, double mim, double *out4)
{
    static _Thread_local cplx_t D[4096];
    const cplx_t m = CPLX(mre, mim);
    const cplx_t y = x * m;
    const double xstop = x + 4. * cbrt(x) + 2.;
    const int nstop = (int)xstop;
    const double base = fmax(xstop, cabs(y));
    int nmx = (int)(base + (base > 8. ? 8. * cbrt(base) : 0.)) + 16;
    if (nmx > 4095) nmx = 4095;
    D[nmx] = CPLX(0., 0.);
    for (int n = nmx; n >= 2; --n) {
        const cplx_t ny = (double)n / y;
        D[n - 1] = ny - 1. / (D[n] + ny);
    }
    double psi0 = cos(x), psi1 = sin(x), chi0 = -sin(x), chi1 = cos(x);
    cplx_t an1 = CPLX(0., 0.), bn1 = CPLX(0., 0.), xb = CPLX(0., 0.);
    double qs = 0., qe = 0., gs = 0., sgn = -1.;
    for (int n = 1; n <= nstop; ++n) {
        const double en = (double)n;
        const double psi = (2. * en - 1.) * psi1 / x - psi0;
        const double chi = (2. * en - 1.) * chi1 / x - chi0;
        const cplx_t xi = CPLX(psi, -chi), xi1 = CPLX(psi1, -chi1);
        const cplx_t ta = D[n] / m + en / x, tb = m * D[n] + en / x;
        const cplx_t an = (ta * psi - psi1) / (ta * xi - xi1);
        const cplx_t bn = (tb * psi - psi1) / (tb * xi - xi1);
        qs += (2. * en + 1.) * (creal(an) * creal(an) + cimag(an) * cimag(an) + creal(bn) * creal(bn) + cimag(bn) * cimag(bn));
        qe += (2. * en + 1.) * (creal(an) + creal(bn));
        gs += ((2. * en + 1.) / (en * (en + 1.))) * (creal(an) * creal(bn) + cimag(an) * cimag(bn));
        if (n > 1)
            gs += ((en - 1.) * (en + 1.) / en) *
                  (creal(an1) * creal(an) + cimag(an1) * cimag(an) + creal(bn1) * creal(bn) + cimag(bn1) * cimag(bn));
        xb += (sgn * (2. * en + 1.)) * (an - bn);
        sgn = -sgn;
        psi0 = psi1; psi1 = psi; chi0 = chi1; chi1 = chi;
        KEEP_COEFFS(n, an, bn);
        an1 = an; bn1 = bn;
    }
    KEEP_END(nstop, x);
    out4[0] = 2. * qe / (x * x);                           /* Qext */
    out4[1] = 2. * qs / (x * x);                           /* Qsca */
    out4[2] = qs > 0. ? 2. * gs / qs : 0.;                 /* asymmetry parameter */
    out4[3] = (creal(xb) * creal(xb) + cimag(xb) * cimag(xb)) / (x * x);   /* Qback */
}

#ifndef ORACLE_COUNT_FLOPS   /* (the counting build covers aer_op_opt = 1, what bench.py runs) */
/* Coated sphere (aer_op_opt = 5, shell_exact, registry.chem:4018): a core of size parameter x and index m1 inside a
 * sphere of size parameter y >= x whose shell has index m2.  Bohren & Huffman (1983) appendix B (BHCOAT): the shell
 * replaces D_n(m y) of the homogeneous sphere by two effective logarithmic derivatives, one for a_n and one for b_n,
 *     Da_n = (D_n(m2 y) - A_n chi_n'(m2 y)) / (1 - A_n chi_n(m2 y)),   Db_n likewise with B_n,
 * where A_n, B_n carry the core (they vanish with it).  Differences from the textbook routine: the three logarithmic
 * derivatives D_n(m1 x), D_n(m2 x), D_n(m2 y) come from the DOWNWARD recurrence with the start order of
 * optics_oracle_mie (BHCOAT carries them upward from cot z, which loses all accuracy for n > |z|, i.e. for every small
 * particle); the core terms are dropped for good once they fall below 1e-14 of D_n(m2 y) (BHCOAT: 1e-8), which is what
 * keeps chi_n(m2 x) -- growing like (2n-1)!!/z^n -- inside the floating-point range; the asymmetry parameter is summed
 * as in optics_oracle_mie. */
static void log_derivative_down(cplx_t z, int nstop, double xstop, cplx_t *D /* [0..nstop] filled for n >= 1 */)
{
    const double base = fmax(xstop, cabs(z));
    int nmx = (int)(base + (base > 8. ? 8. * cbrt(base) : 0.)) + 16;
    cplx_t d = CPLX(0., 0.);
    for (int n = nmx; n >= 2; --n) {
        const cplx_t nz = (double)n / z;
        d = nz - 1. / (d + nz);
        if (n - 1 <= nstop) D[n - 1] = d;
    }
}
void optics_oracle_mie_coated(double x, double y, double m1re, double m1im, double m2re, double m2im, double *out4)
{
    static _Thread_local cplx_t Dx1[NMIE + 8], Dx2[NMIE + 8], Dy2[NMIE + 8];
    const cplx_t m1 = CPLX(m1re, m1im), m2 = CPLX(m2re, m2im);
    const cplx_t x1 = x * m1, x2 = x * m2, y2 = y * m2, refrel = m2 / m1;
    const double ystop = y + 4. * cbrt(y) + 2.;
    const int nstop = (int)ystop;
    log_derivative_down(x1, nstop, ystop, Dx1);
    log_derivative_down(x2, nstop, ystop, Dx2);
    log_derivative_down(y2, nstop, ystop, Dy2);
    double psi0 = cos(y), psi1 = sin(y), chi0 = -sin(y), chi1 = cos(y);
    cplx_t chi0x2 = -csin(x2), chi1x2 = ccos(x2), chi0y2 = -csin(y2), chi1y2 = ccos(y2);
    cplx_t an1 = CPLX(0., 0.), bn1 = CPLX(0., 0.), xb = CPLX(0., 0.);
    double qs = 0., qe = 0., gs = 0., sgn = -1.;
    int core_gone = 0;
    for (int n = 1; n <= nstop; ++n) {
        const double en = (double)n;
        const double psi = (2. * en - 1.) * psi1 / y - psi0;
        const double chi = (2. * en - 1.) * chi1 / y - chi0;
        const cplx_t xi = CPLX(psi, -chi), xi1 = CPLX(psi1, -chi1);
        cplx_t da = Dy2[n], db = Dy2[n];
        if (!core_gone) {
            const cplx_t chix2 = (2. * en - 1.) * chi1x2 / x2 - chi0x2, chiy2 = (2. * en - 1.) * chi1y2 / y2 - chi0y2;
            const cplx_t chipx2 = chi1x2 - en * chix2 / x2, chipy2 = chi1y2 - en * chiy2 / y2;
            const cplx_t wx = chix2 * Dx2[n] - chipx2, wy = chiy2 * Dy2[n] - chipy2;
            const cplx_t brack = (refrel * Dx1[n] - Dx2[n]) / (refrel * Dx1[n] * chix2 - chipx2) / wx * wy;
            const cplx_t crack = (refrel * Dx2[n] - Dx1[n]) / (refrel * chipx2 - Dx1[n] * chix2) / wx * wy;
            const double del = 1.e-14, dmod = cabs(Dy2[n]);
            if (cabs(brack * chipy2) > del * dmod || cabs(brack * chiy2) > del || cabs(crack * chipy2) > del * dmod ||
                cabs(crack * chiy2) > del) {
                da = (Dy2[n] - brack * chipy2) / (1. - brack * chiy2);
                db = (Dy2[n] - crack * chipy2) / (1. - crack * chiy2);
                chi0x2 = chi1x2; chi1x2 = chix2; chi0y2 = chi1y2; chi1y2 = chiy2;
            } else {
                core_gone = 1;
            }
        }
        const cplx_t ta = da / m2 + en / y, tb = m2 * db + en / y;
        const cplx_t an = (ta * psi - psi1) / (ta * xi - xi1);
        const cplx_t bn = (tb * psi - psi1) / (tb * xi - xi1);
        qs += (2. * en + 1.) * (creal(an) * creal(an) + cimag(an) * cimag(an) + creal(bn) * creal(bn) + cimag(bn) * cimag(bn));
        qe += (2. * en + 1.) * (creal(an) + creal(bn));
        gs += ((2. * en + 1.) / (en * (en + 1.))) * (creal(an) * creal(bn) + cimag(an) * cimag(bn));
        if (n > 1)
            gs += ((en - 1.) * (en + 1.) / en) *
                  (creal(an1) * creal(an) + cimag(an1) * cimag(an) + creal(bn1) * creal(bn) + cimag(bn1) * cimag(bn));
        xb += (sgn * (2. * en + 1.)) * (an - bn);
        sgn = -sgn;
        psi0 = psi1; psi1 = psi; chi0 = chi1; chi1 = chi;
        KEEP_COEFFS(n, an, bn);
        an1 = an; bn1 = bn;
    }
    KEEP_END(nstop, y);
    out4[0] = 2. * qe / (y * y);
    out4[1] = 2. * qs / (y * y);
    out4[2] = qs > 0. ? 2. * gs / qs : 0.;
    out4[3] = (creal(xb) * creal(xb) + cimag(xb) * cimag(xb)) / (y * y);
}
#endif

#ifndef ORACLE_COUNT_FLOPS
/* Gauss-Legendre nodes and weights on [-1, 1] (Newton iteration on P_N from the Chebyshev-like first guess) */
static void gauss_legendre(int N, double *xq, double *wq)
{
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < (N + 1) / 2; ++i) {
        double z = cos(pi * (i + 0.75) / (N + 0.5)), pp = 1.;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1., p2 = 0.;
            for (int j = 1; j <= N; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2. * j - 1.) * z * p2 - (j - 1.) * p3) / j;
            }
            pp = N * (z * p1 - p2) / (z * z - 1.);
            const double dz = p1 / pp;
            z -= dz;
            if (fabs(dz) < 1e-15) break;
        }
        xq[i] = -z; xq[N - 1 - i] = z;
        wq[i] = wq[N - 1 - i] = 2. / ((1. - z * z) * pp * pp);
    }
}
/* Legendre moments chi_0..chi_7 of the normalised phase function of the last particle evaluated on this thread:
 * p(mu) = sum_k (2k+1) chi_k P_k(mu), chi_0 = 1, chi_1 = the asymmetry parameter.  The intensity
 * i(mu) = (|S1|^2 + |S2|^2) / 2 from the amplitude functions S1 = sum_n c_n (a_n pi_n + b_n tau_n), S2 = sum_n c_n
 * (a_n tau_n + b_n pi_n), c_n = (2n+1)/(n(n+1)) (Bohren & Huffman eq. 4.74), is a polynomial of degree 2 nstop in mu, so
 * a Gauss-Legendre rule of N >= nstop + 4 nodes integrates i(mu) P_k(mu), k <= 7, exactly; N is the next multiple of 8.
 * chi_k = sum_q w_q i(mu_q) P_k(mu_q) / sum_q w_q i(mu_q). */
void optics_oracle_last_moments(double *chi8)
{
    static _Thread_local double xq[NMIE + 24], wq[NMIE + 24];
    const int nstop = last_nstop;
    const int N = 8 * ((nstop + 4 + 7) / 8);
    gauss_legendre(N, xq, wq);
    double m[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    for (int q = 0; q < N; ++q) {
        const double mu = xq[q];
        double pi0 = 0., pi1 = 1.;
        cplx_t s1 = CPLX(0., 0.), s2 = CPLX(0., 0.);
        for (int n = 1; n <= nstop; ++n) {
            const double en = (double)n;
            const double tau = en * mu * pi1 - (en + 1.) * pi0;
            const double c = (2. * en + 1.) / (en * (en + 1.));
            s1 += c * (last_an[n] * pi1 + last_bn[n] * tau);
            s2 += c * (last_an[n] * tau + last_bn[n] * pi1);
            const double pin = ((2. * en + 1.) * mu * pi1 - (en + 1.) * pi0) / en;
            pi0 = pi1; pi1 = pin;
        }
        const double inten = 0.5 * (creal(s1) * creal(s1) + cimag(s1) * cimag(s1) + creal(s2) * creal(s2) + cimag(s2) * cimag(s2));
        double p0 = 1., p1 = mu;
        m[0] += wq[q] * inten;
        m[1] += wq[q] * inten * mu;
        for (int k = 2; k < 8; ++k) {
            const double pk = ((2. * k - 1.) * mu * p1 - (k - 1.) * p0) / k;
            m[k] += wq[q] * inten * pk;
            p0 = p1; p1 = pk;
        }
    }
    for (int k = 0; k < 8; ++k) chi8[k] = m[0] > 0. ? m[k] / m[0] : (k == 0 ? 1. : 0.);
}
#endif

typedef struct optics_oracle_tables {
    double refr[NSPEC][NWAVE], refi[NSPEC][NWAVE];
    double dens[NSPEC];
    double wavelength_um[NWAVE];
    double dlo_um[8], dhi_um[8];
    int nbin;
} optics_oracle_tables;

/* one grid cell: mass[b][s] ug/kg (s = so4 no3 cl nh4 na oin oc bc water), num[b] #/kg, alt m3/kg, dz m.
 * out_sw[w][5] = tau, omega, g, ext (1/km), backscatter (1/(km sr)); out_lw[b][2] = absorption tau, coefficient (1/km) */
static void cell_impl(const optics_oracle_tables *t, int mix, const double *mass /*[nbin][9]*/,
                      const double *numb /*[nbin]*/, double alt, double dz, double *out_sw, double *out_lw,
                      double *out_leg /* NULL or [4 wavelengths][6]: l2..l7 */)
{
    const double pi = 3.14159265358979323846;
    for (int w = 0; w < NWAVE; ++w) {
        double ext = 0., sca = 0., gsca = 0., bsc = 0.;
        double lsum[6] = {0., 0., 0., 0., 0., 0.};
        for (int b = 0; b < t->nbin; ++b) {
            double vol[NSPEC], vdry = 0.;
            for (int s = 0; s < NSPEC; ++s) {
                vol[s] = fmax(mass[b * NSPEC + s], 0.) / t->dens[s];
                if (s != S_WATER) vdry += vol[s];
            }
            double num = fmax(numb[b], 0.);
            if (!(vdry > 1.e-25)) continue;
            const double vwet = vdry + vol[S_WATER];
            const double vpart = vdry * 1.e-6 / fmax(num, 1.e-30);
            double ddry = 1.e4 * cbrt(vpart * (6. / pi));
            if (ddry < t->dlo_um[b] || ddry > t->dhi_um[b]) {
                ddry = ddry < t->dlo_um[b] ? t->dlo_um[b] : t->dhi_um[b];
                const double dcm = ddry * 1.e-4;
                num = vdry * 1.e-6 / ((pi / 6.) * dcm * dcm * dcm);
            }
            const double dwet = ddry * cbrt(vwet / vdry);
            cplx_t m;
            double sr = 0., si = 0.;
            for (int s = 0; s < NSPEC; ++s) {
                if (mix >= 1 && s == S_BC) continue;
                sr += vol[s] * t->refr[s][w];
                si += vol[s] * t->refi[s][w];
            }
            double xcore = 0.;      /* mix 2: size parameter of the black-carbon core, 0 = homogeneous sphere of index m */
            cplx_t mcore = CPLX(0., 0.);
            if (mix == 0) {
                m = CPLX(sr, si) / vwet;
            } else {
                const double vbc = vol[S_BC], vmat = vwet - vbc;
                const cplx_t mbc = CPLX(t->refr[S_BC][w], t->refi[S_BC][w]);
                if (!(vmat > 1.e-30 * vwet)) m = mbc;
                else {
                    const cplx_t mm = CPLX(sr, si) / vmat;
                    if (!(vbc > 0.)) m = mm;
                    else if (mix == 1) {
                        const double f = vbc / vwet;
                        const cplx_t em = mm * mm, ei = mbc * mbc;
                        m = csqrt(em * (ei + 2. * em + 2. * f * (ei - em)) / (ei + 2. * em - f * (ei - em)));
                    } else {   /* shell_exact: black carbon as a concentric core, everything else (volume-mixed) around it */
                        m = mm; mcore = mbc;
                        xcore = cbrt(vbc / vwet);   /* times x below */
                    }
                }
            }
            double x = pi * dwet / t->wavelength_um[w];
            if (x > XMAX) x = XMAX;
            double q[4];
#ifndef ORACLE_COUNT_FLOPS
            if (xcore > 0.) optics_oracle_mie_coated(xcore * x, x, creal(mcore), cimag(mcore), creal(m), cimag(m), q);
            else
#endif
            optics_oracle_mie(x, creal(m), cimag(m), q);
            const double dcm = dwet * 1.e-4;
            const double cross = (num / alt * 1.e-6) * (0.25 * pi * dcm * dcm);
            ext += cross * q[0];
            sca += cross * q[1];
            gsca += cross * q[1] * q[2];
            bsc += cross * q[3] * (0.25 / pi);
#ifndef ORACLE_COUNT_FLOPS
            if (out_leg && w < NSW) {   /* scattering-weighted Legendre moments of the phase function, k = 2..7 */
                double chi[8];
                optics_oracle_last_moments(chi);
                for (int k = 2; k < 8; ++k) lsum[k - 2] += cross * q[1] * chi[k];
            }
#endif
        }
        if (out_leg && w < NSW)
            for (int k = 0; k < 6; ++k) out_leg[w * 6 + k] = sca > 0. ? lsum[k] / sca : 0.;
        if (w < NSW) {
            double *o = out_sw + w * 5;
            o[0] = ext * 100. * dz;
            o[1] = ext > 0. ? sca / ext : 0.;
            o[2] = sca > 0. ? gsca / sca : 0.;
            o[3] = ext * 1.e5;
            o[4] = bsc * 1.e5;
        } else {
            double *o = out_lw + (w - NSW) * 2;
            const double ab = fmax(ext - sca, 0.);
            o[0] = ab * 100. * dz;
            o[1] = ab * 1.e5;
        }
    }
}

void optics_oracle_cell(const optics_oracle_tables *t, int mix, const double *mass, const double *numb, double alt,
                        double dz, double *out_sw, double *out_lw)
{
    cell_impl(t, mix, mass, numb, alt, dz, out_sw, out_lw, NULL);
}
#ifndef ORACLE_COUNT_FLOPS
/* the same with l2aer..l7aer (registry.chem:1382-1387): out_leg[w][k-2], w = 300/400/600/999 nm */
void optics_oracle_cell_legendre(const optics_oracle_tables *t, int mix, const double *mass, const double *numb, double alt,
                                 double dz, double *out_sw, double *out_lw, double *out_leg)
{
    cell_impl(t, mix, mass, numb, alt, dz, out_sw, out_lw, out_leg);
}
#endif

/* a WRF tile: chem(ni_m,nk_m,nj_m,num_chem) i fastest, idx[b][10] 0-based (-1 absent); outputs are 3-D arrays
 * of the memory shape, [field][ni_m*nk_m*nj_m]: sw5 = 4 wavelengths x (tau,omega,g,ext,bsc), lw2 = 16 x (tau,ext) */
static void tile_impl(const optics_oracle_tables *t, int mix, int ni_m, int nk_m, int nj_m, int i0, int k0, int j0,
                      int ni_t, int nk_t, int nj_t, int num_chem, const int *idx /*[8][10]*/, const double *chem,
                      const double *alt, const double *dz8w, double *sw5, double *lw2, double *leg /* NULL or [4][6][n3] */)
{
    const size_t n3 = (size_t)ni_m * nk_m * nj_m;
    (void)num_chem;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int jt = 0; jt < nj_t; ++jt)
        for (int k = 0; k < nk_t; ++k)
            for (int it = 0; it < ni_t; ++it) {
                const size_t q = ((size_t)(j0 + jt) * nk_m + (k0 + k)) * ni_m + (i0 + it);
                double mass[8 * NSPEC], num[8], osw[NSW * 5], olw[16 * 2], oleg[NSW * 6];
                for (int b = 0; b < t->nbin; ++b) {
                    for (int s = 0; s < NSPEC; ++s) {
                        const int id = idx[b * NSLOT + s];
                        mass[b * NSPEC + s] = id >= 0 ? chem[(size_t)id * n3 + q] : 0.;
                    }
                    const int id = idx[b * NSLOT + S_NUM];
                    num[b] = id >= 0 ? chem[(size_t)id * n3 + q] : 0.;
                }
                cell_impl(t, mix, mass, num, alt[q], dz8w[q], osw, olw, leg ? oleg : NULL);
                if (leg)
                    for (int f = 0; f < NSW * 6; ++f) leg[(size_t)f * n3 + q] = oleg[f];
                for (int w = 0; w < NSW; ++w)
                    for (int f = 0; f < 5; ++f) sw5[((size_t)w * 5 + f) * n3 + q] = osw[w * 5 + f];
                for (int b = 0; b < 16; ++b)
                    for (int f = 0; f < 2; ++f) lw2[((size_t)b * 2 + f) * n3 + q] = olw[b * 2 + f];
            }
}
void optics_oracle_tile(const optics_oracle_tables *t, int mix, int ni_m, int nk_m, int nj_m, int i0, int k0, int j0,
                        int ni_t, int nk_t, int nj_t, int num_chem, const int *idx, const double *chem, const double *alt,
                        const double *dz8w, double *sw5, double *lw2)
{
    tile_impl(t, mix, ni_m, nk_m, nj_m, i0, k0, j0, ni_t, nk_t, nj_t, num_chem, idx, chem, alt, dz8w, sw5, lw2, NULL);
}
#ifndef ORACLE_COUNT_FLOPS
void optics_oracle_tile_legendre(const optics_oracle_tables *t, int mix, int ni_m, int nk_m, int nj_m, int i0, int k0,
                                 int j0, int ni_t, int nk_t, int nj_t, int num_chem, const int *idx, const double *chem,
                                 const double *alt, const double *dz8w, double *sw5, double *lw2, double *leg)
{
    tile_impl(t, mix, ni_m, nk_m, nj_m, i0, k0, j0, ni_t, nk_t, nj_t, num_chem, idx, chem, alt, dz8w, sw5, lw2, leg);
}
#endif

/* Modal (MADE/SORGAM-type) aerosol -> sections: the specification of modal_sections.hip restated for one cell.
 * mass[nmode][9] (ug/kg, species order of the optics), num[nmode] (#/kg), sigma[nmode] -> secmass[nbin][9],
 * secnum[nbin].  Log-normal shares of number (around dgn) and of mass (around dgv = dgn exp(3 ln^2 sigma)); the end
 * sections take the tails.  Parity unpinned (no reference arithmetic exists for this stage). */
static double phi_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }

void optics_oracle_modal_cell(const optics_oracle_tables *t, int nmode, const double *mass, const double *num,
                              const double *sigma, double *secmass, double *secnum)
{
    const double pi = 3.14159265358979323846;
    for (int b = 0; b < t->nbin; ++b) {
        secnum[b] = 0.;
        for (int s = 0; s < 9; ++s) secmass[b * 9 + s] = 0.;
    }
    for (int m = 0; m < nmode; ++m) {
        double vdry = 0., mm[9];
        for (int s = 0; s < 9; ++s) {
            mm[s] = fmax(mass[m * 9 + s], 0.);
            if (s != 8) vdry += mm[s] / t->dens[s];
        }
        const double n = fmax(num[m], 0.);
        if (!(vdry > 1.e-25) || !(n > 0.)) continue;
        const double ls = log(sigma[m]);
        const double lndgn = log(1.e4 * cbrt(vdry * 1.e-6 / n * (6. / pi))) - 1.5 * ls * ls;
        const double lndgv = lndgn + 3. * ls * ls;
        const double rls = 1. / ls;
        double cn_lo = 0., cv_lo = 0.;
        for (int b = 0; b < t->nbin; ++b) {
            const int last = b == t->nbin - 1;
            const double cn_hi = last ? 1. : phi_cdf((log(t->dhi_um[b]) - lndgn) * rls);
            const double cv_hi = last ? 1. : phi_cdf((log(t->dhi_um[b]) - lndgv) * rls);
            const double fn = cn_hi - cn_lo, fv = cv_hi - cv_lo;
            for (int s = 0; s < 9; ++s) secmass[b * 9 + s] += mm[s] * fv;
            secnum[b] += n * fn;
            cn_lo = cn_hi; cv_lo = cv_hi;
        }
    }
}
