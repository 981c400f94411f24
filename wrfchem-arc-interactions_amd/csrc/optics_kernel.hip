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
// Mapping: lanes = consecutive i of the (i,k,j) tile so that every tracer load is a unit-stride segment.  fp64: one
// thread = one cell -- all four shortwave wavelengths (optics_sw_cell), all sixteen longwave bands two at a time
// (optics_lw_cell) -- with the cell's running sums in the thread's LDS column.  The fp32 kernels keep one thread per
// (cell, wavelength pair / band pair).  Compute-bound (fp64 VALU): ~10^5 flop per cell against
// 0.6 KB of tracer data.
#include <hip/hip_runtime.h>

#include <cmath>

#include "optics_common.h"
#include "optics_kernel.h"

namespace rrtmg {

namespace {

constexpr int OPT_THREADS = 128;

__constant__ OpticsTables c_opt;
// weights of the asymmetry sum, functions of the (wave-uniform) term number only:
// c_gw[n] = {(2n+1)/(n(n+1)), (n-1)(n+1)/n}: the weights of the asymmetry sums of term n (the term number is
// wave-uniform, so these are scalar loads; 8 entries of slack for the loop that fetches a block of terms ahead)
__constant__ double2 c_gw[OPT_NMIE + 10];
// everything of term n that depends on n alone, one 64-byte record per term: the term number is wave-uniform, so the
// record is one scalar load and its fields are scalar operands of the lanes' arithmetic (formed per lane they cost
// three conversions and as many multiply-adds per term)
struct MieN { double en, tnm1, w2, gx, gy, pad_[3]; };   // n, 2n - 1, 2n + 1, the two asymmetry weights
__constant__ MieN c_mn[OPT_NMIE + 10];
__device__ __forceinline__ MieN mie_n(int n, double2 gw)   // the same from n (folds where n is a compile-time constant)
{
    const double en = (double)n;
    return MieN{en, 2. * en - 1., 2. * en + 1., gw.x, gw.y, {0., 0., 0.}};
}
// per wavelength slot: pi / lambda (size parameter per um of diameter) and its cube root
__constant__ double c_pil[2][OPT_NWAVE];

struct cplx { double re, im; };

// v_rcp_f64 is good to 2^-24.4 (tools/probes/rcp_probe.hip): one Newton step gives 2.2e-15 (frcp1: inside the
// recurrences and series terms, where hundreds of them are spent per particle), two give 1.1e-16 (frcp)
__device__ __forceinline__ double frcp1(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double frcp(double b)
{
    double r = frcp1(b);
    double e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx crecip(cplx a)
{
    const double r = frcp1(a.re * a.re + a.im * a.im);
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
constexpr int OPT_MB = 16, OPT_RB = 8;   // checkpoint spacing, register block
// Checkpoints per thread: NCK of them cover a pass of NCK * OPT_MB terms; beyond, the series is taken in segments with
// one more downward pass each (mie_sphere_down).  The downward path is the fallback (see mie_upward_ok), so the
// kernels keep 4 (segments of 64 terms); few slots and the 8-value register block are what lets three waves per
// SIMD run.
constexpr int OPT_NCK = 12;   // default of the template parameter


// The values a term hands to the next one live in two slots by the PARITY of their index (psi_k, chi_k, a_k, b_k in
// slot k & 1): term n overwrites slot n & 1, which held the values of n - 2, in place.  The loops that take the terms are
// unrolled by an even count, so the parity is a compile-time constant and nothing is ever moved -- with "previous" and
// "one before" variables every term ended in six register moves (under the per-lane predicate n <= nstop a rename is
// not possible: lanes past their series keep their values).
struct MieAcc {
    double psi[2], chi[2], qs, qe, gs, sgn;
    cplx an[2], bn[2], xb;
};
__device__ __forceinline__ void mie_acc_start(MieAcc &A, double sinx, double cosx)
{
    A.psi[0] = sinx; A.psi[1] = cosx;      // psi_0 = sin x, psi_(-1) = cos x
    A.chi[0] = cosx; A.chi[1] = -sinx;     // chi_0 = cos x, chi_(-1) = -sin x
    A.qs = 0.; A.qe = 0.; A.gs = 0.; A.sgn = -1.;
    A.an[0] = {0., 0.}; A.an[1] = {0., 0.}; A.bn[0] = {0., 0.}; A.bn[1] = {0., 0.}; A.xb = {0., 0.};
}

// FULL = false: only the extinction and scattering sums (longwave bands use absorption alone); BSC = false: no
// backscatter sum (the radiation calls take tau, omega and g; bscoef is a diagnostic that is evaluated when asked for)
template <bool FULL, bool BSC = FULL>
__device__ __forceinline__ void mie_term(MieAcc &A, int n, cplx D, cplx m, cplx minv, double xinv, const MieN &K)
{
    const int par = n & 1;   // a constant after unrolling
    const double nx = K.en * xinv, tx = K.tnm1 * xinv;
    const double psi1 = A.psi[par ^ 1], chi1 = A.chi[par ^ 1];
    const double psi = tx * psi1 - A.psi[par];
    const double chi = tx * chi1 - A.chi[par];
    A.psi[par] = psi; A.chi[par] = chi;
    // a_n = (t psi_n - psi_(n-1)) / (t xi_n - xi_(n-1)) with xi = psi - i chi: the denominator is the numerator minus
    // i (t chi_n - chi_(n-1)) -- four operations instead of a complex product and a complex difference (fp64 only:
    // in the packed-fp32 terms the same rearrangement cost accuracy -- the mixed path's heating-rate bound of
    // tests/test_full_domain_gpu.py went from inside 5e-3 K/day to 8.5e-3 -- and was withdrawn there)
    cplx t = cmul(D, minv);
    t.re += nx;
    const cplx na = {t.re * psi - psi1, t.im * psi};
    const cplx ca = {t.re * chi - chi1, t.im * chi};
    const cplx da = {na.re + ca.im, na.im - ca.re};
    t = cmul(m, D);
    t.re += nx;
    const cplx nb = {t.re * psi - psi1, t.im * psi};
    const cplx cb = {t.re * chi - chi1, t.im * chi};
    const cplx db = {nb.re + cb.im, nb.im - cb.re};
    // a_n = na/da, b_n = nb/db with one reciprocal: 1/|da|^2 = |db|^2 / (|da|^2 |db|^2)
    const double ma = da.re * da.re + da.im * da.im, mb = db.re * db.re + db.im * db.im;
    const double r = frcp1(ma * mb);
    const double ra = r * mb, rb = r * ma;
    if (!FULL) {
        // extinction and scattering only: Re a_n and |a_n|^2 = |na|^2 / |da|^2 without forming a_n
        const double are = (na.re * da.re + na.im * da.im) * ra, bre = (nb.re * db.re + nb.im * db.im) * rb;
        const double a2 = (na.re * na.re + na.im * na.im) * ra, b2 = (nb.re * nb.re + nb.im * nb.im) * rb;
        A.qs += K.w2 * (a2 + b2);
        A.qe += K.w2 * (are + bre);
    } else {
        const cplx an = {(na.re * da.re + na.im * da.im) * ra, (na.im * da.re - na.re * da.im) * ra};
        const cplx bn = {(nb.re * db.re + nb.im * db.im) * rb, (nb.im * db.re - nb.re * db.im) * rb};
        A.qs += K.w2 * (an.re * an.re + an.im * an.im + bn.re * bn.re + bn.im * bn.im);
        A.qe += K.w2 * (an.re + bn.re);
        A.gs += K.gx * (an.re * bn.re + an.im * bn.im);
        if (n > 1) {
            const cplx an1 = A.an[par ^ 1], bn1 = A.bn[par ^ 1];
            A.gs += K.gy * (an1.re * an.re + an1.im * an.im + bn1.re * bn.re + bn1.im * bn.im);
        }
        if (BSC) {
            A.xb.re += A.sgn * K.w2 * (an.re - bn.re);
            A.xb.im += A.sgn * K.w2 * (an.im - bn.im);
            A.sgn = -A.sgn;
        }
        A.an[par] = an; A.bn[par] = bn;
    }
}

// Where the logarithmic derivative may be carried UPWARD with the series (Wiscombe's MIEV0 does the same under a
// similar criterion): D_0 = cot y, D_n = 1 / (n/y - D_(n-1)) - n/y.  The upward recurrence loses accuracy when the
// particle absorbs strongly (Im(m) x large) and at very small size parameters (cancellation in 1/y - cot y); inside
// x >= 0.3, Im(m) x < 10 the efficiencies agree with the downward specification to 4e-12 (asymmetry 4e-12,
// backscatter 6e-11; 60 000 random cases, x to 225, Re m 1.1..2.6, Im m 1e-9..1: tools/probes/mie_updown.py and
// tests/test_optics_oracle.py; below x = 0.3 the difference grows like 1/x^2: 3e-10 at 0.1, 4e-8 at 0.03).  It costs one recurrence step per term instead of the downward pass from the start
// order plus the rebuild (about 3 steps per term), needs no checkpoints, and nearly all aerosol qualifies.
constexpr double OPT_UP_XMIN = 0.3, OPT_UP_ABSMAX = 10.0;
__device__ __forceinline__ bool mie_upward_ok(double x, cplx m) { return x >= OPT_UP_XMIN && m.im * x < OPT_UP_ABSMAX; }

// sin and cos of 0 <= x < 1e5 (size parameters reach 225, |m| x 600): the library's sincos carries the reduction of
// arbitrary arguments -- two of them per evaluation were a tenth of the shortwave kernel's instructions.  Here: n =
// rint(x 2/pi), three-part Cody-Waite reduction with 33-bit pieces of pi/2 (n pi/2_1 and n pi/2_2 are exact products),
// the kernel polynomials of fdlibm on |r| <= pi/4, quadrant by selects.  Absolute error about 1e-16 (1 + x 1e-16),
// far inside the stage's tolerance.
__device__ __forceinline__ void sincos_mid(double x, double &sn, double &cs)
{
    const double fn = rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-fn, 6.07710050630396597660e-11, r);
    r = __builtin_fma(-fn, 2.02226624871116645580e-21, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(ps * z, r, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)fn & 3;
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

// cot(a + ib) = (sin 2a - i sinh 2b) / (cosh 2b - cos 2a) = (sa ca - i sh ch) / (sa^2 + sh^2), b >= 0: one sincos, one
// sinh (its own series below 0.25, where (e^b - e^-b)/2 would cancel), cosh from sinh
__device__ __forceinline__ cplx ccot(cplx y)
{
    double sa, ca;
    sincos_mid(y.re, sa, ca);
    const double b = y.im, b2 = b * b;
    double sh;
    if (b < 0.25) {
        double p = 1. / 39916800.;                       // b^11 / 11!: the next term is below 1e-17 of the sum
        p = __builtin_fma(p, b2, 1. / 362880.);
        p = __builtin_fma(p, b2, 1. / 5040.);
        p = __builtin_fma(p, b2, 1. / 120.);
        p = __builtin_fma(p, b2, 1. / 6.);
        sh = __builtin_fma(p * b2, b, b);
    } else {
        const double e = exp(b);
        sh = 0.5 * (e - frcp(e));
    }
    const double ch = sqrt(__builtin_fma(sh, sh, 1.));
    const double r = frcp(__builtin_fma(sa, sa, sh * sh));
    return {sa * ca * r, -(sh * ch) * r};
}

template <bool FULL, bool BSC = FULL>
__device__ __forceinline__ void mie_term(MieAcc &A, int n, cplx D, cplx m, cplx minv, double xinv)
{
    mie_term<FULL, BSC>(A, n, D, m, minv, xinv, mie_n(n, FULL ? c_gw[n] : make_double2(0., 0.)));
}

// cbx: cube root of x (the caller has it from the particle's diameter: one cube root per section, not per wavelength)
template <bool FULL, bool BSC = FULL>
__device__ __forceinline__ void mie_sphere_up(double x, cplx m, double cbx, double &qext, double &qsca, double &gasy,
                                              double &qback)
{
    const cplx y = {x * m.re, x * m.im};
    const int nstop = (int)(x + 4. * cbx + 2.);
    const cplx yinv = crecip(y);
    MieAcc A;
    {
        double sn, cs;
        sincos_mid(x, sn, cs);
        mie_acc_start(A, sn, cs);
    }
    const double xinv = frcp(x);
    const cplx minv = crecip(m);
    cplx d = ccot(y);
    // the per-term records of the NEXT four terms are fetched while these four are evaluated: read where they are
    // used, each scalar load of a term stood in front of a wait
    MieN kn[4], knn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kn[i] = c_mn[i + 1];
    for (int n0 = 0; n0 < nstop; n0 += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) knn[i] = c_mn[n0 + 5 + i];
#pragma unroll
        for (int i = 1; i <= 4; ++i) {
            const int n = n0 + i;
            if (n <= nstop) {
                const cplx ny = {kn[i - 1].en * yinv.re, kn[i - 1].en * yinv.im};
                d = csub(crecip(csub(ny, d)), ny);
                mie_term<FULL, BSC>(A, n, d, m, minv, xinv, kn[i - 1]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) kn[i] = knn[i];
    }
    const double x2inv = xinv * xinv;
    qsca = 2. * x2inv * A.qs;
    qext = 2. * x2inv * A.qe;
    gasy = FULL && A.qs > 0. ? 2. * A.gs * frcp(A.qs) : 0.;
    qback = BSC ? x2inv * (A.xb.re * A.xb.re + A.xb.im * A.xb.im) : 0.;
}

// the downward evaluation (strongly absorbing or very small particles)
template <bool FULL, int NCK = OPT_NCK, bool BSC = FULL>
__device__ __forceinline__ void mie_sphere_down(double x, cplx m, double2 *ck /* LDS [NCK][OPT_THREADS], this thread's column */,
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
    MieAcc A;
    {
        double sn, cs;
        sincos_mid(x, sn, cs);
        mie_acc_start(A, sn, cs);
    }
    const double xinv = frcp(x);
    const cplx minv = crecip(m);
    for (int seg = 0; seg < nck; seg += NCK) {   // once, unless the series is longer than NCK * OPT_MB terms
    const int nseg = nck - seg < NCK ? nck - seg : NCK;
    {
        // downward pass from the start order to the segment's first block: D_{n-1} = n/y - 1/(D_n + n/y); the
        // checkpoints D_{MB (seg + j)}, j = 1..nseg, stay in LDS
        const int nlow = seg * OPT_MB, ntop = (seg + nseg) * OPT_MB;
        cplx d = {0., 0.};
        for (int n = nmx; n >= 2; --n) {
            const cplx ny = {(double)n * yinv.re, (double)n * yinv.im};
            d = csub(ny, crecip(cadd(d, ny)));
            const int n1 = n - 1;
            if (n1 <= ntop && (n1 & (OPT_MB - 1)) == 0) ck[(size_t)((n1 - nlow) / OPT_MB - 1) * OPT_THREADS] = make_double2(d.re, d.im);
            if (n1 <= nlow + OPT_MB) break;       // the lowest block is rebuilt from its checkpoint below
        }
    }
    for (int js = 0; js < nseg; ++js) {
        const int j = seg + js;
        // the block's 16 values of D_n are rebuilt from its checkpoint in two halves of OPT_RB (the register file
        // holds 8 of them, not 16: a third wave per SIMD); the chain to the lower half passes through the upper one
        // a second time -- the same operations on the same values
        const double2 c = ck[(size_t)js * OPT_THREADS];
        const int n0 = j * OPT_MB;
        cplx dd[OPT_RB];
        cplx d = {c.x, c.y};                          // D_{n0 + 16}
#pragma unroll
        for (int i = OPT_MB; i > OPT_RB; --i) {       // down to D_{n0 + 8}
            const double en = (double)(n0 + i);
            const cplx ny = {en * yinv.re, en * yinv.im};
            d = csub(ny, crecip(cadd(d, ny)));
        }
        dd[OPT_RB - 1] = d;
#pragma unroll
        for (int i = OPT_RB - 1; i >= 1; --i) {       // dd[i] = D_n with n = n0 + i + 1
            const double en = (double)(n0 + i + 1);
            const cplx ny = {en * yinv.re, en * yinv.im};
            dd[i - 1] = csub(ny, crecip(cadd(dd[i], ny)));
        }
#pragma unroll
        for (int i = 0; i < OPT_RB; ++i) {
            const int n = n0 + i + 1;
            if (n <= nstop) mie_term<FULL, BSC>(A, n, dd[i], m, minv, xinv);
        }
        if (n0 + OPT_RB < nstop) {
            dd[OPT_RB - 1] = {c.x, c.y};
#pragma unroll
            for (int i = OPT_RB - 1; i >= 1; --i) {   // dd[i] = D_n with n = n0 + 8 + i + 1
                const double en = (double)(n0 + OPT_RB + i + 1);
                const cplx ny = {en * yinv.re, en * yinv.im};
                dd[i - 1] = csub(ny, crecip(cadd(dd[i], ny)));
            }
#pragma unroll
            for (int i = 0; i < OPT_RB; ++i) {
                const int n = n0 + OPT_RB + i + 1;
                if (n <= nstop) mie_term<FULL, BSC>(A, n, dd[i], m, minv, xinv);
            }
        }
    }
    }
    const double x2inv = xinv * xinv;
    qsca = 2. * x2inv * A.qs;
    qext = 2. * x2inv * A.qe;
    gasy = FULL && A.qs > 0. ? 2. * A.gs * frcp(A.qs) : 0.;
    qback = BSC ? x2inv * (A.xb.re * A.xb.re + A.xb.im * A.xb.im) : 0.;
}
// ... as a real function call, for the longwave kernel: there only particles beyond any aerosol size come here, and
// inlined twice (one per band of the pair) the code raises the register pressure of the whole kernel
struct MieQ2 { double qext, qsca; };
constexpr int OPT_NCK_LW = 2;   // checkpoint slots per thread of the longwave kernel (segments of 32 terms; 36 KB of LDS with the sums: four workgroups per CU with room to spare)
__device__ __noinline__ MieQ2 mie_sphere_down_call(double x, cplx m, double2 *ck)
{
    double qe, qs, g, b;
    mie_sphere_down<false, OPT_NCK_LW>(x, m, ck, qe, qs, g, b);
    return {qe, qs};
}

template <bool FULL, int NCK = OPT_NCK, bool BSC = FULL>
__device__ __forceinline__ void mie_sphere(double x, cplx m, double cbx, double2 *ck, double &qext, double &qsca,
                                           double &gasy, double &qback)
{
    if (mie_upward_ok(x, m)) mie_sphere_up<FULL, BSC>(x, m, cbx, qext, qsca, gasy, qback);   // the particle's own choice: no vote
    else mie_sphere_down<FULL, NCK, BSC>(x, m, ck, qext, qsca, gasy, qback);
}

// Two longwave evaluations of one particle (neighbouring bands) travel side by side (mie_pair_su below; the fp32
// kernels pack them into one register pair): the recurrences of D_n are serial chains of ~10 dependent fp64 operations
// per step and the longwave series are short, so a single evaluation leaves the SIMD waiting on its own results; two
// independent chains in one instruction stream fill those gaps.  Register blocks of OPT_MB2 terms.
constexpr int OPT_MB2 = 8, OPT_NCK2 = 8;   // the fp32 kernels keep 8 + 4 checkpoint slots per thread
constexpr double OPT_XMAX32 = 47.0;   // x + 4 x^(1/3) + 2 < 64 = OPT_MB2 * OPT_NCK2
constexpr int OPT_NCK32 = OPT_NCK2 + OPT_NCK2 / 2;   // LDS slots per thread of the fp32 pair

// ---- one particle in two neighbouring longwave bands, small size parameters (|mx| <= 4, at most OPT_MB2 terms) ----
// Most longwave evaluations are of this kind (x = pi d / lambda with d <= 10 um and lambda >= 3 um).  The whole
// logarithmic-derivative ladder D_1..D_8 then fits the registers: the downward recurrence runs ONCE, from a start
// order chosen from its own contraction rate -- an error e in D_n becomes e |y|^2 / (2n+1)^2 in D_(n-1), so the
// arbitrary start value D_N = 0 has decayed below 1e-20 after a handful of steps (on 17 000 random cases with
// |y| <= 2 and 15 600 with 2 < |y| <= 4 the ladder equals the one started at N = 120 / 140 bit for bit, the latter
// even with one step fewer than the rule gives; the fixed rule of the general path starts 16 above) -- and nothing goes through LDS.  sin x and cos x come from their Taylor series (x <= 2.1, no reduction).
__device__ __forceinline__ void sincos_small(double x, double &sn, double &cs)
{
    const double z = x * x;
    {
        double p = 6.44695028438447359e-26;
        p = __builtin_fma(p, z, -3.86817017063068413e-23);
        p = __builtin_fma(p, z, 1.95729410633912626e-20);
        p = __builtin_fma(p, z, -8.22063524662432950e-18);
        p = __builtin_fma(p, z, 2.81145725434552060e-15);
        p = __builtin_fma(p, z, -7.64716373181981641e-13);
        p = __builtin_fma(p, z, 1.60590438368216133e-10);
        p = __builtin_fma(p, z, -2.50521083854417202e-08);
        p = __builtin_fma(p, z, 2.75573192239858925e-06);
        p = __builtin_fma(p, z, -1.98412698412698413e-04);
        p = __builtin_fma(p, z, 8.33333333333333322e-03);
        p = __builtin_fma(p, z, -1.66666666666666657e-01);
        p = __builtin_fma(p, z, 1.00000000000000000e+00);
        sn = p * x;
    }
    {
        double p = 1.61173757109611839e-24;
        p = __builtin_fma(p, z, -8.89679139245057408e-22);
        p = __builtin_fma(p, z, 4.11031762331216484e-19);
        p = __builtin_fma(p, z, -1.56192069685862253e-16);
        p = __builtin_fma(p, z, 4.77947733238738525e-14);
        p = __builtin_fma(p, z, -1.14707455977297245e-11);
        p = __builtin_fma(p, z, 2.08767569878681002e-09);
        p = __builtin_fma(p, z, -2.75573192239858883e-07);
        p = __builtin_fma(p, z, 2.48015873015873016e-05);
        p = __builtin_fma(p, z, -1.38888888888888894e-03);
        p = __builtin_fma(p, z, 4.16666666666666644e-02);
        p = __builtin_fma(p, z, -5.00000000000000000e-01);
        p = __builtin_fma(p, z, 1.00000000000000000e+00);
        cs = p;
    }
}

struct MieSetupS {
    cplx yinv, minv;
    double xinv;
    int nstop, nstart;
    bool small;
};
// cbx: cube root of x (the caller has it from the particle's diameter)
__device__ __forceinline__ MieSetupS mie_setup_small(double x, cplx m, double cbx)
{
    MieSetupS s;
    const cplx y = {x * m.re, x * m.im};
    const double ay2 = y.re * y.re + y.im * y.im;
    s.nstop = (int)(x + 4. * cbx + 2.);
    s.small = ay2 <= 4.0 && s.nstop <= OPT_MB2;
    // steps until the start error has contracted by 1e-20: each contributes at least 2 ln((2 nstop + 3) / |y|)
    const float g = 2.f * __logf((float)(2 * s.nstop + 3) * __builtin_amdgcn_rsqf((float)ay2));
    const int steps = (int)(46.1f * __builtin_amdgcn_rcpf(g)) + 2;
    const int n0 = s.nstop + steps;
    s.nstart = n0 < OPT_MB2 + 1 ? OPT_MB2 + 1 : n0;
    s.yinv = crecip(y);
    s.minv = crecip(m);
    s.xinv = frcp1(x);
    return s;
}

// Small-size and upward pairs in ONE pass over the terms.  A fifth of the longwave waves hold both kinds (the limit
// |mx| = 2 cuts through a section's size range); run one kind after the other such a wave pays for the first eight
// terms twice.  Here every lane prepares its own source of D_n -- the ladder D_1..D_8 (small) or D_0 = cot y (upward)
// -- and the terms are taken together: per lane the same operations in the same order as if its kind ran alone, so
// a particle's result does not depend on the company it keeps.
__device__ __forceinline__ void mie_pair_su(double x1, cplx m1, double x2, cplx m2, const MieSetupS &s1,
                                            const MieSetupS &s2, const bool small, double &qext1, double &qsca1,
                                            double &qext2, double &qsca2)
{
    cplx dd1[OPT_MB2], dd2[OPT_MB2];
    cplx d1 = {0., 0.}, d2 = {0., 0.};
    MieAcc A1, A2;
    double sn1, cs1, sn2, cs2;
    if (small) {
        const int nstart = s1.nstart > s2.nstart ? s1.nstart : s2.nstart;
        for (int n = nstart; n > OPT_MB2 + 1; --n) {   // D_{n-1} = n/y - 1/(D_n + n/y), down to D_9
            const double en = (double)n;
            const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
            d1 = csub(ny1, crecip(cadd(d1, ny1)));
            d2 = csub(ny2, crecip(cadd(d2, ny2)));
        }
#pragma unroll
        for (int i = OPT_MB2 - 1; i >= 0; --i) {       // dd[i] = D_{i+1}
            const double en = (double)(i + 2);
            const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
            d1 = csub(ny1, crecip(cadd(d1, ny1)));
            d2 = csub(ny2, crecip(cadd(d2, ny2)));
            dd1[i] = d1; dd2[i] = d2;
        }
        sincos_small(x1, sn1, cs1);
        sincos_small(x2, sn2, cs2);
    } else {
#pragma unroll
        for (int i = 0; i < OPT_MB2; ++i) { dd1[i] = {0., 0.}; dd2[i] = {0., 0.}; }
        d1 = ccot({x1 * m1.re, x1 * m1.im});
        d2 = ccot({x2 * m2.re, x2 * m2.im});
        sincos_mid(x1, sn1, cs1);
        sincos_mid(x2, sn2, cs2);
    }
    mie_acc_start(A1, sn1, cs1);
    mie_acc_start(A2, sn2, cs2);
#pragma unroll
    for (int i = 0; i < OPT_MB2; ++i) {
        const int n = i + 1;
        if (!small) {
            const double en = (double)n;
            const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
            d1 = csub(crecip(csub(ny1, d1)), ny1);
            d2 = csub(crecip(csub(ny2, d2)), ny2);
        }
        const cplx D1 = small ? dd1[i] : d1, D2 = small ? dd2[i] : d2;
        if (n <= s1.nstop) mie_term<false>(A1, n, D1, m1, s1.minv, s1.xinv);
        if (n <= s2.nstop) mie_term<false>(A2, n, D2, m2, s2.minv, s2.xinv);
    }
    // beyond eight terms only upward lanes are left
    const int nstopm = s1.nstop > s2.nstop ? s1.nstop : s2.nstop;
    for (int n0 = OPT_MB2; n0 < nstopm; n0 += 4) {
#pragma unroll
        for (int i = 1; i <= 4; ++i) {
            const int n = n0 + i;
            const double en = (double)n;
            const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
            d1 = csub(crecip(csub(ny1, d1)), ny1);
            d2 = csub(crecip(csub(ny2, d2)), ny2);
            if (n <= s1.nstop) mie_term<false>(A1, n, d1, m1, s1.minv, s1.xinv);
            if (n <= s2.nstop) mie_term<false>(A2, n, d2, m2, s2.minv, s2.xinv);
        }
    }
    const double f1 = 2. * s1.xinv * s1.xinv, f2 = 2. * s2.xinv * s2.xinv;
    qsca1 = f1 * A1.qs; qext1 = f1 * A1.qe;
    qsca2 = f2 * A2.qs; qext2 = f2 * A2.qe;
}

// ---- single-precision pair (the "mixed" optics path: longwave bands in fp32, everything else fp64) ----
// Same series, two changes that fp32 needs.  (1) psi_n(x) is carried upward by its recurrence only while n <= x
// (the oscillatory range, where errors do not grow); beyond, psi_n = psi_{n-1} / (n/x + A_n(x)) with the logarithmic
// derivative A_n of the real argument obtained by the same downward recurrence as D_n -- the plain recurrence
// cancels catastrophically there (psi_1 = sin x / x - cos x is off by 5 % at x = 0.002 in fp32).  (2) a_n and b_n
// take one reciprocal each (the product of their denominators overflows fp32).  Reciprocals are v_rcp_f32 (1 ulp).
// Measured against the fp64 oracle over x = 0.0016 .. 40: absorption efficiency within 6e-5, median 2e-7.
__device__ __forceinline__ float rcpf(float b) { return __builtin_amdgcn_rcpf(b); }
// The two evaluations of a pair travel as the two components of packed fp32 values (v_pk_fma_f32 and friends
// process both per instruction); complex numbers are (re, im) of such pairs.
typedef float v2f __attribute__((ext_vector_type(2)));
struct cplx2 { v2f re, im; };
__device__ __forceinline__ v2f rcp2(v2f b) { return v2f{rcpf(b.x), rcpf(b.y)}; }
__device__ __forceinline__ v2f sel2(bool c0, bool c1, v2f a, v2f b) { return v2f{c0 ? a.x : b.x, c1 ? a.y : b.y}; }
__device__ __forceinline__ cplx2 cadd2(cplx2 a, cplx2 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx2 csub2(cplx2 a, cplx2 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx2 cmul2(cplx2 a, cplx2 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx2 crecip2(cplx2 a)
{
    const v2f r = rcp2(a.re * a.re + a.im * a.im);
    return {a.re * r, -a.im * r};
}
struct MieSetup32 {      // both evaluations of the pair
    cplx2 m, yinv, minv;
    v2f x, xinv;
    int nstop1, nstop2, nmx;
};
__device__ __forceinline__ MieSetup32 mie_setup32(float x1, float m1re, float m1im, float x2, float m2re, float m2im)
{
    MieSetup32 s;
    s.x = v2f{x1, x2};
    s.m = {v2f{m1re, m2re}, v2f{m1im, m2im}};
    const cplx2 y = {s.x * s.m.re, s.x * s.m.im};
    const v2f ymod2 = y.re * y.re + y.im * y.im;
    const float ym1 = sqrtf(ymod2.x), ym2 = sqrtf(ymod2.y);
    const float xs1 = s.x.x + 4.f * cbrtf(s.x.x) + 2.f, xs2 = s.x.y + 4.f * cbrtf(s.x.y) + 2.f;
    s.nstop1 = (int)xs1; s.nstop2 = (int)xs2;
    const float b1 = fmaxf(xs1, ym1), b2 = fmaxf(xs2, ym2);
    const int n1 = (int)(b1 + (b1 > 8.f ? 8.f * cbrtf(b1) : 0.f)) + 16, n2 = (int)(b2 + (b2 > 8.f ? 8.f * cbrtf(b2) : 0.f)) + 16;
    s.nmx = n1 > n2 ? n1 : n2;      // a higher start is only more accurate
    s.yinv = crecip2(y);
    s.minv = crecip2(s.m);
    s.xinv = rcp2(s.x);
    return s;
}
struct MieAcc32 { v2f p[2], c[2], qs, qe; };   // psi_k, chi_k in slot k & 1 (as MieAcc: no moves between terms), sums
__device__ __forceinline__ void mie_term32(MieAcc32 &A, int n, cplx2 D, v2f An, const MieSetup32 &s)
{
    const float en = (float)n;
    const v2f nx = en * s.xinv, up = (2.f * en - 1.f) * s.xinv;
    const int par = n & 1;   // a constant after unrolling
    const v2f p1 = A.p[par ^ 1], c1 = A.c[par ^ 1];
    const v2f psi = sel2(en > s.x.x, en > s.x.y, p1 * rcp2(nx + An), up * p1 - A.p[par]);
    const v2f chi = up * c1 - A.c[par];
    A.p[par] = psi; A.c[par] = chi;
    const cplx2 xi = {psi, -chi}, xi1 = {p1, -c1};
    cplx2 t = cmul2(D, s.minv);
    t.re += nx;
    const cplx2 na = {t.re * psi - p1, t.im * psi}, da = csub2(cmul2(t, xi), xi1);
    t = cmul2(s.m, D);
    t.re += nx;
    const cplx2 nb = {t.re * psi - p1, t.im * psi}, db = csub2(cmul2(t, xi), xi1);
    const v2f ra = rcp2(da.re * da.re + da.im * da.im), rb = rcp2(db.re * db.re + db.im * db.im);
    const cplx2 an = {(na.re * da.re + na.im * da.im) * ra, (na.im * da.re - na.re * da.im) * ra};
    const cplx2 bn = {(nb.re * db.re + nb.im * db.im) * rb, (nb.im * db.re - nb.re * db.im) * rb};
    const float w2 = 2.f * en + 1.f;
    const v2f zero = v2f{0.f, 0.f};
    // an evaluation whose series has ended adds nothing
    A.qs += sel2(n <= s.nstop1, n <= s.nstop2, w2 * (an.re * an.re + an.im * an.im + bn.re * bn.re + bn.im * bn.im), zero);
    A.qe += sel2(n <= s.nstop1, n <= s.nstop2, w2 * (an.re + bn.re), zero);
}
// register blocks of OPT_MB2 terms; checkpoints of D_n (both evaluations) in slots 0..NCK2-1 and of A_n (both, two
// checkpoints per slot) in slots NCK2..NCK2+NCK2/2-1 of the thread's column: OPT_NCK32 slots of 16 bytes
__device__ __forceinline__ void mie_pair32(const MieSetup32 &s, double2 *ck, float &qext1, float &qsca1,
                                           float &qext2, float &qsca2)
{
    const int nstopm = s.nstop1 > s.nstop2 ? s.nstop1 : s.nstop2;
    const int nck = (nstopm + OPT_MB2 - 1) / OPT_MB2, ntop = nck * OPT_MB2;
    cplx2 d = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    v2f a = v2f{0.f, 0.f};
    for (int n = s.nmx; n >= 2; --n) {   // D_{n-1} = n/y - 1/(D_n + n/y), A_{n-1} = n/x - 1/(A_n + n/x)
        const float en = (float)n;
        const cplx2 ny = {en * s.yinv.re, en * s.yinv.im};
        d = csub2(ny, crecip2(cadd2(d, ny)));
        const v2f nx = en * s.xinv;
        a = nx - rcp2(a + nx);
        const int n1 = n - 1;
        if (n1 <= ntop && (n1 & (OPT_MB2 - 1)) == 0) {
            const int j = n1 / OPT_MB2 - 1;
            *reinterpret_cast<float4 *>(&ck[(size_t)j * OPT_THREADS]) = make_float4(d.re.x, d.re.y, d.im.x, d.im.y);
            reinterpret_cast<float2 *>(&ck[(size_t)(OPT_NCK2 + (j >> 1)) * OPT_THREADS])[j & 1] = make_float2(a.x, a.y);
        }
        if (n1 <= OPT_MB2) break;
    }
    MieAcc32 A;
    float sn1, cs1, sn2, cs2;
    sincosf(s.x.x, &sn1, &cs1);
    sincosf(s.x.y, &sn2, &cs2);
    A.p[0] = v2f{sn1, sn2}; A.p[1] = v2f{cs1, cs2}; A.c[0] = A.p[1]; A.c[1] = -A.p[0];   // psi_0, psi_(-1), chi_0, chi_(-1)
    A.qs = v2f{0.f, 0.f}; A.qe = v2f{0.f, 0.f};
    for (int j = 0; j < nck; ++j) {
        cplx2 dd[OPT_MB2];
        v2f aa[OPT_MB2];
        const float4 cd = *reinterpret_cast<const float4 *>(&ck[(size_t)j * OPT_THREADS]);
        const float2 ca = reinterpret_cast<const float2 *>(&ck[(size_t)(OPT_NCK2 + (j >> 1)) * OPT_THREADS])[j & 1];
        dd[OPT_MB2 - 1] = {v2f{cd.x, cd.y}, v2f{cd.z, cd.w}};
        aa[OPT_MB2 - 1] = v2f{ca.x, ca.y};
#pragma unroll
        for (int i = OPT_MB2 - 1; i >= 1; --i) {
            const float en = (float)(j * OPT_MB2 + i + 1);
            const cplx2 ny = {en * s.yinv.re, en * s.yinv.im};
            dd[i - 1] = csub2(ny, crecip2(cadd2(dd[i], ny)));
            const v2f nx = en * s.xinv;
            aa[i - 1] = nx - rcp2(aa[i] + nx);
        }
#pragma unroll
        for (int i = 0; i < OPT_MB2; ++i) {
            const int n = j * OPT_MB2 + i + 1;
            if (n <= nstopm) mie_term32(A, n, dd[i], aa[i], s);
        }
    }
    const v2f f = 2.f * s.xinv * s.xinv;
    const v2f qs = f * A.qs, qe = f * A.qe;
    qsca1 = qs.x; qext1 = qe.x;
    qsca2 = qs.y; qext2 = qe.y;
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

// Shortwave, fp64: one thread = one cell with all four wavelengths, as the longwave kernel takes its bands: particle
// geometry and tracers once per section, the sixteen running sums in the thread's LDS column, and every workgroup
// costs the same.  (With one wavelength per workgroup and the four wavelengths of a cell group back-to-back the CUs
// stayed half empty -- the wavelengths differ 2x in series length and workgroups of unequal duration interleaved
// badly: 1.7 of 3 waves per SIMD resident; wavelength-major order fixed that, 5.38 -> 4.19 ms per 1.6 M cells; per
// cell: 3.94 ms.)  The downward fallback keeps 4 checkpoints per thread (segments of 64 terms).
constexpr int OPT_NCK_SWC = 4;
template <bool BSC>   // BSC: a backscatter output (bscoef) is requested
__device__ __forceinline__ void optics_sw_cell(const OpticsKernelArgs &a, double2 *s_ck, double *s_acc)
{
    const long cell = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (cell >= ncell) return;
    unsigned want = 0;
#pragma unroll
    for (int w = 0; w < OPT_NSW; ++w)
        if (a.tauaer[w] || a.extaer[w] || a.bscoef[w]) want |= 1u << w;
    if (!want) return;
    double2 *ck = s_ck + threadIdx.x;
    double *acc = s_acc + threadIdx.x;   // [wavelength][ext, sca, g sca, bsc][thread]
#pragma unroll
    for (int i = 0; i < 4 * OPT_NSW; ++i) acc[i * OPT_THREADS] = 0.;
    const int it = (int)(cell % a.ni_t);
    const long r = cell / a.ni_t;
    const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
    const size_t q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const double alt = ldr(a.alt, q, r4), dz = ldr(a.dz8w, q, r4);
    const double pi = 3.14159265358979323846;
    for (int ib = 0; ib < a.nbin; ++ib) {
        double vol[OPT_NSPEC];   // 1e-6 cm3 per kg of dry air
        double vdry = 0.;
#pragma unroll
        for (int sp = 0; sp < OPT_NSPEC; ++sp) {
            const int id = a.idx[ib][sp];
            const double mass = id >= 0 ? fmax(ldr(a.chem, (size_t)id * sspec + q, r4), 0.) : 0.;
            vol[sp] = mass * c_opt.rdens[sp];
            if (sp != OS_WATER) vdry += vol[sp];
        }
        const int idn = a.idx[ib][OS_NUM];
        double num = idn >= 0 ? fmax(ldr(a.chem, (size_t)idn * sspec + q, r4), 0.) : 0.;
        if (!(vdry > 1.e-25)) continue;
        const double vwet = vdry + vol[OS_WATER];
        const double vpart = vdry * 1.e-6 * frcp(fmax(num, 1.e-30));   // cm3 per particle
        double ddry = 1.e4 * cbrt(vpart * (6. / pi));
        const double dlo = c_opt.dlo_um[ib], dhi = c_opt.dhi_um[ib];
        if (ddry < dlo || ddry > dhi) {
            ddry = ddry < dlo ? dlo : dhi;
            const double dcm = ddry * 1.e-4;
            num = vdry * 1.e-6 * frcp((pi / 6.) * dcm * dcm * dcm);
        }
        const double dwet = ddry * cbrt(vwet * frcp(vdry));
        const double cbd = cbrt(dwet);   // x^(1/3) = dwet^(1/3) (pi / lambda)^(1/3): one cube root for the four wavelengths
        const double dcm = dwet * 1.e-4;
        const double cross = (num * frcp(alt) * 1.e-6) * (0.25 * pi * dcm * dcm);   // #/cm3 * cm2
#pragma unroll 1
        for (int w = 0; w < OPT_NSW; ++w) {
            if (!((want >> w) & 1)) continue;
            const cplx m = mix_index(vol, vwet, a.mix, w);
            double x = pi * dwet * frcp(c_opt.wavelength_um[w]);
            const bool capped = x > OPT_XMAX;
            x = capped ? OPT_XMAX : x;
            const double cbx = capped ? 6.0822019955734001 /* 225^(1/3) */ : cbd * c_pil[1][w];
            double qe, qs, g = 0., qb = 0.;
            mie_sphere<true, OPT_NCK_SWC, BSC>(x, m, cbx, ck, qe, qs, g, qb);
            double *s = acc + (size_t)(4 * w) * OPT_THREADS;
            s[0] += cross * qe;
            s[OPT_THREADS] += cross * qs;
            s[2 * OPT_THREADS] += cross * qs * g;
            s[3 * OPT_THREADS] += cross * qb * (0.25 / pi);
        }
    }
#pragma unroll 1
    for (int w = 0; w < OPT_NSW; ++w) {
        const double *s = acc + (size_t)(4 * w) * OPT_THREADS;
        const double ext = s[0], sca = s[OPT_THREADS], gsca = s[2 * OPT_THREADS], bsc = s[3 * OPT_THREADS];
        if (a.tauaer[w]) {
            str(a.tauaer[w], q, r4, ext * 100. * dz);
            str(a.waer[w], q, r4, ext > 0. ? sca * frcp(ext) : 0.);
            str(a.gaer[w], q, r4, sca > 0. ? gsca * frcp(sca) : 0.);
        }
        if (a.extaer[w]) str(a.extaer[w], q, r4, ext * 1.e5);
        if (a.bscoef[w]) str(a.bscoef[w], q, r4, bsc * 1.e5);
    }
}

// Longwave, fp64: one thread = one cell, all sixteen bands.  The particle geometry of a section (ten tracer loads, three
// cube roots, the clipping rule) is formed once instead of once per band pair; the pairs follow in a loop, and the 32
// running sums of the cell (extinction and scattering per band) live in the thread's LDS column, which a loop may
// index -- in registers they cost an occupancy step (round 2's first per-cell kernel: 256 VGPRs + 100 AGPRs).
__device__ __forceinline__ void optics_lw_cell(const OpticsKernelArgs &a, double2 *s_ck, double *s_acc)
{
    const long cell = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (cell >= ncell) return;
    unsigned want = 0;   // pairs with an output requested
#pragma unroll
    for (int p = 0; p < OPT_NLW / 2; ++p)
        if (a.tauaerlw[2 * p] || a.extaerlw[2 * p] || a.tauaerlw[2 * p + 1] || a.extaerlw[2 * p + 1]) want |= 1u << p;
    if (!want) return;
    double2 *ck = s_ck + threadIdx.x;
    double *acc = s_acc + threadIdx.x;   // [band][ext, sca][thread]
#pragma unroll
    for (int i = 0; i < 2 * OPT_NLW; ++i) acc[i * OPT_THREADS] = 0.;
    const int it = (int)(cell % a.ni_t);
    const long r = cell / a.ni_t;
    const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
    const size_t q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const double alt = ldr(a.alt, q, r4), dz = ldr(a.dz8w, q, r4);
    const double pi = 3.14159265358979323846;
    for (int ib = 0; ib < a.nbin; ++ib) {
        double vol[OPT_NSPEC];   // 1e-6 cm3 per kg of dry air
        double vdry = 0.;
#pragma unroll
        for (int sp = 0; sp < OPT_NSPEC; ++sp) {
            const int id = a.idx[ib][sp];
            const double mass = id >= 0 ? fmax(ldr(a.chem, (size_t)id * sspec + q, r4), 0.) : 0.;
            vol[sp] = mass * c_opt.rdens[sp];
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
        const double cbd = cbrt(dwet);   // x^(1/3) = dwet^(1/3) (pi / lambda)^(1/3): one cube root for all bands
        const double dcm = dwet * 1.e-4;
        const double cross = (num * frcp(alt) * 1.e-6) * (0.25 * pi * dcm * dcm);   // #/cm3 * cm2
#pragma unroll 1
        for (int p = 0; p < OPT_NLW / 2; ++p) {
            if (!((want >> p) & 1)) continue;
            const int w = OPT_NSW + 2 * p, w2 = w + 1;
            const cplx m = mix_index(vol, vwet, a.mix, w), mb = mix_index(vol, vwet, a.mix, w2);
            double x = pi * dwet * frcp(c_opt.wavelength_um[w]), xb = pi * dwet * frcp(c_opt.wavelength_um[w2]);
            x = x > OPT_XMAX ? OPT_XMAX : x;
            xb = xb > OPT_XMAX ? OPT_XMAX : xb;
            double qe, qs, qe2, qs2;
            // the register-only small-size ladder or the upward recurrence, the particle's own choice (a wave with
            // both kinds takes the terms in one pass, mie_pair_su): a cell's result does not depend on which cells
            // share its wave, i.e. on the tiling of the domain
            const MieSetupS t1 = mie_setup_small(x, m, cbd * c_pil[1][w]), t2 = mie_setup_small(xb, mb, cbd * c_pil[1][w2]);
            const bool small = t1.small && t2.small;
            if (small || (mie_upward_ok(x, m) && mie_upward_ok(xb, mb))) {
                mie_pair_su(x, m, xb, mb, t1, t2, small, qe, qs, qe2, qs2);
            } else {   // strongly absorbing beyond any aerosol at these wavelengths: downward, one band after the other
                const MieQ2 q1 = mie_sphere_down_call(x, m, ck), q2 = mie_sphere_down_call(xb, mb, ck);
                qe = q1.qext; qs = q1.qsca; qe2 = q2.qext; qs2 = q2.qsca;
            }
            double *s = acc + (size_t)(4 * p) * OPT_THREADS;
            s[0] += cross * qe;
            s[OPT_THREADS] += cross * qs;
            s[2 * OPT_THREADS] += cross * qe2;
            s[3 * OPT_THREADS] += cross * qs2;
        }
    }
#pragma unroll 1
    for (int ib = 0; ib < OPT_NLW; ++ib) {
        const double ext = acc[(size_t)(2 * ib) * OPT_THREADS], sca = acc[(size_t)(2 * ib + 1) * OPT_THREADS];
        const double ab = fmax(ext - sca, 0.);
        if (a.tauaerlw[ib]) str(a.tauaerlw[ib], q, r4, ab * 100. * dz);
        if (a.extaerlw[ib]) str(a.extaerlw[ib], q, r4, ab * 1.e5);
    }
}

// Mixed path, longwave band pair: the whole cell evaluation in fp32 (particle geometry, refractive index of the
// mixture, Mie series); tracers are read and optical depths written in the arrays' own type.
struct cf32 { float re, im; };
__device__ __forceinline__ cf32 mix_index32(const float *vol, float vwet, int mix, int w)
{
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int sp = 0; sp < OPT_NSPEC; ++sp) {
        if (mix == 1 && sp == OS_BC) continue;
        sr += vol[sp] * (float)c_opt.refr[sp][w];
        si += vol[sp] * (float)c_opt.refi[sp][w];
    }
    if (mix == 0) {
        const float vi = rcpf(vwet);
        return {sr * vi, si * vi};
    }
    const float vbc = vol[OS_BC], vmat = vwet - vbc;
    const cf32 mbc = {(float)c_opt.refr[OS_BC][w], (float)c_opt.refi[OS_BC][w]};
    if (!(vmat > 1.e-30f * vwet)) return mbc;
    const float vi = rcpf(vmat);
    const cf32 mm = {sr * vi, si * vi};
    if (!(vbc > 0.f)) return mm;
    // Maxwell-Garnett: e = em (ei + 2em + 2f(ei-em)) / (ei + 2em - f(ei-em))
    auto mul = [](cf32 p, cf32 q) { return cf32{p.re * q.re - p.im * q.im, p.re * q.im + p.im * q.re}; };
    const float f = vbc * rcpf(vwet);
    const cf32 em = mul(mm, mm), ei = mul(mbc, mbc);
    const cf32 de = {ei.re - em.re, ei.im - em.im};
    const cf32 base = {ei.re + 2.f * em.re, ei.im + 2.f * em.im};
    const cf32 nume = {base.re + 2.f * f * de.re, base.im + 2.f * f * de.im};
    const cf32 deno = {base.re - f * de.re, base.im - f * de.im};
    const float rd = rcpf(deno.re * deno.re + deno.im * deno.im);
    const cf32 quo = mul(nume, cf32{deno.re * rd, -deno.im * rd});
    const cf32 e = mul(em, quo);
    const float mod = sqrtf(e.re * e.re + e.im * e.im);
    const float re = sqrtf(0.5f * (mod + e.re)), im = sqrtf(fmaxf(0.5f * (mod - e.re), 0.f));
    return {re, e.im < 0.f ? -im : im};
}

// ---- mixed path, shortwave: two wavelengths of a particle as a packed fp32 pair, series up to nstop = 256 ----
// 256 terms of D_n and A_n for two evaluations do not fit registers or LDS at 12 waves per CU, so only every 32nd
// term is kept in LDS (OPT_NCK32 slots, as for the longwave pair) and each block of 8 terms is rebuilt from the
// checkpoint above it: the chain is walked 2.5 times on average -- ~12 packed instructions per step against ~70 per
// term of the series.  Asymmetry and backscatter sums as in mie_term<true>.
constexpr int OPT_SB = 32;     // terms between checkpoints
struct MieAccF32 { v2f p[2], c[2], qs, qe, gs; cplx2 an[2], bn[2], xb; float sgn; };   // slots by index parity
__device__ __forceinline__ void mie_term32_full(MieAccF32 &A, int n, cplx2 D, v2f An, const MieSetup32 &s)
{
    const float en = (float)n;
    const v2f nx = en * s.xinv, up = (2.f * en - 1.f) * s.xinv;
    const int par = n & 1;   // a constant after unrolling
    const v2f p1 = A.p[par ^ 1], c1 = A.c[par ^ 1];
    const v2f psi = sel2(en > s.x.x, en > s.x.y, p1 * rcp2(nx + An), up * p1 - A.p[par]);
    const v2f chi = up * c1 - A.c[par];
    A.p[par] = psi; A.c[par] = chi;
    const cplx2 xi = {psi, -chi}, xi1 = {p1, -c1};
    cplx2 t = cmul2(D, s.minv);
    t.re += nx;
    const cplx2 na = {t.re * psi - p1, t.im * psi}, da = csub2(cmul2(t, xi), xi1);
    t = cmul2(s.m, D);
    t.re += nx;
    const cplx2 nb = {t.re * psi - p1, t.im * psi}, db = csub2(cmul2(t, xi), xi1);
    const v2f ra = rcp2(da.re * da.re + da.im * da.im), rb = rcp2(db.re * db.re + db.im * db.im);
    const cplx2 an = {(na.re * da.re + na.im * da.im) * ra, (na.im * da.re - na.re * da.im) * ra};
    const cplx2 bn = {(nb.re * db.re + nb.im * db.im) * rb, (nb.im * db.re - nb.re * db.im) * rb};
    const float w2 = 2.f * en + 1.f;
    const v2f zero = v2f{0.f, 0.f};
    const bool on1 = n <= s.nstop1, on2 = n <= s.nstop2;   // an evaluation whose series has ended adds nothing
    A.qs += sel2(on1, on2, w2 * (an.re * an.re + an.im * an.im + bn.re * bn.re + bn.im * bn.im), zero);
    A.qe += sel2(on1, on2, w2 * (an.re + bn.re), zero);
    v2f dg = (float)c_gw[n].x * (an.re * bn.re + an.im * bn.im);
    if (n > 1) {
        const cplx2 an1 = A.an[par ^ 1], bn1 = A.bn[par ^ 1];
        dg += (float)c_gw[n].y * (an1.re * an.re + an1.im * an.im + bn1.re * bn.re + bn1.im * bn.im);
    }
    A.gs += sel2(on1, on2, dg, zero);
    A.xb.re += sel2(on1, on2, (A.sgn * w2) * (an.re - bn.re), zero);
    A.xb.im += sel2(on1, on2, (A.sgn * w2) * (an.im - bn.im), zero);
    A.sgn = -A.sgn;
    A.an[par] = an; A.bn[par] = bn;
}
__device__ __forceinline__ void mie_pair32_full(const MieSetup32 &s, double2 *ck, v2f &qext, v2f &qsca, v2f &gasy, v2f &qback)
{
    const int nstopm = s.nstop1 > s.nstop2 ? s.nstop1 : s.nstop2;
    const int nck = (nstopm + OPT_SB - 1) / OPT_SB, ntop = nck * OPT_SB;
    auto step = [&](cplx2 &d, v2f &a, int n) {   // D_{n-1}, A_{n-1} from D_n, A_n
        const float en = (float)n;
        const cplx2 ny = {en * s.yinv.re, en * s.yinv.im};
        d = csub2(ny, crecip2(cadd2(d, ny)));
        const v2f nx = en * s.xinv;
        a = nx - rcp2(a + nx);
    };
    {
        cplx2 d = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
        v2f a = v2f{0.f, 0.f};
        // the start must lie above the highest checkpoint (a higher start is only more accurate)
        const int nstart = s.nmx > ntop + 1 ? s.nmx : ntop + 1;
        for (int n = nstart; n >= 2; --n) {
            step(d, a, n);
            const int n1 = n - 1;
            if (n1 <= ntop && (n1 & (OPT_SB - 1)) == 0) {
                const int j = n1 / OPT_SB - 1;
                *reinterpret_cast<float4 *>(&ck[(size_t)j * OPT_THREADS]) = make_float4(d.re.x, d.re.y, d.im.x, d.im.y);
                reinterpret_cast<float2 *>(&ck[(size_t)(OPT_NCK2 + (j >> 1)) * OPT_THREADS])[j & 1] = make_float2(a.x, a.y);
            }
            if (n1 <= OPT_SB) break;
        }
    }
    MieAccF32 A;
    float sn1, cs1, sn2, cs2;
    sincosf(s.x.x, &sn1, &cs1);
    sincosf(s.x.y, &sn2, &cs2);
    const v2f zero = v2f{0.f, 0.f};
    A.p[0] = v2f{sn1, sn2}; A.p[1] = v2f{cs1, cs2}; A.c[0] = A.p[1]; A.c[1] = -A.p[0];
    A.qs = zero; A.qe = zero; A.gs = zero; A.sgn = -1.f;
    A.an[0] = {zero, zero}; A.an[1] = {zero, zero}; A.bn[0] = {zero, zero}; A.bn[1] = {zero, zero}; A.xb = {zero, zero};
    for (int j = 0; j < nck; ++j) {
        const float4 cd = *reinterpret_cast<const float4 *>(&ck[(size_t)j * OPT_THREADS]);
        const float2 ca = reinterpret_cast<const float2 *>(&ck[(size_t)(OPT_NCK2 + (j >> 1)) * OPT_THREADS])[j & 1];
        for (int sub = 0; sub < OPT_SB / OPT_MB2; ++sub) {
            const int nlo = j * OPT_SB + sub * OPT_MB2 + 1;      // first term of this block of OPT_MB2
            if (nlo > nstopm) break;
            cplx2 d = {v2f{cd.x, cd.y}, v2f{cd.z, cd.w}};        // D_{SB (j+1)}
            v2f a = v2f{ca.x, ca.y};
            for (int n = (j + 1) * OPT_SB; n > nlo + OPT_MB2 - 1; --n) step(d, a, n);   // walk down to the block's top
            cplx2 dd[OPT_MB2];
            v2f aa[OPT_MB2];
            dd[OPT_MB2 - 1] = d; aa[OPT_MB2 - 1] = a;
#pragma unroll
            for (int i = OPT_MB2 - 1; i >= 1; --i) {
                dd[i - 1] = dd[i]; aa[i - 1] = aa[i];
                step(dd[i - 1], aa[i - 1], nlo + i);
            }
#pragma unroll
            for (int i = 0; i < OPT_MB2; ++i)
                if (nlo + i <= nstopm) mie_term32_full(A, nlo + i, dd[i], aa[i], s);
        }
    }
    const v2f x2inv = s.xinv * s.xinv;
    qsca = 2.f * x2inv * A.qs;
    qext = 2.f * x2inv * A.qe;
    gasy = v2f{A.qs.x > 0.f ? 2.f * A.gs.x * rcpf(A.qs.x) : 0.f, A.qs.y > 0.f ? 2.f * A.gs.y * rcpf(A.qs.y) : 0.f};
    qback = x2inv * (A.xb.re * A.xb.re + A.xb.im * A.xb.im);
}

// particle geometry of one section in fp32; false: the section is empty
__device__ __forceinline__ bool bin_geometry32(const OpticsKernelArgs &a, int ib, size_t q, size_t sspec, int r4,
                                               float *vol, float &vwet, float &dwet, float &num)
{
    const float pi = 3.14159265358979323846f;
    float vdry = 0.f;
#pragma unroll
    for (int sp = 0; sp < OPT_NSPEC; ++sp) {
        const int id = a.idx[ib][sp];
        const float mass = id >= 0 ? fmaxf((float)ldr(a.chem, (size_t)id * sspec + q, r4), 0.f) : 0.f;
        vol[sp] = mass * rcpf((float)c_opt.dens[sp]);
        if (sp != OS_WATER) vdry += vol[sp];
    }
    const int idn = a.idx[ib][OS_NUM];
    num = idn >= 0 ? fmaxf((float)ldr(a.chem, (size_t)idn * sspec + q, r4), 0.f) : 0.f;
    if (!(vdry > 1.e-25f)) return false;
    vwet = vdry + vol[OS_WATER];
    const float vpart = vdry * 1.e-6f * rcpf(fmaxf(num, 1.e-30f));   // cm3 per particle
    float ddry = 1.e4f * cbrtf(vpart * (6.f / pi));
    const float dlo = (float)c_opt.dlo_um[ib], dhi = (float)c_opt.dhi_um[ib];
    if (ddry < dlo || ddry > dhi) {
        ddry = ddry < dlo ? dlo : dhi;
        const float dcm = ddry * 1.e-4f;
        num = vdry * 1.e-6f * rcpf((pi / 6.f) * dcm * dcm * dcm);
    }
    dwet = ddry * cbrtf(vwet * rcpf(vdry));
    return true;
}

__device__ __forceinline__ void optics_body_sw32(const OpticsKernelArgs &a, double2 *s_ck)
{
    constexpr int NW = OPT_NSW / 2;
    const int b = blockIdx.x;
    const int xcd = b & 7, s = b >> 3;
    // pair-major, as the fp64 kernel is wavelength-major: workgroups of equal duration next to each other
    const int ng8 = (int)(gridDim.x >> 3) / NW;
    const int slot = s / ng8, gl = s - slot * ng8;
    const int w = 2 * slot, w2 = w + 1;
    double2 *ck = s_ck + threadIdx.x;
    const long cell = ((long)gl * 8 + xcd) * OPT_THREADS + threadIdx.x;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (cell >= ncell) return;
    if (!(a.tauaer[w] || a.extaer[w] || a.bscoef[w])) return;   // the groups are complete or absent
    const int it = (int)(cell % a.ni_t);
    const long r = cell / a.ni_t;
    const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
    const size_t q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const float alt = (float)ldr(a.alt, q, r4), dz = (float)ldr(a.dz8w, q, r4);
    const float pi = 3.14159265358979323846f;
    const v2f rlam = v2f{rcpf((float)c_opt.wavelength_um[w]), rcpf((float)c_opt.wavelength_um[w2])};
    const float ralt = rcpf(alt);
    const v2f zero = v2f{0.f, 0.f};
    v2f ext = zero, sca = zero, gsca = zero, bsc = zero;   // 1/cm (bsc: 1/(cm sr))
    for (int ib = 0; ib < a.nbin; ++ib) {
        float vol[OPT_NSPEC], vwet, dwet, num;
        if (!bin_geometry32(a, ib, q, sspec, r4, vol, vwet, dwet, num)) continue;
        const cf32 m1 = mix_index32(vol, vwet, a.mix, w), m2 = mix_index32(vol, vwet, a.mix, w2);
        const float x1 = fminf(pi * dwet * rlam.x, (float)OPT_XMAX), x2 = fminf(pi * dwet * rlam.y, (float)OPT_XMAX);
        const MieSetup32 t12 = mie_setup32(x1, m1.re, m1.im, x2, m2.re, m2.im);
        v2f qe, qs, g, qb;
        mie_pair32_full(t12, ck, qe, qs, g, qb);
        const float dcm = dwet * 1.e-4f;
        const float cross = (num * ralt * 1.e-6f) * (0.25f * pi * dcm * dcm);   // #/cm3 * cm2
        ext += cross * qe;
        sca += cross * qs;
        gsca += cross * qs * g;
        bsc += (cross * (0.25f / pi)) * qb;
    }
    const float e2[2] = {ext.x, ext.y}, s2[2] = {sca.x, sca.y}, g2[2] = {gsca.x, gsca.y}, b2[2] = {bsc.x, bsc.y};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ww = w + u;
        if (a.tauaer[ww]) {
            str(a.tauaer[ww], q, r4, (double)(e2[u] * 100.f * dz));
            str(a.waer[ww], q, r4, (double)(e2[u] > 0.f ? s2[u] * rcpf(e2[u]) : 0.f));
            str(a.gaer[ww], q, r4, (double)(s2[u] > 0.f ? g2[u] * rcpf(s2[u]) : 0.f));
        }
        if (a.extaer[ww]) str(a.extaer[ww], q, r4, (double)(e2[u] * 1.e5f));
        if (a.bscoef[ww]) str(a.bscoef[ww], q, r4, (double)(b2[u] * 1.e5f));
    }
}

// one thread = one cell, the eight band pairs in a loop, the cell's 32 running sums (fp32) in the thread's LDS column
__device__ __forceinline__ void optics_body_lw32(const OpticsKernelArgs &a, double2 *s_ck, float *s_acc)
{
    const long cell = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (cell >= ncell) return;
    unsigned want = 0;
#pragma unroll
    for (int p = 0; p < OPT_NLW / 2; ++p)
        if (a.tauaerlw[2 * p] || a.extaerlw[2 * p] || a.tauaerlw[2 * p + 1] || a.extaerlw[2 * p + 1]) want |= 1u << p;
    if (!want) return;
    double2 *ck = s_ck + threadIdx.x;
    float *acc = s_acc + threadIdx.x;   // [band][ext, sca][thread]
#pragma unroll
    for (int i = 0; i < 2 * OPT_NLW; ++i) acc[i * OPT_THREADS] = 0.f;
    const int it = (int)(cell % a.ni_t);
    const long r = cell / a.ni_t;
    const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
    const size_t q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const float alt = (float)ldr(a.alt, q, r4), dz = (float)ldr(a.dz8w, q, r4);
    const float pi = 3.14159265358979323846f;
    const float ralt = rcpf(alt);
    for (int ib = 0; ib < a.nbin; ++ib) {
        float vol[OPT_NSPEC];   // 1e-6 cm3 per kg of dry air
        float vdry = 0.f;
#pragma unroll
        for (int sp = 0; sp < OPT_NSPEC; ++sp) {
            const int id = a.idx[ib][sp];
            const float mass = id >= 0 ? fmaxf((float)ldr(a.chem, (size_t)id * sspec + q, r4), 0.f) : 0.f;
            vol[sp] = mass * rcpf((float)c_opt.dens[sp]);
            if (sp != OS_WATER) vdry += vol[sp];
        }
        const int idn = a.idx[ib][OS_NUM];
        float num = idn >= 0 ? fmaxf((float)ldr(a.chem, (size_t)idn * sspec + q, r4), 0.f) : 0.f;
        if (!(vdry > 1.e-25f)) continue;
        const float vwet = vdry + vol[OS_WATER];
        const float vpart = vdry * 1.e-6f * rcpf(fmaxf(num, 1.e-30f));   // cm3 per particle
        float ddry = 1.e4f * cbrtf(vpart * (6.f / pi));
        const float dlo = (float)c_opt.dlo_um[ib], dhi = (float)c_opt.dhi_um[ib];
        if (ddry < dlo || ddry > dhi) {
            ddry = ddry < dlo ? dlo : dhi;
            const float dcm = ddry * 1.e-4f;
            num = vdry * 1.e-6f * rcpf((pi / 6.f) * dcm * dcm * dcm);
        }
        const float dwet = ddry * cbrtf(vwet * rcpf(vdry));
        const float dcm = dwet * 1.e-4f;
        const float cross = (num * ralt * 1.e-6f) * (0.25f * pi * dcm * dcm);   // #/cm3 * cm2
#pragma unroll 1
        for (int p = 0; p < OPT_NLW / 2; ++p) {
            if (!((want >> p) & 1)) continue;
            const int w = OPT_NSW + 2 * p, w2 = w + 1;
            const float rlam = rcpf((float)c_opt.wavelength_um[w]), rlam2 = rcpf((float)c_opt.wavelength_um[w2]);
            const cf32 m1 = mix_index32(vol, vwet, a.mix, w), m2 = mix_index32(vol, vwet, a.mix, w2);
            // size parameters capped so that the series fits the register blocks (nstop <= 64); no aerosol reaches
            // x = 47 at wavelengths >= 3.3 um (diameter 50 um)
            const float x1 = fminf(pi * dwet * rlam, (float)OPT_XMAX32), x2 = fminf(pi * dwet * rlam2, (float)OPT_XMAX32);
            const MieSetup32 t12 = mie_setup32(x1, m1.re, m1.im, x2, m2.re, m2.im);
            float qe, qs, qe2, qs2;
            mie_pair32(t12, ck, qe, qs, qe2, qs2);
            float *s = acc + (size_t)(4 * p) * OPT_THREADS;
            s[0] += cross * qe;
            s[OPT_THREADS] += cross * qs;
            s[2 * OPT_THREADS] += cross * qe2;
            s[3 * OPT_THREADS] += cross * qs2;
        }
    }
#pragma unroll 1
    for (int ib = 0; ib < OPT_NLW; ++ib) {
        const float ab = fmaxf(acc[(size_t)(2 * ib) * OPT_THREADS] - acc[(size_t)(2 * ib + 1) * OPT_THREADS], 0.f);
        if (a.tauaerlw[ib]) str(a.tauaerlw[ib], q, r4, (double)(ab * 100.f * dz));
        if (a.extaerlw[ib]) str(a.extaerlw[ib], q, r4, (double)(ab * 1.e5f));
    }
}

__global__ void __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) optics_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK_SWC * OPT_THREADS];
    __shared__ double s_acc[4 * OPT_NSW * OPT_THREADS];
    optics_sw_cell<false>(a, s_ck, s_acc);
}
// the same with the backscatter sums of the series (a bscoef output is requested)
__global__ void __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) optics_bsc_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK_SWC * OPT_THREADS];
    __shared__ double s_acc[4 * OPT_NSW * OPT_THREADS];
    optics_sw_cell<true>(a, s_ck, s_acc);
}

__global__ void __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
optics_lw_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK_LW * OPT_THREADS];
    __shared__ double s_acc[2 * OPT_NLW * OPT_THREADS];
    optics_lw_cell(a, s_ck, s_acc);
}

// the mixed-precision path: shortwave wavelength pairs and longwave band pairs in fp32
__global__ void __launch_bounds__(OPT_THREADS) optics_sw32_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK32 * OPT_THREADS];
    optics_body_sw32(a, s_ck);
}

__global__ void __launch_bounds__(OPT_THREADS) optics_lw32_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK32 * OPT_THREADS];
    __shared__ float s_acc[2 * OPT_NLW * OPT_THREADS];
    optics_body_lw32(a, s_ck, s_acc);
}

}  // namespace

hipError_t optics_upload_tables(const OpticsTables &t)
{
    static double2 gw[OPT_NMIE + 10];
    for (int n = 1; n < OPT_NMIE + 10; ++n) {
        const double en = (double)n;
        gw[n].x = (2. * en + 1.) / (en * (en + 1.));
        gw[n].y = (en - 1.) * (en + 1.) / en;
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_gw), gw, sizeof gw);
    if (e != hipSuccess) return e;
    static MieN mn[OPT_NMIE + 10];
    for (int n = 0; n < OPT_NMIE + 10; ++n) {
        const double en = (double)n;
        mn[n] = MieN{en, 2. * en - 1., 2. * en + 1., gw[n].x, gw[n].y, {0., 0., 0.}};
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_mn), mn, sizeof mn);
    if (e != hipSuccess) return e;
    double pil[2][OPT_NWAVE];
    for (int w = 0; w < OPT_NWAVE; ++w) {
        pil[0][w] = 3.14159265358979323846 / t.wavelength_um[w];
        pil[1][w] = std::cbrt(pil[0][w]);
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_pil), pil, sizeof pil);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_opt), &t, sizeof(OpticsTables));
}

hipError_t optics_launch(const OpticsKernelArgs &a, hipStream_t stream, int precision)
{
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (ncell <= 0) return hipSuccess;
    const long groups = (ncell + OPT_THREADS - 1) / OPT_THREADS;
    const long groups8 = (groups + 7) / 8;
    const long blocks = groups8 * 8 * OPT_NSW;   // the fp32 shortwave kernel: one workgroup per (cell group, wavelength pair)
    if (blocks > 0x7fffffffL) return hipErrorInvalidValue;
    // skip a kernel none of whose outputs were asked for
    bool any_sw = false, any_lw = false;
    for (int w = 0; w < OPT_NSW; ++w) any_sw = any_sw || a.tauaer[w] || a.extaer[w] || a.bscoef[w];
    for (int b = 0; b < OPT_NLW; ++b) any_lw = any_lw || a.tauaerlw[b] || a.extaerlw[b];
    if (any_sw && precision == 1)
        hipLaunchKernelGGL(optics_sw32_kernel, dim3((unsigned)(blocks / 2)), dim3(OPT_THREADS), 0, stream, a);
    else if (any_sw) {   // one thread per cell; the backscatter sums only when a bscoef output is there
        bool bsc = false;
        for (int w = 0; w < OPT_NSW; ++w) bsc = bsc || a.bscoef[w];
        if (bsc)
            hipLaunchKernelGGL(optics_bsc_kernel, dim3((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), dim3(OPT_THREADS),
                               0, stream, a);
        else
            hipLaunchKernelGGL(optics_kernel, dim3((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), dim3(OPT_THREADS), 0,
                               stream, a);
    }
    if (any_lw && precision == 1)
        hipLaunchKernelGGL(optics_lw32_kernel, dim3((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), dim3(OPT_THREADS), 0,
                           stream, a);
    else if (any_lw)   // one thread per cell
        hipLaunchKernelGGL(optics_lw_kernel, dim3((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), dim3(OPT_THREADS), 0,
                           stream, a);
    return hipGetLastError();
}

}  // namespace rrtmg
