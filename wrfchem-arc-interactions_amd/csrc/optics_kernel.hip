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
#include <type_traits>

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
struct MieN { double en, tnm1, w2, gx, gy; float fen, ftn, fw2, fgx, fgy, pad_; };   // n, 2n - 1, 2n + 1, the two asymmetry weights; the same in fp32 (mixed path)
__constant__ MieN c_mn[OPT_NMIE + 10];
__constant__ double c_rn[OPT_NMIE + 10];   // 1 / n
// the same for the term pair (2q + 1, 2q + 2): the fp32 values as the two halves of packed operands
struct MieN2 { double en0, en1, tn0, tn1; float fen[2], ftn[2], fw2[2], fgx[2], fgy[2], pad_[2]; };
constexpr int OPT_NPAIR = (OPT_NMIE + 10) / 2;
__constant__ MieN2 c_mp[OPT_NPAIR];
// Gauss-Legendre rules of N = 8, 16, .. 264 nodes (the Legendre moments l2aer..l7aer): rule j = N / 8 starts at entry
// 4 j (j - 1); nodes in g_gl[0][..], weights in g_gl[1][..]
constexpr int OPT_NGL = 33, OPT_GL_ENTRIES = 4 * OPT_NGL * (OPT_NGL + 1);
__device__ double g_gl[2][OPT_GL_ENTRIES];
__device__ __forceinline__ MieN mie_n(int n, double2 gw)   // the same from n (folds where n is a compile-time constant)
{
    const double en = (double)n;
    return MieN{en, 2. * en - 1., 2. * en + 1., gw.x, gw.y, (float)en, (float)(2. * en - 1.), (float)(2. * en + 1.), (float)gw.x, (float)gw.y, 0.f};
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
constexpr int OPT_MB2 = 8;   // terms of the register-only ladder

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

// ---- packed fp32 helpers of the mixed optics path (precision 1) ----
// The first mixed kernels (rounds 1-2) ran the whole series in fp32: psi_n by its recurrence only while n <= x and as
// psi_{n-1} / (n/x + A_n(x)) beyond (the plain recurrence cancels catastrophically: psi_1 = sin x / x - cos x is off by
// 5 % at x = 0.002 in fp32), D_n and A_n downward through checkpoints and rebuilds.  Once the fp64 kernels carried D_n
// upward they had caught up with that; the mixed path below keeps the two recurrences in fp64 instead.  a_n and b_n
// take one reciprocal each (the product of their denominators overflows fp32); reciprocals are v_rcp_f32 (1 ulp).
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
// ---- mixed path, round 3: psi_n -- and D_n where it is carried upward -- in fp64; chi_n, a_n, b_n and the sums as
// packed fp32 pairs ----
// The packed-fp32 series above pays for what fp32 cannot do: psi_n needs a second downward recurrence (A_n) beyond
// n = x, and D_n goes downward from far above the series through checkpoints and rebuilds (the upward recurrence in
// fp32 is off by up to 1e-3 for weakly absorbing particles, whose cot y runs close to its poles:
// tools/probes/mie_mixed32.py).  Both recurrences are cheap in fp64 -- psi_n two operations per term, D_n upward
// eleven -- while the forty-odd operations of a_n, b_n and the sums are where the packed arithmetic pays.  So the
// recurrences that need the precision keep it and hand their values over by conversion; the small-argument ladder
// D_1..D_8 (contracting downward recurrence) runs in packed fp32 from a start order that leaves 1e-8 of the start error.
struct MiePsi2 { double a[2], b[2]; };   // psi_k of evaluation 1 / 2 in slot k & 1
struct MieAccM { v2f pf[2], c[2], qs, qe; };   // pf: psi_k rounded, c: chi_k (slot k & 1); extinction and scattering sums
struct MiePairC { cplx2 m, minv; v2f xinv; double xinv1, xinv2; int nstop1, nstop2; };
__device__ __forceinline__ cplx2 pack2(cplx a, cplx b) { return {v2f{(float)a.re, (float)b.re}, v2f{(float)a.im, (float)b.im}}; }
__device__ __forceinline__ void mie_accM_start(MieAccM &A, MiePsi2 &P, double sn1, double cs1, double sn2, double cs2)
{
    P.a[0] = sn1; P.a[1] = cs1; P.b[0] = sn2; P.b[1] = cs2;
    const v2f zero = v2f{0.f, 0.f};
    A.pf[0] = v2f{(float)sn1, (float)sn2}; A.pf[1] = v2f{(float)cs1, (float)cs2};
    A.c[0] = A.pf[1]; A.c[1] = -A.pf[0];
    A.qs = zero; A.qe = zero;
}
// extinction and scattering sums only (the longwave bands use absorption alone)
__device__ __forceinline__ void mie_termM(MieAccM &A, MiePsi2 &P, const int n, const cplx2 D, const MiePairC &s, const MieN &K)
{
    const int par = n & 1;   // a constant after unrolling
    const float en = K.fen;
    const double ps1 = __builtin_fma(K.tnm1 * s.xinv1, P.a[par ^ 1], -P.a[par]);
    const double ps2 = __builtin_fma(K.tnm1 * s.xinv2, P.b[par ^ 1], -P.b[par]);
    P.a[par] = ps1; P.b[par] = ps2;
    const v2f psi = v2f{(float)ps1, (float)ps2};
    const v2f p1 = A.pf[par ^ 1], c1 = A.c[par ^ 1];
    const v2f nx = en * s.xinv, up = K.ftn * s.xinv;
    const v2f chi = up * c1 - A.c[par];
    A.pf[par] = psi; A.c[par] = chi;
    const cplx2 xi = {psi, -chi}, xi1 = {p1, -c1};
    cplx2 t = cmul2(D, s.minv);
    t.re += nx;
    const cplx2 na = {t.re * psi - p1, t.im * psi}, da = csub2(cmul2(t, xi), xi1);
    t = cmul2(s.m, D);
    t.re += nx;
    const cplx2 nb = {t.re * psi - p1, t.im * psi}, db = csub2(cmul2(t, xi), xi1);
    const v2f ra = rcp2(da.re * da.re + da.im * da.im), rb = rcp2(db.re * db.re + db.im * db.im);
    const float w2 = K.fw2;
    const v2f zero = v2f{0.f, 0.f};
    const bool on1 = n <= s.nstop1, on2 = n <= s.nstop2;   // an evaluation whose series has ended adds nothing
    // Re a_n and |a_n|^2 = |na|^2 / |da|^2 without forming a_n
    const v2f are = (na.re * da.re + na.im * da.im) * ra, bre = (nb.re * db.re + nb.im * db.im) * rb;
    const v2f a2 = (na.re * na.re + na.im * na.im) * ra, b2 = (nb.re * nb.re + nb.im * nb.im) * rb;
    A.qs += sel2(on1, on2, w2 * (a2 + b2), zero);
    A.qe += sel2(on1, on2, w2 * (are + bre), zero);
}

// One evaluation with its terms taken TWO AT A TIME: terms n and n + 1 (n odd) are the two halves of the packed values.
// Two wavelengths of a particle as the halves (built first: 13.1 ms against this function's 12.2 at one rank's share of
// configs[4]) run to the longer of two series whose lengths differ by a third to two thirds -- a sixth of the packed
// work is spent on an evaluation that has ended; two consecutive terms of ONE series waste at most half a step, and a
// single evaluation's state needs fewer registers.  The recurrences (psi_n
// and D_n in fp64, chi_n in fp32) advance two steps per packed step; the sums of the two halves are added at the end.
struct MieAccT {
    double p0, p1;      // psi_(n-2), psi_(n-1) on entry of a step
    float c0, c1, pf;   // chi_(n-2), chi_(n-1); psi_(n-1) rounded
    v2f qs, qe, gs;
    cplx2 xb;
    float anr, ani, bnr, bni;   // a_(n-1), b_(n-1)
};
template <bool BSC>
__device__ __forceinline__ void mie_term2(MieAccT &A, cplx &d, const cplx yinv, const double xinv, const float xinvf,
                                          const cplx2 m, const cplx2 minv, const int n, const int nstop, const MieN2 &K)
{
    // two steps of each recurrence
    cplx ny = {K.en0 * yinv.re, K.en0 * yinv.im};
    d = csub(crecip(csub(ny, d)), ny);
    const cplx D0 = d;
    ny = {K.en1 * yinv.re, K.en1 * yinv.im};
    d = csub(crecip(csub(ny, d)), ny);
    const double ps0 = __builtin_fma(K.tn0 * xinv, A.p1, -A.p0);
    const double ps1 = __builtin_fma(K.tn1 * xinv, ps0, -A.p1);
    A.p0 = ps0; A.p1 = ps1;
    const v2f up = v2f{K.ftn[0], K.ftn[1]} * xinvf, nx = v2f{K.fen[0], K.fen[1]} * xinvf;
    const float ch0 = up.x * A.c1 - A.c0;
    const float ch1 = up.y * ch0 - A.c1;
    const v2f psi = v2f{(float)ps0, (float)ps1}, p1 = v2f{A.pf, psi.x};
    const v2f chi = v2f{ch0, ch1}, c1 = v2f{A.c1, ch0};
    A.c0 = ch0; A.c1 = ch1; A.pf = psi.y;
    const cplx2 D = pack2(D0, d);
    const cplx2 xi = {psi, -chi}, xi1 = {p1, -c1};
    cplx2 t = cmul2(D, minv);
    t.re += nx;
    const cplx2 na = {t.re * psi - p1, t.im * psi}, da = csub2(cmul2(t, xi), xi1);
    t = cmul2(m, D);
    t.re += nx;
    const cplx2 nb = {t.re * psi - p1, t.im * psi}, db = csub2(cmul2(t, xi), xi1);
    const v2f ra = rcp2(da.re * da.re + da.im * da.im), rb = rcp2(db.re * db.re + db.im * db.im);
    const cplx2 an = {(na.re * da.re + na.im * da.im) * ra, (na.im * da.re - na.re * da.im) * ra};
    const cplx2 bn = {(nb.re * db.re + nb.im * db.im) * rb, (nb.im * db.re - nb.re * db.im) * rb};
    const v2f w2 = v2f{K.fw2[0], K.fw2[1]}, zero = v2f{0.f, 0.f};
    const bool on1 = n + 1 <= nstop;   // (the caller has n <= nstop)
    A.qs += sel2(true, on1, w2 * (an.re * an.re + an.im * an.im + bn.re * bn.re + bn.im * bn.im), zero);
    A.qe += sel2(true, on1, w2 * (an.re + bn.re), zero);
    // a_(n-1), b_(n-1) of the two halves: the previous step's upper half and this step's lower half
    const cplx2 anp = {v2f{A.anr, an.re.x}, v2f{A.ani, an.im.x}}, bnp = {v2f{A.bnr, bn.re.x}, v2f{A.bni, bn.im.x}};
    const v2f dg = v2f{K.fgx[0], K.fgx[1]} * (an.re * bn.re + an.im * bn.im) +
                   v2f{K.fgy[0], K.fgy[1]} * (anp.re * an.re + anp.im * an.im + bnp.re * bn.re + bnp.im * bn.im);
    A.gs += sel2(true, on1, dg, zero);
    if (BSC) {   // (-1)^n (2n + 1) (a_n - b_n): n is odd
        const v2f sw = v2f{-K.fw2[0], K.fw2[1]};
        A.xb.re += sel2(true, on1, sw * (an.re - bn.re), zero);
        A.xb.im += sel2(true, on1, sw * (an.im - bn.im), zero);
    }
    A.anr = an.re.y; A.ani = an.im.y; A.bnr = bn.re.y; A.bni = bn.im.y;
}
template <bool BSC>
__device__ __forceinline__ void mie_sphereM_up(double x, cplx m, double cbx, double &qext, double &qsca, double &gasy,
                                               double &qback)
{
    const cplx y = {x * m.re, x * m.im};
    const int nstop = (int)(x + 4. * cbx + 2.);
    const cplx yinv = crecip(y);
    const double xinv = frcp(x);
    const float xinvf = (float)xinv;
    const cplx mi = crecip(m);
    const cplx2 m2 = pack2(m, m), minv2 = pack2(mi, mi);
    MieAccT A;
    {
        double sn, cs;
        sincos_mid(x, sn, cs);
        A.p0 = cs; A.p1 = sn;                        // psi_(-1), psi_0
        A.c0 = -(float)sn; A.c1 = (float)cs;         // chi_(-1), chi_0
        A.pf = (float)sn;
    }
    const v2f zero = v2f{0.f, 0.f};
    A.qs = zero; A.qe = zero; A.gs = zero; A.xb = {zero, zero};
    A.anr = 0.f; A.ani = 0.f; A.bnr = 0.f; A.bni = 0.f;
    cplx d = ccot(y);
    // the records of the NEXT two steps are fetched while these two are evaluated
    MieN2 kn[2], knn[2];
    kn[0] = c_mp[0]; kn[1] = c_mp[1];
    for (int n0 = 0; n0 < nstop; n0 += 4) {
        knn[0] = c_mp[(n0 >> 1) + 2]; knn[1] = c_mp[(n0 >> 1) + 3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + 2 * i + 1;
            if (n <= nstop) mie_term2<BSC>(A, d, yinv, xinv, xinvf, m2, minv2, n, nstop, kn[i]);
        }
        kn[0] = knn[0]; kn[1] = knn[1];
    }
    const double x2inv = xinv * xinv;
    const float qs = A.qs.x + A.qs.y, qe = A.qe.x + A.qe.y, gs = A.gs.x + A.gs.y;
    qsca = 2. * x2inv * (double)qs;
    qext = 2. * x2inv * (double)qe;
    gasy = qs > 0.f ? (double)(2.f * gs * rcpf(qs)) : 0.;
    const float xr = A.xb.re.x + A.xb.re.y, xi = A.xb.im.x + A.xb.im.y;
    qback = BSC ? x2inv * (double)(xr * xr + xi * xi) : 0.;
}

// The longwave pair of mie_pair_su in this arithmetic: every lane prepares its own source of D_n -- the packed-fp32
// ladder (small) or D_0 = cot y carried upward in fp64 -- and the terms are taken in one pass.
__device__ __forceinline__ void mie_pairM_su(double x1, cplx m1, double x2, cplx m2, const MieSetupS &s1,
                                             const MieSetupS &s2, const bool small, double &qext1, double &qsca1,
                                             double &qext2, double &qsca2)
{
    MiePairC c;
    c.m = pack2(m1, m2); c.minv = pack2(s1.minv, s2.minv);
    c.xinv = v2f{(float)s1.xinv, (float)s2.xinv}; c.xinv1 = s1.xinv; c.xinv2 = s2.xinv;
    c.nstop1 = s1.nstop; c.nstop2 = s2.nstop;
    const v2f zero = v2f{0.f, 0.f};
    cplx2 dd[OPT_MB2];
    cplx d1 = {0., 0.}, d2 = {0., 0.};
    double sn1, cs1, sn2, cs2;
    if (small) {
        const cplx2 yinv = pack2(s1.yinv, s2.yinv);
        // steps until the start error has contracted by 1e-8 (fp32): each contributes at least 2 ln((2 nstop + 3) / |y|)
        const v2f ay = v2f{(float)(2 * s1.nstop + 3), (float)(2 * s2.nstop + 3)} *
                       v2f{__builtin_amdgcn_sqrtf(yinv.re.x * yinv.re.x + yinv.im.x * yinv.im.x),
                           __builtin_amdgcn_sqrtf(yinv.re.y * yinv.re.y + yinv.im.y * yinv.im.y)};
        const int na = s1.nstop + (int)(9.25f * __builtin_amdgcn_rcpf(__logf(ay.x))) + 2;
        const int nb = s2.nstop + (int)(9.25f * __builtin_amdgcn_rcpf(__logf(ay.y))) + 2;
        const int nab = na > nb ? na : nb;
        const int nstart = nab < OPT_MB2 + 1 ? OPT_MB2 + 1 : nab;
        cplx2 d = {zero, zero};
        for (int n = nstart; n > OPT_MB2 + 1; --n) {   // D_{n-1} = n/y - 1/(D_n + n/y), down to D_9
            const float en = (float)n;
            const cplx2 ny = {en * yinv.re, en * yinv.im};
            d = csub2(ny, crecip2(cadd2(d, ny)));
        }
#pragma unroll
        for (int i = OPT_MB2 - 1; i >= 0; --i) {       // dd[i] = D_{i+1}
            const float en = (float)(i + 2);
            const cplx2 ny = {en * yinv.re, en * yinv.im};
            d = csub2(ny, crecip2(cadd2(d, ny)));
            dd[i] = d;
        }
        sincos_small(x1, sn1, cs1);
        sincos_small(x2, sn2, cs2);
    } else {
#pragma unroll
        for (int i = 0; i < OPT_MB2; ++i) dd[i] = {zero, zero};
        d1 = ccot({x1 * m1.re, x1 * m1.im});
        d2 = ccot({x2 * m2.re, x2 * m2.im});
        sincos_mid(x1, sn1, cs1);
        sincos_mid(x2, sn2, cs2);
    }
    MieAccM A;
    MiePsi2 P;
    mie_accM_start(A, P, sn1, cs1, sn2, cs2);
    const int nstopm = s1.nstop > s2.nstop ? s1.nstop : s2.nstop;
#pragma unroll
    for (int i = 0; i < OPT_MB2; ++i) {
        const int n = i + 1;
        cplx2 D = dd[i];
        if (!small) {
            const double en = (double)n;
            const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
            d1 = csub(crecip(csub(ny1, d1)), ny1);
            d2 = csub(crecip(csub(ny2, d2)), ny2);
            D = pack2(d1, d2);
        }
        if (n <= nstopm) mie_termM(A, P, n, D, c, mie_n(n, make_double2(0., 0.)));
    }
    // beyond eight terms only upward lanes are left
    for (int n0 = OPT_MB2; n0 < nstopm; n0 += 4) {
#pragma unroll
        for (int i = 1; i <= 4; ++i) {
            const int n = n0 + i;
            const double en = (double)n;
            const cplx ny1 = {en * s1.yinv.re, en * s1.yinv.im}, ny2 = {en * s2.yinv.re, en * s2.yinv.im};
            d1 = csub(crecip(csub(ny1, d1)), ny1);
            d2 = csub(crecip(csub(ny2, d2)), ny2);
            if (n <= nstopm) mie_termM(A, P, n, pack2(d1, d2), c, c_mn[n]);
        }
    }
    const double f1 = 2. * s1.xinv * s1.xinv, f2 = 2. * s2.xinv * s2.xinv;
    qsca1 = f1 * (double)A.qs.x; qext1 = f1 * (double)A.qe.x;
    qsca2 = f2 * (double)A.qs.y; qext2 = f2 * (double)A.qe.y;
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
template <bool BSC, bool MIXED = false>   // BSC: a backscatter output (bscoef) is requested; MIXED: the series in mixed arithmetic (precision 1)
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
        auto add = [&](int w, double qe, double qs, double g, double qb) {
            double *s = acc + (size_t)(4 * w) * OPT_THREADS;
            s[0] += cross * qe;
            s[OPT_THREADS] += cross * qs;
            s[2 * OPT_THREADS] += cross * qs * g;
            s[3 * OPT_THREADS] += cross * qb * (0.25 / pi);
        };
        auto size_of = [&](int w, double &x, double &cbx) {
            x = pi * dwet * frcp(c_opt.wavelength_um[w]);
            const bool capped = x > OPT_XMAX;
            x = capped ? OPT_XMAX : x;
            cbx = capped ? 6.0822019955734001 /* 225^(1/3) */ : cbd * c_pil[1][w];
        };
        if (MIXED) {
#pragma unroll 1
            for (int w = 0; w < OPT_NSW; ++w) {
                if (!((want >> w) & 1)) continue;
                const cplx m = mix_index(vol, vwet, a.mix, w);
                double x, cbx;
                size_of(w, x, cbx);
                double qe, qs, g = 0., qb = 0.;
                if (mie_upward_ok(x, m)) mie_sphereM_up<BSC>(x, m, cbx, qe, qs, g, qb);
                else mie_sphere_down<true, OPT_NCK_SWC, BSC>(x, m, ck, qe, qs, g, qb);   // very small or strongly absorbing: fp64
                add(w, qe, qs, g, qb);
            }
        } else {
#pragma unroll 1
            for (int w = 0; w < OPT_NSW; ++w) {
                if (!((want >> w) & 1)) continue;
                const cplx m = mix_index(vol, vwet, a.mix, w);
                double x, cbx;
                size_of(w, x, cbx);
                double qe, qs, g = 0., qb = 0.;
                mie_sphere<true, OPT_NCK_SWC, BSC>(x, m, cbx, ck, qe, qs, g, qb);
                add(w, qe, qs, g, qb);
            }
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

template <bool MIXED>   // MIXED: the pair in the mixed arithmetic of mie_pairM_su (precision 1)
__device__ __forceinline__ void optics_lw_cell(const OpticsKernelArgs &a, double2 *s_ck,
                                               typename std::conditional<MIXED, float, double>::type *s_acc)
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
    typedef typename std::conditional<MIXED, float, double>::type acc_t;   // the mixed path keeps its sums in fp32
    acc_t *acc = s_acc + threadIdx.x;   // [band][thread]: absorption = extinction - scattering, summed per particle
#pragma unroll
    for (int i = 0; i < OPT_NLW; ++i) acc[i * OPT_THREADS] = (acc_t)0;
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
        float volf[OPT_NSPEC];
        if (MIXED) {
#pragma unroll
            for (int sp = 0; sp < OPT_NSPEC; ++sp) volf[sp] = (float)vol[sp];
        }
        const double dcm = dwet * 1.e-4;
        const double cross = (num * frcp(alt) * 1.e-6) * (0.25 * pi * dcm * dcm);   // #/cm3 * cm2
#pragma unroll 1
        for (int p = 0; p < OPT_NLW / 2; ++p) {
            if (!((want >> p) & 1)) continue;
            const int w = OPT_NSW + 2 * p, w2 = w + 1;
            cplx m, mb;
            if (MIXED && a.mix == 0) {   // volume mixing is linear: from fp32 volumes, nine registers instead of eighteen across
                                         // the band loop.  (Maxwell-Garnett takes the root of a complex number whose
                                         // imaginary part is small: in fp32 the absorption index of a weakly absorbing
                                         // mixture -- and with it the longwave absorption -- was off by up to 4e-3;
                                         // found by tools/stress_parity.py --mixed)
                const cf32 f1 = mix_index32(volf, (float)vwet, a.mix, w), f2 = mix_index32(volf, (float)vwet, a.mix, w2);
                m = {(double)f1.re, (double)f1.im}; mb = {(double)f2.re, (double)f2.im};
            } else {
                m = mix_index(vol, vwet, a.mix, w); mb = mix_index(vol, vwet, a.mix, w2);
            }
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
                if (MIXED) mie_pairM_su(x, m, xb, mb, t1, t2, small, qe, qs, qe2, qs2);
                else mie_pair_su(x, m, xb, mb, t1, t2, small, qe, qs, qe2, qs2);
            } else {   // strongly absorbing beyond any aerosol at these wavelengths: downward, one band after the other
                const MieQ2 q1 = mie_sphere_down_call(x, m, ck), q2 = mie_sphere_down_call(xb, mb, ck);
                qe = q1.qext; qs = q1.qsca; qe2 = q2.qext; qs2 = q2.qsca;
            }
            acc_t *s = acc + (size_t)(2 * p) * OPT_THREADS;
            s[0] += (acc_t)(cross * (qe - qs));
            s[OPT_THREADS] += (acc_t)(cross * (qe2 - qs2));
        }
    }
#pragma unroll 1
    for (int ib = 0; ib < OPT_NLW; ++ib) {
        const double ab = fmax((double)acc[(size_t)ib * OPT_THREADS], 0.);
        if (a.tauaerlw[ib]) str(a.tauaerlw[ib], q, r4, ab * 100. * dz);
        if (a.extaerlw[ib]) str(a.extaerlw[ib], q, r4, ab * 1.e5);
    }
}

// ---- aer_op_opt = 5 (shell_exact, registry.chem:4018): black carbon as a concentric core ----
// A core of size parameter x and index m1 inside a sphere of size parameter y and shell index m2 (everything but black
// carbon, volume-mixed), after Bohren & Huffman (1983) appendix B: the shell replaces D_n(m y) of the homogeneous
// sphere by two effective logarithmic derivatives, one for a_n and one for b_n,
//     Da_n = (D_n(m2 y) - A_n chi_n'(m2 y)) / (1 - A_n chi_n(m2 y)),   Db_n likewise with B_n,
// where A_n, B_n carry the core.  As in the oracle (oracle/optics_oracle.c: optics_oracle_mie_coated, pinned on the
// published BHCOAT example and on Aden & Kerker's coefficients from scipy's Bessel functions) the three logarithmic
// derivatives D_n(m1 x), D_n(m2 x), D_n(m2 y) come from the DOWNWARD recurrence -- carried upward from cot z, as the
// textbook routine does, they lose all accuracy for n > |z|, i.e. for every small particle -- and the core terms are
// dropped for good once they are below 1e-14 of D_n(m2 y), which keeps chi_n(m2 x) ~ (2n-1)!!/z^n inside the fp64 range.
// The option is a diagnostic configuration, not the production path: its kernel keeps the three derivative ladders in
// per-lane scratch arrays (12 KB per lane) instead of the checkpoint schemes above, and is a kernel of its own so that
// the production kernels' register allocation is untouched.
__device__ __forceinline__ cplx cdivc(cplx a, cplx b) { return cmul(a, crecip(b)); }
__device__ __forceinline__ double cabs2(cplx a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ void csincos(cplx z, cplx &sn, cplx &cs)
{
    double s, c;
    sincos(z.re, &s, &c);
    const double e = exp(z.im), ei = frcp(e);
    const double ch = 0.5 * (e + ei), sh = 0.5 * (e - ei);
    sn = {s * ch, c * sh};
    cs = {c * ch, -s * sh};
}
__device__ __noinline__ void log_derivative_down(cplx z, int nstop, double xstop, cplx *D /* [n * dstride], n = 1..nstop */,
                                                 int dstride)
{
    const double base = fmax(xstop, sqrt(cabs2(z)));
    const int nmx = (int)(base + (base > 8. ? 8. * cbrt(base) : 0.)) + 16;
    const cplx zinv = crecip(z);
    cplx d = {0., 0.};
    for (int n = nmx; n >= 2; --n) {
        const double en = (double)n;
        const cplx nz = {en * zinv.re, en * zinv.im};
        d = csub(nz, crecip(cadd(d, nz)));
        if (n - 1 <= nstop) D[(size_t)(n - 1) * dstride] = d;
    }
}
__device__ __noinline__ void mie_coated(double x, double y, cplx m1, cplx m2, double cby, cplx *Dx1, cplx *Dx2, cplx *Dy2,
                                        double &qext, double &qsca, double &gasy, double &qback,
                                        cplx *an_out = nullptr, cplx *bn_out = nullptr /* [n * cstride], n = 1..nstop, when given */,
                                        int cstride = 1, int dstride = 1 /* of the three ladders */)
{
    const cplx x1 = {x * m1.re, x * m1.im}, x2 = {x * m2.re, x * m2.im}, y2 = {y * m2.re, y * m2.im};
    const cplx refrel = cdivc(m2, m1), m2inv = crecip(m2), x2inv = crecip(x2), y2inv = crecip(y2);
    const double ystop = y + 4. * cby + 2.;
    const int nstop = (int)ystop;
    log_derivative_down(x1, nstop, ystop, Dx1, dstride);
    log_derivative_down(x2, nstop, ystop, Dx2, dstride);
    log_derivative_down(y2, nstop, ystop, Dy2, dstride);
    MieAcc A;
    {
        double sn, cs;
        sincos(y, &sn, &cs);
        mie_acc_start(A, sn, cs);
    }
    cplx chi0x2, chi1x2, chi0y2, chi1y2;
    {
        cplx sn, cs;
        csincos(x2, sn, cs);
        chi0x2 = {-sn.re, -sn.im}; chi1x2 = cs;
        csincos(y2, sn, cs);
        chi0y2 = {-sn.re, -sn.im}; chi1y2 = cs;
    }
    const double yinv = frcp(y);
    bool core_gone = false;
    for (int n = 1; n <= nstop; ++n) {
        const double en = (double)n, tn = 2. * en - 1.;
        const cplx dy = Dy2[(size_t)n * dstride];
        cplx da = dy, db = dy;
        if (!core_gone) {
            const cplx dx1 = Dx1[(size_t)n * dstride], dx2 = Dx2[(size_t)n * dstride];
            const cplx tx = {tn * x2inv.re, tn * x2inv.im}, ty = {tn * y2inv.re, tn * y2inv.im};
            const cplx chix2 = csub(cmul(tx, chi1x2), chi0x2), chiy2 = csub(cmul(ty, chi1y2), chi0y2);
            const cplx nx = {en * x2inv.re, en * x2inv.im}, ny = {en * y2inv.re, en * y2inv.im};
            const cplx chipx2 = csub(chi1x2, cmul(nx, chix2)), chipy2 = csub(chi1y2, cmul(ny, chiy2));
            const cplx wx = csub(cmul(chix2, dx2), chipx2), wy = csub(cmul(chiy2, dy), chipy2);
            const cplx rd = cmul(refrel, dx1);
            const cplx brack = cmul(cdivc(cdivc(csub(rd, dx2), csub(cmul(rd, chix2), chipx2)), wx), wy);
            const cplx crack = cmul(cdivc(cdivc(csub(cmul(refrel, dx2), dx1), csub(cmul(refrel, chipx2), cmul(dx1, chix2))), wx), wy);
            const double del2 = 1.e-28, dm2 = cabs2(dy);
            if (cabs2(cmul(brack, chipy2)) > del2 * dm2 || cabs2(cmul(brack, chiy2)) > del2 ||
                cabs2(cmul(crack, chipy2)) > del2 * dm2 || cabs2(cmul(crack, chiy2)) > del2) {
                const cplx one = {1., 0.};
                da = cdivc(csub(dy, cmul(brack, chipy2)), csub(one, cmul(brack, chiy2)));
                db = cdivc(csub(dy, cmul(crack, chipy2)), csub(one, cmul(crack, chiy2)));
                chi0x2 = chi1x2; chi1x2 = chix2; chi0y2 = chi1y2; chi1y2 = chiy2;
            } else {
                core_gone = true;
            }
        }
        // the homogeneous sphere's term with Da_n in a_n and Db_n in b_n
        const double nxr = en * yinv;
        const double psi1 = A.psi[0], chi1 = A.chi[0];
        const double psi = tn * yinv * psi1 - A.psi[1];
        const double chi = tn * yinv * chi1 - A.chi[1];
        A.psi[1] = psi1; A.psi[0] = psi; A.chi[1] = chi1; A.chi[0] = chi;
        cplx t = cmul(da, m2inv);
        t.re += nxr;
        const cplx na = {t.re * psi - psi1, t.im * psi};
        const cplx dna = {na.re + t.im * chi, na.im - (t.re * chi - chi1)};   // (t xi_n - xi_(n-1)), xi = psi - i chi
        t = cmul(m2, db);
        t.re += nxr;
        const cplx nb = {t.re * psi - psi1, t.im * psi};
        const cplx dnb = {nb.re + t.im * chi, nb.im - (t.re * chi - chi1)};
        const cplx an = cdivc(na, dna), bn = cdivc(nb, dnb);
        const double w2 = 2. * en + 1.;
        A.qs += w2 * (cabs2(an) + cabs2(bn));
        A.qe += w2 * (an.re + bn.re);
        A.gs += (w2 / (en * (en + 1.))) * (an.re * bn.re + an.im * bn.im);
        if (n > 1)
            A.gs += ((en - 1.) * (en + 1.) / en) * (A.an[0].re * an.re + A.an[0].im * an.im + A.bn[0].re * bn.re + A.bn[0].im * bn.im);
        A.xb.re += A.sgn * w2 * (an.re - bn.re);
        A.xb.im += A.sgn * w2 * (an.im - bn.im);
        A.sgn = -A.sgn;
        A.an[0] = an; A.bn[0] = bn;
        if (an_out) { an_out[(size_t)n * cstride] = an; bn_out[(size_t)n * cstride] = bn; }
    }
    const double y2i = yinv * yinv;
    qsca = 2. * y2i * A.qs;
    qext = 2. * y2i * A.qe;
    gasy = A.qs > 0. ? 2. * A.gs / A.qs : 0.;
    qback = y2i * cabs2(A.xb);
}

// ---- l2aer..l7aer (registry.chem:1382-1387): Legendre moments of the phase function at the four shortwave wavelengths ----
// chi_k of p(mu) = sum_k (2k+1) chi_k P_k(mu), chi_0 = 1, chi_1 = the asymmetry parameter; per cell the scattering-
// weighted mean over the sections, k = 2..7.  A diagnostic for schemes outside this path (the radiation calls take tau,
// omega, g), evaluated only when its arrays are passed, by a kernel of its own: coefficients a_n, b_n of a particle
// into per-lane scratch arrays (homogeneous sphere: downward D_n; aer_op_opt 5: mie_coated), then the intensity
// i(mu) = (|S1|^2 + |S2|^2) / 2 from the amplitude functions (Bohren & Huffman eq. 4.74) on a Gauss-Legendre rule of
// N >= nstop + 4 nodes (next multiple of 8), which integrates i(mu) P_k(mu), k <= 7, exactly (i is a polynomial of
// degree 2 nstop in mu).  The oracle does the same (optics_oracle_last_moments; its chi_1 equals the asymmetry parameter
// of the series, an independent formula, and the Rayleigh limit gives chi_2 = 1/10: tests/test_optics_oracle.py).
__device__ __noinline__ void mie_coeffs_down(double x, cplx m, double cbx, cplx *D, int dstride, cplx *an_out, cplx *bn_out,
                                             int cstride, int &nstop, double &qsca)
{
    const double xstop = x + 4. * cbx + 2.;
    nstop = (int)xstop;
    log_derivative_down({x * m.re, x * m.im}, nstop, xstop, D, dstride);
    double psi0, psi1, chi0, chi1;
    {
        double sn, cs;
        sincos(x, &sn, &cs);
        psi0 = cs; psi1 = sn; chi0 = -sn; chi1 = cs;
    }
    const double xinv = frcp(x);
    const cplx minv = crecip(m);
    double qs = 0.;
    for (int n = 1; n <= nstop; ++n) {
        const double en = (double)n, tn = 2. * en - 1.;
        const double psi = tn * xinv * psi1 - psi0, chi = tn * xinv * chi1 - chi0;
        const cplx d = D[(size_t)n * dstride];
        cplx t = cmul(d, minv);
        t.re += en * xinv;
        const cplx na = {t.re * psi - psi1, t.im * psi};
        const cplx da = {na.re + t.im * chi, na.im - (t.re * chi - chi1)};
        t = cmul(m, d);
        t.re += en * xinv;
        const cplx nb = {t.re * psi - psi1, t.im * psi};
        const cplx db = {nb.re + t.im * chi, nb.im - (t.re * chi - chi1)};
        const cplx an = cdivc(na, da), bn = cdivc(nb, db);
        an_out[(size_t)n * cstride] = an; bn_out[(size_t)n * cstride] = bn;
        qs += (2. * en + 1.) * (cabs2(an) + cabs2(bn));
        psi0 = psi1; psi1 = psi; chi0 = chi1; chi1 = chi;
    }
    qsca = 2. * xinv * xinv * qs;
}
// Mapping: a WAVE takes one particle at a time with its lanes on the quadrature nodes.  First every lane forms the
// coefficients of its own cell's particle (lanes = cells, as everywhere in this file) and leaves them in the wave's
// block of a global work area ([n][a_n | b_n][lane], coalesced); then the wave walks through its 64 particles: lane q
// carries node mu_q, reads a_n, b_n of the particle at hand (one address for the whole wave), sums the amplitude
// functions, and six wave reductions give the particle's moments to the lane that owns its cell.  With the lanes on
// their own particles instead (the first version) every lane ran its own N x nstop loop with the coefficients coming
// back from per-lane scratch: 865 ms per 1.6 M cells x 4 sections; this way 158 ms.  The normaliser needs no
// quadrature: the integral of i(mu) is x^2 Q_sca / 2 = sum_n (2n+1) (|a_n|^2 + |b_n|^2).
constexpr int LEG_WAVES = 2048;                        // workgroups (one wave each) of the launch: grid-stride over cell groups
                                                       // (2.7 GB of work area at most)
constexpr int LEG_SLOTS = 5;                           // a_n, b_n and the three derivative ladders
constexpr size_t LEG_AREA = (size_t)(OPT_NMIE + 2) * LEG_SLOTS * 64;   // cplx per wave: [n][slot][lane]
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void __launch_bounds__(64) optics_legendre_kernel(const OpticsKernelArgs a, cplx *work)
{
    __shared__ double s_acc[7 * OPT_NSW * 64];   // per wavelength: scattering sum, six weighted moments; [.][lane]
    __shared__ cplx s_ab[2 * (OPT_NMIE + 2)];    // a_n, b_n of the particle at hand
    const int lane = threadIdx.x;
    // element n of a lane's sequence at [n * LEG_STRIDE]; nothing of it in per-lane scratch: 12 KB of scratch per lane --
    // the first versions -- throttle the waves a CU may hold
    constexpr int LEG_STRIDE = LEG_SLOTS * 64;
    cplx *An = work + (size_t)blockIdx.x * LEG_AREA + lane, *Bn = An + 64, *Da = An + 128, *Db = An + 192, *Dc = An + 256;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    const long ngroups = (ncell + 63) / 64;
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const double pi = 3.14159265358979323846;
    double *acc = s_acc + lane;
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long cell = grp * 64 + lane;
        const bool live = cell < ncell;
        for (int i = 0; i < 7 * OPT_NSW; ++i) acc[i * 64] = 0.;
        size_t q = 0;
        double alt = 1.;
        if (live) {
            const int it = (int)(cell % a.ni_t);
            const long r = cell / a.ni_t;
            const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
            q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
            alt = ldr(a.alt, q, r4);
        }
        for (int ib = 0; ib < a.nbin; ++ib) {
            double vol[OPT_NSPEC];
            double vdry = 0., num = 0.;
            if (live) {
#pragma unroll
                for (int sp = 0; sp < OPT_NSPEC; ++sp) {
                    const int id = a.idx[ib][sp];
                    const double mass = id >= 0 ? fmax(ldr(a.chem, (size_t)id * sspec + q, r4), 0.) : 0.;
                    vol[sp] = mass * c_opt.rdens[sp];
                    if (sp != OS_WATER) vdry += vol[sp];
                }
                const int idn = a.idx[ib][OS_NUM];
                num = idn >= 0 ? fmax(ldr(a.chem, (size_t)idn * sspec + q, r4), 0.) : 0.;
            } else {
#pragma unroll
                for (int sp = 0; sp < OPT_NSPEC; ++sp) vol[sp] = 0.;
            }
            const bool valid = live && vdry > 1.e-25;
            double vwet = 1., dwet = 1., cbd = 1., cross = 0., vbc = 0., vmat = 1.;
            if (valid) {
                vwet = vdry + vol[OS_WATER];
                const double vpart = vdry * 1.e-6 * frcp(fmax(num, 1.e-30));
                double ddry = 1.e4 * cbrt(vpart * (6. / pi));
                const double dlo = c_opt.dlo_um[ib], dhi = c_opt.dhi_um[ib];
                if (ddry < dlo || ddry > dhi) {
                    ddry = ddry < dlo ? dlo : dhi;
                    const double dcm = ddry * 1.e-4;
                    num = vdry * 1.e-6 * frcp((pi / 6.) * dcm * dcm * dcm);
                }
                dwet = ddry * cbrt(vwet * frcp(vdry));
                cbd = cbrt(dwet);
                const double dcm = dwet * 1.e-4;
                cross = (num * frcp(alt) * 1.e-6) * (0.25 * pi * dcm * dcm);
                vbc = vol[OS_BC]; vmat = vwet - vbc;
            }
            if (!__any(valid)) continue;
#pragma unroll 1
            for (int w = 0; w < OPT_NSW; ++w) {
                // ---- the lanes' own particles: coefficients into the wave's work area ----
                int nstop = 0;
                double wsca = 0.;   // cross section x Q_sca / (integral of the intensity): the weight of the node sums
                if (valid) {
                    double x = pi * dwet * frcp(c_opt.wavelength_um[w]);
                    const bool capped = x > OPT_XMAX;
                    x = capped ? OPT_XMAX : x;
                    const double cbx = capped ? 6.0822019955734001 /* 225^(1/3) */ : cbd * c_pil[1][w];
                    nstop = (int)(x + 4. * cbx + 2.);
                    double qs;
                    cplx m = mix_index(vol, vwet, a.mix == 2 ? 1 : a.mix, w);   // (mix 2: replaced below)
                    bool coated = false;
                    if (a.mix == 2) {
                        const cplx mbc = {c_opt.refr[OS_BC][w], c_opt.refi[OS_BC][w]};
                        if (!(vmat > 1.e-30 * vwet)) m = mbc;
                        else {
                            double sr = 0., si = 0.;
#pragma unroll
                            for (int sp = 0; sp < OPT_NSPEC; ++sp) {
                                if (sp == OS_BC) continue;
                                sr += vol[sp] * c_opt.refr[sp][w];
                                si += vol[sp] * c_opt.refi[sp][w];
                            }
                            const double vi = frcp(vmat);
                            m = {sr * vi, si * vi};
                            if (vbc > 0.) {
                                double qe, g, qb;
                                mie_coated(cbrt(vbc * frcp(vwet)) * x, x, mbc, m, cbx, Da, Db, Dc, qe, qs, g, qb, An, Bn, LEG_STRIDE, LEG_STRIDE);
                                coated = true;
                            }
                        }
                    }
                    if (!coated) mie_coeffs_down(x, m, cbx, Da, LEG_STRIDE, An, Bn, LEG_STRIDE, nstop, qs);
                    // Q_sca = 2 / x^2 sum_n (2n+1)(|a_n|^2 + |b_n|^2) and the integral of i(mu) is that sum
                    wsca = qs > 0. ? cross * 2. * frcp(x * x) : 0.;
                    acc[(7 * w) * 64] += cross * qs;
                }
                __syncthreads();
                // ---- the wave on one particle after the other: lanes = quadrature nodes ----
                unsigned long long todo = __ballot(valid && wsca > 0.);
                while (todo) {
                    const int p = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int ns = __shfl(nstop, p, 64);
                    const int j = (ns + 4 + 7) / 8, N = 8 * j;
                    const double *xq = g_gl[0] + 4 * j * (j - 1), *wq = g_gl[1] + 4 * j * (j - 1);
                    {   // the particle's coefficients into LDS: 64 reads at a time instead of one after the other
                        const cplx *ap = work + (size_t)blockIdx.x * LEG_AREA + p, *bp = ap + 64;
                        for (int n = 1 + lane; n <= ns; n += 64) {
                            s_ab[2 * n] = ap[(size_t)n * LEG_STRIDE];
                            s_ab[2 * n + 1] = bp[(size_t)n * LEG_STRIDE];
                        }
                        __syncthreads();
                    }
                    double mk[6] = {0., 0., 0., 0., 0., 0.};
                    for (int q0 = 0; q0 < N; q0 += 64) {
                        const int qn = q0 + lane;
                        const bool on = qn < N;
                        const double mu = on ? xq[qn] : 0., wg = on ? wq[qn] : 0.;
                        double pi0 = 0., pi1 = 1.;
                        cplx s1 = {0., 0.}, s2 = {0., 0.};
#pragma unroll 4
                        for (int n = 1; n <= ns; ++n) {
                            // (per-term constants from the series' records: no division in this loop)
                            const double en = c_mn[n].en, c = c_mn[n].gx;
                            const double tau = en * mu * pi1 - (en + 1.) * pi0;
                            const cplx ca = s_ab[2 * n], cb = s_ab[2 * n + 1];
                            s1.re += c * (ca.re * pi1 + cb.re * tau); s1.im += c * (ca.im * pi1 + cb.im * tau);
                            s2.re += c * (ca.re * tau + cb.re * pi1); s2.im += c * (ca.im * tau + cb.im * pi1);
                            const double pin = (c_mn[n].w2 * mu * pi1 - (en + 1.) * pi0) * c_rn[n];
                            pi0 = pi1; pi1 = pin;
                        }
                        const double wi = wg * 0.5 * (cabs2(s1) + cabs2(s2));
                        double p0 = mu, p1 = 0.5 * (3. * mu * mu - 1.);   // P_1, P_2
                        mk[0] += wi * p1;
#pragma unroll
                        for (int k = 3; k < 8; ++k) {
                            const double pk = ((2. * k - 1.) * mu * p1 - (k - 1.) * p0) * (1. / k);
                            mk[k - 2] += wi * pk;
                            p0 = p1; p1 = pk;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double tot = wave_sum(mk[k]);
                        if (lane == p) acc[(7 * w + 1 + k) * 64] += wsca * tot;
                    }
                    __syncthreads();   // s_ab is rewritten for the next particle
                }
                __syncthreads();   // the work area is rewritten for the next wavelength
            }
        }
        if (live)
            for (int w = 0; w < OPT_NSW; ++w) {
                const double sca = acc[(7 * w) * 64];
                for (int kk = 0; kk < 6; ++kk)
                    str(a.laer[kk], (size_t)w * sspec + q, r4, sca > 0. ? acc[(7 * w + 1 + kk) * 64] * frcp(sca) : 0.);
            }
    }
}

// one thread per cell, all twenty wavelengths; the cell's 32 sums (4 x 4 shortwave, 16 longwave absorption) in LDS
__global__ void __launch_bounds__(OPT_THREADS) optics_shell_kernel(const OpticsKernelArgs a)
{
    __shared__ double s_acc[(4 * OPT_NSW + OPT_NLW) * OPT_THREADS];
    __shared__ double2 s_ck[OPT_NCK_SWC * OPT_THREADS];
    const long cell = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (cell >= ncell) return;
    unsigned want = 0;
#pragma unroll
    for (int w = 0; w < OPT_NSW; ++w)
        if (a.tauaer[w] || a.extaer[w] || a.bscoef[w]) want |= 1u << w;
#pragma unroll
    for (int b = 0; b < OPT_NLW; ++b)
        if (a.tauaerlw[b] || a.extaerlw[b]) want |= 1u << (OPT_NSW + b);
    if (!want) return;
    cplx Dx1[OPT_NMIE + 2], Dx2[OPT_NMIE + 2], Dy2[OPT_NMIE + 2];   // per-lane scratch
    double *acc = s_acc + threadIdx.x;
    double2 *ck = s_ck + threadIdx.x;
    for (int i = 0; i < 4 * OPT_NSW + OPT_NLW; ++i) acc[i * OPT_THREADS] = 0.;
    const int it = (int)(cell % a.ni_t);
    const long r = cell / a.ni_t;
    const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
    const size_t q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const double alt = ldr(a.alt, q, r4), dz = ldr(a.dz8w, q, r4);
    const double pi = 3.14159265358979323846;
    for (int ib = 0; ib < a.nbin; ++ib) {
        double vol[OPT_NSPEC];
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
        const double vpart = vdry * 1.e-6 * frcp(fmax(num, 1.e-30));
        double ddry = 1.e4 * cbrt(vpart * (6. / pi));
        const double dlo = c_opt.dlo_um[ib], dhi = c_opt.dhi_um[ib];
        if (ddry < dlo || ddry > dhi) {
            ddry = ddry < dlo ? dlo : dhi;
            const double dcm = ddry * 1.e-4;
            num = vdry * 1.e-6 * frcp((pi / 6.) * dcm * dcm * dcm);
        }
        const double dwet = ddry * cbrt(vwet * frcp(vdry));
        const double cbd = cbrt(dwet);
        const double dcm = dwet * 1.e-4;
        const double cross = (num * frcp(alt) * 1.e-6) * (0.25 * pi * dcm * dcm);
        const double vbc = vol[OS_BC], vmat = vwet - vbc;
        const double rcore = vbc > 0. ? cbrt(vbc * frcp(vwet)) : 0.;   // core radius / particle radius
#pragma unroll 1
        for (int w = 0; w < OPT_NWAVE; ++w) {
            if (!((want >> w) & 1)) continue;
            double sr = 0., si = 0.;
#pragma unroll
            for (int sp = 0; sp < OPT_NSPEC; ++sp) {
                if (sp == OS_BC) continue;
                sr += vol[sp] * c_opt.refr[sp][w];
                si += vol[sp] * c_opt.refi[sp][w];
            }
            const cplx mbc = {c_opt.refr[OS_BC][w], c_opt.refi[OS_BC][w]};
            double x = pi * dwet * frcp(c_opt.wavelength_um[w]);
            const bool capped = x > OPT_XMAX;
            x = capped ? OPT_XMAX : x;
            const double cbx = capped ? 6.0822019955734001 /* 225^(1/3) */ : cbd * c_pil[1][w];
            double qe, qs, g = 0., qb = 0.;
            if (!(vmat > 1.e-30 * vwet)) {                  // nothing but black carbon
                mie_sphere_down<true, OPT_NCK_SWC, true>(x, mbc, ck, qe, qs, g, qb);
            } else {
                const double vi = frcp(vmat);
                const cplx mm = {sr * vi, si * vi};
                if (!(vbc > 0.)) mie_sphere_down<true, OPT_NCK_SWC, true>(x, mm, ck, qe, qs, g, qb);   // no core
                else mie_coated(rcore * x, x, mbc, mm, cbx, Dx1, Dx2, Dy2, qe, qs, g, qb);
            }
            if (w < OPT_NSW) {
                double *s = acc + (size_t)(4 * w) * OPT_THREADS;
                s[0] += cross * qe;
                s[OPT_THREADS] += cross * qs;
                s[2 * OPT_THREADS] += cross * qs * g;
                s[3 * OPT_THREADS] += cross * qb * (0.25 / pi);
            } else {
                acc[(size_t)(4 * OPT_NSW + (w - OPT_NSW)) * OPT_THREADS] += cross * (qe - qs);
            }
        }
    }
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
    for (int ib = 0; ib < OPT_NLW; ++ib) {
        const double ab = fmax(acc[(size_t)(4 * OPT_NSW + ib) * OPT_THREADS], 0.);
        if (a.tauaerlw[ib]) str(a.tauaerlw[ib], q, r4, ab * 100. * dz);
        if (a.extaerlw[ib]) str(a.extaerlw[ib], q, r4, ab * 1.e5);
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

#ifndef OPT_LW_WAVES
#define OPT_LW_WAVES 2
#endif
__global__ void __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(OPT_LW_WAVES, OPT_LW_WAVES)))
optics_lw_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK_LW * OPT_THREADS];
    __shared__ double s_acc[OPT_NLW * OPT_THREADS];
    optics_lw_cell<false>(a, s_ck, s_acc);
}
// mixed path: the shortwave cell loop with the series in mixed arithmetic (with / without the backscatter sums)
__global__ void __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) optics_mix_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK_SWC * OPT_THREADS];
    __shared__ double s_acc[4 * OPT_NSW * OPT_THREADS];
    optics_sw_cell<false, true>(a, s_ck, s_acc);
}
__global__ void __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) optics_mix_bsc_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK_SWC * OPT_THREADS];
    __shared__ double s_acc[4 * OPT_NSW * OPT_THREADS];
    optics_sw_cell<true, true>(a, s_ck, s_acc);
}
// mixed path: the same cell loop with the pair in mixed arithmetic
__global__ void __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
optics_lw_mix_kernel(const OpticsKernelArgs a)
{
    __shared__ double2 s_ck[OPT_NCK_LW * OPT_THREADS];
    __shared__ float s_acc[OPT_NLW * OPT_THREADS];
    optics_lw_cell<true>(a, s_ck, s_acc);
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
        mn[n] = MieN{en, 2. * en - 1., 2. * en + 1., gw[n].x, gw[n].y, (float)en, (float)(2. * en - 1.), (float)(2. * en + 1.),
                     (float)gw[n].x, (float)gw[n].y, 0.f};
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_mn), mn, sizeof mn);
    if (e != hipSuccess) return e;
    {
        static double rn[OPT_NMIE + 10];
        rn[0] = 0.;
        for (int n = 1; n < OPT_NMIE + 10; ++n) rn[n] = 1. / (double)n;
        e = hipMemcpyToSymbol(HIP_SYMBOL(c_rn), rn, sizeof rn);
        if (e != hipSuccess) return e;
    }
    static MieN2 mp[OPT_NPAIR];
    for (int q = 0; q < OPT_NPAIR; ++q) {
        const MieN &a = mn[2 * q + 1], &b = mn[2 * q + 2 < OPT_NMIE + 10 ? 2 * q + 2 : 2 * q + 1];
        mp[q] = MieN2{a.en, b.en, a.tnm1, b.tnm1, {a.fen, b.fen}, {a.ftn, b.ftn}, {a.fw2, b.fw2}, {a.fgx, b.fgx}, {a.fgy, b.fgy}, {0.f, 0.f}};
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_mp), mp, sizeof mp);
    if (e != hipSuccess) return e;
    {   // Gauss-Legendre nodes and weights: Newton iteration on P_N from the Chebyshev-like first guess
        static double gl[2][OPT_GL_ENTRIES];
        const double pi = 3.14159265358979323846;
        for (int j = 1; j <= OPT_NGL; ++j) {
            const int N = 8 * j;
            double *xq = gl[0] + 4 * j * (j - 1), *wq = gl[1] + 4 * j * (j - 1);
            for (int i = 0; i < (N + 1) / 2; ++i) {
                double z = std::cos(pi * (i + 0.75) / (N + 0.5)), pp = 1.;
                for (int it = 0; it < 100; ++it) {
                    double p1 = 1., p2 = 0.;
                    for (int q = 1; q <= N; ++q) {
                        const double p3 = p2;
                        p2 = p1;
                        p1 = ((2. * q - 1.) * z * p2 - (q - 1.) * p3) / q;
                    }
                    pp = N * (z * p1 - p2) / (z * z - 1.);
                    const double dz = p1 / pp;
                    z -= dz;
                    if (std::fabs(dz) < 1e-15) break;
                }
                xq[i] = -z; xq[N - 1 - i] = z;
                wq[i] = wq[N - 1 - i] = 2. / ((1. - z * z) * pp * pp);
            }
        }
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_gl), gl, sizeof gl);
        if (e != hipSuccess) return e;
    }
    double pil[2][OPT_NWAVE];
    for (int w = 0; w < OPT_NWAVE; ++w) {
        pil[0][w] = 3.14159265358979323846 / t.wavelength_um[w];
        pil[1][w] = std::cbrt(pil[0][w]);
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_pil), pil, sizeof pil);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_opt), &t, sizeof(OpticsTables));
}

namespace {
// interstitial + cloud-borne tracers of every (section, slot) into a compact array of nbin * 10 fields
struct CombineArgs {
    const void *chem;
    void *out;
    int ia[OPT_MAXBIN][OPT_NSLOT], ic[OPT_MAXBIN][OPT_NSLOT];   // species numbers in chem, -1 = absent
    size_t ncell;
    int nbin, real4;
};
__global__ void optics_combine_cw_kernel(const CombineArgs a)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.ncell) return;
    for (int b = 0; b < a.nbin; ++b)
        for (int s = 0; s < OPT_NSLOT; ++s) {
            const int i = a.ia[b][s], c = a.ic[b][s];
            const size_t o = (size_t)(b * OPT_NSLOT + s) * a.ncell + q;
            if (a.real4) {
                const float *p = static_cast<const float *>(a.chem);
                static_cast<float *>(a.out)[o] = (i >= 0 ? p[(size_t)i * a.ncell + q] : 0.f) + (c >= 0 ? p[(size_t)c * a.ncell + q] : 0.f);
            } else {
                const double *p = static_cast<const double *>(a.chem);
                static_cast<double *>(a.out)[o] = (i >= 0 ? p[(size_t)i * a.ncell + q] : 0.) + (c >= 0 ? p[(size_t)c * a.ncell + q] : 0.);
            }
        }
}
}  // namespace

// out: nbin * 10 fields of ncell values in the tracers' type; field b * 10 + s = chem[ia[b][s]] + chem[ic[b][s]]
hipError_t optics_combine_cw(const void *chem, void *out, const int (*ia)[OPT_NSLOT], const int (*ic)[OPT_NSLOT], int nbin,
                             size_t ncell, int real4, hipStream_t stream)
{
    CombineArgs a{};
    a.chem = chem; a.out = out; a.ncell = ncell; a.nbin = nbin; a.real4 = real4;
    for (int b = 0; b < OPT_MAXBIN; ++b)
        for (int s = 0; s < OPT_NSLOT; ++s) { a.ia[b][s] = ia[b][s]; a.ic[b][s] = ic[b][s]; }
    hipLaunchKernelGGL(optics_combine_cw_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t optics_launch(const OpticsKernelArgs &a, hipStream_t stream, int precision)
{
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (ncell <= 0) return hipSuccess;
    if ((ncell + OPT_THREADS - 1) / OPT_THREADS > 0x7fffffffL) return hipErrorInvalidValue;
    // skip a kernel none of whose outputs were asked for
    bool any_sw = false, any_lw = false;
    for (int w = 0; w < OPT_NSW; ++w) any_sw = any_sw || a.tauaer[w] || a.extaer[w] || a.bscoef[w];
    for (int b = 0; b < OPT_NLW; ++b) any_lw = any_lw || a.tauaerlw[b] || a.extaerlw[b];
    if (a.laer[0]) {   // the Legendre moments, when asked for (fp64 whatever the precision mode)
        const long ngroups = (ncell + 63) / 64;
        unsigned nwg = (unsigned)(ngroups < LEG_WAVES ? ngroups : LEG_WAVES);
        void *work = nullptr;   // stream-ordered work area: the coefficients of the waves' particles
        hipError_t e = hipMallocAsync(&work, (size_t)nwg * LEG_AREA * sizeof(cplx), stream);
        while (e == hipErrorOutOfMemory && nwg > 64) {   // a device that cannot spare it: fewer waves, same result
            (void)hipGetLastError();
            nwg /= 2;
            e = hipMallocAsync(&work, (size_t)nwg * LEG_AREA * sizeof(cplx), stream);
        }
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(optics_legendre_kernel, dim3(nwg), dim3(64), 0, stream, a, static_cast<cplx *>(work));
        e = hipGetLastError();
        (void)hipFreeAsync(work, stream);
        if (e != hipSuccess) return e;
    }
    if (a.mix == 2) {   // shell_exact: one kernel for all twenty wavelengths, fp64 whatever the precision mode
        if (any_sw || any_lw)
            hipLaunchKernelGGL(optics_shell_kernel, dim3((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), dim3(OPT_THREADS), 0,
                               stream, a);
        return hipGetLastError();
    }
    if (any_sw) {   // one thread per cell; the backscatter sums only when a bscoef output is there
        bool bsc = false;
        for (int w = 0; w < OPT_NSW; ++w) bsc = bsc || a.bscoef[w];
        const dim3 grid((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), wg(OPT_THREADS);
        if (precision == 1 && bsc) hipLaunchKernelGGL(optics_mix_bsc_kernel, grid, wg, 0, stream, a);
        else if (precision == 1) hipLaunchKernelGGL(optics_mix_kernel, grid, wg, 0, stream, a);
        else if (bsc) hipLaunchKernelGGL(optics_bsc_kernel, grid, wg, 0, stream, a);
        else hipLaunchKernelGGL(optics_kernel, grid, wg, 0, stream, a);
    }
    if (any_lw && precision == 1)
        hipLaunchKernelGGL(optics_lw_mix_kernel, dim3((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), dim3(OPT_THREADS), 0,
                           stream, a);
    else if (any_lw)   // one thread per cell
        hipLaunchKernelGGL(optics_lw_kernel, dim3((unsigned)((ncell + OPT_THREADS - 1) / OPT_THREADS)), dim3(OPT_THREADS), 0,
                           stream, a);
    return hipGetLastError();
}

}  // namespace rrtmg
