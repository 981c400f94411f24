// Longwave column kernel: RRTMG_LWRAD + rrtmg_lw (gas-only "clean" call and
// gas+aerosol call in one launch) for gfx950.
//
// Reference path (phys/module_ra_rrtmg_lw.F): RRTMG_LWRAD preprocessing
// LW:12149-12298 (buffer layers to 0 mb), aerosol copy LW:12576-12630, inatm
// LW:11248-11300, setcoef LW:3540-3808, taugb1..16 LW:4712-7828, rtrnmc
// LW:3135-3410, rrtmg_lw LW:10996-11060, outputs LW:12646-12692.
//
// Mapping: one 192-thread workgroup per column.  Prologue with lanes = layers
// (profile, O3, column amounts, setcoef, Planck functions, the "too abundant to
// be minor" adjustments) into LDS.  Main phase with lanes = g-points (3 waves x
// 47): the 16 taugbNN routines are one data-driven expression; gas optics are
// shared by the two radiative-transfer calls (as in the reference, LW:10990-11044,
// where taumol runs once).  The downward and the upward sweep each evaluate the
// layer optics (no per-lane profile store); level fluxes are wave reductions.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "lw_common.h"
#include "lw_kernel.h"
#include "sw_kernel.h"   // KISS_JUMP_WORDS
#include "mcica_device.h"
#include "cld_retab.h"

namespace rrtmg {

namespace {

constexpr int LW_THREADS = 192;
constexpr int LW_CHUNK = 47;

// Per-layer results of the pre-pass over the layers of a binary-species segment (lw_gas_kernel): everything that
// depends on (band, regime, layer) but not on the g-point -- the reference forms these once per layer around its ig
// loop (e.g. LW:5199-5301, 5354-5355); the (layer, g) lanes only gather and sum.  A wave's area holds PP_LAYERS layers:
//   doubles [field][layer]: 0..5 weights of the major-species stencil for the reference level jp (3 points x 2
//   temperatures, already times fac00 / fac10), 6 speccomb; 7..12, 13 the same for jp + 1; 14..16 interpolation
//   fraction of the three minor species; 17 of the Planck fraction
//   ints [field][layer]: row offsets (in doubles) of the first stencil point for jp, jp + 1; of the three minor
//   species' tables; of the Planck-fraction table
constexpr int PP_LAYERS = 16, PP_NDBL = 18, PP_NINT = 6;
constexpr int PP_DOUBLES = PP_NDBL * PP_LAYERS + PP_NINT * PP_LAYERS / 2;
// The water-vapour continuum rows of a segment's band (selfref: 10 rows, forref: 4 rows of <= 16 g-points), staged by
// the wave that works on the segment: 4 of an item's gathers become LDS reads -- the gas kernel is bound by the bytes
// its gathers pull through the vector cache (64 B per cycle and CU), the LDS has twice that rate and is idle.
constexpr int CONT_SELF_ROWS = 10, CONT_FOR_ROWS = 4, CONT_STRIDE = 16;
constexpr int CONT_DOUBLES = (CONT_SELF_ROWS + CONT_FOR_ROWS) * CONT_STRIDE;


__constant__ double c_lw_preflog[59];
__constant__ double c_lw_tref[59];
__constant__ double c_lw_o3ann[31];
__constant__ double c_lw_ppwrkh[32];
__constant__ double c_lw_retab[95];   // ice effective radius versus temperature (LW:11424-11442)
// mean pressure / temperature profile for the buffer layers, LW:11802-11821
__constant__ double c_pprof[60] = {
    1000.00, 855.47, 731.82, 626.05, 535.57, 458.16, 391.94, 335.29, 286.83, 245.38, 209.91, 179.57,
    153.62, 131.41, 112.42, 96.17, 82.27, 70.38, 60.21, 51.51, 44.06, 37.69, 32.25, 27.59,
    23.60, 20.19, 17.27, 14.77, 12.64, 10.81, 9.25, 7.91, 6.77, 5.79, 4.95, 4.24,
    3.63, 3.10, 2.65, 2.27, 1.94, 1.66, 1.42, 1.22, 1.04, 0.89, 0.76, 0.65,
    0.56, 0.48, 0.41, 0.35, 0.30, 0.26, 0.22, 0.19, 0.16, 0.14, 0.12, 0.10};
__constant__ double c_tprof[60] = {
    286.96, 281.07, 275.16, 268.11, 260.56, 253.02, 245.62, 238.41, 231.57, 225.91, 221.72, 217.79,
    215.06, 212.74, 210.25, 210.16, 210.69, 212.14, 213.74, 215.37, 216.82, 217.94, 219.03, 220.18,
    221.37, 222.64, 224.16, 225.88, 227.63, 229.51, 231.50, 233.73, 236.18, 238.78, 241.60, 244.44,
    247.35, 250.33, 253.32, 256.30, 259.22, 262.12, 264.80, 266.50, 267.59, 268.44, 268.69, 267.76,
    266.13, 263.96, 261.54, 258.93, 256.15, 253.23, 249.89, 246.67, 243.48, 240.25, 236.66, 233.86};
// diffusivity-angle coefficients, LW:2958-2969
__constant__ double c_a0[LW_NBND] = {1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
__constant__ double c_a1[LW_NBND] = {0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
__constant__ double c_a2[LW_NBND] = {0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};

__device__ __forceinline__ double ldr(const void *p, size_t idx, int real4)
{
    // branch-free for both widths: a branch around every load would put a wait behind each one (the loads of a
    // phase then cost one trip to memory each instead of one together), so the value comes in as two dwords -- the
    // same dword twice for a REAL*4 array -- and is picked by a select
    const unsigned *w = reinterpret_cast<const unsigned *>(p) + (real4 ? idx : 2 * idx);
    const unsigned lo = w[0], hi = w[real4 ? 0 : 1];
    const double f = (double)__uint_as_float(lo), d = __hiloint2double((int)hi, (int)lo);
    return real4 ? f : d;
}
__device__ __forceinline__ void str(void *p, size_t idx, int real4, double v)
{
    if (real4) reinterpret_cast<float *>(p)[idx] = (float)v;
    else reinterpret_cast<double *>(p)[idx] = v;
}

// reciprocal / quotient without the compiler's IEEE special-case scaffolding (see sw_kernel.hip: v_rcp_f64 is good to
// 2^-24.4; one Newton step gives 2.2e-15, two 1.1e-16; the quotient built on the one-step reciprocal with its residual
// correction is the correctly rounded one -- it is used wherever a table index is cut from the quotient)
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
__device__ __forceinline__ double fdiv(double a, double b)
{
    double r = frcp1(b);
    double q = a * r;
    double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
}
// One Horner step p*r + c with the coefficient c kept in a register that SURVIVES the step.  Written as C the compiler
// picks the two-address v_fmac_f64 (dst += a*b) and copies the coefficient into a fresh destination first -- a
// v_mov_b64 per step: the 13 steps of exp_neg were 26 instructions, 130 of the 1 166 of the transfer kernel's
// downward loop (four layers).  The three-address v_fma_f64 needs no copy.
__device__ __forceinline__ double horner(double p, double r, double c)
{
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "v"(c));
    return o;
}
// exp(-x) for 0 <= x <~ 700 in fp64: n = rint(-x/ln2), two-term Cody-Waite reduction, degree-13 Taylor
// on |r| <= ln2/2 (truncation 6e-18 relative), v_ldexp_f64.  Within ~2 ulp of a correctly rounded exp.
__device__ __forceinline__ double exp_neg(double x)
{
    const double t = -x;
    const double n = rint(t * 1.4426950408889634074);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, t);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821614599e-10;      // 1/13!
    p = horner(p, r, 2.0876756987868098979e-09); // 1/12!
    p = horner(p, r, 2.5052108385441718775e-08); // 1/11!
    p = horner(p, r, 2.7557319223985890653e-07); // 1/10!
    p = horner(p, r, 2.7557319223985892511e-06); // 1/9!
    p = horner(p, r, 2.4801587301587301566e-05); // 1/8!
    p = horner(p, r, 1.9841269841269841253e-04); // 1/7!
    p = horner(p, r, 1.3888888888888889419e-03); // 1/6!
    p = horner(p, r, 8.3333333333333332177e-03); // 1/5!
    p = horner(p, r, 4.1666666666666664354e-02); // 1/4!
    p = horner(p, r, 1.6666666666666665741e-01); // 1/3!
    p = horner(p, r, 0.5);
    p = horner(p, r, 1.0);
    p = horner(p, r, 1.0);
    return ldexp(p, (int)n);
}

// rad + (bb - rad)*at, LW:3296: the difference is kept out of FMA contraction with the product that formed bb, so that
// the clean, the aerosol and the clear-sky stream round alike (equal inputs -> bitwise equal fluxes)
__device__ __forceinline__ double relax(double rad, double bb, double at)
{
#pragma clang fp contract(off)
    const double d = bb - rad;
    return __builtin_fma(d, at, rad);
}

// sums of two values over the wave: lane 0 ends with sum(a), lane 32 with sum(b)
__device__ __forceinline__ double wave_sum2(double a, double b, int lane)
{
    // no FMA contraction with the caller's products: the two sums must round alike (equal inputs -> equal sums)
#pragma clang fp contract(off)
    const bool hi = lane & 32;
    double send = hi ? a : b, keep = hi ? b : a;
    double v = keep + __shfl_xor(send, 32, 64);
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---- wave sums of four values at a time (gfx950 lane swaps, then DPP row steps; see sw_kernel.hip) ----
typedef unsigned lw_uint2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fold32(double a, double b)
{
    const lw_uint2 lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const lw_uint2 hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
__device__ __forceinline__ double fold16(double a, double b)
{
    const lw_uint2 lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const lw_uint2 hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    const int mlo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int mhi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(mhi, mlo);
}
// the result's row r (lanes 16r..16r+15) holds the wave sum of v[wsum_index(r)]
__device__ __forceinline__ double wave_sum4(double v0, double v1, double v2, double v3)
{
    double v = fold16(fold32(v0, v1), fold32(v2, v3));   // rows: v0, v2, v1, v3
    v = dpp_add<0x140>(v);   // row_mirror
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ int wsum_index(int row) { return ((row & 1) << 1) | (row >> 1); }

struct LCol {   // LDS image of the column
    double *fac00, *fac01, *fac10, *fac11, *selffac, *selffrac, *forfac, *forfrac, *minorfrac, *pavel;
    double *pz;        // [nl+1]
    double *mul;       // [LS_COUNT][nl]
    double *rat;       // [LR_COUNT][2][nl]
    double *q4;        // [16][nl4][4]: planklay, dplankdn, dplankup, taua (lw_kernel.h)
    double *secdiff, *plankbnd;   // [16]
    double *wrk;       // [2][nl] scratch for the column sums
    double *tz;        // [nl+1] level temperatures
    double *tav;       // [nl] t3d of the layer (the layers kte-1.. take the level mean instead)
    double *pp;        // [3 waves][PP_DOUBLES] per-layer results of the binary-species pre-pass (gas kernel)
    double *cont;      // [3 waves][CONT_DOUBLES] staged continuum rows of the wave's segment, or nullptr
    double *flux;      // [3 waves][4][nl+1]: down clean, down total, up clean, up total
    int *rb0, *rb1, *inds, *indf, *indm;
    int nl;
};

// gas kernel: the layer state lives in LDS; the per-band Planck / aerosol arrays are written straight
// into the column's global record (lw_rec_doubles), which lw_rt_kernel stages into its own LDS
__device__ __forceinline__ LCol carve_gas(double *base, double *rec, int nl, bool stage_cont)
{
    LCol c;
    c.nl = nl;
    double *p = base;
    c.fac00 = p; p += nl; c.fac01 = p; p += nl; c.fac10 = p; p += nl; c.fac11 = p; p += nl;
    c.selffac = p; p += nl; c.selffrac = p; p += nl; c.forfac = p; p += nl; c.forfrac = p; p += nl;
    c.minorfrac = p; p += nl; c.pavel = p; p += nl;
    c.pz = p; p += nl + 1;
    c.mul = p; p += LS_COUNT * nl;
    c.rat = p; p += LR_COUNT * 2 * nl;
    c.wrk = p; p += 2 * nl;
    c.tz = p; p += nl + 1;
    c.tav = p; p += nl;
    c.pp = p; p += 3 * PP_DOUBLES;
    c.cont = stage_cont ? p : nullptr; p += stage_cont ? 3 * CONT_DOUBLES : 0;
    int *q = reinterpret_cast<int *>(p);
    c.rb0 = q; q += nl; c.rb1 = q; q += nl; c.inds = q; q += nl; c.indf = q; q += nl; c.indm = q; q += nl;
    c.secdiff = rec; c.plankbnd = rec + LW_NBND;
    c.q4 = rec + lw_rec_q4_offset(nl);
    c.flux = nullptr;
    return c;
}
// transfer kernel: LDS image of the record (same order) followed by the flux slots
__device__ __forceinline__ LCol carve_rt(double *base, int nl, bool cloud)
{
    LCol c{};
    c.nl = nl;
    double *p = base;
    c.secdiff = p; p += LW_NBND; c.plankbnd = p; p += LW_NBND;
    c.pz = p; p += nl + 1;
    c.q4 = base + lw_rec_q4_offset(nl);
    // after the image of the whole record; [3 waves][4 or 6][nl+1]
    c.flux = base + lw_rec_tc_offset(nl);   // (the record without its last part, which this kernel does not read)
    (void)cloud;
    return c;
}

// species index / fraction of a binary parameter, e.g. LW:5199-5204
__device__ __forceinline__ void spec_idx(double cola, double ratio, double colb, double mult, double oneminus,
                                         double &speccomb, double &specparm, int &j, double &f)
{
    speccomb = cola + ratio * colb;
    specparm = fdiv(cola, speccomb);
    specparm = specparm >= oneminus ? oneminus : specparm;
    double specmult = mult * specparm;
    j = (int)specmult;
    f = specmult - (double)j;
}

// The 32 (band, regime) segments of the gas optics, costliest first: the regime of the band's FIRST g-point (offsets of
// g-point ig are + ig) and the band's numbers, in ONE record per segment, read with a wave-uniform index, i.e. into
// scalar registers.  One record, one symbol: a wave that has claimed a segment is two dependent scalar reads away from
// everything it needs (the symbol's address, then the fields) -- as three tables (order, band, first g-point) it was
// six.
struct LwSegRec {
    LwRegime R;
    int32_t upper, g0, ng, stride, off_self, off_for;
};
__constant__ LwSegRec c_lwseg[2 * LW_NBND];

// A lane of the gas optics owns TWO neighbouring g-points (ig, ig + 1; every band has an even number): each of its
// table reads is then one 16-byte load -- the kernel is bound by the rate at which the vector cache takes gather
// instructions (r02: 1.4e10 cache accesses per launch = 23 of its 28 ms), and a 16-byte gather costs it what an
// 8-byte one does -- and the per-layer weights, fractions and row offsets are fetched once for two points.
// What differs between the g-points of a band besides the table column:
struct LaneG {
    int ig;                                       // the even one of the pair
    double2 fracconst, halo0, halo1, postscale;
};
__device__ __forceinline__ double2 ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
__device__ __forceinline__ double2 mul2(double w, double2 v) { return make_double2(w * v.x, w * v.y); }
__device__ __forceinline__ double2 add2(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 sub2(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 mad2(double w, double2 v, double2 a) { return make_double2(a.x + w * v.x, a.y + w * v.y); }

// The 16 taugbNN routines (LW:4712-7828) as data-driven expressions.  R is the regime of the wave's band -- the same
// for every lane, so every feature test (continua, each minor species, halocarbon terms, pressure corrections, Planck
// fraction mode) is a scalar branch and no lane pays for another band's terms.  The kernel waits on dependent
// gathers (LDS -> index arithmetic -> table rows -> sums), so an item is written in two passes: first every table read
// is issued, then the sums are taken in the reference's order.  Two kinds of regime (lw_upload_segments checks that
// the descriptors fall into them): one key species or none (gas_item_single; minor species and Planck fractions then
// depend on temperature only), and two key species (gas_prepass + gas_item_binary below).
__device__ __forceinline__ void gas_item_single(const LwDeviceTables &tb, const LCol &c, const LwRegime &R, const int ng,
                                                const int off_self, const int off_for, const LaneG &G, const int k,
                                                const double *cont, double2 &taug, double2 &frac)
{
    const uint32_t cf = R.cfg;
    const int nl = c.nl, ig = G.ig;
    const int type = (cf >> LC_TYPE_SHIFT) & 3;
    const int spa = (cf >> LC_SPA_SHIFT) & 15;
    const int nsp = (cf >> LC_NSP_SHIFT) & 15;
    const int fmode = (cf >> LC_FRAC_SHIFT) & 3;
    const double2 zero = make_double2(0., 0.);
    // ---- pass 1: table reads ----
    double2 va[2] = {zero, zero}, vb[2] = {zero, zero};
    if (type != 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int rb = m == 0 ? c.rb0[k] : c.rb1[k];
            const double *Kp = tb.K + R.off_major + ig + (size_t)(rb * nsp) * ng;
            va[m] = ld2(Kp); vb[m] = ld2(Kp + ng);   // the two temperatures of this reference level
        }
    }
    double2 s0 = zero, s1 = zero, f0 = zero, f1 = zero;
    if (cf & (LC_SELF | LC_FOR)) {
        if (cont) {   // staged by this wave for the segment (lw_gas_kernel)
            const double *Ff = cont + (CONT_SELF_ROWS + c.indf[k]) * CONT_STRIDE + ig;
            f0 = ld2(Ff); f1 = ld2(Ff + CONT_STRIDE);
            if (cf & LC_SELF) {
                const double *Sf = cont + c.inds[k] * CONT_STRIDE + ig;
                s0 = ld2(Sf); s1 = ld2(Sf + CONT_STRIDE);
            }
        } else {
            const double *Ff = tb.K + off_for + ig + (size_t)c.indf[k] * ng;
            f0 = ld2(Ff); f1 = ld2(Ff + ng);
            if (cf & LC_SELF) {
                const double *Sf = tb.K + off_self + ig + (size_t)c.inds[k] * ng;
                s0 = ld2(Sf); s1 = ld2(Sf + ng);
            }
        }
    }
    double2 ma[3], mb[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const LwMinor &M = R.minor[m];
        ma[m] = mb[m] = zero;
        if (M.slot != LS_ZERO) {
            const double *Km = tb.K + M.off + ig + (size_t)c.indm[k] * ng;
            ma[m] = ld2(Km); mb[m] = ld2(Km + ng);
        }
    }
    // ---- pass 2: sums ----
    double2 tau = zero;
    if (type != 0) {
        const double2 maj0 = mad2(c.fac10[k], vb[0], mul2(c.fac00[k], va[0]));
        const double2 maj1 = mad2(c.fac11[k], vb[1], mul2(c.fac01[k], va[1]));
        tau = mul2(c.mul[spa * nl + k], add2(maj0, maj1));
    }
    // continua (selffac / forfac already carry colh2o, LW:3806-3807)
    if (cf & (LC_SELF | LC_FOR)) {
        if (cf & LC_SELF) tau = mad2(c.selffac[k], mad2(c.selffrac[k], sub2(s1, s0), s0), tau);
        if (cf & LC_FOR) tau = mad2(c.forfac[k], mad2(c.forfrac[k], sub2(f1, f0), f0), tau);
    }
    // minor species, e.g. LW:4880-4885, 5877-5878
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const LwMinor &M = R.minor[m];
        if (M.slot != LS_ZERO) tau = mad2(c.mul[M.slot * nl + k], mad2(c.minorfrac[k], sub2(mb[m], ma[m]), ma[m]), tau);
    }
    const double h0 = c.mul[R.halo_slot[0] * nl + k], h1 = c.mul[R.halo_slot[1] * nl + k];
    double2 t = make_double2(tau.x + h0 * G.halo0.x + h1 * G.halo1.x, tau.y + h0 * G.halo0.y + h1 * G.halo1.y);
    // pressure corrections of bands 1 and 2, LW:4873-4876, 4906, 5016
    const int corr = (cf >> LC_CORR_SHIFT) & 3;
    if (corr) {
        const double pp = c.pavel[k];
        double corradj = 1.;
        if (corr == 1) corradj = pp < 250. ? 1. - 0.15 * (250. - pp) / 154.4 : 1.;
        else if (corr == 2) corradj = 1. - 0.15 * (pp / 95.6);
        else corradj = 1. - .05 * (pp - 100.) / 900.;
        t = mul2(corradj, t);
    }
    taug = make_double2(t.x * G.postscale.x, t.y * G.postscale.y);
    frac = fmode == 1 ? G.fracconst : zero;
}

// ---- binary-species segments: pre-pass (lanes = layers x jobs) and item (lanes = layers x g-points) ----
// Pre-pass over up to PP_LAYERS layers starting at kc.  Lane = (layer l = lane & 15, job slot = lane >> 4 [+ 4 in a
// second round]): slots 0 / 1 the major species at reference levels jp / jp + 1, slot 2 the first minor species, slot
// 3 the Planck fraction, slots 4 / 5 the other two minor species (band 13 only) -- each lane forms ONE species index
// (the division of spec_idx) where a lane of the old item formed up to six.
__device__ __forceinline__ void gas_prepass(const LwDeviceTables &tb, const LCol &c, const LwRegime &R, const int ng,
                                            double *ppd, int *ppi, const int kc, const int kend, const int lane)
{
    const uint32_t cf = R.cfg;
    const int nl = c.nl;
    const int spa = (cf >> LC_SPA_SHIFT) & 15, spb = (cf >> LC_SPB_SHIFT) & 15, rs = (cf >> LC_RAT_SHIFT) & 7;
    const int nsp = (cf >> LC_NSP_SHIFT) & 15;
    const bool edges = cf & LC_EDGES;
    const double mult = (cf & LC_MULT4) ? 4. : 8.;
    const int fmode = (cf >> LC_FRAC_SHIFT) & 3;
    // the lane number afresh (two instructions) rather than a value carried, in scratch, through the segment loop
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int l = ln & 15;
    const int k = kc + l < kend ? kc + l : kend - 1;
    const double colA = c.mul[spa * nl + k], colB = c.mul[spb * nl + k];
    const bool more = R.minor[1].slot != LS_ZERO || R.minor[2].slot != LS_ZERO;   // wave-uniform (band 13)
    int msp0 = R.minor[0].nsp, msp1 = R.minor[1].nsp, msp2 = R.minor[2].nsp;
    double rpl = R.planck_ratio, rm0 = R.minor[0].refrat, rm1 = R.minor[1].refrat, rm2 = R.minor[2].refrat;
    // scalar reads, not one read through a selected address
    asm("" : "+s"(msp0), "+s"(msp1), "+s"(msp2), "+s"(rpl), "+s"(rm0), "+s"(rm1), "+s"(rm2));
    for (int round = 0; round < (more ? 2 : 1); ++round) {
        const int slot = (ln >> 4) + 4 * round;
        // the job's species ratio
        const double rat01 = c.rat[(rs * 2 + (slot & 1)) * nl + k];     // read by every lane: a select of values,
        double ratio = rpl;                                              // slot 3     not of addresses (flat load)
        if (slot <= 1) ratio = rat01;
        else if (slot == 2) ratio = rm0;
        else if (slot == 4) ratio = rm1;
        else if (slot == 5) ratio = rm2;
        double speccomb, specparm, f;
        int j;
        spec_idx(colA, ratio, colB, mult, tb.oneminus, speccomb, specparm, j, f);
        if (slot <= 1) {
            // stencil of the major species: two points (1 - fs, fs) or, next to the ends of the species axis, three
            // (LW:5231-5290); the first point's row and the weights times the temperature / pressure factors
            double w0 = 1. - f, w1 = f, w2 = 0.;
            int base = j;
            if (edges) {
                const bool lo = specparm < 0.125, hi = specparm > 0.875;
                const double p = lo ? f - 1 : -f;
                const double p4 = (p * p) * (p * p);
                const double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
                w0 = lo ? fk0 : (hi ? fk2 : w0);
                w1 = lo ? fk1 : (hi ? fk1 : w1);
                w2 = lo ? fk2 : (hi ? fk0 : 0.);
                base = hi ? j - 1 : j;
            }
            const double fTa = slot == 0 ? c.fac00[k] : c.fac01[k], fTb = slot == 0 ? c.fac10[k] : c.fac11[k];
            const int rb = slot == 0 ? c.rb0[k] : c.rb1[k];
            double *d = ppd + (slot * 7) * PP_LAYERS + l;
            d[0 * PP_LAYERS] = w0 * fTa; d[1 * PP_LAYERS] = w1 * fTa; d[2 * PP_LAYERS] = w2 * fTa;
            d[3 * PP_LAYERS] = w0 * fTb; d[4 * PP_LAYERS] = w1 * fTb; d[5 * PP_LAYERS] = w2 * fTb;
            d[6 * PP_LAYERS] = speccomb;
            ppi[slot * PP_LAYERS + l] = (rb * nsp + base) * ng;
        } else if (slot == 3) {
            if (fmode == 2) { ppd[17 * PP_LAYERS + l] = f; ppi[5 * PP_LAYERS + l] = j * ng; }
        } else {
            const int m = slot == 2 ? 0 : slot - 3;
            if (m < 3) {
                // the three species counts are wave-uniform scalar reads and a lane picks its own (R.minor[m] with
                // the lane's m is a gather through a per-lane pointer, which the segment loop then keeps in scratch)
                const int msp = m == 0 ? msp0 : (m == 1 ? msp1 : msp2);
                const bool bin = msp > 1;
                ppd[(14 + m) * PP_LAYERS + l] = bin ? f : 0.;
                ppi[(2 + m) * PP_LAYERS + l] = (c.indm[k] * msp + (bin ? j : 0)) * ng;
            }
        }
    }
}

// The item of a binary-species segment: gaseous optical depth and Planck fraction of layer k (pre-pass slot l) for
// this lane's g-point.  Indices, fractions and stencil weights come from the pre-pass; the lane gathers and sums.
__device__ __forceinline__ void gas_item_binary(const LwDeviceTables &tb, const LCol &c, const LwRegime &R, const int ng,
                                                const int off_self, const int off_for, const LaneG &G,
                                                const double *ppd, const int *ppi, const int k, const int l,
                                                const double *cont, double2 &taug, double2 &frac)
{
    const uint32_t cf = R.cfg;
    const int nl = c.nl, ig = G.ig;
    const int nsp = (cf >> LC_NSP_SHIFT) & 15;
    const bool edges = cf & LC_EDGES;
    const int ts = nsp * ng;
    const int fmode = (cf >> LC_FRAC_SHIFT) & 3;
    const double2 zero = make_double2(0., 0.);
    // ---- pass 1: table reads ----
    double2 v[2][6];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const double *Kp = tb.K + R.off_major + ig + ppi[m * PP_LAYERS + l];
        v[m][0] = ld2(Kp); v[m][1] = ld2(Kp + ng); v[m][3] = ld2(Kp + ts); v[m][4] = ld2(Kp + ts + ng);
        v[m][2] = zero; v[m][5] = zero;
        if (edges) { v[m][2] = ld2(Kp + 2 * ng); v[m][5] = ld2(Kp + ts + 2 * ng); }   // (weight 0 away from the ends)
    }
    double2 s0 = zero, s1 = zero, f0 = zero, f1 = zero;
    if (cf & (LC_SELF | LC_FOR)) {
        if (cont) {   // staged by this wave for the segment (lw_gas_kernel)
            const double *Ff = cont + (CONT_SELF_ROWS + c.indf[k]) * CONT_STRIDE + ig;
            f0 = ld2(Ff); f1 = ld2(Ff + CONT_STRIDE);
            if (cf & LC_SELF) {
                const double *Sf = cont + c.inds[k] * CONT_STRIDE + ig;
                s0 = ld2(Sf); s1 = ld2(Sf + CONT_STRIDE);
            }
        } else {
            const double *Ff = tb.K + off_for + ig + (size_t)c.indf[k] * ng;
            f0 = ld2(Ff); f1 = ld2(Ff + ng);
            if (cf & LC_SELF) {
                const double *Sf = tb.K + off_self + ig + (size_t)c.inds[k] * ng;
                s0 = ld2(Sf); s1 = ld2(Sf + ng);
            }
        }
    }
    double2 ma0[3], ma1[3], mb0[3], mb1[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const LwMinor &M = R.minor[m];
        ma0[m] = ma1[m] = mb0[m] = mb1[m] = zero;
        if (M.slot != LS_ZERO) {
            const double *Km = tb.K + M.off + ig + ppi[(2 + m) * PP_LAYERS + l];
            const int tm = M.nsp * ng;
            ma0[m] = ld2(Km); mb0[m] = ld2(Km + tm);
            if (M.nsp > 1) { ma1[m] = ld2(Km + ng); mb1[m] = ld2(Km + tm + ng); }
        }
    }
    double2 fr0 = zero, fr1 = zero;
    if (fmode == 2) {
        const double *Fp = tb.K + R.off_frac + ig + ppi[5 * PP_LAYERS + l];
        fr0 = ld2(Fp); fr1 = ld2(Fp + ng);
    }
    // ---- pass 2: sums ----
    double2 maj[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const double *d = ppd + (m * 7) * PP_LAYERS + l;
        double2 sum = mul2(d[0 * PP_LAYERS], v[m][0]);
        sum = mad2(d[1 * PP_LAYERS], v[m][1], sum);
        sum = mad2(d[3 * PP_LAYERS], v[m][3], sum);
        sum = mad2(d[4 * PP_LAYERS], v[m][4], sum);
        if (edges) { sum = mad2(d[2 * PP_LAYERS], v[m][2], sum); sum = mad2(d[5 * PP_LAYERS], v[m][5], sum); }
        maj[m] = sum;
    }
    double2 tau = mad2(ppd[13 * PP_LAYERS + l], maj[1], mul2(ppd[6 * PP_LAYERS + l], maj[0]));
    // continua (selffac / forfac already carry colh2o, LW:3806-3807)
    if (cf & (LC_SELF | LC_FOR)) {
        if (cf & LC_SELF) tau = mad2(c.selffac[k], mad2(c.selffrac[k], sub2(s1, s0), s0), tau);
        if (cf & LC_FOR) tau = mad2(c.forfac[k], mad2(c.forfrac[k], sub2(f1, f0), f0), tau);
    }
    // minor species, e.g. LW:5297-5301, 5877-5878
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const LwMinor &M = R.minor[m];
        if (M.slot != LS_ZERO) {
            double2 m1 = ma0[m], m2 = mb0[m];
            if (M.nsp > 1) {
                const double fm = ppd[(14 + m) * PP_LAYERS + l];
                m1 = mad2(fm, sub2(ma1[m], ma0[m]), ma0[m]);
                m2 = mad2(fm, sub2(mb1[m], mb0[m]), mb0[m]);
            }
            tau = mad2(c.mul[M.slot * nl + k], mad2(c.minorfrac[k], sub2(m2, m1), m1), tau);
        }
    }
    const double h0 = c.mul[R.halo_slot[0] * nl + k], h1 = c.mul[R.halo_slot[1] * nl + k];
    taug = make_double2((tau.x + h0 * G.halo0.x + h1 * G.halo1.x) * G.postscale.x,
                        (tau.y + h0 * G.halo0.y + h1 * G.halo1.y) * G.postscale.y);
    frac = fmode == 2 ? mad2(ppd[17 * PP_LAYERS + l], sub2(fr1, fr0), fr0) : (fmode == 1 ? G.fracconst : zero);
}

// One entry of the pair table (lw_fill_pairs): what layer_src needs of exp_tbl(itr) / tfn_tbl(itr) -- the absorptivity
// 1 - exp_tbl(itr) and tfn_tbl(itr), formed by the arithmetic the kernels use when they re-evaluate an entry from its
// index, so a table read and a re-evaluation give the same bits.
__device__ __forceinline__ double2 lw_pair_entry(double bpade, int itr)
{
    const bool last = itr >= LW_NTBL;
    itr = last ? LW_NTBL - 1 : itr;
    const double fi = (double)itr;
    const double tau_t = (bpade * fi) * frcp(10000.0 - fi);
    double transc = fmax(exp_neg(tau_t), 1.e-20);
    const double omt = 1. - transc;
    double tausfac_t = tau_t < 0.06 ? tau_t * (1. / 6.) : 1. - 2. * ((omt - tau_t * transc) * frcp1(tau_t * omt));
    transc = last ? 1.e-20 : transc;
    tausfac_t = last ? 1.0 : tausfac_t;
    return make_double2(1. - transc, tausfac_t);
}

// The copy of the pair table in LDS starts at entry LW_PAIR_FIRST: an optical depth above 0.06 -- the only kind that
// reads the table -- has an index of 164 or more (0.06 / (bpade + 0.06) * 10000 = 164.07); smaller ones take the
// expansion.  (0 <= itr <= 10000 for every finite optical depth >= 0; anything else reads the last entry.)
constexpr int LW_PAIR_FIRST = 164, LW_PAIR_COUNT = LW_NTBL + 1 - LW_PAIR_FIRST;
__device__ __forceinline__ int lw_pair_slot(int itr)
{
    itr = itr < LW_PAIR_FIRST ? LW_PAIR_FIRST : itr;
    return ((unsigned)itr > (unsigned)LW_NTBL ? LW_NTBL : itr) - LW_PAIR_FIRST;
}

// absorptivity and Planck source terms of one layer for one g-point, LW:3284-3296
// ltbl: the pair table in LDS (lw_rt_fast_kernel, cloud-free instantiation) or nullptr (compile-time in either case)
template <bool USE_TBL>
__device__ __forceinline__ void layer_src(const LwDeviceTables &tb, const double2 *ltbl, double tau, double secdiff,
                                          double plfrac, double blay, double dplankup, double dplankdn, double &atrans,
                                          double &bbd, double &bbu)
{
    double odepth = secdiff * tau;
    odepth = odepth < 0.0 ? 0.0 : odepth;
    double tblind = fdiv(odepth, tb.bpade + odepth);
    int itr = (int)(10000.0 * tblind + 0.5);
    const bool small = odepth <= 0.06;
    // The reference reads exp_tbl(itr) and tfn_tbl(itr) (10 001-entry tables, LW:7940-7957).  A 64-lane gather of
    // random entries from memory costs the vector cache more than re-evaluating the entry costs the SIMD, so the
    // kernels either read the pair from a copy of the table in LDS (160 KB: a CU's whole LDS, which the sixteen-wave
    // workgroup of lw_rt_fast_kernel has to itself anyway) or re-evaluate the quantised value from its index: same itr,
    // the formulas of rrtmg_lw_ini with tfn = itr/10000 substituted (tau = bpade*itr/(10000-itr)) and the two quotients
    // of tfn_tbl, 1/tau - trans/(1-trans), taken over one common denominator.
    double2 e;
    if constexpr (USE_TBL) {
        e = ltbl[lw_pair_slot(itr)];
        asm("" : "+v"(e.x), "+v"(e.y));   // (one 16-byte read for every lane, not two reads and a branch on `small`)
    }
    else e = lw_pair_entry(tb.bpade, itr);
    atrans = small ? odepth - 0.5 * odepth * odepth : e.x;
    double tausfac = small ? 0.166667 * odepth : e.y;
    // small branch: plfrac*(blay + dplank*odepth_rec); table branch: plfrac*(blay + tausfac*dplank)
    bbd = small ? plfrac * (blay + dplankdn * tausfac) : plfrac * (blay + tausfac * dplankdn);
    bbu = small ? plfrac * (blay + dplankup * tausfac) : plfrac * (blay + tausfac * dplankup);
}

// One quantised entry of the reference's lookup tables (rrtmg_lw_ini, LW:7940-7957): tau_tbl (only where NEED_TAU),
// 1 - exp_tbl and tfn_tbl, read from the pair table in LDS (USE_TBL) or re-evaluated from the index as in layer_src.
template <bool USE_TBL, bool NEED_TAU>
__device__ __forceinline__ void lw_tbl_entry(const LwDeviceTables &tb, const double2 *ltbl, double od, double &tau_t,
                                             double &atr, double &tfn)
{
    const double tblind = fdiv(od, tb.bpade + od);
    int itr = (int)(10000.0 * tblind + 0.5);
    const bool last = itr >= LW_NTBL;
    if constexpr (USE_TBL) {
        const double2 e = ltbl[lw_pair_slot(itr)];
        atr = e.x; tfn = e.y;
        tau_t = 0.;
        if (NEED_TAU) {
            const double fi = (double)(last ? LW_NTBL - 1 : itr);
            tau_t = last ? 1.e10 : (tb.bpade * fi) * frcp(10000.0 - fi);
        }
    } else {
        itr = last ? LW_NTBL - 1 : itr;
        const double fi = (double)itr;
        tau_t = (tb.bpade * fi) * frcp(10000.0 - fi);
        double transc = fmax(exp_neg(tau_t), 1.e-20);
        const double omt = 1. - transc;
        tfn = tau_t < 0.06 ? tau_t * (1. / 6.) : 1. - 2. * ((omt - tau_t * transc) * frcp1(tau_t * omt));
        tau_t = last ? 1.e10 : tau_t;
        transc = last ? 1.e-20 : transc;
        tfn = last ? 1.0 : tfn;
        atr = 1. - transc;
    }
}

// A layer in which some sub-column is cloudy: the three optical-depth regimes of rtrnmc, LW:3216-3282.
// odcld = secdiff * cloud optical depth of this sub-column (0 where it is clear).
struct LwCldSrc { double atrans, atot, bbd, bbdtot, bbugas, bbutot; };
template <bool USE_TBL>
__device__ __forceinline__ void layer_src_cld(const LwDeviceTables &tb, const double2 *ltbl, double tau, double secdiff,
                                              double plfrac, double blay, double dplankup, double dplankdn, double odcld,
                                              LwCldSrc &o)
{
    const double rec_6 = 0.166667;
    double odepth = secdiff * tau;
    odepth = odepth < 0.0 ? 0.0 : odepth;
    double odtot = odepth + odcld;
    // Each case yields four numbers -- the two absorptivities and the two factors of the Planck differences -- in
    // local scalars, and the six results are formed from them once, behind the cases.  (Assigning the fields of `o` in
    // every case made the compiler merge the cases' stores into one store through a *selected address*; the record
    // then lived in scratch memory, with vector-memory stores and loads in the middle of the sweeps.)
    double atrans, atot, fgas, ftot;
    if (odtot < 0.06) {
        atrans = odepth - 0.5 * odepth * odepth;
        fgas = rec_6 * odepth;
        atot = odtot - 0.5 * odtot * odtot;
        ftot = rec_6 * odtot;
    } else if (odepth <= 0.06) {
        atrans = odepth - 0.5 * odepth * odepth;
        fgas = rec_6 * odepth;
        double tt;
        lw_tbl_entry<USE_TBL, false>(tb, ltbl, odtot, tt, atot, ftot);
    } else {
        double tgas;
        lw_tbl_entry<USE_TBL, true>(tb, ltbl, odepth, tgas, atrans, fgas);
        odtot = tgas + odcld;            // odepth = tau_tbl(itgas), LW:3262
        double tt;
        lw_tbl_entry<USE_TBL, false>(tb, ltbl, odtot, tt, atot, ftot);
    }
    o.atrans = atrans;
    o.atot = atot;
    o.bbdtot = plfrac * (blay + ftot * dplankdn);
    o.bbd = plfrac * (blay + fgas * dplankdn);
    o.bbugas = plfrac * (blay + fgas * dplankup);
    o.bbutot = plfrac * (blay + ftot * dplankup);
}

// cloud absorption optical depth of a cloudy sub-column layer: cldprmc, LW:2800-2908 (inflag >= 2, liqflag 1,
// iceflag 3/4/5).  cb = cloud part of the column record; T = [2][16][58] absliq1, absice3.
// bad: 1 ice, 2 snow, 3 liquid size out of bounds (fatal in the reference).
__device__ __forceinline__ double lw_cloud_tau(const double *T, const double *cb, int nl, int band, int k, int iceflag,
                                               int &bad)
{
    const double cldmin = 1.e-20;
    const double clwp = cb[1 * nl + k], ciwp = cb[2 * nl + k], cswp = cb[3 * nl + k];
    if (!(ciwp + clwp + cswp >= cldmin)) return 0.;
    double abscoice = 0., abscosno = 0., abscoliq = 0.;
    if ((ciwp + cswp) != 0.0) {
        const int ix = (int)cb[8 * nl + k];
        if (ix == 0) { bad = 1; return 0.; }
        const double f = cb[5 * nl + k];
        const double *A = T + (1 * LW_NBND + band) * 58;
        abscoice = A[ix - 1] + f * (A[ix] - (A[ix - 1]));
    }
    if (cswp > 0.0 && iceflag == 5) {
        const int ix = (int)cb[9 * nl + k];
        if (ix == 0) { bad = 2; return 0.; }
        const double f = cb[6 * nl + k];
        const double *A = T + (1 * LW_NBND + band) * 58;
        abscosno = A[ix - 1] + f * (A[ix] - (A[ix - 1]));
    }
    if (clwp != 0.0) {
        const int ix = (int)cb[7 * nl + k];
        if (ix == 0) { bad = 3; return 0.; }
        const double f = cb[4 * nl + k];
        const double *A = T + (0 * LW_NBND + band) * 58;
        abscoliq = A[ix - 1] + f * (A[ix] - (A[ix - 1]));
    }
    return ciwp * abscoice + clwp * abscoliq + cswp * abscosno;
}

// pressure (mb) of level l (0-based, l = 0 surface), LW:12149-12152, 12207-12212
__device__ double plev_at(const LwKernelArgs &a, size_t base3, size_t sk, int l, int nk, int nl)
{
    if (l <= nk) return ldr(a.p8w, base3 + (size_t)l * sk, a.real4) / 100.;
    if (l == nl) return 0.00;
    double p = ldr(a.p8w, base3 + (size_t)nk * sk, a.real4) / 100.;
    for (int i = nk; i < l; ++i) p = p - 4.;   // deltap = 4 mb
    return p;
}
// temperature of the 60-level mean profile interpolated to pressure p, LW:12219-12241
__device__ double varint_at(double p)
{
    // first level (from the surface) of the mean profile below p; without the search loop's early exit
    int first = 60;
#pragma unroll
    for (int j = 59; j >= 1; --j) first = c_pprof[j] < p ? j : first;
    const int klev = c_pprof[59] < p ? first : 60;
    double vark = c_tprof[klev - 1], vark1 = klev != 60 ? c_tprof[klev] : vark;
    double wght = klev != 60 ? (p - c_pprof[klev - 1]) / (c_pprof[klev] - c_pprof[klev - 1]) : 0.0;
    return wght * (vark1 - vark) + vark;
}
__device__ double tlev_at(const LwKernelArgs &a, size_t base3, size_t sk, int l, int nk, int nl)
{
    if (l <= nk - 1) return ldr(a.t8w, base3 + (size_t)l * sk, a.real4);
    // tlev(L) = varint(L) + (tlev(kte) - varint(kte)) for L >= kte+1, LW:12244-12247
    double tk = ldr(a.t8w, base3 + (size_t)(nk - 1) * sk, a.real4);
    double vk = varint_at(plev_at(a, base3, sk, nk - 1, nk, nl));
    return varint_at(plev_at(a, base3, sk, l, nk, nl)) + (tk - vk);
}
__device__ double o3clim(double plev_b, double plev_t)
{   // O3DATA, LW:12806-12836
    double o3 = 0.;
    for (int jj = 0; jj < 31; ++jj) {
        double pb1 = (-(plev_b - c_lw_ppwrkh[jj]) >= 0.) ? 0. : plev_b - c_lw_ppwrkh[jj];
        double pb2 = (-(plev_b - c_lw_ppwrkh[jj + 1]) >= 0.) ? 0. : plev_b - c_lw_ppwrkh[jj + 1];
        double pt1 = (-(plev_t - c_lw_ppwrkh[jj]) >= 0.) ? 0. : plev_t - c_lw_ppwrkh[jj];
        double pt2 = (-(plev_t - c_lw_ppwrkh[jj + 1]) >= 0.) ? 0. : plev_t - c_lw_ppwrkh[jj + 1];
        o3 = o3 + (pb2 - pb1 - pt2 + pt1) * c_lw_o3ann[jj];
    }
    return o3 / (plev_b - plev_t);
}
// "too abundant to be minor" column adjustment, e.g. LW:5209-5216
__device__ __forceinline__ double adjcol(double col, double coldry, double chiref, double thr, double a0, double ex)
{
    double chi = fdiv(col, coldry);
    double rat = fdiv(1.e20 * chi, chiref);
    double out = col;
    if (rat > thr) {   // rare (a species several times more abundant than in the reference atmosphere): pow only there
        const double adjfac = a0 + pow(rat - a0, ex);
        out = adjfac * chiref * coldry * 1.e-20;
    }
    return out;
}

}  // namespace

// Where a lane of the transfer kernels finds the Planck fraction of its g-point in layer k: in the scratch if the
// layer's regime (lower / upper atmosphere, decided by the column's laytrop) interpolates it, else the descriptor's
// constant (the gas kernel does not store those).
struct FracSrc {
    double c_lo, c_up;      // the constant of the regime (0 where the fraction is stored, or where it is zero)
    bool st_lo, st_up;      // stored in the scratch (fdel doubles behind the lane's taug slot)
    int ltrop;
    ptrdiff_t fdel;         // (the same for every lane: scalar registers)
};
__device__ __forceinline__ FracSrc frac_src(const LwGDesc *gd, double laytrop, ptrdiff_t fdel)
{
    const int mlo = (gd->lo.cfg >> LC_FRAC_SHIFT) & 3, mup = (gd->up.cfg >> LC_FRAC_SHIFT) & 3;
    return FracSrc{mlo == 1 ? gd->lo.fracconst : 0., mup == 1 ? gd->up.fracconst : 0., mlo == 2, mup == 2, (int)laytrop, fdel};
}
// The Planck fraction of layer k from a pointer to the lane's taug slot of that layer.  The load is unconditional and
// from the scratch in every lane, and its value is used in every lane (times 0 where the fraction is a constant: those
// lanes read their finite taug slot again -- same line, no traffic).  A conditional load came out of the compiler as a
// FLAT load from "the scratch or a stack slot holding the constant", and with a flat load in flight every wait of the
// loop is vmcnt(0): the layer fetched one iteration ahead was waited for right after its issue (r03: the transfer
// kernel "waits on its reads").
// (in two halves, so that a loop can issue the read ahead and touch the value only when the layer's turn comes)
__device__ __forceinline__ double frac_read(const FracSrc &F, const double *tgk, int k)
{
    return tgk[(k < F.ltrop ? F.st_lo : F.st_up) ? F.fdel : (ptrdiff_t)0];
}
__device__ __forceinline__ double frac_value(const FracSrc &F, double v, int k)
{
    const bool lower = k < F.ltrop;
    return __builtin_fma(v, (lower ? F.st_lo : F.st_up) ? 1. : 0., lower ? F.c_lo : F.c_up);
}

// column slot of this workgroup within the batch: consecutive slots stay on one XCD (its own L2)
// nwg: the workgroups of the launch that have work (all of them, or what the call's column list holds, cloud_lists.h);
// -1 for the others
__device__ __forceinline__ int batch_slot(const int nwg)
{
    const int b = blockIdx.x;
    if (b >= nwg) return -1;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
// columns of this launch: the batch's own count, or what the call's column list holds beyond col0
__device__ __forceinline__ int batch_columns(const LwKernelArgs &a)
{
    int n = a.nbatch;
    if (a.colmap) {
        const int left = *a.colcount - a.col0;
        n = left < n ? left : n;
    }
    return n;
}

// the gas optics leave the chip once and are read back by another kernel: streaming (non-temporal) stores, so that the
// 46 GB of a launch do not push the spectroscopic tables out of L2 on their way
__device__ __forceinline__ void st2_stream(double *p, const double2 v)
{
    __builtin_nontemporal_store(v.x, p);
    __builtin_nontemporal_store(v.y, p + 1);
}
// Kernel 1 of 2: column set-up (inatm + setcoef, lanes = layers) and gaseous optical depths / Planck
// fractions of every (layer, g-point) (taumol, lanes = g-points) -> scratch.
__global__ void __launch_bounds__(LW_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) lw_gas_kernel(const LwKernelArgs a)
{
    extern __shared__ double lds[];
    const LwDeviceTables &tb = a.tab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nk = a.nk, nl = a.nlayers;
    const int slot = batch_slot(batch_columns(a));
    if (slot < 0) return;   // (uniform over the workgroup, in front of its first barrier)
    int col = a.col0 + slot;
    if (a.colmap) col = a.colmap[col];   // entry of the call's column list (cloud_lists.h)
    double *rec = a.rec + (size_t)slot * lw_rec_doubles(nl);
    LCol c = carve_gas(lds, rec, nl, a.stage_cont != 0);
    const size_t nl4 = lw_rec_q4_layers(nl);

    const int it = col % a.ni_t, jt = col / a.ni_t;
    const size_t i0 = (size_t)(a.its - a.ims + it), j0 = (size_t)(a.jts - a.jms + jt), k0 = (size_t)(a.kts - a.kms);
    const size_t i2 = j0 * a.ni_m + i0;
    const size_t sk = a.ni_m, sj = (size_t)a.ni_m * a.nk_m;
    const size_t base3 = j0 * sj + k0 * sk + i0;
    const int r4 = a.real4;
    const double tbound = ldr(a.tsk, i2, r4);
    const double semiss = ldr(a.emiss, i2, r4);   // same emissivity in all 16 bands, LW:12296-12298

    // ======================= prologue =======================
    // three phases between two barriers, so that no wave sits through the whole of it:
    //   A  wave 0, lanes = levels: level pressures / temperatures and the layers' p, T, qv -> LDS (one round trip to
    //      memory); waves 1-2 meanwhile: aerosol optical depths (memory only)
    //   B  wave 0, lanes = layers: inatm + setcoef; waves 1-2: Planck functions of the layers and levels
    //   C  column sums (precipitable water, aerosol optical depth) by 16 lanes each of waves 0 and 1
#ifdef LW_EXP_CLOCK   // cycle stamps of the phases, printed by three workgroups (profiling aid)
    long long tstamp[8];
#define LW_STAMP(i) tstamp[i] = clock64()
#else
#define LW_STAMP(i)
#endif
    __shared__ int s_laytrop, s_nextseg;
    if (tid == 0) { s_laytrop = 0; s_nextseg = 0; }
    LW_STAMP(0);
    if (wave == 2) {
        // the ozone climatology of the layers (O3DATA: a 31-step sum per layer, a third of the instructions of phase B)
        // off wave 0's path: this wave forms the level pressures a second time and parks the result in the upper half
        // of wrk, which phase B overwrites only after it has read the value
        for (int k = lane; k < nl; k += 64) {
            const int lb = k < nk ? k : nk, lu = k + 1 < nk ? k + 1 : nk;
            const double pb8 = ldr(a.p8w, base3 + (size_t)lb * sk, r4), pu8 = ldr(a.p8w, base3 + (size_t)lu * sk, r4);
            double pb = fdiv(pb8, 100.), pu = fdiv(pu8, 100.);   // (lb = lu = nk above the model top: p8w(kte+1) / 100)
            if (k > nk) pb = pb - 4. * (double)(k - nk);
            if (k + 1 > nk) pu = k + 1 == nl ? 0.00 : pu - 4. * (double)(k + 1 - nk);
            c.wrk[nl + k] = o3clim(pb, pu);
        }
    } else if (wave == 0) {
        // every load of the phase goes out first, at clamped indices (a buffer level reads the model top's values again;
        // written with the branches of plev_at / tlev_at, the loads of the buffer levels waited for those of the model
        // levels: three to four trips to memory instead of one), then the levels are formed as LW:12207-12247 forms them
        for (int l = lane; l <= nl; l += 64) {
            const int lp = l < nk ? l : nk, lt = l < nk - 1 ? l : nk - 1;
            const double p8 = ldr(a.p8w, base3 + (size_t)lp * sk, r4), t8 = ldr(a.t8w, base3 + (size_t)lt * sk, r4);
            const double p8top = ldr(a.p8w, base3 + (size_t)nk * sk, r4), p8sub = ldr(a.p8w, base3 + (size_t)(nk - 1) * sk, r4);
            const double t3 = ldr(a.t3d, base3 + (size_t)lt * sk, r4), p3 = ldr(a.p3d, base3 + (size_t)lt * sk, r4),
                         q3 = ldr(a.qv3d, base3 + (size_t)lt * sk, r4);
            // plev: the model's levels, then steps of deltap = 4 mb down from the model top (the reference subtracts 4
            // level by level; every partial result is exact, so one subtraction gives the same number), 0 at the top
            double pz = fdiv(p8, 100.);
            if (l > nk) pz = l == nl ? 0.00 : fdiv(p8top, 100.) - 4. * (double)(l - nk);
            // tlev(L) = varint(L) + (tlev(kte) - varint(kte)) for L >= kte+1, LW:12244-12247
            double tz = t8;
            if (l > nk - 1) tz = varint_at(pz) + (t8 - varint_at(fdiv(p8sub, 100.)));
            c.pz[l] = pz;
            c.tz[l] = tz;
            if (l < nl) { c.tav[l] = t3; c.pavel[l] = fdiv(p3, 100.); c.wrk[l] = q3; }
        }
    } else if (wave == 1) {
        // aerosol optical depth per band, LW:12576-12604 (zero in the buffer layers): lanes = layers, the sixteen
        // bands of a layer by one lane (bands 1 and 16, which gate the copy, are read once; the REAL width is decided
        // outside the loads so that the sixteen of a layer go out together)
        double asum[LW_NBND];   // this lane's share of the column sums (the guard of LW:12607-12615)
#pragma unroll
        for (int nb = 0; nb < LW_NBND; ++nb) asum[nb] = 0.;
        for (int k = lane; k < nl; k += 64) {
            double t[LW_NBND];
#pragma unroll
            for (int nb = 0; nb < LW_NBND; ++nb) t[nb] = 0.;
            if (a.aer_ra_feedback == 1 && k < nk) {
                const size_t q = base3 + (size_t)k * sk;
                if (r4) {
#pragma unroll
                    for (int nb = 0; nb < LW_NBND; ++nb) t[nb] = (double)reinterpret_cast<const float *>(a.tauaerlw[nb])[q];
                } else {
#pragma unroll
                    for (int nb = 0; nb < LW_NBND; ++nb) t[nb] = reinterpret_cast<const double *>(a.tauaerlw[nb])[q];
                }
            }
            const bool on = t[0] > 1.e-9 && t[15] > 1.e-9;
#pragma unroll
            for (int nb = 0; nb < LW_NBND; ++nb) {
                const double v = on ? t[nb] : 0.;
                __builtin_nontemporal_store(v, &c.q4[((size_t)nb * nl4 + k) * 4 + 3]);
                asum[nb] += v;
            }
        }
        // negative column aerosol optical depth is fatal, LW:12607-12615: the column sums of the 16 bands from the
        // registers that fetched them (a sum of the same numbers in another order decides the same way unless it is
        // zero to rounding)
        if (a.aer_ra_feedback == 1) {
#pragma unroll
            for (int nb = 0; nb < LW_NBND; ++nb) {
                double v = asum[nb];
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                if (lane == 0 && v < 0.) atomicMax(a.status, 1 + col);
            }
        }
    }
    LW_STAMP(1);
    __syncthreads();
    LW_STAMP(2);
    int nlower = 0;
    if (wave == 0)
    for (int k = lane; k < nl; k += 64) {
        const double plev_b = c.pz[k], plev_t = c.pz[k + 1];
        const double pavel = k < nk ? c.pavel[k] : 0.5 * (plev_b + plev_t);
        const double tz_t = c.tz[k + 1];
        const double tavel = k <= nk - 2 ? c.tav[k] : 0.5 * (tz_t + c.tz[k]);   // LW:12246 (also overwrites tlay(kte))
        double qv = c.wrk[k];
        qv = fmax(0., qv);
        qv = fmax(qv, 1.E-12);
        const double amdw = 1.607793, amdo = 0.603461;
        const double h2ovmr = qv * amdw;
        const double o3mmr = c.wrk[nl + k];   // o3clim(plev_b, plev_t), wave 2 in phase A
        double o3vmr = o3mmr * amdo;
        if (a.o33d && a.o3input == 2) {   // LW:12268-12280
            if (k < nk) o3vmr = ldr(a.o33d, base3 + (size_t)k * sk, r4);
            else {
                const double pb = plev_at(a, base3, sk, nk - 1, nk, nl), pt = plev_at(a, base3, sk, nk, nk, nl);
                double v = ldr(a.o33d, base3 + (size_t)(nk - 1) * sk, r4) - o3clim(pb, pt) * amdo + o3mmr * amdo;
                o3vmr = v <= 0. ? o3mmr * amdo : v;
            }
        }
        // fixed trace gases, LW:11720-11738
        const double co2 = 379.e-6, ch4 = 1774.e-9, n2o = 319.e-9, cfc11 = 0.251e-9, cfc12 = 0.538e-9,
                     cfc22 = 0.169e-9, ccl4 = 0.093e-9, o2 = 0.209488;
        // inatm, LW:11262-11300
        const double avogad = 6.02214199e+23, grav = 9.8066, amd = 28.9660, amw = 18.0160;
        const double amm = (1. - h2ovmr) * amd + h2ovmr * amw;
        const double coldry = (plev_b - plev_t) * 1.e3 * avogad / (1.e2 * grav * amm * (1. + h2ovmr));
        double summol = 0.0;
        summol = summol + co2; summol = summol + o3vmr; summol = summol + n2o; summol = summol + 0.0;
        summol = summol + ch4; summol = summol + o2;
        const double wbrodl = coldry * (1. - summol);
        const double wh2o = coldry * h2ovmr, wco2 = coldry * co2, wo3 = coldry * o3vmr, wn2o = coldry * n2o,
                     wco = coldry * 0.0, wch4 = coldry * ch4, wo2 = coldry * o2;
        c.wrk[k] = coldry + wh2o;
        c.wrk[nl + k] = wh2o;
        c.mul[LS_WX1 * nl + k] = coldry * ccl4 * 1.e-20;
        c.mul[LS_WX2 * nl + k] = coldry * cfc11 * 1.e-20;
        c.mul[LS_WX3 * nl + k] = coldry * cfc12 * 1.e-20;
        c.mul[LS_WX4 * nl + k] = coldry * cfc22 * 1.e-20;
        // setcoef: interpolation factors, LW:3647-3808 (the Planck functions are with waves 1-2 below)
        const double stpfac = 296. / 1013.;
        const double plog = log(pavel);
        int jp = (int)(36. - 5 * (plog + 0.04));
        jp = jp < 1 ? 1 : (jp > 58 ? 58 : jp);
        const int jp1 = jp + 1;
        const double fp = 5. * (c_lw_preflog[jp - 1] - plog);
        const double d0 = (tavel - c_lw_tref[jp - 1]) / 15.;
        int jtt = (int)(3. + d0);
        jtt = jtt < 1 ? 1 : (jtt > 4 ? 4 : jtt);
        const double ft = d0 - (double)(jtt - 3);
        const double d1 = (tavel - c_lw_tref[jp1 - 1]) / 15.;
        int jt1 = (int)(3. + d1);
        jt1 = jt1 < 1 ? 1 : (jt1 > 4 ? 4 : jt1);
        const double ft1 = d1 - (double)(jt1 - 3);
        const double water = wh2o / coldry;
        const double scalefac = pavel * stpfac / tavel;
        const bool lower = plog > 4.56;
        nlower += lower ? 1 : 0;
        double forfac = scalefac / (1. + water), forfrac, selffac = water * forfac, selffrac = 0.;
        int indfor, indself = 1;
        if (lower) {
            double factor = (332.0 - tavel) / 36.0;
            int ifac = (int)factor;
            indfor = ifac > 1 ? (ifac < 2 ? ifac : 2) : 1;
            forfrac = factor - (double)indfor;
            factor = (tavel - 188.0) / 7.2;
            int is = (int)factor - 7;
            indself = is > 1 ? (is < 9 ? is : 9) : 1;
            selffrac = factor - (double)(indself + 7);
        } else {
            double factor = (tavel - 188.0) / 36.0;
            indfor = 3;
            forfrac = factor - 1.0;
        }
        const double scaleminor = pavel / tavel;
        const double scaleminorn2 = (pavel / tavel) * (wbrodl / (coldry + wh2o));
        double factor = (tavel - 180.8) / 7.2;
        int im = (int)factor;
        const int indminor = im > 1 ? (im < 18 ? im : 18) : 1;
        const double minorfrac = factor - (double)indminor;
        const double *X = tb.chi_mls;   // [7][59]
#define CHI(m, l) X[((m)-1) * 59 + (l)-1]
        // chi_mls ratios at jp and jp + 1 from the tabulated quotients (lw_tables.hip: [LR_*][59] behind the 7 x 59)
        // all fourteen reads first (left to itself the compiler serialises them, read -> wait -> LDS store, to save
        // registers the set-up has to spare)
        double ra[6], rb[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) { ra[r] = X[7 * 59 + r * 59 + jp - 1]; rb[r] = X[7 * 59 + r * 59 + jp]; }
        double chin2o = CHI(4, jp + 1), chico2 = CHI(2, jp + 1);
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(rb[0]),
                          "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(chin2o), "+v"(chico2));
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            c.rat[(r * 2 + 0) * nl + k] = ra[r];
            c.rat[(r * 2 + 1) * nl + k] = rb[r];
        }
        c.rat[(LR_ONE * 2 + 0) * nl + k] = 1.;
        c.rat[(LR_ONE * 2 + 1) * nl + k] = 1.;
        const double colh2o = 1.e-20 * wh2o;
        double colco2 = 1.e-20 * wco2, colo3 = 1.e-20 * wo3, coln2o = 1.e-20 * wn2o, colco = 1.e-20 * wco,
               colch4 = 1.e-20 * wch4;
        const double colo2 = 1.e-20 * wo2;
        if (colco2 == 0.) colco2 = 1.e-32 * coldry;
        if (colo3 == 0.) colo3 = 1.e-32 * coldry;
        if (coln2o == 0.) coln2o = 1.e-32 * coldry;
        if (colco == 0.) colco = 1.e-32 * coldry;
        if (colch4 == 0.) colch4 = 1.e-32 * coldry;
        const double colbrd = 1.e-20 * wbrodl;
        c.mul[LS_H2O * nl + k] = colh2o; c.mul[LS_CO2 * nl + k] = colco2; c.mul[LS_O3 * nl + k] = colo3;
        c.mul[LS_N2O * nl + k] = coln2o; c.mul[LS_CO * nl + k] = colco; c.mul[LS_CH4 * nl + k] = colch4;
        c.mul[LS_O2 * nl + k] = colo2; c.mul[LS_BRD * nl + k] = colbrd; c.mul[LS_ZERO * nl + k] = 0.;
        c.mul[LS_ADJ_N2O * nl + k] = adjcol(coln2o, coldry, chin2o, 1.5, 0.5, 0.65);
        c.mul[LS_ADJ_CO2_6 * nl + k] = adjcol(colco2, coldry, chico2, 3.0, 2.0, 0.77);
        c.mul[LS_ADJ_CO2_7L * nl + k] = adjcol(colco2, coldry, chico2, 3.0, 3.0, 0.79);
        c.mul[LS_ADJ_CO2_7U * nl + k] = adjcol(colco2, coldry, chico2, 3.0, 2.0, 0.79);
        c.mul[LS_ADJ_CO2_8 * nl + k] = adjcol(colco2, coldry, chico2, 3.0, 2.0, 0.65);
        c.mul[LS_ADJ_CO2_13 * nl + k] = adjcol(colco2, coldry, 3.55e-4, 3.0, 2.0, 0.68);
        c.mul[LS_SCALEN2A * nl + k] = colbrd * scaleminorn2;
        c.mul[LS_SCALEN2B * nl + k] = colbrd * scaleminor;
        c.mul[LS_SCALEO2 * nl + k] = colo2 * scaleminor;
#undef CHI
        const double compfp = 1. - fp;
        c.fac10[k] = compfp * ft;
        c.fac00[k] = compfp * (1. - ft);
        c.fac11[k] = fp * ft1;
        c.fac01[k] = fp * (1. - ft1);
        c.selffac[k] = colh2o * selffac;
        c.forfac[k] = colh2o * forfac;
        c.selffrac[k] = selffrac; c.forfrac[k] = forfrac; c.minorfrac[k] = minorfrac; c.pavel[k] = pavel;
        c.rb0[k] = lower ? ((jp - 1) * 5 + (jtt - 1)) : ((jp - 13) * 5 + (jtt - 1));
        c.rb1[k] = lower ? (jp * 5 + (jt1 - 1)) : ((jp - 12) * 5 + (jt1 - 1));
        c.inds[k] = indself - 1; c.indf[k] = indfor - 1; c.indm[k] = indminor - 1;
        if (a.icloud != 0) {
            // hydrometeors -> in-cloud water paths and effective sizes, LW:11921-12140 and 12335-12553 (the block the
            // SW wrapper has too; relcalc / reicalc see tlay with the top model layer already overwritten); written
            // to the cloud part of the column record for lw_rt_kernel.  Buffer layers carry no cloud.
            double cf = 0., clwp = 0., ciwp = 0., cswp = 0., reliq = 10., reice = 10., resno = 10.;
            int iceflg = 3;
            if (k < nk) {
                const size_t q = base3 + (size_t)k * sk;
                const double t1d = ldr(a.t3d, q, r4);
                double qc = a.qc3d ? fmax(0., ldr(a.qc3d, q, r4)) : 0.;
                double qr = a.qr3d ? fmax(0., ldr(a.qr3d, q, r4)) : 0.;
                double qi = 0., qs = 0.;
                if (!a.qi3d && !a.warm_rain && t1d < 273.15) { qi = qc; qs = qr; qc = 0.; qr = 0.; }
                if (a.qi3d) qi = fmax(0., ldr(a.qi3d, q, r4));
                if (a.qs3d) qs = fmax(0., ldr(a.qs3d, q, r4));
                if (a.f_ice_phy) {        // "MP option 5": one frozen species carried as snow, LW:12003-12014
                    const double s3 = a.qs3d ? ldr(a.qs3d, q, r4) : 0.;
                    qi = fmax(0., 0.1 * s3); qs = 0.9 * s3;
                    qc = a.qc3d ? fmax(0., ldr(a.qc3d, q, r4)) : 0.;   // qc1d = qc3d again, LW:12009
                }
                if (a.fer_hires) {        // Ferrier-Aligo: qi3d holds all frozen water, LW:12016-12034
                    qi = a.qi3d ? fmax(0., ldr(a.qi3d, q, r4)) : 0.; qs = 0.0;
                    qc = a.qc3d ? fmax(0., ldr(a.qc3d, q, r4)) : 0.;   // LW:12029
                }
                cf = a.cldfra3d ? ldr(a.cldfra3d, q, r4) : 0.;
                const double xland = a.xland ? ldr(a.xland, i2, r4) : 1.;
                int inflg = 2;
                double recloud = 5.0, reice1 = 10., resnow1 = 10.0;
                if (a.has_reqc != 0) {
                    inflg = 3;
                    recloud = fmax(2.5, ldr(a.re_cloud, q, r4) * 1.E6);
                    if (recloud <= 2.5 && cf > 0. && (xland - 1.5) > 0.) recloud = 10.5;
                    else if (recloud <= 2.5 && cf > 0. && (xland - 1.5) < 0.) recloud = 7.5;
                }
                if (a.has_reqi != 0) {
                    inflg = 4; iceflg = 4;
                    reice1 = fmax(5., ldr(a.re_ice, q, r4) * 1.E6);
                    if (reice1 <= 5. && cf > 0.) {
                        int idx = (int)(t1d - 179.);
                        idx = idx < 1 ? 1 : (idx > 75 ? 75 : idx);
                        const double corr = t1d - (double)(int)t1d;
                        reice1 = c_lw_retab[idx - 1] * (1. - corr) + c_lw_retab[idx] * corr;
                        reice1 = fmax(reice1, 5.0);
                    }
                }
                if (a.has_reqs != 0) {
                    inflg = 5; iceflg = 5;
                    resnow1 = fmax(10., ldr(a.re_snow, q, r4) * 1.E6);
                }
                if (a.has_reqs == 0 && a.has_reqi != 0 && a.has_reqc != 0) {
                    inflg = 5; iceflg = 5;
                    resnow1 = fmax(10., ldr(a.re_ice, q, r4) * 1.E6);
                    qs = a.qi3d ? ldr(a.qi3d, q, r4) : 0.;
                    qi = 0.;
                    reice1 = 10.;
                }
                const double pdel = plev_b - plev_t;
                double gicewp = (qi + qs) * pdel * 100.0 / a.grav * 1000.0;
                const double gliqwp = qc * pdel * 100.0 / a.grav * 1000.0;
                if (iceflg >= 4) gicewp = qi * pdel * 100.0 / a.grav * 1000.0;
                ciwp = gicewp / fmax(0.01, cf);
                clwp = gliqwp / fmax(0.01, cf);
                if (iceflg == 5) {
                    double snow_mass_factor = 1.0;
                    if (resnow1 > 130.) {
                        snow_mass_factor = (130.0 / resnow1) * (130.0 / resnow1);
                        resnow1 = 130.0;
                    }
                    const double gsnowp = qs * snow_mass_factor * pdel * 100.0 / a.grav * 1000.0;
                    cswp = gsnowp / fmax(0.01, cf);
                }
                if (a.progn == 1) {
                    const double pi = 4. * atan(1.0), third = 1. / 3., rhoh2o = 1.e3, relconst = 3 / (4. * pi * rhoh2o);
                    reliq = 10.;
                    const double qnd = a.qndrop3d ? ldr(a.qndrop3d, q, r4) : 0.;
                    if (a.qndrop3d && qc * pdel > 3.e-5 && qnd > 1000.) {
                        reliq = pow(relconst * qc / qnd, third);
                        reliq = 1.1 * reliq;
                        reliq = reliq * 1.e6;
                        reliq = fmax(reliq, 4.);
                        reliq = fmin(reliq, 20.);
                    }
                } else {              // relcalc, LW:14403-14462
                    const double landm = 2. - xland, snowh = 0.001 * (a.snow ? ldr(a.snow, i2, r4) : 0.);
                    const double icefrac = a.xice ? ldr(a.xice, i2, r4) : 0.;
                    reliq = 8.0 + (14.0 - 8.0) * fmin(1.0, fmax(0.0, (273.16 - tavel) * 0.05));
                    reliq = reliq + (14.0 - reliq) * fmin(1.0, fmax(0.0, snowh * 10.));
                    reliq = reliq + (14.0 - reliq) * fmin(1.0, fmax(0.0, 1.0 - landm));
                    reliq = reliq + (14.0 - reliq) * fmin(1.0, fmax(0.0, icefrac));
                }
                {                     // reicalc, LW:14464-14491
                    int index = (int)(tavel - 179.);
                    index = index < 1 ? 1 : (index > 94 ? 94 : index);
                    const double corr = tavel - (double)(int)tavel;
                    reice = c_lw_retab[index - 1] * (1. - corr) + c_lw_retab[index] * corr;
                }
                if (inflg >= 3) reliq = recloud;
                if (iceflg >= 4) reice = reice1;
                if (iceflg == 3) { reice = reice * 1.0315; reice = fmin(140.0, reice); }
                if (a.iradius) {          // CAM-MG microphysics supplies the radii, LW:12473-12488
                    reice = (qi > 1.e-20 || qs > 1.e-20) ? ldr(a.iradius, q, r4) : 25.;
                    reice = fmax(5., fmin(140.0, reice));
                    reliq = qc > 1.e-20 ? ldr(a.lradius, q, r4) : 10.;
                    reliq = fmax(2.5, fmin(60.0, reliq));
                }
                if (inflg == 5) resno = resnow1; else { cswp = 0.0; resno = 10.0; }
            }
            double *cb = rec + lw_rec_base_doubles(nl);
            cb[0 * nl + k] = cf < 1.0e-20 ? 0. : cf;
            cb[1 * nl + k] = clwp; cb[2 * nl + k] = ciwp; cb[3 * nl + k] = cswp;
            {   // cldprmc table positions, LW:2830-2842, 2866-2872, 2890-2899; 0 = size out of bounds
                const bool ok = !(reice < 5.0 || reice > 140.0);
                const double factor = (reice - 2.) / 3.;
                int index = (int)factor;
                if (index == 46) index = 45;
                cb[8 * nl + k] = ok ? (double)index : 0.;
                cb[5 * nl + k] = factor - (double)index;
            }
            {
                const bool ok = !(resno < 5.0 || resno > 140.0);
                const double factor = (resno - 2.) / 3.;
                int index = (int)factor;
                if (index == 46) index = 45;
                cb[9 * nl + k] = ok ? (double)index : 0.;
                cb[6 * nl + k] = factor - (double)index;
            }
            {
                const bool ok = !(reliq < 2.5 || reliq > 60.);
                int index = (int)(reliq - 1.5);
                if (index == 0) index = 1;
                if (index == 58) index = 57;
                cb[7 * nl + k] = ok ? (double)index : 0.;
                cb[4 * nl + k] = reliq - 1.5 - (double)index;
            }
            if (k < 4) {
                // KISS seeds, LW:2330-2334 (fractional Pa of the four lowest layers; single precision in a REAL*4 build)
#pragma clang fp contract(off)
                int seed;
                if (r4) {
                    const float pl = reinterpret_cast<const float *>(a.p3d)[base3 + (size_t)k * sk] / 100.f;
                    const float pm = pl * 1.e2f;
                    seed = (int)((pm - (float)(int)pm) * (float)1000000000);
                } else {
                    const double pm = pavel * 1.e2;
                    seed = (int)((pm - (double)(int)pm) * 1000000000.);
                }
                cb[10 * nl + k] = (double)seed;
            }
        }
    }
    // laytrop: layers with ln(p) > 4.56 (LW:3673)
    if (nlower) atomicAdd(&s_laytrop, nlower);
    if (wave != 0) {
        // setcoef: Planck functions of the layers and of the levels above them in the 16 bands, LW:3583-3644;
        // (band, layer) items
        for (int item = tid - 64; item < LW_NBND * nl; item += LW_THREADS - 64) {
            const int ib = item / nl, k = item - ib * nl;
            const double *P = tb.totplnk + ib * 181;
            const double tz_b = c.tz[k], tz_t = c.tz[k + 1];
            const double tavel = k <= nk - 2 ? c.tav[k] : 0.5 * (tz_t + tz_b);
            int indlay = (int)(tavel - 159.);
            indlay = indlay < 1 ? 1 : (indlay > 180 ? 180 : indlay);
            const double tlayfrac = tavel - 159. - (double)indlay;
            const double dbdtlay = P[indlay] - P[indlay - 1];
            const double blay = P[indlay - 1] + tlayfrac * dbdtlay;
            // the level below and the level above (every inner level is formed twice, by its two layers, alike)
            double plev[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const double tz = m == 0 ? tz_b : tz_t;
                int indlev = (int)(tz - 159.);
                indlev = indlev < 1 ? 1 : (indlev > 180 ? 180 : indlev);
                const double tlevfrac = tz - 159. - (double)indlev;
                const double dbdtlev = P[indlev] - P[indlev - 1];
                plev[m] = P[indlev - 1] + tlevfrac * dbdtlev;
            }
            double *q = c.q4 + ((size_t)ib * nl4 + k) * 4;
            __builtin_nontemporal_store(blay, q); __builtin_nontemporal_store(plev[0] - blay, q + 1);
            __builtin_nontemporal_store(plev[1] - blay, q + 2);
        }
        if (tid >= 128 && tid < 128 + LW_NBND) {   // the boundary
            const int ib = tid - 128;
            const double *P = tb.totplnk + ib * 181;
            int indbound = (int)(tbound - 159.);
            indbound = indbound < 1 ? 1 : (indbound > 180 ? 180 : indbound);
            const double tbndfrac = tbound - 159. - (double)indbound;
            const double dbdtlev = P[indbound] - P[indbound - 1];
            c.plankbnd[ib] = semiss * (P[indbound - 1] + tbndfrac * dbdtlev);
        }
    }
    LW_STAMP(3);
    __syncthreads();
    LW_STAMP(4);
    const int ltrop = s_laytrop;
    if (tid < LW_NBND) {
        const int ib = tid;
        // precipitable water -> diffusivity angle of this band, LW:11307-11309 and LW:3142-3150 (every one of the
        // 16 threads forms the column sums in the reference's order)
        double amttl = 0.0, wvttl = 0.0;
#pragma unroll 8
        for (int k = 0; k < nl; ++k) { amttl = amttl + c.wrk[k]; wvttl = wvttl + c.wrk[nl + k]; }
        const double amd = 28.9660, amw = 18.0160, grav = 9.8066;
        const double wvsh = (amw * wvttl) / (amd * amttl);
        const double pwvcm = wvsh * (1.e3 * c.pz[0]) / (1.e2 * grav);
        double sd = 1.66;
        if (!(ib == 0 || ib == 3 || ib >= 9)) {
            sd = c_a0[ib] + c_a1[ib] * exp(c_a2[ib] * pwvcm);
            if (sd > 1.80) sd = 1.80;
            if (sd < 1.50) sd = 1.50;
        }
        c.secdiff[ib] = sd;
    }
    if (a.icloud != 0 && wave != 0) {
        // cloud absorption optical depth per (band, layer) for the transfer kernel (lw_kernel.h); the in-cloud state
        // was written to the record by wave 0 in front of the barrier above
        int iceflag = 3;
        if (a.has_reqi != 0) iceflag = 4;
        if (a.has_reqs != 0 || (a.has_reqi != 0 && a.has_reqc != 0)) iceflag = 5;
        const double *cbg = rec + lw_rec_base_doubles(nl);
        double *tc = rec + lw_rec_tc_offset(nl);
        for (int item = tid - 64; item < LW_NBND * (int)nl4; item += LW_THREADS - 64) {
            const int ib = item / (int)nl4, k = item - ib * (int)nl4;
            double v = 0.;
            if (k < nk) {
                int badk = 0;
                v = lw_cloud_tau(a.cldtab, cbg, nl, ib, k, iceflag, badk);
                if (badk) v = -(double)badk;
            }
            tc[item] = v;
        }
    }

    // pressure levels complete the record (heating-rate denominators of the transfer kernel)
    for (int l = tid; l <= nl; l += LW_THREADS) rec[lw_rec_base_doubles(nl) - (nl + 2) + l] = c.pz[l];
    if (tid == 0) rec[lw_rec_base_doubles(nl) - 1] = (double)ltrop;   // the transfer kernel needs the regime of a layer

    LW_STAMP(5);
    // ======================= gas optics: (band, regime) segments; lanes = (layer, g-point) =======================
    // A wave takes one segment at a time off the column's list (LDS counter; costliest first) and lays lpi = 64 / ng
    // layers x the band's ng g-points over its lanes: the regime is the same for all of them (scalar registers,
    // scalar branches) and 56 to 64 lanes work, where a fixed g-point per lane left 47 of 64 busy and made every
    // lane sit through the terms of every band in its wave.
    double *tg = a.taug + (size_t)slot * nl * LW_NGPT, *fr = a.frac + (size_t)slot * nl * LW_NGPT;
    for (;;) {
        int seg = 0;
        if (lane == 0) seg = atomicAdd(&s_nextseg, 1);
        seg = __builtin_amdgcn_readfirstlane(seg);
        if (seg >= 2 * LW_NBND) break;
        const LwSegRec &B = c_lwseg[seg];
        const LwRegime &R = B.R;
        const int upper = B.upper;
        const int kbeg = upper ? ltrop : 0, kend = upper ? nl : ltrop;
        if (kbeg >= kend) continue;
        const int ng = B.ng, g0 = B.g0;
        const int np = ng >> 1;                 // g-point pairs of the band: a lane owns one pair of one layer
        const int lpi = 64 / np;
        const int dk = lane / np;
        LaneG G;
        G.ig = 2 * (lane - dk * np);
        {
            const LwGDesc *gd = tb.gdesc + g0 + G.ig;
            const LwRegime *r0 = upper ? &gd[0].up : &gd[0].lo, *r1 = upper ? &gd[1].up : &gd[1].lo;
            G.fracconst = make_double2(r0->fracconst, r1->fracconst);
            G.halo0 = make_double2(r0->halo_coef[0], r1->halo_coef[0]);
            G.halo1 = make_double2(r0->halo_coef[1], r1->halo_coef[1]);
            G.postscale = make_double2(r0->postscale, r1->postscale);
        }
        // band-major scratch [band][layer][ig]: a lane stores its pair with one 16-byte store, an item's lanes one
        // contiguous run
        double *tgb = tg + (size_t)nl * g0 + G.ig, *frb = fr + (size_t)nl * g0 + G.ig;
        const int type = (R.cfg >> LC_TYPE_SHIFT) & 3;
        // the Planck fraction travels through the scratch only where it is interpolated; elsewhere the transfer kernel
        // takes the g-point's constant from the descriptor (48 % of the (layer, g-point) pairs: 17 GB less to read)
        const bool frac_stored = ((R.cfg >> LC_FRAC_SHIFT) & 3) == 2;
        // continuum rows of this band into the wave's LDS area: lanes = (row, g-pair), 16-byte loads
        const double *cont = nullptr;
        if (c.cont && (R.cfg & (LC_SELF | LC_FOR))) {
            double *cw = c.cont + (size_t)wave * CONT_DOUBLES;
            __builtin_amdgcn_wave_barrier();   // the previous segment's items have read their rows
            // At most 14 x 8 = 112 (row, pair) items: two per lane, both reads issued before either is stored -- one
            // trip to memory per segment.  (As a loop with the band's offsets picked per lane, the compiler read
            // the picked offset through a per-lane pointer first: two dependent trips per turn, two turns.)
            static_assert((CONT_SELF_ROWS + CONT_FOR_ROWS) * (16 / 2) <= 128, "two items per lane");
            int o_self = B.off_self, o_for = B.off_for, strd = B.stride;
            asm("" : "+s"(o_self), "+s"(o_for), "+s"(strd));
            const int nitem = (CONT_SELF_ROWS + CONT_FOR_ROWS) * np;
            const bool has_self = R.cfg & LC_SELF;
            double2 cv[2];
            int cdst[2];
            bool cok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = lane + 64 * u;
                const int itc = it < nitem ? it : nitem - 1;
                const int row = itc / np, pr = itc - row * np;
                const bool self = row < CONT_SELF_ROWS;
                cok[u] = it < nitem && (has_self || !self);
                // (a self row of a band without self continuum is not stored; its lane reads a foreign row instead)
                const int off = self && has_self ? o_self + row * strd : o_for + (self ? 0 : row - CONT_SELF_ROWS) * strd;
                cv[u] = ld2(tb.K + off + 2 * pr);
                cdst[u] = row * CONT_STRIDE + 2 * pr;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (cok[u]) *reinterpret_cast<double2 *>(cw + cdst[u]) = cv[u];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cont = cw;
        }
        if (type == 2) {
            // binary-species regime: per-layer pre-pass into this wave's LDS area, then the items of the same layers
            double *ppd = c.pp + (size_t)wave * PP_DOUBLES;
            int *ppi = reinterpret_cast<int *>(ppd + PP_NDBL * PP_LAYERS);
            const int lpe = lpi < PP_LAYERS ? lpi : PP_LAYERS;
            const bool lane_on = dk < lpe;
            for (int kc = kbeg; kc < kend; kc += PP_LAYERS) {
                const int nin = kend - kc < PP_LAYERS ? kend - kc : PP_LAYERS;
                gas_prepass(tb, c, R, B.stride, ppd, ppi, kc, kend, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int kb = 0; kb < nin; kb += lpe) {
                    const bool act = lane_on && kb + dk < nin;
                    const int l = act ? kb + dk : nin - 1;
                    const int k = kc + l;
                    double2 taug, plfrac;
                    gas_item_binary(tb, c, R, B.stride, B.off_self, B.off_for, G, ppd, ppi, k, l, cont, taug, plfrac);
                    if (act) {
                        st2_stream(tgb + (size_t)k * ng, taug);
                        if (frac_stored) st2_stream(frb + (size_t)k * ng, plfrac);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }
        const bool lane_on = dk < lpi;
        for (int kb = kbeg; kb < kend; kb += lpi) {
            const bool act = lane_on && kb + dk < kend;
            const int k = act ? kb + dk : kend - 1;
            double2 taug, plfrac;
            gas_item_single(tb, c, R, B.stride, B.off_self, B.off_for, G, k, cont, taug, plfrac);
            // no fraction to store: only a binary-species regime interpolates it (checked where the segment order is
            // made, lw_upload_segments), here it is the g-point's constant, which the transfer kernel takes itself
            if (act) st2_stream(tgb + (size_t)k * ng, taug);
        }
    }
#ifdef LW_EXP_CLOCK
    LW_STAMP(6);
    if ((blockIdx.x == 5 || blockIdx.x == 3000 || blockIdx.x == 20000) && lane == 0)
        printf("blk %d wave %d: A %lld bar %lld B %lld bar %lld C %lld gas %lld (ltrop %d nl %d)\n", blockIdx.x, wave,
               tstamp[1] - tstamp[0], tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3],
               tstamp[5] - tstamp[4], tstamp[6] - tstamp[5], ltrop, nl);
#endif
}

// Kernel 2 of 2: radiative transfer (rtrnmc) for the clean and the aerosol optical depths, lanes = g-points;
// flux profiles, heating rate and the WRF output fields.
template <bool DEEP>   // DEEP: McICA clouds in more than 64 model layers (the bits of layers 64.. live in LDS)
__device__ __forceinline__ void lw_rt_body(const LwKernelArgs &a)
{
    extern __shared__ double lds[];
    const LwDeviceTables &tb = a.tab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nk = a.nk, nl = a.nlayers;
    const int slot = batch_slot(batch_columns(a));
    if (slot < 0) return;   // (uniform over the workgroup, in front of its first barrier)
    int col = a.col0 + slot;
    if (a.colmap) col = a.colmap[col];   // entry of the call's column list (cloud_lists.h)
    LCol c = carve_rt(lds, nl, a.icloud != 0);
    const int nslot = a.icloud != 0 ? 6 : 4;   // flux slots per wave: + clear-sky down / up with clouds
    {
        const double *rec = a.rec + (size_t)slot * lw_rec_doubles(nl);
        // (everything in front of the per-band cloud optical depths: this kernel forms them itself from the in-cloud
        // state; the cloud part is written only when icloud != 0 and is not read otherwise)
        const int n = (int)lw_rec_tc_offset(nl);
        for (int q = tid; q < n; q += LW_THREADS) lds[q] = rec[q];
    }
    // cloud part of the record: cldf clwp ciwp cswp fliq fice fsno iliq iice isno [nl] each, seeds[4]
    const double *cb = lds + lw_rec_base_doubles(nl);
    __shared__ unsigned long long s_lmask[2];       // layers 0..63, 64..127
    __shared__ unsigned long long s_cmhi[LW_THREADS];   // each lane's cloud bits of layers 64..127
    if (tid == 0) { s_lmask[0] = 0ull; s_lmask[1] = 0ull; }
    const int it = col % a.ni_t, jt = col / a.ni_t;
    const size_t i0 = (size_t)(a.its - a.ims + it), j0 = (size_t)(a.jts - a.jms + jt), k0 = (size_t)(a.kts - a.kms);
    const size_t i2 = j0 * a.ni_m + i0;
    const size_t sk = a.ni_m, sj = (size_t)a.ni_m * a.nk_m, sjf = (size_t)a.ni_m * (a.nk_m + 2);
    const size_t base3 = j0 * sj + k0 * sk + i0, base3f = j0 * sjf + k0 * sk + i0;
    const int r4 = a.real4;
    const double semiss = ldr(a.emiss, i2, r4);   // same emissivity in all 16 bands, LW:12296-12298

    const int gbeg = wave * LW_CHUNK;
    const int gcnt = (LW_NGPT - gbeg) < LW_CHUNK ? (LW_NGPT - gbeg) : LW_CHUNK;
    const bool active = lane < gcnt;
    const int g = gbeg + (active ? lane : gcnt - 1);
    const LwGDesc *gd = tb.gdesc + g;
    const int band = gd->band;
    const double wgt = active ? 0.5 * gd->delwave : 0.;   // wtdiff * delwave(iband), LW:3366-3371
    const bool clean_on = a.clean > 0;
    const int nlev = nl + 1;
    // scratch of the column, band-major: [band][layer][ig] (lw_kernel.h)
    const int ng = gd->ng;
    const size_t sbase = (size_t)slot * nl * LW_NGPT + (size_t)nl * (g - gd->ig) + gd->ig;
    const double *tg = a.taug + sbase, *fr = a.frac + sbase;
    __syncthreads();
    const FracSrc FS = frac_src(gd, c.pz[nl + 1], fr - tg);   // laytrop follows pz in the record
    const double secdiff = c.secdiff[band];
    double *fl = c.flux + (size_t)wave * nslot * nlev;   // down clean, down total, up clean, up total, down clear, up clear
    const double *q4 = c.q4 + (size_t)band * lw_rec_q4_layers(nl) * 4;   // planklay, dplankdn, dplankup, taua per layer

    // McICA (generate_stochastic_clouds, LW:2208-2578, icld = 2): this lane's sub-column of binary cloud over the
    // model layers (the buffer layers above carry none); lmask = layers in which any sub-column is cloudy
    // (icldlyr, LW:3163-3186) -- those take the "cloudy layer" arithmetic for every g-point.
    uint64_t cmask = 0;
    constexpr bool deep = DEEP;
    int iceflag = 3;
    if (a.icloud != 0) {
        if (a.has_reqi != 0) iceflag = 4;
        if (a.has_reqs != 0 || (a.has_reqi != 0 && a.has_reqc != 0)) iceflag = 5;
        KissState ks = {(uint32_t)(int)cb[10 * nl + 0], (uint32_t)(int)cb[10 * nl + 1], (uint32_t)(int)cb[10 * nl + 2],
                        (uint32_t)(int)cb[10 * nl + 3]};
        kiss_jump(ks, a.jump + (size_t)g * KISS_JUMP_WORDS);
        double prev = 0.;
        uint64_t cmhi = 0;
        for (int l = 0; l < nk; ++l) {
            double cur = kiss_next(ks);
            if (l > 0) {
                const double cbl = cb[l - 1];
                cur = prev > 1. - cbl ? prev : cur * (1. - cbl);
            }
            if (cur >= 1. - cb[l]) {
                if (l < 64) cmask |= (uint64_t)1 << l; else cmhi |= (uint64_t)1 << (l - 64);
            }
            prev = cur;
        }
        uint64_t m = active ? cmask : 0;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) m |= __shfl_xor(m, s, 64);
        if (lane == 0 && m) atomicOr(&s_lmask[0], (unsigned long long)m);
        if (deep) {
            s_cmhi[tid] = cmhi;
            m = active ? cmhi : 0;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) m |= __shfl_xor(m, s, 64);
            if (lane == 0 && m) atomicOr(&s_lmask[1], (unsigned long long)m);
        }
        __syncthreads();
    }
    const uint64_t lmask = a.icloud != 0 ? (uint64_t)s_lmask[0] : 0;
    const bool col_cloudy = lmask != 0 || (a.icloud != 0 && deep && s_lmask[1] != 0);
    int bad = 0;

    // ---- downward sweep (top -> surface), LW:3207-3312 ----
    double radld_c = 0., radld_t = 0., radclr = 0., frac0 = 0.;
    bool iclddn = false;
    if (lane == 0) { fl[0 * nlev + nl] = 0.; fl[1 * nlev + nl] = 0.; if (nslot == 6) fl[4 * nlev + nl] = 0.; }
    double tnext = tg[(size_t)(nl - 1) * ng], fnext = frac_read(FS, tg + (size_t)(nl - 1) * ng, nl - 1);
    for (int k = nl - 1; k >= 0; --k) {
        const double taug = tnext, plfrac = frac_value(FS, fnext, k);
        if (k > 0) { tnext = tg[(size_t)(k - 1) * ng]; fnext = frac_read(FS, tg + (size_t)(k - 1) * ng, k - 1); }
        const double blay = q4[4 * k], dplankdn = q4[4 * k + 1], dplankup = q4[4 * k + 2], tak = q4[4 * k + 3];
        const bool lay_cld = k < 64 ? (lmask >> k) & 1 : (col_cloudy && deep && k < 128 && ((s_lmask[1] >> (k - 64)) & 1));
        if (!lay_cld) {
            double at, bbd, bbu;
            layer_src<false>(tb, nullptr, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, at, bbd, bbu);   // taut, LW:11004-11008
            radld_t = relax(radld_t, bbd, at);
            radclr = iclddn ? relax(radclr, bbd, at) : radld_t;   // LW:3304-3311
            if (clean_on) {
                // without aerosol in this layer (buffer layers, clean air) the clean call equals the one above
                if (__any(tak != 0.)) layer_src<false>(tb, nullptr, taug, secdiff, plfrac, blay, dplankup, dplankdn, at, bbd, bbu);
                radld_c = relax(radld_c, bbd, at);
            }
        } else {
            iclddn = true;
            const bool cl = k < 64 ? (cmask >> k) & 1 : (s_cmhi[tid] >> (k - 64)) & 1;
            const double cfm = cl ? 1. : 0.;
            double odcld = 0., efcl = 0.;
            if (cl) {
                odcld = secdiff * lw_cloud_tau(a.cldtab, cb, nl, band, k, iceflag, bad);   // LW:3172-3176
                efcl = 1. - exp_neg(odcld);
            }
            LwCldSrc s;
            layer_src_cld<false>(tb, nullptr, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, odcld, s);
            double gassrc = s.bbd * s.atrans;
            radld_t = radld_t - radld_t * (s.atrans + efcl * (1. - s.atrans)) + gassrc + cfm * (s.bbdtot * s.atot - gassrc);
            radclr = radclr + (s.bbd - radclr) * s.atrans;
            if (clean_on) {
                layer_src_cld<false>(tb, nullptr, taug, secdiff, plfrac, blay, dplankup, dplankdn, odcld, s);
                gassrc = s.bbd * s.atrans;
                radld_c = radld_c - radld_c * (s.atrans + efcl * (1. - s.atrans)) + gassrc + cfm * (s.bbdtot * s.atot - gassrc);
            }
        }
        double s2 = wave_sum2(wgt * radld_c, wgt * radld_t, lane);
        if (lane == 0) fl[0 * nlev + k] = s2;
        if (lane == 32) fl[1 * nlev + k] = s2;
        if (col_cloudy) {
            const double sc = wave_sum2(wgt * radclr, 0., lane);
            if (lane == 0) fl[4 * nlev + k] = sc;
        }
        frac0 = plfrac;
    }
    // ---- surface: emission + reflection, LW:3320-3324 ----
    const double rad0 = frac0 * c.plankbnd[band];
    const double reflect = 1. - semiss;
    double radlu_c = __builtin_fma(reflect, radld_c, rad0), radlu_t = __builtin_fma(reflect, radld_t, rad0),
           radclru = __builtin_fma(reflect, radclr, rad0);
    {
        double s2 = wave_sum2(wgt * radlu_c, wgt * radlu_t, lane);
        if (lane == 0) fl[2 * nlev] = s2;
        if (lane == 32) fl[3 * nlev] = s2;
        if (col_cloudy) {
            const double sc = wave_sum2(wgt * radclru, 0., lane);
            if (lane == 0) fl[5 * nlev] = sc;
        }
    }
    // ---- upward sweep (surface -> top), LW:3331-3356 ----
    for (int k = 0; k < nl; ++k) {   // tnext / fnext hold layer 0 after the downward sweep
        const double taug = tnext, plfrac = frac_value(FS, fnext, k);
        if (k + 1 < nl) { tnext = tg[(size_t)(k + 1) * ng]; fnext = frac_read(FS, tg + (size_t)(k + 1) * ng, k + 1); }
        const double blay = q4[4 * k], dplankdn = q4[4 * k + 1], dplankup = q4[4 * k + 2], tak = q4[4 * k + 3];
        const bool lay_cld = k < 64 ? (lmask >> k) & 1 : (col_cloudy && deep && k < 128 && ((s_lmask[1] >> (k - 64)) & 1));
        if (!lay_cld) {
            double at, bbd, bbu;
            layer_src<false>(tb, nullptr, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, at, bbd, bbu);
            radlu_t = relax(radlu_t, bbu, at);
            radclru = col_cloudy ? relax(radclru, bbu, at) : radlu_t;   // LW:3348-3354 (iclddn of the column)
            if (clean_on) {
                // without aerosol in this layer (buffer layers, clean air) the clean call equals the one above
                if (__any(tak != 0.)) layer_src<false>(tb, nullptr, taug, secdiff, plfrac, blay, dplankup, dplankdn, at, bbd, bbu);
                radlu_c = relax(radlu_c, bbu, at);
            }
        } else {
            const bool cl = k < 64 ? (cmask >> k) & 1 : (s_cmhi[tid] >> (k - 64)) & 1;
            const double cfm = cl ? 1. : 0.;
            double odcld = 0., efcl = 0.;
            if (cl) {
                odcld = secdiff * lw_cloud_tau(a.cldtab, cb, nl, band, k, iceflag, bad);
                efcl = 1. - exp_neg(odcld);
            }
            LwCldSrc s;
            layer_src_cld<false>(tb, nullptr, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, odcld, s);
            double gassrc = s.bbugas * s.atrans;
            radlu_t = radlu_t - radlu_t * (s.atrans + efcl * (1. - s.atrans)) + gassrc + cfm * (s.bbutot * s.atot - gassrc);
            radclru = radclru + (s.bbugas - radclru) * s.atrans;
            if (clean_on) {
                layer_src_cld<false>(tb, nullptr, taug, secdiff, plfrac, blay, dplankup, dplankdn, odcld, s);
                gassrc = s.bbugas * s.atrans;
                radlu_c = radlu_c - radlu_c * (s.atrans + efcl * (1. - s.atrans)) + gassrc + cfm * (s.bbutot * s.atot - gassrc);
            }
        }
        double s2 = wave_sum2(wgt * radlu_c, wgt * radlu_t, lane);
        if (lane == 0) fl[2 * nlev + k + 1] = s2;
        if (lane == 32) fl[3 * nlev + k + 1] = s2;
        if (col_cloudy) {
            const double sc = wave_sum2(wgt * radclru, 0., lane);
            if (lane == 0) fl[5 * nlev + k + 1] = sc;
        }
    }
    if (bad) atomicMax(a.status, (bad << 28) | (1 + col));
    __syncthreads();

    // ======================= epilogue: lanes = levels =======================
    const double *f0 = c.flux, *f1 = c.flux + (size_t)nslot * nlev, *f2 = c.flux + (size_t)2 * nslot * nlev;
    for (int l = tid; l < nlev; l += LW_THREADS) {
        auto tot = [&](int q, int lev) { return (f0[q * nlev + lev] + f1[q * nlev + lev] + f2[q * nlev + lev]) * tb.fluxfac; };
        const double dnc = clean_on ? tot(0, l) : 0., dn = tot(1, l), upc = clean_on ? tot(2, l) : 0., up = tot(3, l);
        // clear sky (aerosol call): own stream in a cloudy column, else the total-sky values (LW:3304-3311)
        const double dnk = col_cloudy ? tot(4, l) : dn, upk = col_cloudy ? tot(5, l) : up;
        if (a.lwupflx && l <= nk + 1) {   // kts:kte+2, LW:12677-12684
            const size_t q = base3f + (size_t)l * sk;
            str(a.lwupflx, q, r4, up); str(a.lwupflxc, q, r4, upk); str(a.lwdnflx, q, r4, dn); str(a.lwdnflxc, q, r4, dnk);
            str(a.lwupflxcln, q, r4, upc); str(a.lwdnflxcln, q, r4, dnc);
        }
        if (l < nk && a.rthratenlw) {
            // htr(l) = heatfac*(fnet(l)-fnet(l+1))/(pz(l)-pz(l+1)), LW:3402; then LW:12687-12690
            const double up1 = tot(3, l + 1), dn1 = tot(1, l + 1);
            const double fnet0 = up - dn, fnet1 = up1 - dn1;
            const double htr = tb.heatfac * (fnet0 - fnet1) / (c.pz[l] - c.pz[l + 1]);
            const double tten = htr / 86400.;
            str(a.rthratenlw, base3 + (size_t)l * sk, r4, tten / ldr(a.pi3d, base3 + (size_t)l * sk, r4));
        }
        if (l == 0) {
            if (a.glw) str(a.glw, i2, r4, dn);
            if (a.lwupt) {
                str(a.lwupb, i2, r4, up); str(a.lwupbc, i2, r4, upk); str(a.lwdnb, i2, r4, dn); str(a.lwdnbc, i2, r4, dnk);
                str(a.lwupbcln, i2, r4, upc); str(a.lwdnbcln, i2, r4, dnc);
            }
        }
        if (l == nl) {
            if (a.olr) str(a.olr, i2, r4, up);
            if (a.lwcf) str(a.lwcf, i2, r4, upk - up);   // LW:12658
            if (a.lwupt) {
                str(a.lwupt, i2, r4, up); str(a.lwuptc, i2, r4, upk); str(a.lwdnt, i2, r4, dn); str(a.lwdntc, i2, r4, dnk);
                str(a.lwuptcln, i2, r4, upc); str(a.lwdntcln, i2, r4, dnc);
            }
        }
    }
}

// The transfer kernel as the patched radiation_driver calls it: no layer-flux profiles (LWUPFLX.. are optional
// arguments the driver does not receive, DRV:1528-1577).  Then only the call with aerosol needs fluxes at the model
// levels (heating rates); the clean-sky call and, in cloudy columns, the clear-sky stream are used at the surface and
// at the top only.  Their downward recurrences are carried along anyway; their upward flux at the top follows in the
// SAME top-down sweep from the linearity of the upward recurrence (LW:3331-3356):
//     radlu(top) = radlu(0) * prod_k (1 - A_k) + sum_k S_k * prod_{j > k} (1 - A_j)
// with (A_k, S_k) = (atrans, bbugas*atrans) of a clear layer or (atrans + efclfrac*(1 - atrans), gassrc +
// cldfmc*(bbutot*atot - gassrc)) of a cloudy one.  The upward sweep is left with one stream, and it evaluates no
// layer optics of its own: the downward sweep leaves (A_k, bbu_k) -- (A_k, S_k) in a cloudy layer -- of the call with
// aerosol in the scratch slots the gas optics occupied (same bytes), and the way up is one relaxation per layer.
// The spectral sums (LW:3366-3377) are taken four levels at a time.
//
// SEVEN columns per workgroup of sixteen waves.  A column's 140 g-points are nine chunks of 16 (the last one holds
// 12); a chunk sits on one ROW of a wave (16 lanes, the unit of the DPP row operations), so 7 x 9 = 63 of the
// workgroup's 64 rows are taken and 61.25 of 64 lanes work (a column alone on three waves: 47; three columns on eight
// waves in half-wave chunks of 28: 52.5).  One such workgroup fills a CU (4 waves per SIMD, what 121 VGPRs allow); a
// five-wave workgroup -- two columns, 56 lanes -- puts two of its waves on one SIMD and only two of them fit (measured
// 2.05 waves per SIMD).  Every spectral sum is taken per chunk, i.e. inside a row, and the nine chunk sums of a column
// are added in chunk order: a column's sums are formed the same way whichever rows it has (results do not depend on
// the tiling of the domain).
constexpr int RTF_COLS = 7, RTF_WAVES = 16, RTF_THREADS = RTF_WAVES * 64, RTF_CHUNK = 16, RTF_NCHUNK = 9;
// The McICA instantiations: FIVE columns on TWELVE waves (45 of 48 rows), i.e. three waves per SIMD and 168 VGPRs -- on
// sixteen waves (128 VGPRs) the clear-sky stream and the cloudy-layer terms spilled 300 bytes per lane, 40 spill
// accesses per layer of the downward sweep, each of them a vector-memory operation in the loop's wait queue.
constexpr int RTF_CLD_COLS = 5, RTF_CLD_WAVES = 12, RTF_CLD_THREADS = RTF_CLD_WAVES * 64;
static_assert(RTF_CHUNK * RTF_NCHUNK >= LW_NGPT && RTF_CHUNK * (RTF_NCHUNK - 1) < LW_NGPT, "chunks must cover the g-points");
static_assert(RTF_COLS * RTF_NCHUNK <= RTF_WAVES * 4 && RTF_CLD_COLS * RTF_NCHUNK <= RTF_CLD_WAVES * 4, "a row per chunk");
__host__ __device__ constexpr int rtf_cols(bool cloud) { return cloud ? RTF_CLD_COLS : RTF_COLS; }
__host__ __device__ constexpr int rtf_threads(bool cloud) { return cloud ? RTF_CLD_THREADS : RTF_THREADS; }

// LDS per column: the in-cloud state of the column (McICA instantiations only) and the flux slots of its five chunks.
// The per-band arrays of the column record (Planck functions, aerosol optical depth) are NOT staged: the lanes of a
// band read the same address, a wave's load touches a handful of lines that the workgroup's other waves find in L1,
// and without them two workgroups fit a CU whatever the layer count (staged, 3 x 25 KB at 63 layers left one).
static_assert(LW_RTF_NCHUNK == RTF_NCHUNK, "lw_flux_doubles (lw_kernel.h) sizes the flux slots of RTF_NCHUNK chunks");
__host__ __device__ inline size_t rtf_col_doubles(int nl, bool cloud)
{
    return (cloud ? (size_t)nl : 0) + (size_t)RTF_NCHUNK * (2 * (nl + 1) + 8);   // (cloud fractions) + flux slots
}

// Sums of four values over each ROW of the wave (16 lanes): afterwards every lane of quad q of a row holds the row's
// sum of v[{0, 2, 1, 3}[q]].  The values are folded onto each other on the way (halves of the row keep v0 / v1 and
// v2 / v3, then quads keep one each), so the four sums cost what one and a half would; all rows use the same tree.
__device__ __forceinline__ double row_sum4(double v0, double v1, double v2, double v3, int lane)
{
    const bool up = lane & 8, q1 = lane & 4;
    // row_mirror pairs lane i with 15 - i (the other half of the row)
    const double x01 = (up ? v1 : v0) + __hiloint2double(
        __builtin_amdgcn_mov_dpp(__double2hiint(up ? v0 : v1), 0x140, 0xF, 0xF, true),
        __builtin_amdgcn_mov_dpp(__double2loint(up ? v0 : v1), 0x140, 0xF, 0xF, true));
    const double x23 = (up ? v3 : v2) + __hiloint2double(
        __builtin_amdgcn_mov_dpp(__double2hiint(up ? v2 : v3), 0x140, 0xF, 0xF, true),
        __builtin_amdgcn_mov_dpp(__double2loint(up ? v2 : v3), 0x140, 0xF, 0xF, true));
    // row_half_mirror pairs lane i with 7 - i (the other quad of the half)
    const double send = q1 ? x01 : x23;
    double y = (q1 ? x23 : x01) + __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(send), 0x141, 0xF, 0xF, true),
                                                   __builtin_amdgcn_mov_dpp(__double2loint(send), 0x141, 0xF, 0xF, true));
    y = dpp_add<0x4E>(y);    // quad_perm [2,3,0,1]
    y = dpp_add<0xB1>(y);    // quad_perm [1,0,3,2]
    return y;
}

// TBL: the pair table of layer_src -- the entries from LW_PAIR_FIRST on of 10 001 x (1 - exp_tbl, tfn_tbl), 157 392 bytes
// -- fills the workgroup's LDS, and the flux slots live in the column's block of the scratch instead (lw_flux_doubles:
// written by one lane in four once per four layers, read by the epilogue; they never leave L2).  An evaluation of the
// layer optics is then ~25 instructions instead of ~80 (r03: three evaluations per (layer, g-point) were half of the
// kernel).  With McICA only the layer cloud fractions (the sub-column generator walks through them) follow the table in
// LDS; the in-cloud water paths and sizes of the few cloudy layers are read from the column record.
template <bool DEEP, bool CLOUD, bool TBL>   // CLOUD: McICA (icloud != 0); the cloud-free instantiation carries none of it
__device__ __forceinline__ void lw_rt_fast(const LwKernelArgs &a)
{
    static_assert(!(TBL && DEEP), "more than 64 cloud layers: the instantiation without the table");
    extern __shared__ double lds[];
    const LwDeviceTables &tb = a.tab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nk = a.nk, nl = a.nlayers;
    constexpr bool cloud = CLOUD;
    constexpr int NCOL = rtf_cols(CLOUD), NTHR = rtf_threads(CLOUD);   // columns and threads of the workgroup
    const int nbatch = batch_columns(a);
    const int grp = batch_slot((nbatch + NCOL - 1) / NCOL);
    if (grp < 0) return;   // (uniform over the workgroup, in front of its first barrier)
    const int ncols = nbatch - NCOL * grp < NCOL ? nbatch - NCOL * grp : NCOL;   // 1..7
    // staged part of a column's record: the layer cloud fractions (the sub-column generator walks through them)
    const size_t recn = cloud ? (size_t)nl : 0;
    const size_t cold = TBL ? recn : rtf_col_doubles(nl, cloud);   // a column's LDS (TBL: behind the table)
    double *const lds_cols = TBL ? lds + 2 * LW_PAIR_COUNT : lds;
    const double2 *const LTBL = TBL ? reinterpret_cast<const double2 *>(lds) : nullptr;
    if (TBL) {
        const double2 *src = reinterpret_cast<const double2 *>(tb.tbl_pairs) + LW_PAIR_FIRST;
        double2 *dst = reinterpret_cast<double2 *>(lds);
        for (int q = tid; q < LW_PAIR_COUNT; q += NTHR) dst[q] = src[q];
    }
    if (cloud)
        for (int cc = 0; cc < ncols; ++cc) {
            const double *rec = a.rec + (size_t)(NCOL * grp + cc) * lw_rec_doubles(nl) + lw_rec_base_doubles(nl);
            double *dst = lds_cols + cc * cold;
            for (int q = tid; q < (int)recn; q += NTHR) dst[q] = rec[q];
        }
    __shared__ unsigned long long s_lmask[NCOL][2];     // [column][layers 0..63, 64..127]
    __shared__ unsigned long long s_cmhi[NTHR];
    if (tid < 2 * NCOL) s_lmask[tid >> 1][tid & 1] = 0ull;

    // this lane's seat: row -> (column, chunk); lanes 12..15 of a column's last chunk and the 64th row are idle
    const int seat = 4 * wave + (lane >> 4), l16 = lane & 15;
    const int scol = seat / RTF_NCHUNK, chunk = seat - scol * RTF_NCHUNK;
    const bool seat_on = scol < ncols;                        // uniform over the row
    const int graw = chunk * RTF_CHUNK + l16;
    const bool live = seat_on && graw < LW_NGPT;              // a lane with a g-point of an existing column
    const int colsel = seat_on ? scol : 0;                    // idle seats shadow column 0 (no stores, nothing summed)
    const int g = graw < LW_NGPT ? graw : LW_NGPT - 1;
    const int slot = NCOL * grp + colsel;
    int col = a.col0 + slot;
    if (a.colmap) col = a.colmap[col];   // (an existing entry: slot is clamped to the group's columns)
    const double *const rec = a.rec + (size_t)slot * lw_rec_doubles(nl);   // global: secdiff plankbnd pz laytrop [cloud] q4
    // cloud part of the record: cbf = the cloud fractions (LDS), cb = all of it (the record: the four seeds)
    const double *const cbf = lds_cols + colsel * cold;
    const double *const cb = rec + lw_rec_base_doubles(nl);
    const int r4 = a.real4;
    double semiss;
    {
        const int it = col % a.ni_t, jt = col / a.ni_t;
        const size_t i2 = (size_t)(a.jts - a.jms + jt) * a.ni_m + (size_t)(a.its - a.ims + it);
        semiss = ldr(a.emiss, i2, r4);
    }
    const LwGDesc *gd = tb.gdesc + g;
    const int band = gd->band, ng = gd->ng;
    const double wgt = live ? 0.5 * gd->delwave : 0.;
    // the clean-sky call differs from the one with aerosol only if there is aerosol; without it the reference's
    // two calls are the same arithmetic on the same numbers, and the outputs are copies
    const bool clean_on = a.clean > 0, clean_own = clean_on && a.aer_ra_feedback == 1;
    const int nlev = nl + 1;
    // scratch of the column, band-major: [band][layer][ig] (lw_kernel.h)
    const size_t sbase = (size_t)slot * nl * LW_NGPT + (size_t)nl * (g - gd->ig) + gd->ig;
    double *tg = a.taug + sbase, *fr = a.frac + sbase;
    __syncthreads();
    const FracSrc FS = frac_src(gd, rec[lw_rec_base_doubles(nl) - 1], fr - tg);
    // per chunk: downward and upward flux of the call with aerosol at every level, then the small sums
    // [0..2] clean sky: surface down, surface up, top up; [4..6] clear sky alike
    const size_t ws = 2 * nlev + 8;
    double *fl;                                                           // this seat's slots
    if constexpr (TBL) fl = a.flx + (size_t)slot * lw_flux_doubles(nl) + (size_t)chunk * ws;
    else fl = lds_cols + colsel * cold + recn + (size_t)chunk * ws;
    // the lane's band in the packed part of the record: per layer (planklay, dplankdn) and (dplankup, taua)
    // The lanes of a band read the same quadruple, and a band's block of four layers is one 128-byte line: the WAVE
    // fetches each line once per block -- lane 4s + j the quadruple of layer j of the wave's s-th (column, band) pair,
    // a block ahead -- and a lane takes its layer's four values from that lane's registers through the LDS crossbar
    // (ds_bpermute: no LDS memory, no trip to the vector cache).  Read per lane and layer, the quadruples cost the
    // vector cache a line fill per band, layer and wave (the streaming gas optics turn the L1 over in between), 5 of the
    // kernel's 19.5 ms.  A wave's four rows hold at most 14 pairs (16 x 4 loader lanes).
    int bp_idx;                 // byte index (lane * 4) of the loader lane of this lane's pair, layer 0 of a block
    long long bp_off;           // loader lanes: where the pair's quadruples (layer 0) sit behind a.rec, in doubles; others: -1
    long long bp_tc;            // ... and its cloud optical depths (McICA instantiations)
    {
        const int key = colsel * LW_NBND + band, prev = __shfl_up(key, 1, 64);
        const unsigned long long firsts = __ballot(lane == 0 || key != prev);
        const int myslot = __popcll(firsts & (~0ull >> (63 - lane))) - 1;
        bp_idx = 16 * myslot;
        // loader lane l serves pair l >> 2: the lane that opens the pair knows its quadruples
        const int s_ld = lane >> 2;
        unsigned long long t = firsts;
        for (int i = 0; i < s_ld && t; ++i) t &= t - 1;
        const int opener = t ? __builtin_ctzll(t) : 0;
        // (as an offset into the scratch's record area, formed by integer arithmetic alone: a pointer put together from
        // shuffled halves is read through flat loads, and even the conversion of a record pointer to an integer makes
        // the compiler treat every access to the record as flat -- with a flat load in flight every wait of the loop is
        // vmcnt(0))
        const unsigned long long qoff = (unsigned long long)slot * lw_rec_doubles(nl) + lw_rec_q4_offset(nl) +
                                        (unsigned long long)band * lw_rec_q4_layers(nl) * 4;
        const unsigned lo = __shfl((unsigned)qoff, opener, 64), hi = __shfl((unsigned)(qoff >> 32), opener, 64);
        bp_off = t ? (long long)((((unsigned long long)hi << 32) | lo) + 4 * (lane & 3)) : -1;
        bp_tc = 0;
        if (cloud) {
            const unsigned long long toff = (unsigned long long)slot * lw_rec_doubles(nl) + lw_rec_tc_offset(nl) +
                                            (unsigned long long)band * lw_rec_q4_layers(nl);
            const unsigned tlo = __shfl((unsigned)toff, opener, 64), thi = __shfl((unsigned)(toff >> 32), opener, 64);
            bp_tc = (long long)((((unsigned long long)thi << 32) | tlo) + (lane & 3));
        }
    }
    auto bp_fetch = [&](const int b, double2 &u, double2 &w, double &tcv) {   // the loader lanes' part of block b
        if (bp_off >= 0) {
            const double2 *src = reinterpret_cast<const double2 *>(a.rec + bp_off) + 8 * b;
            u = src[0]; w = src[1];
            if (cloud) tcv = a.rec[bp_tc + 4 * b];
        }
    };
    auto bp_take1 = [&](const int j, const double x) {
        const int idx = bp_idx + 4 * j;
        return __hiloint2double(__builtin_amdgcn_ds_bpermute(idx, __double2hiint(x)),
                                __builtin_amdgcn_ds_bpermute(idx, __double2loint(x)));
    };
    auto bp_take = [&](const int j, const double2 u, const double2 w, double &blay, double &dpdn, double &dpup, double &tak) {
        const int idx = bp_idx + 4 * j;
        auto pick = [&](const double x) {
            return __hiloint2double(__builtin_amdgcn_ds_bpermute(idx, __double2hiint(x)),
                                    __builtin_amdgcn_ds_bpermute(idx, __double2loint(x)));
        };
        blay = pick(u.x); dpdn = pick(u.y); dpup = pick(w.x); tak = pick(w.y);
    };
    const double secdiff = rec[band], plankbnd = rec[LW_NBND + band];

    uint64_t cmask = 0;
    constexpr bool deep = DEEP;
    if (cloud) {
        KissState ks = {(uint32_t)(int)cb[10 * nl + 0], (uint32_t)(int)cb[10 * nl + 1], (uint32_t)(int)cb[10 * nl + 2],
                        (uint32_t)(int)cb[10 * nl + 3]};
        kiss_jump(ks, a.jump + (size_t)g * KISS_JUMP_WORDS);
        double prev = 0.;
        uint64_t cmhi = 0;
        for (int l = 0; l < nk; ++l) {
            double cur = kiss_next(ks);
            if (l > 0) {
                const double cbl = cbf[l - 1];
                cur = prev > 1. - cbl ? prev : cur * (1. - cbl);
            }
            if (cur >= 1. - cbf[l]) {
                if (l < 64) cmask |= (uint64_t)1 << l; else cmhi |= (uint64_t)1 << (l - 64);
            }
            prev = cur;
        }
        if (live && cmask) atomicOr(&s_lmask[colsel][0], (unsigned long long)cmask);
        if (deep) {
            s_cmhi[tid] = cmhi;
            if (live && cmhi) atomicOr(&s_lmask[colsel][1], (unsigned long long)cmhi);
        }
        __syncthreads();
    }
    const uint64_t lmask = cloud ? (uint64_t)s_lmask[colsel][0] : 0;
    const uint64_t lmask_hi = cloud && deep ? (uint64_t)s_lmask[colsel][1] : 0;
    const bool col_cloudy = lmask != 0 || lmask_hi != 0;
    int bad = 0;
    // the first lane of quad q of a row holds the row's sum of value jw = {0, 2, 1, 3}[q] (row_sum4)
    const bool writer = (lane & 3) == 0 && seat_on;
    const int jw = ((lane >> 1) & 2) | ((lane >> 3) & 1);

    // ---- downward sweep (top -> surface), LW:3207-3312, in blocks of four layers ----
    double radld_c = 0., radld_t = 0., radclr = 0., frac0 = 0.;
    double ec = 0., trc = 1., ek = 0., trk = 1.;   // upward flux at the top of the clean / clear stream, see above
    bool iclddn = false;
    if (writer && jw == 0) fl[nl] = 0.;
    const int nblk = (nl + 3) >> 2;
    // The layer's inputs are fetched ahead.  The gas optics come from HBM, 16 bytes per lane and layer: a whole block of
    // four layers ahead (T4 / F4 hold the block in turn, the reads of the next one are in flight meanwhile) -- with one
    // layer ahead the kernel had 16 KB in flight per CU and ran at the 2.9 TB/s that allows (Little's law at ~2 us),
    // whatever its instruction count.  The Planck functions and the aerosol optical depth of the lane's band come from
    // the column record, which the lanes of a band share (L1 / L2 hits): one layer ahead.
    const double *tgw = tg + (size_t)(4 * (nblk - 1)) * ng;   // the lane's taug slot of the block's first layer
    double T4[4], F4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 4 * (nblk - 1) + j;
        T4[j] = 0.; F4[j] = 0.;
        if (k < nl) { T4[j] = tgw[j * ng]; F4[j] = frac_read(FS, tgw + j * ng, k); }
    }
    double2 qu = make_double2(0., 0.), qw = qu;
    double qt = 0.;
    bp_fetch(nblk - 1, qu, qw, qt);
    for (int b = nblk - 1; b >= 0; --b) {
        double v[4] = {0., 0., 0., 0.};
        double Tn[4] = {0., 0., 0., 0.}, Fn[4] = {0., 0., 0., 0.};
        double2 qun = make_double2(0., 0.), qwn = qun;
        double qtn = 0.;
        if (b > 0) {
            bp_fetch(b - 1, qun, qwn, qtn);
            tgw -= 4 * ng;
#pragma unroll
            for (int j = 0; j < 4; ++j) { Tn[j] = tgw[j * ng]; Fn[j] = frac_read(FS, tgw + j * ng, 4 * (b - 1) + j); }
        }
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int k = b * 4 + j;
            if (k < nl) {
                const double taug = T4[j], plfrac = frac_value(FS, F4[j], k);
                double blay, dplankdn, dplankup, tak;
                bp_take(j, qu, qw, blay, dplankdn, dplankup, tak);
                double tcv = cloud ? bp_take1(j, qt) : 0.;   // (here, where every lane takes part in the exchange)
                const bool lay_cld = k < 64 ? (lmask >> k) & 1 : (deep && k < 128 && ((lmask_hi >> (k - 64)) & 1));
                if (!lay_cld) {
                    double at, bbd, bbu;
                    layer_src<TBL>(tb, LTBL, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, at, bbd, bbu);
                    radld_t = relax(radld_t, bbd, at);
                    if (col_cloudy) {
                        radclr = iclddn ? relax(radclr, bbd, at) : radld_t;   // LW:3304-3311
                        ek = __builtin_fma(bbu * at, trk, ek);
                        trk = __builtin_fma(-trk, at, trk);
                    }
                    if (clean_own) {
                        // without aerosol in this layer (buffer layers, clean air) the clean call equals the one above
                        if (__any(tak != 0.)) layer_src<TBL>(tb, LTBL, taug, secdiff, plfrac, blay, dplankup, dplankdn, at, bbd, bbu);
                        radld_c = relax(radld_c, bbd, at);
                        ec = __builtin_fma(bbu * at, trc, ec);
                        trc = __builtin_fma(-trc, at, trc);
                    }
                } else {
                    iclddn = true;
                    const bool cl = k < 64 ? (cmask >> k) & 1 : (deep ? (s_cmhi[tid] >> (k - 64)) & 1 : 0);
                    const double cfm = cl ? 1. : 0.;
                    double odcld = 0., efcl = 0.;
                    if (cl) {
                        if (tcv < 0.) { bad = (int)(-tcv); tcv = 0.; }   // a size out of bounds, used (lw_kernel.h)
                        odcld = secdiff * tcv;   // LW:3172-3176
                        efcl = 1. - exp_neg(odcld);
                    }
                    LwCldSrc s;
                    layer_src_cld<TBL>(tb, LTBL, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, odcld, s);
                    double gassrc = s.bbd * s.atrans;
                    radld_t = radld_t - radld_t * (s.atrans + efcl * (1. - s.atrans)) + gassrc + cfm * (s.bbdtot * s.atot - gassrc);
                    radclr = radclr + (s.bbd - radclr) * s.atrans;
                    ek = __builtin_fma(s.bbugas * s.atrans, trk, ek);
                    trk = __builtin_fma(-trk, s.atrans, trk);
                    if (clean_own) {
                        layer_src_cld<TBL>(tb, LTBL, taug, secdiff, plfrac, blay, dplankup, dplankdn, odcld, s);
                        gassrc = s.bbd * s.atrans;
                        const double A = s.atrans + efcl * (1. - s.atrans);
                        radld_c = radld_c - radld_c * A + gassrc + cfm * (s.bbdtot * s.atot - gassrc);
                        const double gasu = s.bbugas * s.atrans;
                        ec = __builtin_fma(gasu + cfm * (s.bbutot * s.atot - gasu), trc, ec);
                        trc = __builtin_fma(-trc, A, trc);
                    }
                }
                v[j] = wgt * radld_t;
                frac0 = plfrac;
            }
        }
        // downward flux at the bottom of the block's layers = levels 4b .. 4b+3
        const double qs = row_sum4(v[0], v[1], v[2], v[3], lane);
        if (writer && b * 4 + jw < nl) fl[b * 4 + jw] = qs;
        if (b > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { T4[j] = Tn[j]; F4[j] = Fn[j]; }
            qu = qun; qw = qwn; qt = qtn;
        }
    }
    // ---- surface: emission + reflection, LW:3320-3324 ----
    const double rad0 = frac0 * plankbnd;
    const double reflect = 1. - semiss;
    double radlu_t = __builtin_fma(reflect, radld_t, rad0);
    {
        const double radlu_c = __builtin_fma(reflect, radld_c, rad0), radclru = __builtin_fma(reflect, radclr, rad0);
        const double topc = __builtin_fma(trc, radlu_c, ec), topk = __builtin_fma(trk, radclru, ek);
        const double q0 = row_sum4(wgt * radld_c, wgt * radlu_c, wgt * topc, wgt * radlu_t, lane);
        const double q1 = row_sum4(wgt * radclr, wgt * radclru, wgt * topk, 0., lane);
        if (writer) {
            if (jw < 3) { fl[2 * nlev + jw] = q0; fl[2 * nlev + 4 + jw] = q1; }   // surface down, surface up, top up
            else fl[nlev] = q0;                                        // upward flux of the call with aerosol at the surface
        }
    }
    // ---- upward sweep (surface -> top), LW:3331-3356: the call with aerosol ----
    // (the gas optics are read a second time and the layer optics evaluated again: with the pair table that is 25
    // instructions; storing (A, S) on the way down instead was 46 GB more traffic and slower, r03)
    {
        // T4 / F4 and qu / qw hold block 0 after the downward sweep
        for (int b = 0; b < nblk; ++b) {
            double v[4] = {0., 0., 0., 0.};
            double Tn[4] = {0., 0., 0., 0.}, Fn[4] = {0., 0., 0., 0.};
            double2 qun = make_double2(0., 0.), qwn = qun;
            double qtn = 0.;
            if (b + 1 < nblk) {
                bp_fetch(b + 1, qun, qwn, qtn);
                tgw += 4 * ng;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * (b + 1) + j;
                    if (k < nl) { Tn[j] = tgw[j * ng]; Fn[j] = frac_read(FS, tgw + j * ng, k); }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = b * 4 + j;
                if (k < nl) {
                    const double taug = T4[j], plfrac = frac_value(FS, F4[j], k);
                    double blay, dplankdn, dplankup, tak;
                    bp_take(j, qu, qw, blay, dplankdn, dplankup, tak);
                    double tcv = cloud ? bp_take1(j, qt) : 0.;
                    const bool lay_cld = k < 64 ? (lmask >> k) & 1 : (deep && k < 128 && ((lmask_hi >> (k - 64)) & 1));
                    if (!lay_cld) {
                        double at, bbd, bbu;
                        layer_src<TBL>(tb, LTBL, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, at, bbd, bbu);
                        radlu_t = relax(radlu_t, bbu, at);
                    } else {
                        const bool cl = k < 64 ? (cmask >> k) & 1 : (deep ? (s_cmhi[tid] >> (k - 64)) & 1 : 0);
                        const double cfm = cl ? 1. : 0.;
                        double odcld = 0., efcl = 0.;
                        if (cl) {
                            if (tcv < 0.) { bad = (int)(-tcv); tcv = 0.; }
                            odcld = secdiff * tcv;
                            efcl = 1. - exp_neg(odcld);
                        }
                        LwCldSrc s;
                        layer_src_cld<TBL>(tb, LTBL, taug + tak, secdiff, plfrac, blay, dplankup, dplankdn, odcld, s);
                        const double gassrc = s.bbugas * s.atrans;
                        radlu_t = radlu_t - radlu_t * (s.atrans + efcl * (1. - s.atrans)) + gassrc + cfm * (s.bbutot * s.atot - gassrc);
                    }
                    v[j] = wgt * radlu_t;
                }
            }
            // upward flux at the top of the block's layers = levels 4b+1 .. 4b+4
            const double qs = row_sum4(v[0], v[1], v[2], v[3], lane);
            if (writer && b * 4 + jw < nl) fl[nlev + b * 4 + 1 + jw] = qs;
#pragma unroll
            for (int j = 0; j < 4; ++j) { T4[j] = Tn[j]; F4[j] = Fn[j]; }
            qu = qun; qw = qwn; qt = qtn;
        }
    }
    if (bad && live) atomicMax(a.status, (bad << 28) | (1 + col));
    __syncthreads();

    // ======================= epilogue: lanes = levels; two waves per column (waves 14, 15 idle) =======================
    const int ec_col = wave >> 1;
    if (ec_col < ncols) {
        const int etid = tid & 127;
        int ecol = a.col0 + NCOL * grp + ec_col;
        if (a.colmap) ecol = a.colmap[ecol];
        const int eit = ecol % a.ni_t, ejt = ecol / a.ni_t;
        const size_t ei0 = (size_t)(a.its - a.ims + eit), ej0 = (size_t)(a.jts - a.jms + ejt), k0 = (size_t)(a.kts - a.kms);
        const size_t ei2 = ej0 * a.ni_m + ei0;
        const size_t sk = a.ni_m, sj = (size_t)a.ni_m * a.nk_m;
        const size_t base3 = ej0 * sj + k0 * sk + ei0;
        const double *pz = a.rec + (size_t)(NCOL * grp + ec_col) * lw_rec_doubles(nl) + lw_rec_base_doubles(nl) - (nl + 2);
        const double *f0;
        if constexpr (TBL) f0 = a.flx + (size_t)(NCOL * grp + ec_col) * lw_flux_doubles(nl);
        else f0 = lds_cols + ec_col * cold + recn;
        const bool e_cloudy = cloud && (s_lmask[ec_col][0] != 0 || (deep && s_lmask[ec_col][1] != 0));
        // the chunk sums in chunk order
        auto tot = [&](size_t q) {
            double t = f0[q];
#pragma unroll
            for (int c9 = 1; c9 < RTF_NCHUNK; ++c9) t = t + f0[c9 * ws + q];
            return t * tb.fluxfac;
        };
        auto dnl = [&](int lev) { return tot(lev); };
        auto upl = [&](int lev) { return tot(nlev + lev); };
        auto sm = [&](int q) { return tot(2 * nlev + q); };
        for (int l = etid; l < nk; l += 128) {
            if (a.rthratenlw) {
                // htr(l) = heatfac*(fnet(l)-fnet(l+1))/(pz(l)-pz(l+1)), LW:3402; then LW:12687-12690
                const double fnet0 = upl(l) - dnl(l), fnet1 = upl(l + 1) - dnl(l + 1);
                const double htr = tb.heatfac * (fnet0 - fnet1) / (pz[l] - pz[l + 1]);
                const double tten = htr / 86400.;
                str(a.rthratenlw, base3 + (size_t)l * sk, r4, tten / ldr(a.pi3d, base3 + (size_t)l * sk, r4));
            }
        }
        if (etid == 0) {
            const double dnb = dnl(0), upb = upl(0), upt = upl(nl), dnt = dnl(nl);
            // clean sky: own stream, a copy of the call with aerosol when there is no aerosol, zero when switched off
            const double dnbc = clean_own ? sm(0) : (clean_on ? dnb : 0.), upbc = clean_own ? sm(1) : (clean_on ? upb : 0.),
                         uptc = clean_own ? sm(2) : (clean_on ? upt : 0.), dntc = clean_on ? dnt : 0.;
            // clear sky (aerosol call): own stream in a cloudy column, else the total-sky values (LW:3304-3311)
            const double dnbk = e_cloudy ? sm(4) : dnb, upbk = e_cloudy ? sm(5) : upb, uptk = e_cloudy ? sm(6) : upt;
            if (a.glw) str(a.glw, ei2, r4, dnb);
            if (a.olr) str(a.olr, ei2, r4, upt);
            if (a.lwcf) str(a.lwcf, ei2, r4, uptk - upt);   // LW:12658
            if (a.lwupt) {
                str(a.lwupb, ei2, r4, upb); str(a.lwupbc, ei2, r4, upbk); str(a.lwdnb, ei2, r4, dnb); str(a.lwdnbc, ei2, r4, dnbk);
                str(a.lwupbcln, ei2, r4, upbc); str(a.lwdnbcln, ei2, r4, dnbc);
                str(a.lwupt, ei2, r4, upt); str(a.lwuptc, ei2, r4, uptk); str(a.lwdnt, ei2, r4, dnt); str(a.lwdntc, ei2, r4, dnt);
                str(a.lwuptcln, ei2, r4, uptc); str(a.lwdntcln, ei2, r4, dntc);
            }
        }
    }
}

__global__ void __launch_bounds__(LW_THREADS) lw_rt_kernel(const LwKernelArgs a) { lw_rt_body<false>(a); }
__global__ void __launch_bounds__(RTF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
lw_rt_fast_kernel(const LwKernelArgs a) { lw_rt_fast<false, false, false>(a); }
__global__ void __launch_bounds__(RTF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
lw_rt_fast_tbl_kernel(const LwKernelArgs a) { lw_rt_fast<false, false, true>(a); }
__global__ void __launch_bounds__(RTF_CLD_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
lw_rt_fast_cld_tbl_kernel(const LwKernelArgs a) { lw_rt_fast<false, true, true>(a); }
__global__ void __launch_bounds__(RTF_CLD_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
lw_rt_fast_cld_kernel(const LwKernelArgs a) { lw_rt_fast<false, true, false>(a); }
__global__ void __launch_bounds__(RTF_CLD_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
lw_rt_fast_deep_kernel(const LwKernelArgs a) { lw_rt_fast<true, true, false>(a); }
__global__ void __launch_bounds__(LW_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
lw_rt_deep_kernel(const LwKernelArgs a) { lw_rt_body<true>(a); }

// the pair table of layer_src, filled once at init by the arithmetic the kernels use for a re-evaluated entry
__global__ void lw_fill_pairs_kernel(double *out, double bpade)
{
    const int itr = blockIdx.x * blockDim.x + threadIdx.x;
    if (itr > LW_NTBL) return;
    reinterpret_cast<double2 *>(out)[itr] = lw_pair_entry(bpade, itr);
}
hipError_t lw_fill_pairs(double *out, double bpade)
{
    hipLaunchKernelGGL(lw_fill_pairs_kernel, dim3((LW_NTBL + 256) / 256), dim3(256), 0, nullptr, out, bpade);
    hipError_t e = hipGetLastError();
    return e != hipSuccess ? e : hipDeviceSynchronize();
}

// ---------------------------------------------------------------------------
size_t lw_gas_lds_bytes(int nl, bool stage_cont)
{
    size_t nd = (size_t)nl * (10 + LS_COUNT + LR_COUNT * 2 + 2 + 1) + 2 * (nl + 1) + 3 * PP_DOUBLES +
                (stage_cont ? 3 * CONT_DOUBLES : 0);
    size_t ni = (size_t)nl * 5;
    return nd * sizeof(double) + ((ni * sizeof(int) + 7) & ~size_t(7));
}
size_t lw_rt_fast_lds_bytes(int nl, bool cloud)
{
    return rtf_cols(cloud) * rtf_col_doubles(nl, cloud) * sizeof(double);
}
size_t lw_rt_lds_bytes(int nl, bool cloud)
{
    return (lw_rec_tc_offset(nl) + (size_t)3 * (cloud ? 6 : 4) * (nl + 1)) * sizeof(double);
}

// band descriptors and the order of the gas-optics segments (see lw_gas_kernel)
hipError_t lw_upload_segments(const LwGDesc *gdesc)
{
    LwGDesc band[LW_NBND];
    int g0[LW_NBND], order[2 * LW_NBND];
    double cost[2 * LW_NBND];
    int g = 0;
    for (int ib = 0; ib < LW_NBND; ++ib) {
        if (g >= LW_NGPT || gdesc[g].band != ib || gdesc[g].ig != 0) return hipErrorInvalidValue;
        band[ib] = gdesc[g];
        g0[ib] = g;
        const int ng = gdesc[g].ng, lpi = 64 / ng;
        for (int up = 0; up < 2; ++up) {
            // rough relative cost: blocks of a 2:1 lower / upper split x the table reads of one item
            const LwRegime &R = up ? gdesc[g].up : gdesc[g].lo;
            const int type = (R.cfg >> LC_TYPE_SHIFT) & 3;
            // the two item kinds of the gas kernel: a regime without a second key species has no species-indexed
            // minor table, no interpolated Planck fraction and no pressure correction together with two key species
            if (type != 2) {
                for (int m = 0; m < 3; ++m)
                    if (R.minor[m].slot != LS_ZERO && R.minor[m].nsp != 1) return hipErrorInvalidValue;
                if (((R.cfg >> LC_FRAC_SHIFT) & 3) == 2) return hipErrorInvalidValue;
            } else if ((R.cfg >> LC_CORR_SHIFT) & 3) return hipErrorInvalidValue;
            double w = 1. + (type == 2 ? 10. : (type == 1 ? 8. : 0.));
            if (R.cfg & LC_EDGES) w += 3.;
            if (R.cfg & LC_SELF) w += 2.;
            if (R.cfg & LC_FOR) w += 2.;
            for (int m = 0; m < 3; ++m)
                if (R.minor[m].slot != LS_ZERO) w += R.minor[m].nsp > 1 ? 6. : 4.;
            if (((R.cfg >> LC_FRAC_SHIFT) & 3) == 2) w += 4.;
            const int layers = up ? 20 : 40;
            cost[2 * ib + up] = w * ((layers + lpi - 1) / lpi);
            order[2 * ib + up] = 2 * ib + up;
        }
        g += ng;
    }
    if (g != LW_NGPT) return hipErrorInvalidValue;
    std::stable_sort(order, order + 2 * LW_NBND, [&](int x, int y) { return cost[x] > cost[y]; });
    LwSegRec rec[2 * LW_NBND];
    for (int s = 0; s < 2 * LW_NBND; ++s) {
        const int ib = order[s] >> 1, up = order[s] & 1;
        rec[s].R = up ? band[ib].up : band[ib].lo;
        rec[s].upper = up; rec[s].g0 = g0[ib]; rec[s].ng = band[ib].ng; rec[s].stride = band[ib].stride;
        rec[s].off_self = band[ib].off_self; rec[s].off_for = band[ib].off_for;
    }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_lwseg), rec, sizeof rec);
}

hipError_t lw_upload_constants(const double *preflog, const double *tref)
{
    static const double o3sum[31] = {
        5.297E-8, 5.852E-8, 6.579E-8, 7.505E-8, 8.577E-8, 9.895E-8, 1.175E-7, 1.399E-7, 1.677E-7, 2.003E-7,
        2.571E-7, 3.325E-7, 4.438E-7, 6.255E-7, 8.168E-7, 1.036E-6, 1.366E-6, 1.855E-6, 2.514E-6, 3.240E-6,
        4.033E-6, 4.854E-6, 5.517E-6, 6.089E-6, 6.689E-6, 1.106E-5, 1.462E-5, 1.321E-5, 9.856E-6, 5.960E-6,
        5.960E-6};
    static const double ppsum[31] = {
        955.890, 850.532, 754.599, 667.742, 589.841, 519.421, 455.480, 398.085, 347.171, 301.735, 261.310,
        225.360, 193.419, 165.490, 141.032, 120.125, 102.689, 87.829, 75.123, 64.306, 55.086, 47.209, 40.535,
        34.795, 29.865, 19.122, 9.277, 4.660, 2.421, 1.294, 0.647};
    static const double o3win[31] = {
        4.629E-8, 4.686E-8, 5.017E-8, 5.613E-8, 6.871E-8, 8.751E-8, 1.138E-7, 1.516E-7, 2.161E-7, 3.264E-7,
        4.968E-7, 7.338E-7, 1.017E-6, 1.308E-6, 1.625E-6, 2.011E-6, 2.516E-6, 3.130E-6, 3.840E-6, 4.703E-6,
        5.486E-6, 6.289E-6, 6.993E-6, 7.494E-6, 8.197E-6, 9.632E-6, 1.113E-5, 1.146E-5, 9.389E-6, 6.135E-6,
        6.135E-6};
    static const double ppwin[31] = {
        955.747, 841.783, 740.199, 649.538, 568.404, 495.815, 431.069, 373.464, 322.354, 277.190, 237.635,
        203.433, 174.070, 148.949, 127.408, 108.915, 93.114, 79.551, 67.940, 58.072, 49.593, 42.318, 36.138,
        30.907, 26.362, 16.423, 7.583, 3.620, 1.807, 0.938, 0.469};
    double o3ann[31], ppwrkh[32];
    o3ann[0] = 0.5 * (o3sum[0] + o3win[0]);
    for (int k = 1; k < 31; ++k)
        o3ann[k] = o3win[k - 1] + (o3win[k] - o3win[k - 1]) / (ppwin[k] - ppwin[k - 1]) * (ppsum[k] - ppwin[k - 1]);
    for (int k = 1; k < 31; ++k) o3ann[k] = 0.5 * (o3ann[k] + o3sum[k]);
    ppwrkh[0] = 1100.;
    for (int k = 1; k < 31; ++k) ppwrkh[k] = (ppsum[k] + ppsum[k - 1]) / 2.;
    ppwrkh[31] = 0.;
    hipError_t e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_lw_o3ann), o3ann, sizeof o3ann)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_lw_ppwrkh), ppwrkh, sizeof ppwrkh)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_lw_preflog), preflog, 59 * sizeof(double))) != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_lw_tref), tref, 59 * sizeof(double));
}

hipError_t lw_upload_retab()
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_lw_retab), CLD_RETAB, sizeof(double) * 95);
}

hipError_t lw_launch(LwKernelArgs a, int ncol, double *scratch, int cap_cols, hipStream_t stream)
{
    const int nl = a.nlayers;
    // the continuum rows go through LDS when that still leaves four workgroups per CU
    a.stage_cont = lw_gas_lds_bytes(nl, true) + 64 <= 160 * 1024 / 4 ? 1 : 0;
    const size_t lds_gas = lw_gas_lds_bytes(nl, a.stage_cont != 0), lds_rt = lw_rt_lds_bytes(nl, a.icloud != 0);
    const size_t lds_fast = lw_rt_fast_lds_bytes(nl, a.icloud != 0);
    // the cloud-free transfer kernel with the pair table in LDS (RRTMG_HIP_LW_TABLE=0: the re-evaluating one)
    static const bool use_tbl = [] { const char *e = std::getenv("RRTMG_HIP_LW_TABLE"); return !(e && e[0] == '0'); }();
    const size_t lds_tbl = (size_t)LW_PAIR_COUNT * 2 * sizeof(double) +
                           (a.icloud != 0 ? (size_t)RTF_CLD_COLS * nl * sizeof(double) : 0);   // (+ the layer cloud fractions)
    const bool tbl_fits = lds_tbl <= 160 * 1024 - 256;
    // the three-column transfer kernel holds three column records; beyond what a workgroup may have (very deep
    // columns) the one-column kernel serves the call
    const bool fast_fits = lds_fast <= 160 * 1024 - 2048;
    {
        // opt-in to more than 48 KB of dynamic LDS, once per size: concurrent tile threads launch from here (mem = 0
        // calls run outside the engine lock), so the bookkeeping has its own lock; the launches stay outside it
        static std::mutex attr_mutex;
        static size_t attr_gas = 0, attr_rt = 0, attr_fast = 0, attr_tbl = 0;
        std::lock_guard<std::mutex> lock(attr_mutex);
        auto opt_in = [](const void *f, size_t bytes) {
            return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        };
        if (lds_gas > attr_gas && lds_gas > 48 * 1024) {
            hipError_t e = opt_in(reinterpret_cast<const void *>(lw_gas_kernel), lds_gas);
            if (e != hipSuccess) return e;
            attr_gas = lds_gas;
        }
        if (lds_rt > attr_rt && lds_rt > 48 * 1024) {
            hipError_t e = opt_in(reinterpret_cast<const void *>(lw_rt_kernel), lds_rt);
            if (e == hipSuccess) e = opt_in(reinterpret_cast<const void *>(lw_rt_deep_kernel), lds_rt);
            if (e != hipSuccess) return e;
            attr_rt = lds_rt;
        }
        if (fast_fits && lds_fast > attr_fast && lds_fast > 48 * 1024) {
            hipError_t e = opt_in(reinterpret_cast<const void *>(lw_rt_fast_kernel), lds_fast);
            if (e == hipSuccess) e = opt_in(reinterpret_cast<const void *>(lw_rt_fast_cld_kernel), lds_fast);
            if (e == hipSuccess) e = opt_in(reinterpret_cast<const void *>(lw_rt_fast_deep_kernel), lds_fast);
            if (e != hipSuccess) return e;
            attr_fast = lds_fast;
        }
        if (use_tbl && tbl_fits && lds_tbl > attr_tbl) {
            hipError_t e = opt_in(reinterpret_cast<const void *>(lw_rt_fast_tbl_kernel), lds_tbl);
            if (e == hipSuccess) e = opt_in(reinterpret_cast<const void *>(lw_rt_fast_cld_tbl_kernel), lds_tbl);
            if (e != hipSuccess) return e;
            attr_tbl = lds_tbl;
        }
    }
    const bool deep = a.icloud != 0 && a.nk > 64;
    if (!scratch || cap_cols <= 0) return hipErrorInvalidValue;
    if (std::getenv("RRTMG_HIP_DEBUG_OCC")) {   // development aid: what the runtime expects to keep resident per CU
        int ng = 0, nf = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&ng, lw_gas_kernel, LW_THREADS, lds_gas);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, lw_rt_fast_kernel, RTF_THREADS, lds_fast);
        fprintf(stderr, "lw_gas_kernel: %d workgroups/CU (LDS %zu B); lw_rt_fast_kernel: %d workgroups/CU (LDS %zu B)\n",
                ng, lds_gas, nf, lds_fast);
    }
    a.taug = scratch;
    a.frac = a.taug + (size_t)cap_cols * nl * LW_NGPT;
    a.rec = a.frac + (size_t)cap_cols * nl * LW_NGPT;
    a.flx = a.rec + (size_t)cap_cols * lw_rec_doubles(nl);
    for (int c0 = 0; c0 < ncol; c0 += cap_cols) {
        const int n = ncol - c0 < cap_cols ? ncol - c0 : cap_cols;
        a.col0 = c0;
        a.nbatch = n;
        hipLaunchKernelGGL(lw_gas_kernel, dim3(n), dim3(LW_THREADS), lds_gas, stream, a);
        if (!a.lwupflx && !deep && use_tbl && tbl_fits) {   // no layer-flux profiles requested: the pair table in LDS
            const int nc = rtf_cols(a.icloud != 0), ngrp = (n + nc - 1) / nc;
            if (a.icloud != 0) hipLaunchKernelGGL(lw_rt_fast_cld_tbl_kernel, dim3(ngrp), dim3(RTF_CLD_THREADS), lds_tbl, stream, a);
            else hipLaunchKernelGGL(lw_rt_fast_tbl_kernel, dim3(ngrp), dim3(RTF_THREADS), lds_tbl, stream, a);
        } else if (!a.lwupflx && fast_fits) {   // no layer-flux profiles requested: what the radiation driver calls
            const int nc = rtf_cols(a.icloud != 0), ngrp = (n + nc - 1) / nc;
            if (deep) hipLaunchKernelGGL(lw_rt_fast_deep_kernel, dim3(ngrp), dim3(RTF_CLD_THREADS), lds_fast, stream, a);
            else if (a.icloud != 0) hipLaunchKernelGGL(lw_rt_fast_cld_kernel, dim3(ngrp), dim3(RTF_CLD_THREADS), lds_fast, stream, a);
            else hipLaunchKernelGGL(lw_rt_fast_kernel, dim3(ngrp), dim3(RTF_THREADS), lds_fast, stream, a);
        } else if (deep) hipLaunchKernelGGL(lw_rt_deep_kernel, dim3(n), dim3(LW_THREADS), lds_rt, stream, a);
        else hipLaunchKernelGGL(lw_rt_kernel, dim3(n), dim3(LW_THREADS), lds_rt, stream, a);
    }
    return hipGetLastError();
}

}  // namespace rrtmg
