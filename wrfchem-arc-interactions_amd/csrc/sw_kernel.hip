// Shortwave column kernel: RRTMG_SWRAD + rrtmg_sw (aerosol and clean-sky
// calls in one launch) for gfx950.
//
// Reference path (phys/module_ra_rrtmg_sw.F): RRTMG_SWRAD preprocessing
// SW:10320-11071, inatm_sw SW:9735-9790, setcoef_sw SW:2734-2990, taumol_sw
// SW:3081-4540, spcvmc_sw SW:8344-8650, reftra_sw SW:2422-2701, vrtqdr_sw
// SW:7922-8046, rrtmg_sw post-processing SW:9391-9486, outputs SW:11125-11197.
//
// Mapping (MI355X-first, not the reference's loop nest):
//   * one 128-thread workgroup (two wavefronts) per daylit column; blockIdx is
//     remapped so that the workgroups of one XCD walk consecutive i (they share
//     the 64-B sectors of the (i,k,j) state in that XCD's L2).
//   * prologue: lanes = layers.  Gathers the column, builds the 1-D profile,
//     O3 climatology, column amounts and the setcoef interpolation factors once
//     and leaves them in LDS; aerosol wavelength->band mapping into LDS.
//   * main phase: lanes = g-points, wave w = g-chunk w (56 g-points).  A lane
//     carries BOTH calls of its g-point through one upward sweep: the gas
//     optics (one data-driven formula per g-point descriptor, so the 14 bands
//     never diverge; table reads issued one layer ahead) are evaluated once and
//     feed two streams:
//       P  the call with aerosol.  Its fluxes are needed at every level
//          (heating rates), so each layer's two-stream result (5 doubles per
//          lane) and every 4th level of the upward recurrence go to the
//          column's scratch slice in HBM (coalesced 512-B rows) and the
//          downward sweep reads them back block by block (vrtqdr_sw as the
//          reference orders it).
//       S  the clean-sky call.  RRTMG_SWRAD uses only its TOA and surface
//          fluxes (SW:11140-11150; the layer-flux profiles are optional
//          arguments the patched radiation_driver does not receive), and those
//          follow from the two-stream properties of the whole atmosphere.  The
//          adding step is associative, so the same upward sweep composes them
//          layer by layer (reflectance from above and below, total / diffuse /
//          direct transmittance): no second sweep, nothing stored.
//     A cloudy column runs a second, S-only sweep for the clear-sky fluxes of
//     the aerosol call.  When the optional flux profiles are requested every
//     call runs as a P stream (kernel instantiation PROF).
//   * per-level spectral sums (SW:8629-8634) are wave reductions of eight
//     values at a time (v_permlane32_swap / v_permlane16_swap butterflies, then
//     DPP row steps); the two chunk waves meet in LDS; the epilogue (lanes =
//     levels) forms net fluxes, heating rates and writes the WRF arrays.
// No MFMA: there is no contraction anywhere on this path.
#include <hip/hip_runtime.h>

#include <mutex>

#include <cmath>

#include "rrtmg_common.h"
#include "sw_kernel.h"
#include "cld_retab.h"
#include "sw_aer_tables.h"
#include "mcica_device.h"

namespace rrtmg {

namespace {

constexpr int SW_SCR_VALS = 5;     // ref, refd, tra, trad, dbt per (layer, lane)
constexpr int BLK = 4;            // levels between checkpoints of the upward sweep = layers per block on the way down
constexpr int CHUNK = 56;         // g-points per wave (112 = 2 x 56)
constexpr int SCR_ROW = CHUNK;    // lanes of a scratch row: 448 bytes = seven whole 64-byte lines, no dead bytes (the
                                  // eight idle lanes of a wave neither store nor own a slot)
constexpr int NTHREADS = 128;     // two waves: g-chunks 0 and 1
constexpr int NWAVE = NTHREADS / 64;
#ifndef SW_WAVES_PER_SIMD_V
#define SW_WAVES_PER_SIMD_V 3
#endif
constexpr int SW_WAVES_PER_SIMD = SW_WAVES_PER_SIMD_V; // register budget: 12 waves per CU

// reference atmosphere, SW:3019-3045 (preflog is 6.96 - 0.2*i)
__constant__ double c_tref[59] = {
    2.9420e+02, 2.8799e+02, 2.7894e+02, 2.6925e+02, 2.5983e+02, 2.5017e+02, 2.4077e+02, 2.3179e+02,
    2.2306e+02, 2.1578e+02, 2.1570e+02, 2.1570e+02, 2.1570e+02, 2.1706e+02, 2.1858e+02, 2.2018e+02,
    2.2174e+02, 2.2328e+02, 2.2479e+02, 2.2655e+02, 2.2834e+02, 2.3113e+02, 2.3401e+02, 2.3703e+02,
    2.4022e+02, 2.4371e+02, 2.4726e+02, 2.5085e+02, 2.5457e+02, 2.5832e+02, 2.6216e+02, 2.6606e+02,
    2.6999e+02, 2.7340e+02, 2.7536e+02, 2.7568e+02, 2.7372e+02, 2.7163e+02, 2.6955e+02, 2.6593e+02,
    2.6211e+02, 2.5828e+02, 2.5360e+02, 2.4854e+02, 2.4348e+02, 2.3809e+02, 2.3206e+02, 2.2603e+02,
    2.2000e+02, 2.1435e+02, 2.0887e+02, 2.0340e+02, 1.9792e+02, 1.9290e+02, 1.8809e+02, 1.8329e+02,
    1.7849e+02, 1.7394e+02, 1.7212e+02};
__constant__ double c_preflog[59] = {
    6.9600e+00, 6.7600e+00, 6.5600e+00, 6.3600e+00, 6.1600e+00, 5.9600e+00, 5.7600e+00, 5.5600e+00,
    5.3600e+00, 5.1600e+00, 4.9600e+00, 4.7600e+00, 4.5600e+00, 4.3600e+00, 4.1600e+00, 3.9600e+00,
    3.7600e+00, 3.5600e+00, 3.3600e+00, 3.1600e+00, 2.9600e+00, 2.7600e+00, 2.5600e+00, 2.3600e+00,
    2.1600e+00, 1.9600e+00, 1.7600e+00, 1.5600e+00, 1.3600e+00, 1.1600e+00, 9.6000e-01, 7.6000e-01,
    5.6000e-01, 3.6000e-01, 1.6000e-01, -4.0000e-02, -2.4000e-01, -4.4000e-01, -6.4000e-01, -8.4000e-01,
    -1.0400e+00, -1.2400e+00, -1.4400e+00, -1.6400e+00, -1.8400e+00, -2.0400e+00, -2.2400e+00, -2.4400e+00,
    -2.6400e+00, -2.8400e+00, -3.0400e+00, -3.2400e+00, -3.4400e+00, -3.6400e+00, -3.8400e+00, -4.0400e+00,
    -4.2400e+00, -4.4400e+00, -4.6400e+00};
// O3 climatology of INIRAD/O3DATA, LW:12730-12756 (annual mean built on the host)
__constant__ double c_o3ann[31];
__constant__ double c_ppwrkh[32];
// band mid wavelengths (um) of the 14 SW intervals, from wavemin/wavemax SW:10080-10085
__constant__ double c_wavemid[SW_NBND];
__constant__ double c_lnwave[SW_NBND];   // ln(0.4 / wavemid)

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

// ---- wave reductions in the VALU (data-parallel primitives, no LDS round trip) ----
// gfx950 lane-swap instructions: after swap32(a, b) a = [a.lanes0-31, b.lanes0-31], b = [a.lanes32-63, b.lanes32-63];
// after swap16(a, b) a = rows [a0, b0, a2, b2], b = rows [a1, b1, a3, b3] (row = 16 lanes).  a + b then holds the
// pairwise-folded partial sums of BOTH inputs: two values are folded for the price of one.
typedef unsigned sw_uint2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fold32(double a, double b)
{
    const sw_uint2 lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const sw_uint2 hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
__device__ __forceinline__ double fold16(double a, double b)
{
    const sw_uint2 lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const sw_uint2 hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    const int mlo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int mhi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(mhi, mlo);
}
// sum inside each row of 16 lanes, valid in every lane (row_mirror, row_half_mirror, quad_perm [2,3,0,1], [1,0,3,2])
__device__ __forceinline__ double row_sum(double v)
{
    v = dpp_add<0x140>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0xB1>(v);
    return v;
}
// wave sums of four values: the result's row r (lanes 16r..16r+15) holds the sum of v[WSUM_ROW(r)]
__device__ __forceinline__ double wave_sum4(double v0, double v1, double v2, double v3)
{
    return row_sum(fold16(fold32(v0, v1), fold32(v2, v3)));   // rows: v0, v2, v1, v3
}
__device__ __forceinline__ int wsum_index(int row) { return ((row & 1) << 1) | (row >> 1); }

// Reciprocal / quotient without the IEEE special-case scaffolding of the compiler's fp64 division
// (v_div_scale x2, v_div_fmas, v_div_fixup).  Operands on this path are normal-range and non-zero (the reference
// guards its own denominators with max(.,1e-30)).  Measured on MI355X over 4 M random operand pairs
// (tools/probes/rcp_probe.hip): v_rcp_f64 is good to 2^-24.4; one Newton step brings the reciprocal to 2.2e-15 and
// the quotient with its residual correction to the correctly rounded result in every case.
// frcp: two Newton steps (1.1e-16).  frcp1: one step (2.2e-15), for factors of physical quantities only.
// fdiv: reciprocal with one step + residual correction -- used wherever an integer is cut from the quotient (table
// indices), so that the index is the reference's.
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
// square root of a positive normal-range number: v_rsq_f64 seed, two coupled Newton steps (Goldschmidt form) and a
// residual correction; correctly rounded in all but rare cases (<= 1 ulp), without the scaling scaffolding of the
// compiler's expansion
__device__ __forceinline__ double fsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// transmittance by lookup table or small-argument expansion (SW:8459-8466, 2575-2581).  The index arithmetic is kept
// out of FMA contraction: the reference rounds 1e4*tblind before it adds 0.5.
// ... in two halves: the table read goes out as soon as its argument is known, the value is picked where it is used
__device__ __forceinline__ double exp_issue(const SwDeviceTables &tb, double ze1)
{
    int itind;
    {
#pragma clang fp contract(off)
        const double tblind = fdiv(ze1, tb.bpade + ze1);
        const double x = 10000.0 * tblind;
        itind = (int)(x + 0.5);
    }
    itind = itind < 0 ? 0 : (itind > NTBL ? NTBL : itind);
    // a lane that will take the small-argument expansion does not use the entry: let it read entry 0, so that the
    // gather of the wave touches fewer cache lines
    itind = ze1 <= 0.06 ? 0 : itind;
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tb.exp_tbl) + (uint32_t)itind * 8u);
}
__device__ __forceinline__ double exp_pick(double ze1, double tv)
{
    const double sv = 1. - ze1 + 0.5 * ze1 * ze1;
    return ze1 <= 0.06 ? sv : tv;
}
__device__ __forceinline__ double exp_lookup(const SwDeviceTables &tb, double ze1) { return exp_pick(ze1, exp_issue(tb, ze1)); }

struct RT {
    double ref, refd, tra, trad, dbt;
};

// reftra_sw (SW:2526-2699) with the divisions folded: the reference forms zep1 = 1/zem1 and
// zep2 = 1/zem2 and divides by zdenr = zr4*zep1 + zr5*zem1 (= zdent); multiplying numerators and
// denominator by zem1 gives the same ratios over D = zr4 + zr5*zem1^2 with one reciprocal, and
// zdend = 1/((1 - zbeta*zemm)*zrkg) = 1/(zrkg - (zgamma1 - zrk)*zemm) without forming zbeta.
// zem2 is the caller's direct-beam transmittance exp(-tau/mu0): the reference looks it up twice
// (SW:2663-2665 with the 500 cap, SW:8568-8575 without; beyond the cap both are the 1e-20 floor).
__device__ __forceinline__ void reftra_fast(const SwDeviceTables &tb, double zto1, double zw, double zg,
                                            double prmuz, double zem2, RT &o)
{
    const double eps = 1.e-08, zwcrit = 0.9999995;
    double zg3 = 3. * zg;
    double zgamma1 = (8. - zw * (5. + zg3)) * 0.25;
    double zgamma2 = 3. * (zw * (1. - zg)) * 0.25;
    double zgamma3 = (2. - zg3 * prmuz) * 0.25;
    double zgamma4 = 1. - zgamma3;
    // the table read of the general branch goes out in front of the branch (a lane on the conservative side does
    // not use it): the quotient and the choice below then cover part of its latency
    const double zrk = fsqrt(fmax(zgamma1 * zgamma1 - zgamma2 * zgamma2, 0.));
    const double ze1 = fmin(zrk * zto1, 500.);
    const double tv1 = exp_issue(tb, ze1);
    double zq = fdiv(zg, 1. - zg);
    double denom = fmax((1. - (1. - zw) * (zq * zq)), 1.0E-30);
    // zwo = zw / denom; if (zwo >= zwcrit): the reference's test, SW:2557-2560, on the ROUNDED quotient.  Away from the
    // threshold the sign of zw - zwcrit*denom (one FMA) decides the same way -- the rounded quotient lies on the exact
    // quotient's side of zwcrit as soon as they are more than an ulp apart --; a lane within 1e-12 of it forms the
    // quotient as the reference does (fdiv is the correctly rounded one) and compares that.
    const double zt = __builtin_fma(-zwcrit, denom, zw);
    bool conservative = zt >= 0.;
    if (fabs(zt) <= 1.e-12 * denom) conservative = fdiv(zw, denom) >= zwcrit;
    if (conservative) {
        // conservative scattering, SW:2560-2600
        double za = zgamma1 * prmuz;
        double za1 = za - zgamma3;
        double zgt = zgamma1 * zto1;
        double rg = frcp1(1. + zgt);
        double r = (zgt - za1 * (1. - zem2)) * rg;
        double rd = zgt * rg;
        bool one = (zem2 == 1.0);
        o.ref = one ? 0.0 : r;
        o.tra = one ? 1.0 : 1. - r;
        o.refd = one ? 0.0 : rd;
        o.trad = one ? 1.0 : 1. - rd;
    } else {
        double za1 = zgamma1 * zgamma4 + zgamma2 * zgamma3;
        double za2 = zgamma1 * zgamma3 + zgamma2 * zgamma4;
        double zrp = zrk * prmuz;
        double zrp1 = 1. + zrp;
        double zrm1 = 1. - zrp;
        double zrk2 = 2. * zrk;
        double zrpp = 1. - zrp * zrp;
        double zrkg = zrk + zgamma1;
        double zr1 = zrm1 * (za2 + zrk * zgamma3);
        double zr2 = zrp1 * (za2 - zrk * zgamma3);
        double zr3 = zrk2 * (zgamma3 - za2 * prmuz);
        double zr4 = zrpp * zrkg;
        double zr5 = zrpp * (zrk - zgamma1);
        double zt1 = zrp1 * (za1 + zrk * zgamma4);
        double zt2 = zrm1 * (za1 - zrk * zgamma4);
        double zt3 = zrk2 * (zgamma4 + za1 * prmuz);
        double zem1 = exp_pick(ze1, tv1);
        double zemm = zem1 * zem1;
        double D = zr4 + zr5 * zemm;                 // zdenr * zem1
        double rD = frcp1(D);
        bool small = fabs(D) <= eps * zem1;          // |zdenr| <= eps, SW:2679
        double r = zw * (zr1 - zr2 * zemm - zr3 * (zem2 * zem1)) * rD;
        double t = zem2 - zw * (zem2 * (zt1 - zt2 * zemm) - zt3 * zem1) * rD;
        o.ref = small ? eps : r;
        o.tra = small ? zem2 : t;
        double zdend = frcp1(zrkg - (zgamma1 - zrk) * zemm);
        o.refd = zgamma2 * (1. - zemm) * zdend;
        o.trad = zrk2 * zem1 * zdend;
    }
}
// LDS image of one column (all arrays indexed by 0-based layer, bottom -> top)
struct ColView {
    double *fac00, *fac01, *fac10, *fac11;
    double *selffac, *selffrac, *forfac, *forfrac;
    double *col;        // [SP_COUNT][nlp]
    double *colmol, *pdp;
    double *taer, *waer, *gaer;   // [14][nlp]; compact: waer = {slope, value at 600 nm}[nlp] of omega, gaer of g
    bool compact;                 // WRF-Chem route only: omega and g of a band follow from two values per layer
    int *rb0, *rb1, *inds, *indf, *jp;   // row bases (0-based), continuum rows, jp
    double *flux;       // [NWAVE][nlp + 1][2] level fluxes of the P stream (call with aerosol)
    double *sums;       // [NWAVE][SUMS_N] surface / TOA sums of the waves (SUM_* slots)
    double *fluxc;      // PROF: [NWAVE][nlp + 1][2] clean-sky profile
    double *fluxk;      // PROF and CLOUD: [NWAVE][nlp + 1][2] clear-sky profile of a cloudy column
    // in-cloud state per layer (CLOUD): cloud fraction, liquid / ice / snow water path, table fractions
    double *cldf, *clwp, *ciwp, *cswp, *fliq, *fice, *fsno;
    int *iliq, *iice, *isno;   // table indices (1-based as in cldprmc_sw), 0 = size out of bounds
    int *seeds;                // [4] KISS seeds of the column
    uint64_t *cmhi;            // [NTHREADS] McICA cloud bits of layers 64.. of each lane's sub-column
    int nlp;            // padded layer count (multiple of BLK)
};
// slots of ColView::sums: surface direct / UV-visible / near-IR sums of the call with aerosol, the P stream's surface
// level, then {TOA up, TOA down, surface up, surface down} of the clean-sky (S) and clear-sky (K) streams
enum : int { SUM_DIRBB = 0, SUM_UVFD, SUM_UVDIR, SUM_NIFD, SUM_NIDIR, SUM_S = 8, SUM_K = 12, SUMS_N = 16 };

__device__ __forceinline__ ColView carve(double *base, int nlp, bool cloud, bool prof, bool compact)
{
    ColView v;
    v.nlp = nlp;
    v.compact = compact;
    double *p = base;
    v.fac00 = p; p += nlp; v.fac01 = p; p += nlp; v.fac10 = p; p += nlp; v.fac11 = p; p += nlp;
    v.selffac = p; p += nlp; v.selffrac = p; p += nlp; v.forfac = p; p += nlp; v.forfrac = p; p += nlp;
    v.col = p; p += SP_COUNT * nlp;
    v.colmol = p; p += nlp; v.pdp = p; p += nlp;
    v.taer = p; p += SW_NBND * nlp;
    v.waer = p; p += (compact ? 2 : SW_NBND) * nlp;
    v.gaer = p; p += (compact ? 2 : SW_NBND) * nlp;
    v.flux = p; p += NWAVE * (nlp + 1) * 2;
    v.sums = p; p += NWAVE * SUMS_N;
    v.fluxc = v.fluxk = nullptr;
    if (prof) {
        v.fluxc = p; p += NWAVE * (nlp + 1) * 2;
        if (cloud) { v.fluxk = p; p += NWAVE * (nlp + 1) * 2; }
    }
    v.cldf = v.clwp = v.ciwp = v.cswp = v.fliq = v.fice = v.fsno = nullptr;
    v.cmhi = nullptr;
    if (cloud) {
        v.cldf = p; p += nlp; v.clwp = p; p += nlp; v.ciwp = p; p += nlp; v.cswp = p; p += nlp;
        v.fliq = p; p += nlp; v.fice = p; p += nlp; v.fsno = p; p += nlp;
        v.cmhi = reinterpret_cast<uint64_t *>(p); p += NTHREADS;
    }
    int *q = reinterpret_cast<int *>(p);
    v.rb0 = q; q += nlp; v.rb1 = q; q += nlp; v.inds = q; q += nlp; v.indf = q; q += nlp; v.jp = q; q += nlp + 1;
    v.iliq = v.iice = v.isno = v.seeds = nullptr;
    if (cloud) { v.iliq = q; q += nlp; v.iice = q; q += nlp; v.isno = q; q += nlp; v.seeds = q; q += 4; }
    return v;
}

struct LaneG {   // per-lane (g-point) constants
    GRegime lo, up;
    int ng, off_self, off_for, band, off_rayla;
};

// gaseous + Rayleigh optical depth of layer k for this lane's g-point: the
// 14 taumolNN routines (SW:3293-4540) as one data-driven expression, in two steps so that the table reads of the
// next layer are in flight while the current layer is worked on (gas_fetch issues them, gas_finish consumes them).
struct GasPre {
    double kv[8], s0, s1, f0, f1, r0, r1, fs, speccomb;
};

__device__ __forceinline__ void gas_fetch(const SwDeviceTables &tb, const ColView &c, const LaneG &L, int k,
                                          bool lower, GasPre &p)
{
    const GRegime &R = lower ? L.lo : L.up;
    const uint32_t cf = R.cfg;
    const int nlp = c.nlp;
    const int spa = (cf >> CFG_SPA_SHIFT) & 7, spb = (cf >> CFG_SPB_SHIFT) & 7;
    const int nsp = (cf >> CFG_NSP_SHIFT) & 15;
    const double colA = c.col[spa * nlp + k], colB = c.col[spb * nlp + k];
    double speccomb = colA, fs = 0.;
    int jsi = 0;
    if (cf & CFG_BINARY) {
        // SW:3321-3326
        speccomb = colA + R.strrat * colB;
        double specparm = fdiv(colA, speccomb);
        specparm = specparm >= tb.oneminus ? tb.oneminus : specparm;
        double specmult = ((cf & CFG_MULT4) ? 4. : 8.) * specparm;
        jsi = (int)specmult;
        fs = specmult - (double)jsi;
    }
    p.fs = fs; p.speccomb = speccomb;
    const int ng = L.ng;
    // element offsets fit 32 bits (the table is < 1 MB): uniform base + 32-bit lane offset, no 64-bit address math
    auto at = [&](uint32_t e) { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tb.K) + e * 8u); };
    const uint32_t K0 = (uint32_t)R.off_major + (uint32_t)(c.rb0[k] * nsp + jsi) * (uint32_t)ng;
    const uint32_t K1 = (uint32_t)R.off_major + (uint32_t)(c.rb1[k] * nsp + jsi) * (uint32_t)ng;
    const uint32_t dT = (uint32_t)(nsp * ng);
    p.kv[0] = at(K0); p.kv[1] = at(K0 + ng); p.kv[2] = at(K0 + dT); p.kv[3] = at(K0 + dT + ng);
    p.kv[4] = at(K1); p.kv[5] = at(K1 + ng); p.kv[6] = at(K1 + dT); p.kv[7] = at(K1 + dT + ng);
    p.s0 = p.s1 = p.f0 = p.f1 = 0.;
    if (cf & (CFG_SELF | CFG_FOR)) {
        const uint32_t Sf = (uint32_t)L.off_self + (uint32_t)c.inds[k] * (uint32_t)ng;
        const uint32_t Ff = (uint32_t)L.off_for + (uint32_t)c.indf[k] * (uint32_t)ng;
        p.s0 = at(Sf); p.s1 = at(Sf + ng); p.f0 = at(Ff); p.f1 = at(Ff + ng);
    }
    p.r0 = p.r1 = 0.;
    if (cf & CFG_RAYL24) {   // SW:4130-4131
        const uint32_t Ra = (uint32_t)L.off_rayla + (uint32_t)jsi * (uint32_t)ng;
        p.r0 = at(Ra); p.r1 = at(Ra + ng);
    }
}

__device__ __forceinline__ void gas_finish(const SwDeviceTables &tb, const ColView &c, const LaneG &L, int k,
                                           bool lower, const GasPre &p, double &taug, double &taur)
{
    const GRegime &R = lower ? L.lo : L.up;
    const uint32_t cf = R.cfg;
    const int nlp = c.nlp;
    const int spm = (cf >> CFG_SPM_SHIFT) & 7;
    const double colM = c.col[spm * nlp + k];
    const double colh2o = c.col[SP_H2O * nlp + k];
    const double fs = p.fs;
    const double f00 = c.fac00[k], f10 = c.fac10[k], f01 = c.fac01[k], f11 = c.fac11[k];
    const double omf = 1. - fs;
    // SW:3342-3350 (8-point) / SW:3374-3378 (4-point: fs = 0, zero-weight terms are exact)
    double S = (omf * f00) * p.kv[0] + (fs * f00) * p.kv[1] + (omf * f10) * p.kv[2] + (fs * f10) * p.kv[3] +
               (omf * f01) * p.kv[4] + (fs * f01) * p.kv[5] + (omf * f11) * p.kv[6] + (fs * f11) * p.kv[7];
    double major = (cf & CFG_MAJOR) ? (p.speccomb * R.mfac) * S : 0.;
    // continua, SW:3351-3357
    double cont = 0.;
    if (cf & (CFG_SELF | CFG_FOR)) {
        double selft = (cf & CFG_SELF) ? c.selffac[k] * (p.s0 + c.selffrac[k] * (p.s1 - p.s0)) : 0.;
        double fort = (cf & CFG_FOR) ? c.forfac[k] * (p.f0 + c.forfrac[k] * (p.f1 - p.f0)) : 0.;
        cont = colh2o * (selft + fort);
    }
    double t = major + colM * R.minorabs + cont;
    if (cf & CFG_O2CONT) t += 4.35e-4 * c.col[SP_O2 * nlp + k] / (350.0 * 2.0);   // SW:3933
    taug = t;
    double rayl = R.rayl;
    if (cf & CFG_RAYL24) rayl = p.r0 + fs * (p.r1 - p.r0);
    taur = c.colmol[k] * rayl;
}

// ice effective radius versus temperature (retab, LW:11424-11442), uploaded by sw_upload_retab()
__constant__ double c_retab[95];
// rsrtaua, rsrpiza, rsrasya of the ECMWF aerosol types (swaerpr, SW:4932-5017), [quantity][band][type]
__constant__ double c_swaer[3][SW_NBND][6];

// Cloud optical properties of one cloudy sub-column layer: cldprmc_sw, SW:2062-2388, for inflag >= 2, liqflag = 1,
// iceflag = 3, 4 or 5 (what RRTMG_SWRAD can select, SW:10514-10604).  Table indices and interpolation fractions
// were prepared per layer by the prologue.  bad: 1 ice, 2 snow, 3 liquid size out of bounds (fatal in the reference).
__device__ __forceinline__ void cloud_optics(const double *T, const ColView &c, int band, int k, int iceflag,
                                             double &taucmc, double &ssacmc, double &asmcmc, double &taormc, int &bad)
{
    const double cldmin = 1.e-20;
    const double ciwp = c.ciwp[k], clwp = c.clwp[k], cswp = iceflag == 5 ? c.cswp[k] : 0.;
    taucmc = 0.; ssacmc = 1.; asmcmc = 0.; taormc = 0.;
    if (!(ciwp + clwp + cswp >= cldmin)) return;
    auto tab = [&](int t, int i) { return T[(t * SW_NBND + band) * CLD_STRIDE + i]; };
    double extcoice = 0., ssacoice = 0., gice = 0., forwice = 0.;
    double extcosno = 0., ssacosno = 0., gsno = 0., forwsno = 0.;
    double extcoliq = 0., ssacoliq = 0., gliq = 0., forwliq = 0.;
    if ((ciwp + cswp) != 0.0) {
        const int ix = c.iice[k];
        if (ix == 0) { bad = 1; return; }
        const double f = c.fice[k];
        extcoice = tab(3, ix - 1) + f * (tab(3, ix) - tab(3, ix - 1));
        ssacoice = tab(4, ix - 1) + f * (tab(4, ix) - tab(4, ix - 1));
        gice = tab(5, ix - 1) + f * (tab(5, ix) - tab(5, ix - 1));
        const double fdelta = tab(6, ix - 1) + f * (tab(6, ix) - tab(6, ix - 1));
        forwice = fdelta + fdiv(0.5, ssacoice);
        forwice = forwice > gice ? gice : forwice;
    }
    if (cswp > 0.0 && iceflag == 5) {
        const int ix = c.isno[k];
        if (ix == 0) { bad = 2; return; }
        const double f = c.fsno[k];
        extcosno = tab(3, ix - 1) + f * (tab(3, ix) - tab(3, ix - 1));
        ssacosno = tab(4, ix - 1) + f * (tab(4, ix) - tab(4, ix - 1));
        gsno = tab(5, ix - 1) + f * (tab(5, ix) - tab(5, ix - 1));
        const double fdelta = tab(6, ix - 1) + f * (tab(6, ix) - tab(6, ix - 1));
        forwsno = fdelta + fdiv(0.5, ssacosno);
        forwsno = forwsno > gsno ? gsno : forwsno;
    }
    if (clwp != 0.0) {
        const int ix = c.iliq[k];
        if (ix == 0) { bad = 3; return; }
        const double f = c.fliq[k];
        extcoliq = tab(0, ix - 1) + f * (tab(0, ix) - tab(0, ix - 1));
        ssacoliq = tab(1, ix - 1) + f * (tab(1, ix) - tab(1, ix - 1));
        if (f < 0. && ssacoliq > 1.) ssacoliq = tab(1, ix - 1);
        gliq = tab(2, ix - 1) + f * (tab(2, ix) - tab(2, ix - 1));
        forwliq = gliq * gliq;
    }
    const double tauliqorig = clwp * extcoliq, tauiceorig = ciwp * extcoice;
    const double ssaliq = fdiv(ssacoliq * (1. - forwliq), 1. - forwliq * ssacoliq);
    const double tauliq = (1. - forwliq * ssacoliq) * tauliqorig;
    const double ssaice = fdiv(ssacoice * (1. - forwice), 1. - forwice * ssacoice);
    const double tauice = (1. - forwice * ssacoice) * tauiceorig;
    const double scatliq = ssaliq * tauliq;
    double scatice = ssaice * tauice, scatsno = 0.0;
    if (iceflag < 5) {
        taormc = tauliqorig + tauiceorig;
        taucmc = tauliq + tauice;
    } else {
        const double tausnoorig = cswp * extcosno;
        taormc = tauliqorig + tauiceorig + tausnoorig;
        const double ssasno = fdiv(ssacosno * (1. - forwsno), 1. - forwsno * ssacosno);
        const double tausno = (1. - forwsno * ssacosno) * tausnoorig;
        scatsno = ssasno * tausno;
        taucmc = tauliq + tauice + tausno;
    }
    if (taucmc == 0.) taucmc = cldmin;
    if (scatice == 0.) scatice = cldmin;
    if (scatsno == 0.) scatsno = cldmin;
    const double gl = fdiv(scatliq * (gliq - forwliq), 1.0 - forwliq);
    const double gi = scatice * fdiv(gice - forwice, 1.0 - forwice);
    if (iceflag < 5) {
        ssacmc = fdiv(scatliq + scatice, taucmc);
        asmcmc = frcp1(scatliq + scatice) * (gl + gi);
    } else {
        ssacmc = fdiv(scatliq + scatice + scatsno, taucmc);
        asmcmc = frcp1(scatliq + scatice + scatsno) * (gl + gi + scatsno * fdiv(gsno - forwsno, 1.0 - forwsno));
    }
}
// reftra_sw for a layer without asymmetry (zg = 0: the clean-sky call outside cloud, where the only scatterer is
// Rayleigh): zgamma3 = zgamma4 = 1/2, za1 = za2, zq = 0, denom = 1.  Same operations as reftra_fast otherwise.
__device__ __forceinline__ void reftra_clean(const SwDeviceTables &tb, double zto1, double zw, double prmuz, double zem2,
                                             RT &o)
{
    const double eps = 1.e-08, zwcrit = 0.9999995;
    const double zgamma1 = (8. - zw * 5.) * 0.25;
    const double zgamma2 = 3. * zw * 0.25;
    if (zw >= zwcrit) {
        const double za1 = zgamma1 * prmuz - 0.5;
        const double zgt = zgamma1 * zto1;
        const double rg = frcp1(1. + zgt);
        const double r = (zgt - za1 * (1. - zem2)) * rg;
        const double rd = zgt * rg;
        const bool one = (zem2 == 1.0);
        o.ref = one ? 0.0 : r;
        o.tra = one ? 1.0 : 1. - r;
        o.refd = one ? 0.0 : rd;
        o.trad = one ? 1.0 : 1. - rd;
    } else {
        const double za = 0.5 * (zgamma1 + zgamma2);
        const double zrk = fsqrt(zgamma1 * zgamma1 - zgamma2 * zgamma2);
        const double zrp = zrk * prmuz;
        const double zrp1 = 1. + zrp, zrm1 = 1. - zrp;
        const double zrk2 = 2. * zrk;
        const double zrpp = 1. - zrp * zrp;
        const double zrkg = zrk + zgamma1;
        const double hp = za + 0.5 * zrk, hm = za - 0.5 * zrk;
        const double zr1 = zrm1 * hp, zr2 = zrp1 * hm, zr3 = zrk2 * (0.5 - za * prmuz);
        const double zr4 = zrpp * zrkg, zr5 = zrpp * (zrk - zgamma1);
        const double zt1 = zrp1 * hp, zt2 = zrm1 * hm, zt3 = zrk2 * (0.5 + za * prmuz);
        const double zem1 = exp_lookup(tb, fmin(zrk * zto1, 500.));
        const double zemm = zem1 * zem1;
        const double D = zr4 + zr5 * zemm;
        const double rD = frcp1(D);
        const bool small = fabs(D) <= eps * zem1;
        const double r = zw * (zr1 - zr2 * zemm - zr3 * (zem2 * zem1)) * rD;
        const double t = zem2 - zw * (zem2 * (zt1 - zt2 * zemm) - zt3 * zem1) * rD;
        o.ref = small ? eps : r;
        o.tra = small ? zem2 : t;
        const double zdend = frcp1(zrkg - (zgamma1 - zrk) * zemm);
        o.refd = zgamma2 * (1. - zemm) * zdend;
        o.trad = zrk2 * zem1 * zdend;
    }
}

// cloud optical properties of one cloudy sub-column layer, as the solver needs them
struct CldLayer {
    double tc, oc, gc, taor;
};

// optical properties of layer k as seen by the two-stream solver (spcvmc_sw layer preparation, SW:8443-8501)
// followed by reftra and the delta-scaled direct-beam transmittance (SW:8568-8578).  ta, wa, ga: aerosol optical
// depth (0 in the clean call: ztauacln, SW:9346), single-scattering albedo, asymmetry.  dbt_nodel: unscaled direct
// beam (SW:8459-8469), requested by the call with aerosol only.  cloudy: this sub-column (lane) has cloud in the
// layer -- total-sky properties per SW:8505-8510, 8476-8486.
__device__ __forceinline__ void layer_rt(const bool NODEL, const SwDeviceTables &tb, double taug, double taur, double ta,
                                         double wa, double ga, double prmu0, double rprmu0, bool cloudy,
                                         const CldLayer &cl, RT &o, double &dbt_nodel)
{
    // the four quotients of SW:8476-8501 over two reciprocals (each of a product of the two denominators)
    double ztauc = taur + taug + ta;
    double zomcc = taur * 1.0 + ta * wa;
    const double r1 = frcp1(zomcc * ztauc);
    double zgcc = (ga * wa * ta) * (r1 * ztauc);
    zomcc = zomcc * (r1 * zomcc);
    const double ztau_un = ztauc;
    const double zf = zgcc * zgcc;
    const double zwf = zomcc * zf;
    ztauc = (1.0 - zwf) * ztauc;
    const double d1 = fmax((1.0 - zwf), 1.0E-30), d2 = fmax((1.0 - zf), 1.0E-30);
    const double r2 = frcp1(d1 * d2);
    zomcc = (zomcc - zwf) * (r2 * d2);
    zgcc = (zgcc - zf) * (r2 * d1);
    double taor = 0.;
    if (cloudy) {
        const double ztauo = ztauc + cl.tc;
        const double zomco = ztauc * zomcc + cl.tc * cl.oc;
        zgcc = fdiv(cl.tc * cl.oc * cl.gc + ztauc * zomcc * zgcc, zomco);
        zomcc = fdiv(zomco, ztauo);
        ztauc = ztauo;
        taor = cl.taor;
    }
    if (NODEL) dbt_nodel = exp_lookup(tb, (ztau_un + taor) * rprmu0);
    o.dbt = exp_lookup(tb, ztauc * rprmu0);
    reftra_fast(tb, ztauc, zomcc, zgcc, prmu0, o.dbt, o);
}

// the same with ta = 0 and no cloud: zgcc = 0, no delta scaling (zf = zwf = 0), one division left
__device__ __forceinline__ void layer_rt_clean(const SwDeviceTables &tb, double taug, double taur, double prmu0,
                                               double rprmu0, RT &o)
{
    const double ztauc = taur + taug;
    const double zomcc = fdiv(taur, ztauc);
    o.dbt = exp_lookup(tb, ztauc * rprmu0);
    reftra_clean(tb, ztauc, zomcc, prmu0, o.dbt, o);
}
// ... with its two table reads issued ahead (right after the gas optics, in front of the stream with aerosol, whose
// arithmetic then covers their latency): the arguments and the raw table values travel to the finishing half
struct CleanPre { double e_1, tv_dbt, tv_1; };   // (the direct-beam argument is a sum and a product: formed again)
__device__ __forceinline__ CleanPre layer_rt_clean_issue(const SwDeviceTables &tb, double taug, double taur, double rprmu0)
{
    CleanPre p;
    const double ztauc = taur + taug;
    const double zw = fdiv(taur, ztauc);
    p.tv_dbt = exp_issue(tb, ztauc * rprmu0);
    const double zgamma1 = (8. - zw * 5.) * 0.25, zgamma2 = 3. * zw * 0.25;
    const double zrk = fsqrt(fmax(zgamma1 * zgamma1 - zgamma2 * zgamma2, 0.));   // unused where zw >= zwcrit
    p.e_1 = fmin(zrk * ztauc, 500.);
    p.tv_1 = exp_issue(tb, p.e_1);
    return p;
}
__device__ __forceinline__ void layer_rt_clean_finish(double taug, double taur, double prmuz, double rprmu0,
                                                      const CleanPre &p, RT &o)
{
    const double zto1 = taur + taug;
    const double zw = fdiv(taur, zto1);
    const double zem2 = exp_pick(zto1 * rprmu0, p.tv_dbt);
    o.dbt = zem2;
    // reftra_clean with the exponential of zrk * zto1 from the issued read
    const double eps = 1.e-08, zwcrit = 0.9999995;
    const double zgamma1 = (8. - zw * 5.) * 0.25;
    const double zgamma2 = 3. * zw * 0.25;
    if (zw >= zwcrit) {
        const double za1 = zgamma1 * prmuz - 0.5;
        const double zgt = zgamma1 * zto1;
        const double rg = frcp1(1. + zgt);
        const double r = (zgt - za1 * (1. - zem2)) * rg;
        const double rd = zgt * rg;
        const bool one = (zem2 == 1.0);
        o.ref = one ? 0.0 : r;
        o.tra = one ? 1.0 : 1. - r;
        o.refd = one ? 0.0 : rd;
        o.trad = one ? 1.0 : 1. - rd;
    } else {
        const double za = 0.5 * (zgamma1 + zgamma2);
        const double zrk = fsqrt(zgamma1 * zgamma1 - zgamma2 * zgamma2);
        const double zrp = zrk * prmuz;
        const double zrp1 = 1. + zrp, zrm1 = 1. - zrp;
        const double zrk2 = 2. * zrk;
        const double zrpp = 1. - zrp * zrp;
        const double zrkg = zrk + zgamma1;
        const double hp = za + 0.5 * zrk, hm = za - 0.5 * zrk;
        const double zr1 = zrm1 * hp, zr2 = zrp1 * hm, zr3 = zrk2 * (0.5 - za * prmuz);
        const double zr4 = zrpp * zrkg, zr5 = zrpp * (zrk - zgamma1);
        const double zt1 = zrp1 * hp, zt2 = zrm1 * hm, zt3 = zrk2 * (0.5 + za * prmuz);
        const double zem1 = exp_pick(p.e_1, p.tv_1);
        const double zemm = zem1 * zem1;
        const double D = zr4 + zr5 * zemm;
        const double rD = frcp1(D);
        const bool small = fabs(D) <= eps * zem1;
        const double r = zw * (zr1 - zr2 * zemm - zr3 * (zem2 * zem1)) * rD;
        const double t = zem2 - zw * (zem2 * (zt1 - zt2 * zemm) - zt3 * zem1) * rD;
        o.ref = small ? eps : r;
        o.tra = small ? zem2 : t;
        const double zdend = frcp1(zrkg - (zgamma1 - zrk) * zemm);
        o.refd = zgamma2 * (1. - zemm) * zdend;
        o.trad = zrk2 * zem1 * zdend;
    }
}

// Two-stream properties of a stack of layers under a direct beam from above: reflectance to the beam (ref) and to
// diffuse light from above (refd), total (tra), diffuse (trad) and direct-beam (dbt) transmittance, reflectance to
// diffuse light from below (rb).  add_on_top() puts one homogeneous layer on top of the stack: the adding step of
// vrtqdr_sw (SW:7997-8034) written for a composite lower part -- ref/refd follow the reference's upward recurrence
// (SW:8003-8013), tra/rb its downward one (SW:8026-8034) with the roles of "above" and "layer" taken by the new
// layer and the stack.  The diffuse transmittance of a stack is the same in both directions (reciprocity).
struct Stack {
    double ref, refd, tra, trad, rb, dbt;
};
__device__ __forceinline__ void add_on_top(Stack &s, const RT &b)
{
    const double z = frcp1(1. - s.refd * b.refd);
    const double dif = b.tra - b.dbt;                       // diffuse part of the layer's transmitted beam
    const double nref = b.ref + (b.trad * (dif * s.refd + b.dbt * s.ref)) * z;
    const double nrefd = b.refd + b.trad * b.trad * s.refd * z;
    const double ntra = b.dbt * s.tra + (s.trad * (dif + b.dbt * s.ref * b.refd)) * z;
    const double nrb = s.rb + s.trad * s.trad * b.refd * z;
    const double ntrad = s.trad * b.trad * z;
    s.ref = nref; s.refd = nrefd; s.tra = ntra; s.trad = ntrad; s.rb = nrb;
    s.dbt = b.dbt * s.dbt;
}

}  // namespace

// O3DATA (LW:12806-12836): the annual-mean ozone mixing ratio of the layer between plev_b and plev_t (mb)
__device__ __forceinline__ double sw_o3clim(const double plev_b, const double plev_t)
{
    double o3mmr = 0.;
    for (int jj = 0; jj < 31; ++jj) {
        double pb1 = (-(plev_b - c_ppwrkh[jj]) >= 0.) ? 0. : plev_b - c_ppwrkh[jj];
        double pb2 = (-(plev_b - c_ppwrkh[jj + 1]) >= 0.) ? 0. : plev_b - c_ppwrkh[jj + 1];
        double pt1 = (-(plev_t - c_ppwrkh[jj]) >= 0.) ? 0. : plev_t - c_ppwrkh[jj];
        double pt2 = (-(plev_t - c_ppwrkh[jj + 1]) >= 0.) ? 0. : plev_t - c_ppwrkh[jj + 1];
        o3mmr = o3mmr + (pb2 - pb1 - pt2 + pt1) * c_o3ann[jj];
    }
    return fdiv(o3mmr, plev_b - plev_t);
}

// ozone volume mixing ratio of layer k from the climatology o3mmr, or (o3input = 2) from the model's field, SW:10661-10680
__device__ __forceinline__ double sw_o3vmr(const SwKernelArgs &a, const size_t base3, const size_t sk, const int r4, const int k,
                                           const int nk, const double plev_b, const double o3mmr)
{
    const double amdo = 0.603461;
    double o3vmr = o3mmr * amdo;
    if (a.o33d && a.o3input == 2) {
        if (k != nk) {
            o3vmr = ldr(a.o33d, base3 + (size_t)k * sk, r4);
        } else {
            // o31d(kte) - o3mmr(kte)*amdo + o3mmr(k)*amdo needs the climatology of the layer below
            const double pb_b = ldr(a.p8w, base3 + (size_t)(nk - 1) * sk, r4) / 100.;
            const double o3b = sw_o3clim(pb_b, plev_b);
            const double v = ldr(a.o33d, base3 + (size_t)(nk - 1) * sk, r4) - o3b * amdo + o3mmr * amdo;
            o3vmr = v <= 0. ? o3mmr * amdo : v;
        }
    }
    return o3vmr;
}

// One column: vb = position of the column in the XCD-interleaved order, scr_block = this workgroup's slice of the
// layer scratch.  CLOUD: icloud /= 0 (McICA sub-columns); PROF: the optional layer-flux profiles are requested.
template <bool CLOUD, bool PROF>
__device__ __forceinline__ void sw_column(const SwKernelArgs &a, const int vb, double *lds, double *scr_block)
{
    const SwDeviceTables &tb = a.tab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nk = a.nk, nlay = nk + 1, nlev = nlay + 1;
    const int nlp = (nlay + BLK - 1) / BLK * BLK, nblk = nlp / BLK;
    // the WRF-Chem aerosol route alone (300/400/600/999 nm arrays; no band-resolved input, no ECMWF types) keeps
    // omega and g as two numbers per layer instead of 14: 10 KB less LDS per column at 50 layers = one more workgroup
    // per CU
    const bool compact = a.tauaer3d_sw == nullptr && a.aerod == nullptr;
    ColView c = carve(lds, nlp, CLOUD, PROF, compact);

    // ---- column of this workgroup (XCD-aware: b, b+8, b+16.. are neighbours in i) ----
    int col;
    {
        const int b = vb;
        int nwg = a.ncol;   // columns of this launch (a chunk of the tile), or what the call's column list holds of them
        if (a.colmap) {
            const int left = *a.colcount - a.col0;
            nwg = left < nwg ? left : nwg;
        }
        if (b >= nwg) return;   // (uniform over the workgroup, in front of its first barrier)
        const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        col = a.col0 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        if (a.colmap) col = a.colmap[col];   // (cloud_lists.h)
    }
    const int it = col % a.ni_t, jt = col / a.ni_t;            // position inside the tile
    const size_t i0 = (size_t)(a.its - a.ims + it), j0 = (size_t)(a.jts - a.jms + jt);
    const size_t k0 = (size_t)(a.kts - a.kms);
    const size_t i2 = j0 * a.ni_m + i0;
    const size_t sk = a.ni_m, sj = (size_t)a.ni_m * a.nk_m, sjf = (size_t)a.ni_m * (a.nk_m + 2);
    const size_t base3 = j0 * sj + k0 * sk + i0, base3f = j0 * sjf + k0 * sk + i0;
    const int r4 = a.real4;

    const double coszrs = ldr(a.xcoszen, i2, r4);
    if (tid == 0 && a.coszr) str(a.coszr, i2, r4, coszrs);
    if (coszrs <= 0.0) {
        // night column, SW:11173-11197: zero the 2-D fluxes, leave gsw / rthratensw untouched
        if (tid == 0) {
            if (a.swupt) {
                str(a.swupt, i2, r4, 0.); str(a.swuptc, i2, r4, 0.); str(a.swuptcln, i2, r4, 0.);
                str(a.swdnt, i2, r4, 0.); str(a.swdntc, i2, r4, 0.); str(a.swdntcln, i2, r4, 0.);
                str(a.swupb, i2, r4, 0.); str(a.swupbc, i2, r4, 0.); str(a.swupbcln, i2, r4, 0.);
                str(a.swdnb, i2, r4, 0.); str(a.swdnbc, i2, r4, 0.); str(a.swdnbcln, i2, r4, 0.);
                if (a.swvisdir) {
                    str(a.swvisdir, i2, r4, 0.); str(a.swvisdif, i2, r4, 0.);
                    str(a.swnirdir, i2, r4, 0.); str(a.swnirdif, i2, r4, 0.);
                }
            }
            if (a.swddir) { str(a.swddir, i2, r4, 0.); str(a.swddni, i2, r4, 0.); str(a.swddif, i2, r4, 0.); }
            if (a.swcf) str(a.swcf, i2, r4, 0.);
        }
        return;
    }

    // ======================= prologue: lanes = layers =======================
    bool lowerp = false;
    // With at most 64 layers the second wave holds none: it forms the ozone climatology of the layers (a 31-step sum each,
    // a third of the set-up's instructions) while the first does the rest; the ozone amount follows behind the barrier.
    const bool o3_split = nlp <= 64;
    double *const o3tmp = c.flux;   // (the flux slots are idle until the sweeps)
    double coldry_k = 0., plev_b_k = 0.;
    if (o3_split && wave == 1 && lane < nlay) {
        const int k = lane;
        const double pb = fdiv(ldr(a.p8w, base3 + (size_t)k * sk, r4), 100.);
        const double pt = k == nk ? 1.0e-5 : fdiv(ldr(a.p8w, base3 + (size_t)(k + 1) * sk, r4), 100.);
        o3tmp[k] = sw_o3clim(pb, pt);
    }
    if (tid < nlp) {
        const int k = tid;
        if (k < nlay) {
            // 1-D profile, SW:10340-10390, 10628-10656 (extra layer to 1e-5 mb on top)
            const bool top = (k == nk);
            const int km = top ? nk - 1 : k;
            // (the quotients of the set-up through fdiv: the correctly rounded quotient in 8 instructions where the
            // compiler's IEEE division takes 30 -- this block is one wave's serial path)
            double plev_b = fdiv(ldr(a.p8w, base3 + (size_t)k * sk, r4), 100.);
            double plev_t = top ? 1.0e-5 : fdiv(ldr(a.p8w, base3 + (size_t)(k + 1) * sk, r4), 100.);
            double pavel = top ? 0.5 * plev_b : fdiv(ldr(a.p3d, base3 + (size_t)k * sk, r4), 100.);
            double tavel = top ? ldr(a.t8w, base3 + (size_t)k * sk, r4) + 0.0 : ldr(a.t3d, base3 + (size_t)k * sk, r4);
            double qv = ldr(a.qv3d, base3 + (size_t)km * sk, r4);
            qv = fmax(0., qv);
            qv = fmax(qv, 1.E-12);
            const double amdw = 1.607793;
            double h2ovmr = qv * amdw;
            // O3 climatology, O3DATA LW:12806-12836 (o3_split: by the other wave, see above; the amount follows below)
            double o3vmr = 0.;
            if (!o3_split) o3vmr = sw_o3vmr(a, base3, sk, r4, k, nk, plev_b, sw_o3clim(plev_b, plev_t));
            // inatm_sw, SW:9777-9779 and SW:9810-9814
            const double avogad = 6.02214199e+23, grav = 9.8066, amd = 28.9660, amw = 18.0160;
            const double co2 = 379.e-6, ch4 = 1774.e-9, o2 = 0.209488;
            double pdp = plev_b - plev_t;
            double amm = (1. - h2ovmr) * amd + h2ovmr * amw;
            double coldry = fdiv((plev_b - plev_t) * 1.e3 * avogad, 1.e2 * grav * amm * (1. + h2ovmr));
            double wh2o = coldry * h2ovmr, wco2 = coldry * co2, wo3 = coldry * o3vmr, wch4 = coldry * ch4,
                   wo2 = coldry * o2;
            coldry_k = coldry; plev_b_k = plev_b;
            // setcoef_sw, SW:2850-2987
            const double stpfac = 296. / 1013.;
            double plog = log(pavel);
            int jp = (int)(36. - 5 * (plog + 0.04));
            jp = jp < 1 ? 1 : (jp > 58 ? 58 : jp);
            int jp1 = jp + 1;
            double fp = 5. * (c_preflog[jp - 1] - plog);
            double d0 = fdiv(tavel - c_tref[jp - 1], 15.);
            int jtt = (int)(3. + d0);
            jtt = jtt < 1 ? 1 : (jtt > 4 ? 4 : jtt);
            double ft = d0 - (double)(jtt - 3);
            double d1 = fdiv(tavel - c_tref[jp1 - 1], 15.);
            int jt1 = (int)(3. + d1);
            jt1 = jt1 < 1 ? 1 : (jt1 > 4 ? 4 : jt1);
            double ft1 = d1 - (double)(jt1 - 3);
            double water = fdiv(wh2o, coldry);
            double scalefac = fdiv(pavel * stpfac, tavel);
            lowerp = plog > 4.56;
            double forfac = fdiv(scalefac, 1. + water), forfrac, selffac = 0., selffrac = 0.;
            int indfor, indself = 1;
            if (lowerp) {
                double factor = fdiv(332.0 - tavel, 36.0);
                int ifac = (int)factor;
                indfor = ifac > 1 ? (ifac < 2 ? ifac : 2) : 1;
                forfrac = factor - (double)indfor;
                selffac = water * forfac;
                factor = fdiv(tavel - 188.0, 7.2);
                int is = (int)factor - 7;
                indself = is > 1 ? (is < 9 ? is : 9) : 1;
                selffrac = factor - (double)(indself + 7);
            } else {
                double factor = fdiv(tavel - 188.0, 36.0);
                indfor = 3;
                forfrac = factor - 1.0;
            }
            double colh2o = 1.e-20 * wh2o, colco2 = 1.e-20 * wco2, colo3 = 1.e-20 * wo3, colch4 = 1.e-20 * wch4,
                   colo2 = 1.e-20 * wo2;
            double colmol = 1.e-20 * coldry + colh2o;
            if (colco2 == 0.) colco2 = 1.e-32 * coldry;
            if (colch4 == 0.) colch4 = 1.e-32 * coldry;
            if (colo2 == 0.) colo2 = 1.e-32 * coldry;
            double compfp = 1. - fp;
            c.fac10[k] = compfp * ft;
            c.fac00[k] = compfp * (1. - ft);
            c.fac11[k] = fp * ft1;
            c.fac01[k] = fp * (1. - ft1);
            c.selffac[k] = selffac; c.selffrac[k] = selffrac; c.forfac[k] = forfac; c.forfrac[k] = forfrac;
            c.col[SP_H2O * nlp + k] = colh2o; c.col[SP_CO2 * nlp + k] = colco2; c.col[SP_O3 * nlp + k] = colo3;
            c.col[SP_CH4 * nlp + k] = colch4; c.col[SP_O2 * nlp + k] = colo2; c.col[SP_ZERO * nlp + k] = 0.;
            c.colmol[k] = colmol; c.pdp[k] = pdp;
            c.jp[k] = jp;
            // 0-based row bases of the k-tables: ind0-js, ind1-js of SW:3335-3336 (lower), SW:3369-3370 (upper)
            c.rb0[k] = lowerp ? ((jp - 1) * 5 + (jtt - 1)) : ((jp - 13) * 5 + (jtt - 1));
            c.rb1[k] = lowerp ? (jp * 5 + (jt1 - 1)) : ((jp - 12) * 5 + (jt1 - 1));
            c.inds[k] = indself - 1;
            c.indf[k] = indfor - 1;
            if (CLOUD) {
                // hydrometeors -> in-cloud water paths and effective sizes, SW:10371-10618 and 10743-10947
                double cf = 0., clwp = 0., ciwp = 0., cswp = 0., reliq = 10., reice = 10., resno = 10.;
                if (!top) {
                    const size_t q = base3 + (size_t)k * sk;
                    double qc = a.qc3d ? fmax(0., ldr(a.qc3d, q, r4)) : 0.;
                    double qr = a.qr3d ? fmax(0., ldr(a.qr3d, q, r4)) : 0.;
                    double qi = 0., qs = 0.;
                    if (!a.qi3d && !a.warm_rain && tavel < 273.15) { qi = qc; qs = qr; qc = 0.; qr = 0.; }
                    if (a.qi3d) qi = fmax(0., ldr(a.qi3d, q, r4));
                    if (a.qs3d) qs = fmax(0., ldr(a.qs3d, q, r4));
                    if (a.f_ice_phy) {        // "MP option 5": one frozen species carried as snow, SW:10461-10472;
                        // the cloud water the T < 273.15 rule above moved to ice comes back (qc1d = qc3d, SW:10467)
                        const double s3 = a.qs3d ? ldr(a.qs3d, q, r4) : 0.;
                        qi = fmax(0., 0.1 * s3); qs = 0.9 * s3;
                        qc = a.qc3d ? fmax(0., ldr(a.qc3d, q, r4)) : 0.;
                    }
                    if (a.fer_hires) {        // Ferrier-Aligo: qi3d holds all frozen water, SW:10474-10492
                        qi = a.qi3d ? fmax(0., ldr(a.qi3d, q, r4)) : 0.; qs = 0.0;
                        qc = a.qc3d ? fmax(0., ldr(a.qc3d, q, r4)) : 0.;
                    }
                    cf = a.cldfra3d ? ldr(a.cldfra3d, q, r4) : 0.;
                    const double xland = a.xland ? ldr(a.xland, i2, r4) : 1.;
                    int iceflg = 3, inflg = 2;
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
                            int idx = (int)(tavel - 179.);
                            idx = idx < 1 ? 1 : (idx > 75 ? 75 : idx);
                            const double corr = tavel - (double)(int)tavel;
                            reice1 = c_retab[idx - 1] * (1. - corr) + c_retab[idx] * corr;
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
                    if (a.progn == 1) {   // prognostic droplet number, SW:10797-10816
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
                        reice = c_retab[index - 1] * (1. - corr) + c_retab[index] * corr;
                    }
                    if (inflg >= 3) reliq = recloud;
                    if (iceflg >= 4) reice = reice1;
                    if (iceflg == 3) { reice = reice * 1.0315; reice = fmin(140.0, reice); }
                    if (a.iradius) {          // CAM-MG microphysics supplies the radii, SW:10883-10898
                        reice = (qi > 1.e-20 || qs > 1.e-20) ? ldr(a.iradius, q, r4) : 25.;
                        reice = fmax(5., fmin(140.0, reice));
                        reliq = qc > 1.e-20 ? ldr(a.lradius, q, r4) : 10.;
                        reliq = fmax(2.5, fmin(60.0, reliq));
                    }
                    if (inflg == 5) resno = resnow1; else { cswp = 0.0; resno = 10.0; }
                }
                c.cldf[k] = cf < 1.0e-20 ? 0. : cf;   // generate_stochastic_clouds_sw, SW:1677-1684
                c.clwp[k] = clwp; c.ciwp[k] = ciwp; c.cswp[k] = cswp;
                // table positions of cldprmc_sw (SW:2168-2176, 2212-2219, 2268-2276); 0 = out of bounds
                {
                    const bool ok = !(reice < 5.0 || reice > 140.0);
                    const double factor = (reice - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 46) index = 45;
                    c.iice[k] = ok ? index : 0;
                    c.fice[k] = factor - (double)index;
                }
                {
                    const bool ok = !(resno < 5.0 || resno > 140.0);
                    const double factor = (resno - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 46) index = 45;
                    c.isno[k] = ok ? index : 0;
                    c.fsno[k] = factor - (double)index;
                }
                {
                    const bool ok = !(reliq < 1.5 || reliq > 60.);
                    int index = (int)(reliq - 1.5);
                    if (index == 0) index = 1;
                    if (index == 58) index = 57;
                    c.iliq[k] = ok ? index : 0;
                    c.fliq[k] = reliq - 1.5 - (double)index;
                }
                if (k < 4) {
                    // KISS seeds from the fractional part of the mid-layer pressure in Pa, SW:1696-1699; a REAL*4
                    // build of the reference does this arithmetic in single precision
                    // (no FMA contraction here: fusing pm*100 into the subtraction changes the truncated integer)
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
                    c.seeds[k] = seed;
                }
            }
        } else {
            // identity padding layers above the real top: never evaluated, keep LDS defined
            c.fac00[k] = c.fac01[k] = c.fac10[k] = c.fac11[k] = 0.;
            c.selffac[k] = c.selffrac[k] = c.forfac[k] = c.forfrac[k] = 0.;
            for (int s = 0; s < SP_COUNT; ++s) c.col[s * nlp + k] = 0.;
            c.colmol[k] = 0.; c.pdp[k] = 1.;
            c.jp[k] = 0; c.rb0[k] = c.rb1[k] = 0; c.inds[k] = c.indf[k] = 0;
        }
    }
    if (tid == 0) c.jp[nlp] = 0;
    // laytrop = number of layers with ln(p) > 4.56 (SW:2890)
    const int laytrop = __syncthreads_count(lowerp ? 1 : 0);
    if (o3_split && tid < nlay) {
        // colo3 = 1e-20 * (coldry * o3vmr), SW:9810-9814, with the climatology the other wave has formed
        const double o3vmr = sw_o3vmr(a, base3, sk, r4, tid, nk, plev_b_k, o3tmp[tid]);
        c.col[SP_O3 * nlp + tid] = 1.e-20 * (coldry_k * o3vmr);
    }

    // aerosol: defaults, then 300/400/600/999 nm -> 14 bands (SW:10974-11023)
    if (compact) {
        // per layer: Angstrom exponent, tau(400 nm) and the two straight lines of omega and g over wavelength (parked
        // in the flux slots, which are free until the sweeps); per (band, layer): tau = tau400 * (0.4 / wavemid)^ang
        // as exp(ang * ln(0.4 / wavemid)) -- one log per layer and one exp per item instead of two logs and a pow
        double *ang = c.flux, *t400 = c.flux + nlp;
        if (tid < nlp) {
            const int k = tid;
            double an = 0., t4 = 0., slw = 0., w6 = 1., slg = 0., g6 = 0.5;   // no aerosol: tau = 0 (omega, g idle)
            if (a.aer_ra_feedback == 1 && k < nk) {
                const size_t q = base3 + (size_t)k * sk;
                const double t300 = ldr(a.tauaer300, q, r4), t999 = ldr(a.tauaer999, q, r4);
                if (t300 > 1.e-9 && t999 > 1.e-9) {
                    an = log(t300 / t999) / log(999. / 300.);
                    t4 = ldr(a.tauaer400, q, r4);
                    const double w4 = ldr(a.waer400, q, r4), g4 = ldr(a.gaer400, q, r4);
                    w6 = ldr(a.waer600, q, r4); g6 = ldr(a.gaer600, q, r4);
                    slw = (w6 - w4) / .2; slg = (g6 - g4) / .2;
                }
            }
            ang[k] = an; t400[k] = t4;
            c.waer[k] = slw; c.waer[nlp + k] = w6; c.gaer[k] = slg; c.gaer[nlp + k] = g6;
        }
        __syncthreads();
        for (int item = tid; item < SW_NBND * nlp; item += NTHREADS) {
            const int nb = item / nlp, k = item - nb * nlp;
            const double t4 = t400[k];
            c.taer[item] = t4 != 0. ? t4 * exp(ang[k] * c_lnwave[nb]) : 0.;
        }
    } else
    for (int item = tid; item < SW_NBND * nlp; item += NTHREADS) {
        const int nb = item / nlp, k = item - nb * nlp;
        double tau = 0., ssa = 1., asy = 0.;
        if (a.tauaer3d_sw && k < nk) {   // aer_opt = 2/3 input, SW:10982-10991
            const size_t q4 = (size_t)nb * a.ni_m * a.nk_m * a.nj_m + base3 + (size_t)k * sk;
            tau = ldr(a.tauaer3d_sw, q4, r4); ssa = ldr(a.ssaaer3d_sw, q4, r4); asy = ldr(a.asyaer3d_sw, q4, r4);
        }
        if (a.aer_ra_feedback == 1 && k < nk) {
            const size_t q = base3 + (size_t)k * sk;
            const double t300 = ldr(a.tauaer300, q, r4), t999 = ldr(a.tauaer999, q, r4);
            if (t300 > 1.e-9 && t999 > 1.e-9) {
                const double wavemid = c_wavemid[nb];
                double ang = log(t300 / t999) / log(999. / 300.);
                tau = ldr(a.tauaer400, q, r4) * pow(0.4 / wavemid, ang);
                const double w4 = ldr(a.waer400, q, r4), w6 = ldr(a.waer600, q, r4);
                double slope = (w6 - w4) / .2;
                ssa = slope * (wavemid - .6) + w6;
                if (ssa < 0.4) ssa = 0.4;
                if (ssa >= 1.0) ssa = 1.0;
                const double g4 = ldr(a.gaer400, q, r4), g6 = ldr(a.gaer600, q, r4);
                slope = (g6 - g4) / .2;
                asy = slope * (wavemid - .6) + g6;
                if (asy < 0.5) asy = 0.5;
                if (asy >= 1.0) asy = 1.0;
            }
        }
        c.taer[item] = tau; c.waer[item] = ssa; c.gaer[item] = asy;
    }
    __syncthreads();
    // column optical-depth guard per band (SW:11026-11047): sequential sum as in the reference
    if (a.aer_ra_feedback == 1 && tid < SW_NBND) {
        double slope = 0.;
        for (int k = 0; k < nk; ++k) slope = slope + c.taer[tid * nlp + k];
        if (slope < 0.) {
            atomicMax(a.status, 1 + col);   // fatal in the reference ("Negative total optical depth")
        } else if (slope > 6.) {
            for (int k = 0; k < nk; ++k) c.taer[tid * nlp + k] = c.taer[tid * nlp + k] * 6.0 / slope;
        }
    }
    __syncthreads();
    if (a.aerod) {
        // aer_opt = 1: the six ECMWF aerosol types replace every other aerosol input (iaer = 6, SW:9313-9340);
        // no aerosol in the extra layer (SW:11096)
        for (int item = tid; item < SW_NBND * nlp; item += NTHREADS) {
            const int nb = item / nlp, k = item - nb * nlp;
            double tau = 0., ssa = 0., asy = 0.;
            if (k < nk) {
                for (int ia = 0; ia < 6; ++ia) {
                    const double ec = ldr(a.aerod, (size_t)ia * a.ni_m * a.nk_m * a.nj_m + base3 + (size_t)k * sk, r4);
                    tau = tau + c_swaer[0][nb][ia] * ec;
                    ssa = ssa + c_swaer[0][nb][ia] * ec * c_swaer[1][nb][ia];
                    asy = asy + c_swaer[0][nb][ia] * ec * c_swaer[1][nb][ia] * c_swaer[2][nb][ia];
                }
                if (ssa != 0.) asy = asy / ssa;
                if (tau != 0.) ssa = ssa / tau;
            }
            c.taer[item] = tau; c.waer[item] = ssa; c.gaer[item] = asy;
        }
        __syncthreads();
    }

    // ======================= main phase: lanes = g-points =======================
    const int chunk = wave;
    const bool active = lane < CHUNK;
    const int g = chunk * CHUNK + (active ? lane : CHUNK - 1);
    const GDesc gd = tb.gdesc[g];
    LaneG L;
    L.lo = gd.lo; L.up = gd.up; L.ng = gd.ng; L.off_self = gd.off_self; L.off_for = gd.off_for;
    L.band = gd.band; L.off_rayla = gd.off_rayla;
    const double dwave = c_wavemid[gd.band] - .6;

    // McICA: this lane's sub-column of binary cloud (generate_stochastic_clouds_sw, SW:1737-1772, icld = 2):
    // bit k of cmask = cloudy in layer k.  The lane jumps the column's KISS sequence to its own nlay draws.
    uint64_t cmask = 0, cmhi = 0;
    int iceflag = 3;
    int col_cloudy = 0;
    if (CLOUD) {
        if (a.has_reqi != 0) iceflag = 4;
        if (a.has_reqs != 0 || (a.has_reqi != 0 && a.has_reqc != 0)) iceflag = 5;
        KissState ks = {(uint32_t)c.seeds[0], (uint32_t)c.seeds[1], (uint32_t)c.seeds[2], (uint32_t)c.seeds[3]};
        kiss_jump(ks, a.jump + (size_t)g * KISS_JUMP_WORDS);
        double prev = 0.;
        for (int l = 0; l < nlay; ++l) {
            double cur = kiss_next(ks);
            if (l > 0) {
                const double cb = c.cldf[l - 1];
                cur = prev > 1. - cb ? prev : cur * (1. - cb);
            }
            if (cur >= 1. - c.cldf[l]) {
                if (l < 64) cmask |= (uint64_t)1 << l; else cmhi |= (uint64_t)1 << (l - 64);
            }
            prev = cur;
        }
        if (nlay > 64) c.cmhi[tid] = cmhi;
        col_cloudy = __syncthreads_or((cmask | cmhi) != 0);
    }

    const double prmu0 = coszrs <= 1.e-10 ? 1.e-10 : coszrs;   // SW:9254-9255
    const double rprmu0 = 1.0 / prmu0;
    // solar source at the layer selected per band (SW:3363-3380, 3533-3538, 4297) -> incoming flux
    double sfluxzen;
    {
        int laysolfr;
        if (gd.solar_kind == SOL_UPPER) {
            laysolfr = nlay;   // 1-based, as in the reference
            for (int lay = laytrop + 1; lay <= nlay; ++lay)
                if (c.jp[lay - 2] < gd.layreffr && c.jp[lay - 1] >= gd.layreffr) laysolfr = lay;
        } else {
            laysolfr = laytrop;
            if (gd.solar_kind == SOL_LOWER)
                for (int lay = 1; lay <= laytrop; ++lay)
                    if (c.jp[lay - 1] < gd.layreffr && c.jp[lay] >= gd.layreffr)
                        laysolfr = (lay + 1 < laytrop) ? lay + 1 : laytrop;
        }
        const int ks = laysolfr - 1;
        const bool lowers = (laysolfr <= laytrop);
        const GRegime &R = lowers ? gd.lo : gd.up;
        int jsi = 0;
        double fs = 0.;
        if (gd.nsf > 1) {
            const int spa = (R.cfg >> CFG_SPA_SHIFT) & 7, spb = (R.cfg >> CFG_SPB_SHIFT) & 7;
            double colA = c.col[spa * nlp + ks], colB = c.col[spb * nlp + ks];
            double speccomb = colA + R.strrat * colB;
            double specparm = colA / speccomb;
            specparm = specparm >= tb.oneminus ? tb.oneminus : specparm;
            double specmult = ((R.cfg & CFG_MULT4) ? 4. : 8.) * specparm;
            jsi = (int)specmult;
            fs = specmult - (double)jsi;
        }
        const double *Sf = tb.K + gd.off_sflux + (size_t)jsi * gd.ng;
        double s0 = Sf[0];
        double s1 = gd.nsf > 1 ? Sf[gd.ng] : s0;
        sfluxzen = gd.nsf > 1 ? s0 + fs * (s1 - s0) : gd.scalekur * s0;
    }
    const double adjflux = 1.0 * (a.solcon / 1.36822e+03);   // adjes = 1, dyofyr = 0 (SW:10623-10626, 9735-9739)
    const double zincflx = active ? adjflux * sfluxzen * prmu0 : 0.;   // SW:8363
    // surface albedo of this lane's band, direct and diffuse: the broadband value (SW:10699-10702) or, over SSiB
    // land, the UV/visible pair in bands 25-28 and the near-IR pair elsewhere (SW:10692-10697, 9266-9277)
    double albp = ldr(a.albedo, i2, r4), albd = albp;
    if (a.alswvisdir && ldr(a.xland, i2, r4) < 1.5) {
        albp = gd.uvband ? ldr(a.alswvisdir, i2, r4) : ldr(a.alswnirdir, i2, r4);
        albd = gd.uvband ? ldr(a.alswvisdif, i2, r4) : ldr(a.alswnirdif, i2, r4);
    }
    // (an idle lane points at the row's last slot: it reads that lane's values and stores nothing)
    const int slane = active ? lane : CHUNK - 1;
    double *ck = scr_block + (size_t)NWAVE * nlay * SW_SCR_VALS * SCR_ROW + (size_t)wave * nblk * 2 * SCR_ROW + slane;
    double *scr = scr_block + (size_t)wave * nlay * SW_SCR_VALS * SCR_ROW + slane;
    // a layer's block: [56 lanes][ref, refd] [56 lanes][tra, trad] [56 lanes] dbt; this lane's pairs start at 2 * slane
    double *scr_layer = scr_block + (size_t)wave * nlay * SW_SCR_VALS * SCR_ROW + 2 * slane;
    (void)scr;
    int bad = 0;
    // the lanes that store wave sums: lane 16r holds row r, i.e. value wsum_index(r) of a wave_sum4
    const bool writer = (lane & 15) == 0;
    const int wrow = wsum_index(lane >> 4);
    double *wsums = c.sums + wave * SUMS_N;

    // One upward sweep of this wave's 56 sub-columns carrying up to two calls on the same gas optics:
    //   P (hasP): fluxes at every level -> pflux[2l], pflux[2l+1]; p_aer = with aerosol optical depth, p_mask = the
    //     sub-column's cloudy layers are in (0: clear sky); want_sfc: the surface direct / UV / near-IR sums too.
    //   S (hasS): TOA and surface fluxes only -> wsums[sdst .. sdst+3]; s_aer, s_mask alike.
    auto sweep = [&](const bool hasP, const bool p_aer, const bool p_mask, const bool want_sfc, double *pflux,
                     const bool hasS, const bool s_aer, const bool s_mask, const int sdst) {
        // ---- upward sweep ----
        double rup = albp, rupd = albd, tdbt_nodel = 1.;
        Stack st = {0., 0., 1., 1., 0., 1.};
        GasPre gp;
        gas_fetch(tb, c, L, 0, 0 < laytrop, gp);
        for (int k = 0; k < nlay; ++k) {
            double taug, taur;
            gas_finish(tb, c, L, k, k < laytrop, gp, taug, taur);
            // The table values of the next layer are read ahead of the arithmetic.  gp is dead from here on, so the
            // reads land in the same registers (no copies): behind the P stream when an S stream follows (its
            // arithmetic covers the latency and the P stream runs with fewer live registers), else right away.
            const int kn = k + 1 < nlay ? k + 1 : k;
            if (!hasS) gas_fetch(tb, c, L, kn, kn < laytrop, gp);
            const int ia = L.band * nlp + k;
            const double ta = c.taer[ia];
            double wa, ga;
            if (compact) {   // omega and g of this lane's band from the layer's two lines (SW:11006-11021)
                wa = fmin(fmax(c.waer[k] * dwave + c.waer[nlp + k], 0.4), 1.0);
                ga = fmin(fmax(c.gaer[k] * dwave + c.gaer[nlp + k], 0.5), 1.0);
            } else {
                wa = c.waer[ia]; ga = c.gaer[ia];
            }
            bool cloudy = false;
            CldLayer cl = {0., 1., 0., 0.};
            if (CLOUD) {
                cloudy = k < 64 ? (cmask >> k) & 1 : (nlay > 64 ? (c.cmhi[tid] >> (k - 64)) & 1 : 0);
                if (cloudy && ((hasP && p_mask) || (hasS && s_mask))) {
                    int b = 0;
                    cloud_optics(a.cldtab, c, L.band, k, iceflag, cl.tc, cl.oc, cl.gc, cl.taor, b);
                    if (b) bad = b;
                }
            }
            // the cheap form of the S stream needs ta = 0 and no cloudy lane in this layer (wave-uniform choice); its two
            // table reads go out here, in front of the P stream
            const bool s_general = s_aer || (CLOUD && s_mask && __any(cloudy));
            CleanPre cpre = {0., 0., 0.};
            if (hasS && !s_general) cpre = layer_rt_clean_issue(tb, taug, taur, rprmu0);
            RT rtp;
            if (hasP) {
                if ((k & (BLK - 1)) == 0 && active) {
                    ck[((k / BLK) * 2 + 0) * SCR_ROW] = rup;
                    ck[((k / BLK) * 2 + 1) * SCR_ROW] = rupd;
                }
                RT rt;
                double dnl = 1.;
                layer_rt(want_sfc, tb, taug, taur, p_aer ? ta : 0.0, wa, ga, prmu0, rprmu0, cloudy && p_mask, cl, rt, dnl);
                rtp = rt;   // stored at the end of the iteration, see below
                tdbt_nodel = dnl * tdbt_nodel;
                // vrtqdr_sw, SW:7997-8015
                const double zreflect = frcp1(1. - rupd * rt.refd);
                const double nrup = rt.ref + (rt.trad * ((rt.tra - rt.dbt) * rupd + rt.dbt * rup)) * zreflect;
                const double nrupd = rt.refd + rt.trad * rt.trad * rupd * zreflect;
                rup = nrup; rupd = nrupd;
            }
            if (hasS) {
                gas_fetch(tb, c, L, kn, kn < laytrop, gp);
                RT rt;
                if (s_general) {
                    double dnl;
                    layer_rt(false, tb, taug, taur, s_aer ? ta : 0.0, wa, ga, prmu0, rprmu0, cloudy && s_mask, cl, rt, dnl);
                } else {
                    layer_rt_clean_finish(taug, taur, prmu0, rprmu0, cpre, rt);
                }
                add_on_top(st, rt);
            }
            if (hasP) {
                // keep the layer's two-stream result for the way down.  Last in the iteration: vector memory
                // operations retire in order, and a table gather issued behind these stores would have its first
                // use wait for their acknowledgement too -- here the next gather is a whole layer preparation away
                // (ref, refd) and (tra, trad) as 16-byte pairs, dbt alone: three store instructions per layer instead of
                // five -- the vector cache takes a 16-byte access per lane at the price of an 8-byte one
                double *s = scr_layer + (size_t)k * SW_SCR_VALS * SCR_ROW;
                if (active) {
                    *reinterpret_cast<double2 *>(s) = make_double2(rtp.ref, rtp.refd);
                    *reinterpret_cast<double2 *>(s + 2 * SCR_ROW) = make_double2(rtp.tra, rtp.trad);
                    s[4 * SCR_ROW - slane] = rtp.dbt;
                }
            }
        }
        if (hasS) {
            // the surface under the stack (SW:8038-8044 at the lowest level, SW:8003-8006 for the top)
            const double zreflect = frcp1(1. - st.rb * albd);
            const double fu0 = (st.dbt * albp + (st.tra - st.dbt) * albd) * zreflect;
            const double fd0 = st.dbt + (st.tra - st.dbt + st.dbt * albp * st.rb) * zreflect;
            const double futop = st.ref + st.trad * fu0;
            const double q = wave_sum4(zincflx * futop, zincflx, zincflx * fu0, zincflx * fd0);
            if (writer) wsums[sdst + wrow] = q;
        }
        if (!hasP) return;

        // ---- downward sweep by blocks: read the block's layers back, rebuild its upward reflectances, descend ----
        double tdn = 1., rdnd = 0., tdbt = 1.;
        for (int b = nblk - 1; b >= 0; --b) {
            RT rt[BLK];
            double ru[BLK + 1], rud[BLK + 1];
            ru[0] = ck[(b * 2 + 0) * SCR_ROW];
            rud[0] = ck[(b * 2 + 1) * SCR_ROW];
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const int k = b * BLK + j;
                if (k < nlay) {
                    const double *s = scr_layer + (size_t)k * SW_SCR_VALS * SCR_ROW;
                    const double2 rr = *reinterpret_cast<const double2 *>(s), tt = *reinterpret_cast<const double2 *>(s + 2 * SCR_ROW);
                    rt[j].ref = rr.x; rt[j].refd = rr.y; rt[j].tra = tt.x; rt[j].trad = tt.y;
                    rt[j].dbt = s[4 * SCR_ROW - slane];
                } else {
                    rt[j].ref = 0.; rt[j].refd = 0.; rt[j].tra = 1.; rt[j].trad = 1.; rt[j].dbt = 1.;
                }
                const double zreflect = frcp1(1. - rud[j] * rt[j].refd);
                ru[j + 1] = rt[j].ref + (rt[j].trad * ((rt[j].tra - rt[j].dbt) * rud[j] + rt[j].dbt * ru[j])) * zreflect;
                rud[j + 1] = rt[j].refd + rt[j].trad * rt[j].trad * rud[j] * zreflect;
            }
            double fu[BLK], fd[BLK];
#pragma unroll
            for (int j = BLK - 1; j >= 0; --j) {
                {   // fluxes at level l = b*BLK + j + 1 (top of layer k = b*BLK + j), SW:8038-8044
                    const double zreflect = frcp1(1. - rdnd * rud[j + 1]);
                    fu[j] = zincflx * ((tdbt * ru[j + 1] + (tdn - tdbt) * rud[j + 1]) * zreflect);
                    fd[j] = zincflx * (tdbt + (tdn - tdbt + tdbt * ru[j + 1] * rdnd) * zreflect);
                }
                // through layer k, SW:8026-8034 and SW:8578
                const double zreflect = frcp1(1. - rt[j].refd * rdnd);
                const double ntdn = tdbt * rt[j].tra + (rt[j].trad * ((tdn - tdbt) + tdbt * rt[j].ref * rdnd)) * zreflect;
                const double nrdnd = rt[j].refd + rt[j].trad * rt[j].trad * rdnd * zreflect;
                tdn = ntdn; rdnd = nrdnd;
                tdbt = rt[j].dbt * tdbt;
            }
            // spectral sums of the block's eight values (SW:8629-8634); levels above the real top (identity padding)
            // land in the padded tail of the flux array
            static_assert(BLK == 4, "two wave_sum4 per block");
            const double q0 = wave_sum4(fu[0], fd[0], fu[1], fd[1]);
            const double q1 = wave_sum4(fu[2], fd[2], fu[3], fd[3]);
            if (writer) {
                double *f = pflux + 2 * (b * BLK + 1) + wrow;
                f[0] = q0; f[4] = q1;
            }
        }
        {   // surface level
            const double zreflect = frcp1(1. - rdnd * albd);
            const double fu = (tdbt * albp + (tdn - tdbt) * albd) * zreflect;
            const double fd = tdbt + (tdn - tdbt + tdbt * albp * rdnd) * zreflect;
            if (want_sfc) {
                // with the surface-only direct / UV-visible / near-IR sums of the total-sky call (SW:8633-8647, ikl = 1)
                const double uv = gd.uvband ? zincflx : 0., ni = gd.uvband ? 0. : zincflx;
                const double q0 = wave_sum4(zincflx * fu, zincflx * fd, zincflx * tdbt_nodel, uv * fd);
                const double q1 = wave_sum4(uv * tdbt_nodel, ni * fd, ni * tdbt_nodel, 0.);
                if (writer) {
                    if (wrow < 2) pflux[wrow] = q0; else wsums[wrow == 2 ? SUM_DIRBB : SUM_UVFD] = q0;
                    if (wrow < 3) wsums[wrow == 0 ? SUM_UVDIR : (wrow == 1 ? SUM_NIFD : SUM_NIDIR)] = q1;
                }
            } else {
                const double q0 = wave_sum4(zincflx * fu, zincflx * fd, 0., 0.);
                if (writer && wrow < 2) pflux[wrow] = q0;
            }
        }
    };

    double *flux = c.flux + (size_t)wave * (nlp + 1) * 2;
    const bool do_cln = a.clean > 0;
    // the clean-sky call differs from the one with aerosol only if there is aerosol: without any, the reference's two
    // calls are the same arithmetic on the same numbers and the CLN outputs are copies
    const bool cln_own = do_cln && (a.aer_ra_feedback == 1 || a.tauaer3d_sw != nullptr || a.aerod != nullptr);
    if (lane < SUMS_N) wsums[lane] = 0.;   // clean_atm_diag <= 0: CLN fluxes are exactly 0 (SW:9473-9478)
    if (PROF) {
        // every call as a P stream: with aerosol; clean sky (call #2, SW:9439-9467); clear sky of a cloudy column
        // (zbbcu / zbbcd of call #1, SW:8405-8416, 8590-8593).  One call site keeps one copy of the sweep.
        double *fc = c.fluxc + (size_t)wave * (nlp + 1) * 2;
        if (!do_cln) for (int l = lane; l < 2 * nlev; l += 64) fc[l] = 0.;
        const int nround = CLOUD && col_cloudy ? 3 : 2;
        for (int r = 0; r < nround; ++r) {
            if (r == 1 && !do_cln) continue;
            sweep(true, r != 1, r != 2, r == 0, r == 0 ? flux : (r == 1 ? fc : c.fluxk + (size_t)wave * (nlp + 1) * 2),
                  false, false, false, 0);
        }
    } else {
        // round 0: P = total sky with aerosol (call #1), S = total sky without aerosol (call #2); round 1, cloudy
        // columns only: S = the clear-sky fluxes of call #1.  Cloud-free columns reuse the total-sky result.
        const int nround = CLOUD && col_cloudy ? 2 : 1;
        for (int r = 0; r < nround; ++r)
            sweep(r == 0, true, true, true, flux, r == 0 ? cln_own : true, r == 1, r == 0, r == 0 ? SUM_S : SUM_K);
    }
    if (CLOUD && bad) atomicMax(a.status, (bad << 28) | (1 + col));
    __syncthreads();

    // ======================= epilogue: lanes = levels =======================
    const int fstride = (nlp + 1) * 2;
    const double *fA0 = c.flux, *fA1 = c.flux + fstride;
    const double *s0 = c.sums, *s1 = c.sums + SUMS_N;
    for (int l = tid; l < nlev; l += NTHREADS) {
        const double up = fA0[2 * l] + fA1[2 * l], dn = fA0[2 * l + 1] + fA1[2 * l + 1];
        if (PROF && a.swupflx) {   // (i, kts:kte+2, j), SW:11156-11165
            const double *fC0 = c.fluxc, *fC1 = c.fluxc + fstride;
            const double upc = fC0[2 * l] + fC1[2 * l], dnc = fC0[2 * l + 1] + fC1[2 * l + 1];
            // clear sky: third sweep of a cloudy column, else identical to the total-sky result (SW:8428, 8556)
            const double *fK0 = CLOUD && col_cloudy ? c.fluxk : fA0, *fK1 = CLOUD && col_cloudy ? c.fluxk + fstride : fA1;
            const double upk = fK0[2 * l] + fK1[2 * l], dnk = fK0[2 * l + 1] + fK1[2 * l + 1];
            const size_t q = base3f + (size_t)l * sk;
            str(a.swupflx, q, r4, up); str(a.swupflxc, q, r4, upk); str(a.swupflxcln, q, r4, upc);
            str(a.swdnflx, q, r4, dn); str(a.swdnflxc, q, r4, dnk); str(a.swdnflxcln, q, r4, dnc);
        }
        if (l < nk && a.rthratensw) {
            // heating rate of layer l, SW:9428-9434, then SW:11167-11170
            const double up1 = fA0[2 * (l + 1)] + fA1[2 * (l + 1)], dn1 = fA0[2 * (l + 1) + 1] + fA1[2 * (l + 1) + 1];
            const double net0 = dn - up, net1 = dn1 - up1;
            const double zdpgcp = tb.heatfac / c.pdp[l];
            const double swhr = (net1 - net0) * zdpgcp;
            const double tten = swhr / 86400.;
            str(a.rthratensw, base3 + (size_t)l * sk, r4, tten / ldr(a.pi3d, base3 + (size_t)l * sk, r4));
        }
        if (l == 0 || l == nlay) {
            // clean sky and clear sky at the surface (l = 0) / the top (l = nlay)
            double upc, dnc, upk, dnk;
            if (PROF) {
                const double *fC0 = c.fluxc, *fC1 = c.fluxc + fstride;
                upc = fC0[2 * l] + fC1[2 * l]; dnc = fC0[2 * l + 1] + fC1[2 * l + 1];
                const double *fK0 = CLOUD && col_cloudy ? c.fluxk : fA0, *fK1 = CLOUD && col_cloudy ? c.fluxk + fstride : fA1;
                upk = fK0[2 * l] + fK1[2 * l]; dnk = fK0[2 * l + 1] + fK1[2 * l + 1];
            } else {
                const int o = l == 0 ? 2 : 0;
                upc = s0[SUM_S + o] + s1[SUM_S + o]; dnc = s0[SUM_S + o + 1] + s1[SUM_S + o + 1];
                if (do_cln && !cln_own) { upc = up; dnc = dn; }
                if (CLOUD && col_cloudy) {
                    upk = s0[SUM_K + o] + s1[SUM_K + o]; dnk = s0[SUM_K + o + 1] + s1[SUM_K + o + 1];
                } else {
                    upk = up; dnk = dn;
                }
            }
            if (l == 0) {
                if (a.gsw) str(a.gsw, i2, r4, dn - up);
                if (a.swupt) {
                    str(a.swupb, i2, r4, up); str(a.swupbc, i2, r4, upk); str(a.swupbcln, i2, r4, upc);
                    str(a.swdnb, i2, r4, dn); str(a.swdnbc, i2, r4, dnk); str(a.swdnbcln, i2, r4, dnc);
                }
                const double dirbb = s0[SUM_DIRBB] + s1[SUM_DIRBB], uvfd = s0[SUM_UVFD] + s1[SUM_UVFD],
                             uvdir = s0[SUM_UVDIR] + s1[SUM_UVDIR], nifd = s0[SUM_NIFD] + s1[SUM_NIFD],
                             nidir = s0[SUM_NIDIR] + s1[SUM_NIDIR];
                if (a.swupt && a.swvisdir) {
                    str(a.swvisdir, i2, r4, uvdir); str(a.swvisdif, i2, r4, uvfd - uvdir);
                    str(a.swnirdir, i2, r4, nidir); str(a.swnirdif, i2, r4, nifd - nidir);
                }
                if (a.swddir) {
                    str(a.swddir, i2, r4, dirbb);
                    str(a.swddni, i2, r4, dirbb / coszrs);
                    str(a.swddif, i2, r4, dn - dirbb);
                }
            }
            if (l == nlay) {
                if (a.swcf) str(a.swcf, i2, r4, (dn - up) - (dnk - upk));   // SW:11126
                if (a.swupt) {
                    str(a.swupt, i2, r4, up); str(a.swuptc, i2, r4, upk); str(a.swuptcln, i2, r4, upc);
                    str(a.swdnt, i2, r4, dn); str(a.swdntc, i2, r4, dnk); str(a.swdntcln, i2, r4, dnc);
                }
            }
        }
    }
}

// One workgroup per column of the launch; its scratch slice: the layers' two-stream results [2 waves][nlay][5 x 56]
// (per layer 56 pairs (ref, refd), 56 pairs (tra, trad), 56 dbt)
// and the upward sweep's checkpoints [2 waves][nblk][2][56].
// the McICA instantiations: two waves per SIMD and up to 256 VGPRs (231 used, no spill) -- at three waves and 168 VGPRs
// they spilled 140 / 220 bytes per lane (15.7 -> 15.1 ms for the 50 000 cloudy columns of the section-8d atmosphere)
#ifndef SW_CLOUD_WAVES
#define SW_CLOUD_WAVES 2
#endif
template <bool CLOUD, bool PROF>
__global__ void __launch_bounds__(NTHREADS, CLOUD ? SW_CLOUD_WAVES : SW_WAVES_PER_SIMD) sw_column_kernel(const SwKernelArgs a)
{
    extern __shared__ double lds[];
    double *scr = a.scratch + (size_t)blockIdx.x * a.scratch_stride;
    sw_column<CLOUD, PROF>(a, blockIdx.x, lds, scr);
}

size_t sw_scratch_doubles_per_column(int nk)
{
    const size_t nlay = nk + 1, nblk = (nlay + BLK - 1) / BLK;
    return NWAVE * nlay * SW_SCR_VALS * SCR_ROW + NWAVE * nblk * 2 * SCR_ROW;
}

// ---------------------------------------------------------------------------
size_t sw_lds_bytes(int nk, bool cloud, bool prof, bool compact)
{
    const int nlay = nk + 1;
    const int nlp = (nlay + BLK - 1) / BLK * BLK;
    size_t nd = (size_t)nlp * (8 + SP_COUNT + 2 + (compact ? SW_NBND + 4 : 3 * SW_NBND)) + (size_t)NWAVE * (nlp + 1) * 2 +
                NWAVE * SUMS_N;
    if (prof) nd += (size_t)NWAVE * (nlp + 1) * 2 * (cloud ? 2 : 1);
    if (cloud) nd += (size_t)nlp * 7 + NTHREADS;
    size_t ni = (size_t)nlp * 5 + 1 + (cloud ? (size_t)nlp * 3 + 4 : 0);
    return nd * sizeof(double) + ((ni * sizeof(int) + 7) & ~size_t(7));
}

hipError_t sw_upload_constants(hipStream_t)
{
    // O3DATA annual profile, LW:12758-12790
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
    static const double wavemin[SW_NBND] = {3.077, 2.500, 2.150, 1.942, 1.626, 1.299, 1.242,
                                            0.778, 0.625, 0.442, 0.345, 0.263, 0.200, 3.846};
    static const double wavemax[SW_NBND] = {3.846, 3.077, 2.500, 2.150, 1.942, 1.626, 1.299,
                                            1.242, 0.778, 0.625, 0.442, 0.345, 0.263, 12.195};
    double wavemid[SW_NBND];
    for (int i = 0; i < SW_NBND; ++i) wavemid[i] = 0.5 * (wavemin[i] + wavemax[i]);
    hipError_t e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_o3ann), o3ann, sizeof o3ann)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_ppwrkh), ppwrkh, sizeof ppwrkh)) != hipSuccess) return e;
    double lnwave[SW_NBND];
    for (int i = 0; i < SW_NBND; ++i) lnwave[i] = std::log(0.4 / wavemid[i]);
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_lnwave), lnwave, sizeof lnwave)) != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_wavemid), wavemid, sizeof wavemid);
}

hipError_t sw_upload_retab()
{
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_swaer), SWAER_RSRTAUA, sizeof(double) * SW_NBND * 6, 0);
    if (e == hipSuccess)
        e = hipMemcpyToSymbol(HIP_SYMBOL(c_swaer), SWAER_RSRPIZA, sizeof(double) * SW_NBND * 6, sizeof(double) * SW_NBND * 6);
    if (e == hipSuccess)
        e = hipMemcpyToSymbol(HIP_SYMBOL(c_swaer), SWAER_RSRASYA, sizeof(double) * SW_NBND * 6, 2 * sizeof(double) * SW_NBND * 6);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_retab), CLD_RETAB, sizeof(double) * 95);
}

hipError_t sw_launch(const SwKernelArgs &a, hipStream_t stream)
{
    const bool cloud = a.icloud != 0, prof = a.swupflx != nullptr;
    const size_t lds = sw_lds_bytes(a.nk, cloud, prof, a.tauaer3d_sw == nullptr && a.aerod == nullptr);
    const void *fn = cloud ? (prof ? reinterpret_cast<const void *>(sw_column_kernel<true, true>)
                                   : reinterpret_cast<const void *>(sw_column_kernel<true, false>))
                           : (prof ? reinterpret_cast<const void *>(sw_column_kernel<false, true>)
                                   : reinterpret_cast<const void *>(sw_column_kernel<false, false>));
    {
        // grow-only opt-in for > 64 KB of dynamic LDS, per instantiation.  Concurrent tile threads launch from here
        // (mem = 0 calls run outside the engine lock), so the bookkeeping has its own lock; the launch stays outside it
        static std::mutex attr_mutex;
        static size_t attr_bytes[4] = {0, 0, 0, 0};
        std::lock_guard<std::mutex> lock(attr_mutex);
        size_t &ab = attr_bytes[(cloud ? 2 : 0) + (prof ? 1 : 0)];
        if (lds > ab) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            ab = lds;
        }
    }
    if (cloud) {
        if (prof) hipLaunchKernelGGL((sw_column_kernel<true, true>), dim3(a.ncol), dim3(NTHREADS), lds, stream, a);
        else hipLaunchKernelGGL((sw_column_kernel<true, false>), dim3(a.ncol), dim3(NTHREADS), lds, stream, a);
    } else {
        if (prof) hipLaunchKernelGGL((sw_column_kernel<false, true>), dim3(a.ncol), dim3(NTHREADS), lds, stream, a);
        else hipLaunchKernelGGL((sw_column_kernel<false, false>), dim3(a.ncol), dim3(NTHREADS), lds, stream, a);
    }
    return hipGetLastError();
}

}  // namespace rrtmg
