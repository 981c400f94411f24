// Host-side tables of the McICA cloud path.
//
// 1. KISS jump-ahead.  The reference draws the sub-column random numbers of a column strictly in sequence
//    (generate_stochastic_clouds_sw, SW:1737-1747: for isubcol, for ilev: kissvec), 112 x nlay steps of a 128-bit
//    state.  On the device every lane is one sub-column (= g-point), so each lane must enter the sequence at
//    change_seed + isubcol*nlay.  All four component generators of kissvec (SW:1900-1932) can be advanced n
//    steps in O(log n):
//      seed1  s -> 69069 s + 1327217885 (mod 2^32)           affine map, composed by squaring
//      seed2  xorshift 13, -17, 5                             linear over GF(2): 32x32 bit matrix, squared
//      seed3  s -> 18000 (s & 65535) + (s >> 16)              multiply-with-carry == s * 18000^n mod (18000*2^16 - 1)
//      seed4  s -> 30903 (s & 65535) + (s >> 16)              likewise mod (30903*2^16 - 1)
//    (for the MWC pair: 2^16 * next == s mod p, and 2^16 has inverse a mod p = a*2^16 - 1; states stay in [0, p)
//    when the seed is < 10^9, which the seeding rule guarantees.)
// 2. Cloud optics coefficient tables in the layout the kernel reads.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "cloud_lists.h"

#include "sw_kernel.h"
#include "sw_cld_tables.h"
#include "lw_cld_tables.h"
#include "lw_kernel.h"

namespace rrtmg {

namespace {

struct Affine { uint32_t a, c; };
Affine compose(Affine f, Affine g) { return {f.a * g.a, f.a * g.c + f.c}; }   // f(g(s))

struct BitMat { uint32_t col[32]; };   // col[j] = image of bit j
uint32_t apply(const BitMat &m, uint32_t s)
{
    uint32_t r = 0;
    for (int j = 0; j < 32; ++j)
        if ((s >> j) & 1u) r ^= m.col[j];
    return r;
}
BitMat compose(const BitMat &f, const BitMat &g)
{
    BitMat r;
    for (int j = 0; j < 32; ++j) r.col[j] = apply(f, g.col[j]);
    return r;
}
uint32_t xorshift(uint32_t s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}
uint32_t powmod(uint32_t a, uint64_t n, uint32_t p)
{
    uint64_t r = 1, b = a % p;
    while (n) {
        if (n & 1) r = r * b % p;
        b = b * b % p;
        n >>= 1;
    }
    return (uint32_t)r;
}

}  // namespace

void kiss_jump_table(int nlay, int change_seed, int nsub, uint32_t *out)
{
    BitMat x1;
    for (int j = 0; j < 32; ++j) x1.col[j] = xorshift(1u << j);
    const uint32_t p3 = 18000u * 65536u - 1u, p4 = 30903u * 65536u - 1u;
    for (int g = 0; g < nsub; ++g) {
        uint64_t n = (uint64_t)change_seed + (uint64_t)g * (uint64_t)nlay;
        Affine f = {1u, 0u}, fb = {69069u, 1327217885u};
        BitMat m, mb = x1;
        for (int j = 0; j < 32; ++j) m.col[j] = 1u << j;
        for (uint64_t e = n; e; e >>= 1) {
            if (e & 1) { f = compose(fb, f); m = compose(mb, m); }
            fb = compose(fb, fb);
            mb = compose(mb, mb);
        }
        uint32_t *o = out + (size_t)g * KISS_JUMP_WORDS;
        o[0] = f.a; o[1] = f.c;
        std::memcpy(o + 2, m.col, sizeof m.col);
        o[34] = powmod(18000u, n, p3);
        o[35] = powmod(30903u, n, p4);
    }
}

void sw_cloud_tables(double *out)
{
    std::memset(out, 0, sizeof(double) * CLD_NTAB * SW_NBND * CLD_STRIDE);
    for (int b = 0; b < SW_NBND; ++b) {
        for (int i = 0; i < 58; ++i) {
            out[(0 * SW_NBND + b) * CLD_STRIDE + i] = SWCLD_EXTLIQ1[b][i];
            out[(1 * SW_NBND + b) * CLD_STRIDE + i] = SWCLD_SSALIQ1[b][i];
            out[(2 * SW_NBND + b) * CLD_STRIDE + i] = SWCLD_ASYLIQ1[b][i];
        }
        for (int i = 0; i < 46; ++i) {
            out[(3 * SW_NBND + b) * CLD_STRIDE + i] = SWCLD_EXTICE3[b][i];
            out[(4 * SW_NBND + b) * CLD_STRIDE + i] = SWCLD_SSAICE3[b][i];
            out[(5 * SW_NBND + b) * CLD_STRIDE + i] = SWCLD_ASYICE3[b][i];
            out[(6 * SW_NBND + b) * CLD_STRIDE + i] = SWCLD_FDLICE3[b][i];
        }
    }
}

void lw_cloud_tables(double *out)
{
    std::memset(out, 0, sizeof(double) * 2 * LW_NBND * 58);
    for (int b = 0; b < LW_NBND; ++b) {
        for (int i = 0; i < 58; ++i) out[(0 * LW_NBND + b) * 58 + i] = LWCLD_ABSLIQ1[b][i];
        for (int i = 0; i < 46; ++i) out[(1 * LW_NBND + b) * 58 + i] = LWCLD_ABSICE3[b][i];
    }
}


// ---- cloud lists (cloud_lists.h) ----
namespace {
__global__ void cloud_lists_kernel(const void *cldfra3d, int real4, int ni_m, int nk_m, int i0, int j0, int k0, int ni_t,
                                   int ncol, int nk, int *lists, int *counts)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = col < ncol;
    bool cloudy = false;
    if (valid) {
        const int it = col % ni_t, jt = col / ni_t;
        const size_t base = ((size_t)(j0 + jt) * nk_m + k0) * ni_m + (size_t)(i0 + it);
        for (int k = 0; k < nk; ++k) {
            const size_t q = base + (size_t)k * ni_m;
            const double cf = real4 ? (double)static_cast<const float *>(cldfra3d)[q] : static_cast<const double *>(cldfra3d)[q];
            cloudy = cloudy || !(cf < 1.0e-20);   // the kernels' own threshold for "no cloud in this layer"
        }
    }
    // one atomic per wave and list; the wave's columns stay together and in order
    const unsigned long long mc = __ballot(valid && cloudy), mf = __ballot(valid && !cloudy);
    const int lane = threadIdx.x & 63;
    int basec = 0, basef = 0;
    if (lane == 0) {
        if (mc) basec = atomicAdd(&counts[1], __popcll(mc));
        if (mf) basef = atomicAdd(&counts[0], __popcll(mf));
    }
    basec = __shfl(basec, 0, 64);
    basef = __shfl(basef, 0, 64);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (valid) {
        if (cloudy) lists[ncol + basec + __popcll(mc & below)] = col;
        else lists[basef + __popcll(mf & below)] = col;
    }
}
}  // namespace

hipError_t cloud_lists_launch(const void *cldfra3d, int real4, int ni_m, int nk_m, int i0, int j0, int k0, int ni_t, int nj_t,
                              int nk, int *lists, int *counts, hipStream_t stream)
{
    const int ncol = ni_t * nj_t;
    hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cloud_lists_kernel, dim3((ncol + 255) / 256), dim3(256), 0, stream, cldfra3d, real4, ni_m, nk_m, i0, j0,
                       k0, ni_t, ncol, nk, lists, counts);
    return hipGetLastError();
}

bool cloud_lists_enabled()
{
    static const bool on = [] {
        const char *e = std::getenv("RRTMG_HIP_CLOUD_LISTS");
        return !(e && e[0] == '0');
    }();
    return on;
}

}  // namespace rrtmg
