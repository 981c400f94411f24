// Device side of the McICA sub-column generator shared by the shortwave and longwave kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rrtmg {

// kissvec, SW:1900-1932 (identical in LW:2580-2612): one draw of the column's generator (32-bit wrap-around
// arithmetic, logical shifts)
struct KissState { uint32_t s1, s2, s3, s4; };

__device__ __forceinline__ double kiss_next(KissState &k)
{
#pragma clang fp contract(off)   // kiss*2.328306e-10 + 0.5 is a product and a sum in the reference, not an FMA
    k.s1 = 69069u * k.s1 + 1327217885u;
    k.s2 ^= k.s2 << 13;
    k.s2 ^= k.s2 >> 17;
    k.s2 ^= k.s2 << 5;
    k.s3 = 18000u * (k.s3 & 65535u) + (k.s3 >> 16);
    k.s4 = 30903u * (k.s4 & 65535u) + (k.s4 >> 16);
    const int32_t kiss = (int32_t)(k.s1 + k.s2 + (k.s3 << 16) + k.s4);
    return (double)kiss * 2.328306e-10 + 0.5;
}

// advance by the number of draws encoded in J (kiss_jump_table, mcica.hip): affine map, GF(2) matrix, two
// multiply-with-carry factors
__device__ __forceinline__ void kiss_jump(KissState &k, const uint32_t *J)
{
    k.s1 = J[0] * k.s1 + J[1];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) r ^= J[2 + j] & (0u - ((k.s2 >> j) & 1u));
    k.s2 = r;
    k.s3 = (uint32_t)(((uint64_t)k.s3 * J[34]) % (18000u * 65536u - 1u));
    k.s4 = (uint32_t)(((uint64_t)k.s4 * J[35]) % (30903u * 65536u - 1u));
}

}  // namespace rrtmg
