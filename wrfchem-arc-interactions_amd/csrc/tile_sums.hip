// Per-tile sums of the TOA / surface flux fields: the only quantity that
// crosses GPUs.  The reference's post-processing forms domain means
// of these fields (NCL_extraction_package/misc_stats_library.ncl:433 `avg`);
// here every rank reduces its own (i,j) slab on the device and the 16-double
// vectors are summed with one RCCL all-reduce by the host code.
//
// Deterministic: one workgroup per field, fixed strided accumulation order and
// a fixed LDS tree, so two runs give identical bits.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/rrtmg_hip.h"

namespace {

struct SumArgs {
    const void *f[12];
    int real4;
    int ni_m, i0, j0, ni_t, nj_t;
    double *out;   // [16]
};

__global__ void __launch_bounds__(256) tile_sums_kernel(const SumArgs a)
{
    __shared__ double red[256];
    const int field = blockIdx.x, tid = threadIdx.x;
    const void *p = a.f[field];
    double s = 0.;
    const int n = a.ni_t * a.nj_t;
    if (p) {
        for (int c = tid; c < n; c += 256) {
            const int it = c % a.ni_t, jt = c / a.ni_t;
            const size_t q = (size_t)(a.j0 + jt) * a.ni_m + (a.i0 + it);
            s += a.real4 ? (double)reinterpret_cast<const float *>(p)[q] : reinterpret_cast<const double *>(p)[q];
        }
    }
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) a.out[field] = red[0];
}

__global__ void tile_sums_finish(double *out, double ncol)
{
    out[12] = out[2] - out[0];   // *UPTCLN - *UPT : clean-minus-aerosol TOA forcing term
    out[13] = ncol;
    out[14] = 0.;
    out[15] = 0.;
}

int fail(char *err, const std::string &msg)
{
    if (err) {
        std::strncpy(err, msg.c_str(), RRTMG_HIP_ERRLEN - 1);
        err[RRTMG_HIP_ERRLEN - 1] = 0;
    }
    return 1;
}

}  // namespace

// fields12: *upt,*uptc,*uptcln,*dnt,*dntc,*dntcln,*upb,*upbc,*upbcln,*dnb,*dnbc,*dnbcln (SW or LW), device pointers
extern "C" int rrtmg_hip_tile_sums(const void *const *fields12, int real_bytes, int ims, int ime, int jms, int jme,
                                   int its, int ite, int jts, int jte, double *out16_dev, void *stream, char *err)
{
    if (real_bytes != 4 && real_bytes != 8) return fail(err, "rrtmg_hip_tile_sums: real_bytes must be 4 or 8");
    if (!out16_dev) return fail(err, "rrtmg_hip_tile_sums: out16 is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ite < its || jte < jts) {
        // an empty tile (a rank without rows: tiles.slab_bounds with fewer rows than ranks) contributes zeros, as
        // swrad / lwrad / the optical averaging treat it as a no-op -- the rank still takes part in the all-reduce
        hipError_t e = hipMemsetAsync(out16_dev, 0, 16 * sizeof(double), s);
        if (e != hipSuccess) return fail(err, std::string("rrtmg_hip_tile_sums: ") + hipGetErrorString(e));
        return 0;
    }
    if (its < ims || ite > ime || jts < jms || jte > jme)
        return fail(err, "rrtmg_hip_tile_sums: inconsistent tile / memory dimensions");
    SumArgs a{};
    for (int i = 0; i < 12; ++i) a.f[i] = fields12[i];
    a.real4 = real_bytes == 4;
    a.ni_m = ime - ims + 1; a.i0 = its - ims; a.j0 = jts - jms; a.ni_t = ite - its + 1; a.nj_t = jte - jts + 1;
    a.out = out16_dev;
    hipLaunchKernelGGL(tile_sums_kernel, dim3(12), dim3(256), 0, s, a);
    hipLaunchKernelGGL(tile_sums_finish, dim3(1), dim3(1), 0, s, out16_dev, (double)a.ni_t * a.nj_t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(err, std::string("rrtmg_hip_tile_sums: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int rrtmg_hip_sw_tile_sums(const rrtmg_hip_swrad_args *a, double *out16, char *err)
{
    if (a->mem != 1) return fail(err, "rrtmg_hip_sw_tile_sums: needs device-resident outputs (mem = 1)");
    const void *f[12] = {a->swupt, a->swuptc, a->swuptcln, a->swdnt, a->swdntc, a->swdntcln,
                         a->swupb, a->swupbc, a->swupbcln, a->swdnb, a->swdnbc, a->swdnbcln};
    return rrtmg_hip_tile_sums(f, a->real_bytes, a->ims, a->ime, a->jms, a->jme, a->its, a->ite, a->jts, a->jte, out16,
                               a->stream, err);
}

extern "C" int rrtmg_hip_lw_tile_sums(const rrtmg_hip_lwrad_args *a, double *out16, char *err)
{
    if (a->mem != 1) return fail(err, "rrtmg_hip_lw_tile_sums: needs device-resident outputs (mem = 1)");
    const void *f[12] = {a->lwupt, a->lwuptc, a->lwuptcln, a->lwdnt, a->lwdntc, a->lwdntcln,
                         a->lwupb, a->lwupbc, a->lwupbcln, a->lwdnb, a->lwdnbc, a->lwdnbcln};
    return rrtmg_hip_tile_sums(f, a->real_bytes, a->ims, a->ime, a->jms, a->jme, a->its, a->ite, a->jts, a->jte, out16,
                               a->stream, err);
}
