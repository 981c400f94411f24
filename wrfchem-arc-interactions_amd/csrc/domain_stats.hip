// Domain statistics of a 2-D output field on the device: what the reference's post-processing
// computes per field and domain on the host from the wrfout files --
//   analysis_scripts/NCL_extraction_package/misc_stats_library.ncl:405-458 (calc_standard_stats):
//   avg, stddev (sample, N-1), min, max, median, 25th, 75th, 5th, 95th percentile, standard error, N
//   with percentiles taken as sorted(round(p/100*(N-1)))  (calc_boxplot_stats, :146-190).
//   Moran's I (calc_morans_i_2D, :196-370, with the options calc_standard_stats sets: neighbour, manhattan) and the
//   autocorrelation-corrected standard error SE * I (:450-451).
//
// Pipeline: gather the (masked) tile into a dense fp64 buffer, radix-sort it (rocPRIM through hipCUB), then one
// workgroup forms mean and variance in a fixed order and picks the order statistics; Moran's I from per-workgroup
// partial sums added in workgroup order.  Deterministic.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstring>
#include <string>

#include "../../include/rrtmg_hip.h"

namespace {

struct GatherArgs {
    const void *f;
    const unsigned char *mask;   // optional, 2-D (memory shape), nonzero = use
    int real4, ni_m, i0, j0, ni_t, nj_t;
    double *buf;
    unsigned int *count;
};

__global__ void __launch_bounds__(256) gather_kernel(const GatherArgs a)
{
    const int n = a.ni_t * a.nj_t;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const int it = c % a.ni_t, jt = c / a.ni_t;
    const size_t q = (size_t)(a.j0 + jt) * a.ni_m + (a.i0 + it);
    if (a.mask && !a.mask[q]) return;
    const double v = a.real4 ? (double)reinterpret_cast<const float *>(a.f)[q] : reinterpret_cast<const double *>(a.f)[q];
    a.buf[atomicAdd(a.count, 1u)] = v;   // order is irrelevant: the buffer is sorted next
}

__device__ double block_sum(double s, double *red, int tid)
{
    red[tid] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// NCL round(x, 3): nearest integer, halves away from zero
__device__ long ncl_round(double x) { return (long)(x < 0. ? x - 0.5 : x + 0.5); }

// Moran's I as the reference forms it: the field is flattened (i fastest), every cell n is paired with cell
// (n + m) mod N for the eight displacements m = 1, ni-1, ni, ni+1, N-ni-1, N-ni, N-ni+1, N-1, and a pair counts
// with weight 1 when its Manhattan distance in (i,j) is exactly 1 (:272-277, 318-331).  Masked-out cells are
// "missing": their products drop out of the sum, their weights do not (NCL sum() skips missing values).
struct MoranArgs {
    const void *f;
    const unsigned char *mask;
    int real4, ni_m, i0, j0, ni_t, nj_t;
    const double *stats;   // [0] mean
    double *part;          // [2][nblocks]: autocovariance sums, weight sums
    int nblocks;
};
__device__ __forceinline__ bool moran_value(const MoranArgs &a, int c, double &v)
{
    const int it = c % a.ni_t, jt = c / a.ni_t;
    const size_t q = (size_t)(a.j0 + jt) * a.ni_m + (a.i0 + it);
    if (a.mask && !a.mask[q]) return false;
    v = a.real4 ? (double)reinterpret_cast<const float *>(a.f)[q] : reinterpret_cast<const double *>(a.f)[q];
    return true;
}
__global__ void __launch_bounds__(256) moran_partial_kernel(const MoranArgs a)
{
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const long N = (long)a.ni_t * a.nj_t;
    const long c = (long)blockIdx.x * 256 + tid;
    double au = 0., w = 0.;
    if (c < N) {
        const double mean = a.stats[0];
        double x0 = 0.;
        const bool have0 = moran_value(a, (int)c, x0);
        const int i0 = (int)(c % a.ni_t), j0 = (int)(c / a.ni_t);
        const long ni = a.ni_t;
        const long disp[8] = {1, ni - 1, ni, ni + 1, N - ni - 1, N - ni, N - ni + 1, N - 1};
        for (int d = 0; d < 8; ++d) {
            const long m = disp[d];
            if (m == 0) continue;                       // :300-302
            const long c1 = ((c + m) % N + N) % N;
            const int i1 = (int)(c1 % a.ni_t), j1 = (int)(c1 / a.ni_t);
            const int dist = (i0 > i1 ? i0 - i1 : i1 - i0) + (j0 > j1 ? j0 - j1 : j1 - j0);
            if (dist == 1) {                            // 0 < dist < 2
                w += 1.;
                double x1 = 0.;
                if (have0 && moran_value(a, (int)c1, x1)) au += (x0 - mean) * (x1 - mean);
            }
        }
    }
    red[0][tid] = au; red[1][tid] = w;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { a.part[blockIdx.x] = red[0][0]; a.part[a.nblocks + blockIdx.x] = red[1][0]; }
}
__global__ void moran_finish_kernel(const double *part, int nblocks, double *out13)
{
    double au = 0., w = 0.;
    for (int b = 0; b < nblocks; ++b) { au += part[b]; w += part[nblocks + b]; }
    const double sd = out13[1];
    const double I = (sd > 0. && w > 0.) ? au / (w * sd * sd) : 0.;   // all differences zero -> 0 (:243-246)
    out13[10] = I;
    out13[11] = out13[9] * I;
}

__global__ void __launch_bounds__(1024) stats_kernel(const double *sorted, const unsigned int *count, double *out13)
{
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    const long n = (long)*count;
    if (n == 0) {
        if (tid < 13) out13[tid] = 0.0;   // "if no data then let's get out of here": stats = 0
        return;
    }
    double s = 0.;
    for (long i = tid; i < n; i += 1024) s += sorted[i];
    const double mean = block_sum(s, red, tid) / (double)n;
    double v = 0.;
    for (long i = tid; i < n; i += 1024) { const double d = sorted[i] - mean; v += d * d; }
    const double var = n > 1 ? block_sum(v, red, tid) / (double)(n - 1) : 0.;
    if (tid == 0) {
        const double sd = sqrt(var);
        const double perc[5] = {5., 25., 50., 75., 95.};
        double st[5];
        for (int j = 0; j < 5; ++j) {
            long pt = ncl_round(.01 * perc[j] * (double)(n - 1));
            pt = pt < 0 ? 0 : (pt > n - 1 ? n - 1 : pt);
            st[j] = sorted[pt];
        }
        out13[0] = mean; out13[1] = sd; out13[2] = sorted[0]; out13[3] = sorted[n - 1];
        out13[4] = st[2]; out13[5] = st[1]; out13[6] = st[3]; out13[7] = st[0]; out13[8] = st[4];
        out13[9] = sd / sqrt((double)n);
        out13[10] = -9999.; out13[11] = -9999.;
        out13[12] = (double)n;
    }
}

int fail(char *err, const std::string &msg)
{
    if (err) {
        std::strncpy(err, msg.c_str(), RRTMG_HIP_ERRLEN - 1);
        err[RRTMG_HIP_ERRLEN - 1] = 0;
    }
    return 1;
}

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            cleanup();                                                                                   \
            return fail(err, std::string("rrtmg_hip_domain_stats: ") + #expr + ": " + hipGetErrorString(e_)); \
        }                                                                                                \
    } while (0)

}  // namespace

extern "C" int rrtmg_hip_domain_stats(const void *field2d, const unsigned char *mask2d, int real_bytes, int ims, int ime,
                                      int jms, int jme, int its, int ite, int jts, int jte, double *out13_dev,
                                      void *stream, char *err)
{
    void *buf = nullptr, *srt = nullptr, *cnt = nullptr, *tmp = nullptr, *part = nullptr;
    auto cleanup = [&]() {
        if (buf) (void)hipFree(buf);
        if (srt) (void)hipFree(srt);
        if (cnt) (void)hipFree(cnt);
        if (tmp) (void)hipFree(tmp);
        if (part) (void)hipFree(part);
    };
    if (real_bytes != 4 && real_bytes != 8) return fail(err, "rrtmg_hip_domain_stats: real_bytes must be 4 or 8");
    if (!field2d || !out13_dev) return fail(err, "rrtmg_hip_domain_stats: field / output is NULL");
    if (its < ims || ite > ime || jts < jms || jte > jme || ite < its || jte < jts)
        return fail(err, "rrtmg_hip_domain_stats: inconsistent tile / memory dimensions");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ni_t = ite - its + 1, nj_t = jte - jts + 1, n = ni_t * nj_t;
    HIPCHK(hipMalloc(&buf, sizeof(double) * n));
    HIPCHK(hipMalloc(&srt, sizeof(double) * n));
    HIPCHK(hipMalloc(&cnt, sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned int), s));
    GatherArgs g{field2d, mask2d, real_bytes == 4, ime - ims + 1, its - ims, jts - jms, ni_t, nj_t,
                 static_cast<double *>(buf), static_cast<unsigned int *>(cnt)};
    hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, g);
    unsigned int hcount = 0;
    HIPCHK(hipMemcpyAsync(&hcount, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (hcount > 0) {
        size_t tbytes = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tbytes, static_cast<const double *>(buf),
                                                 static_cast<double *>(srt), (int)hcount, 0, 64, s));
        HIPCHK(hipMalloc(&tmp, tbytes ? tbytes : 8));
        HIPCHK(hipcub::DeviceRadixSort::SortKeys(tmp, tbytes, static_cast<const double *>(buf),
                                                 static_cast<double *>(srt), (int)hcount, 0, 64, s));
    }
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(1024), 0, s, static_cast<const double *>(srt),
                       static_cast<const unsigned int *>(cnt), out13_dev);
    HIPCHK(hipGetLastError());
    if (hcount > 0) {
        const int nblocks = (n + 255) / 256;
        HIPCHK(hipMalloc(&part, sizeof(double) * 2 * nblocks));
        MoranArgs m{field2d, mask2d, real_bytes == 4, ime - ims + 1, its - ims, jts - jms, ni_t, nj_t, out13_dev,
                    static_cast<double *>(part), nblocks};
        hipLaunchKernelGGL(moran_partial_kernel, dim3(nblocks), dim3(256), 0, s, m);
        hipLaunchKernelGGL(moran_finish_kernel, dim3(1), dim3(1), 0, s, static_cast<const double *>(part), nblocks,
                           out13_dev);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));   // temporaries are freed below
    cleanup();
    return 0;
}
