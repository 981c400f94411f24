// Issue cost of fp64 vector instructions on one SIMD (development probe): a wave runs N independent instructions of one
// kind back to back (8 accumulators), timed with s_memtime; printed in cycles per instruction for 1 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define KERNEL(name, stmt)                                                                  \
    __global__ void name(double *out, long long *cyc, double seed)                          \
    {                                                                                       \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
               a7 = a0 + 7;                                                                  \
        const double b = seed * 0.5 + 1.0, c = seed * 0.25 + 0.5;                           \
        long long t0 = __builtin_readcyclecounter();                                        \
        for (int i = 0; i < REP; ++i) {                                                     \
            stmt(a0) stmt(a1) stmt(a2) stmt(a3) stmt(a4) stmt(a5) stmt(a6) stmt(a7)         \
        }                                                                                   \
        long long t1 = __builtin_readcyclecounter();                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                    \
    }
#define S_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
#define S_RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define S_RSQ(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
#define S_SQRT(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
#define S_LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x));
#define S_RNDNE(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define S_CNDMASK(x) { int lo = __double2loint(x); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(7)); x = __hiloint2double(__double2hiint(x), lo); }
KERNEL(k_fma, S_FMA) KERNEL(k_mul, S_MUL) KERNEL(k_add, S_ADD) KERNEL(k_rcp, S_RCP) KERNEL(k_rsq, S_RSQ)
KERNEL(k_sqrt, S_SQRT) KERNEL(k_ldexp, S_LDEXP) KERNEL(k_rndne, S_RNDNE)
template <class K> void run(const char *name, K k)
{
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&cyc, 1024 * 8);
    for (int wg_threads : {64, 192}) {   // one wave per workgroup; 1 or 3 waves per SIMD (256 CUs x 4 SIMDs x n)
        const int nwg = 256 * 4 * (wg_threads == 64 ? 1 : 3) / (wg_threads / 64);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(wg_threads), 0, 0, out, cyc, 1.5);
        hipDeviceSynchronize();
        long long h[16]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("%-8s waves/SIMD ~%d: %6.2f cycles per instruction (wave 0 of workgroup 0..3: %lld %lld %lld %lld)\n", name,
               wg_threads == 64 ? 1 : 3, (double)h[0] / (REP * 8), h[0], h[1], h[2], h[3]);
    }
    hipFree(out); hipFree(cyc);
}
int main()
{
    run("fma", k_fma); run("mul", k_mul); run("add", k_add); run("rcp", k_rcp); run("rsq", k_rsq); run("sqrt", k_sqrt);
    run("ldexp", k_ldexp); run("rndne", k_rndne);
    return 0;
}
