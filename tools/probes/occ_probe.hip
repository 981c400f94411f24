// Occupancy probe: how many workgroups of a given size and dynamic-LDS footprint does a CU hold at once?
// Every workgroup runs a fixed dependent chain (latency-bound), so the launch time falls as 1 / resident workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/occ_probe tools/probes/occ_probe.hip && /tmp/occ_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(double *out, int iters)
{
    extern __shared__ double lds[];
    double x = threadIdx.x * 1e-3;
    lds[threadIdx.x] = x;
    __syncthreads();
    for (int i = 0; i < iters; ++i) x = __builtin_fma(x, 0.999999, 1e-9);
    if (x == 123.456) out[0] = x + lds[(threadIdx.x + 1) % blockDim.x];
}
int main()
{
    double *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {192, 320, 512}) {
        for (int kb : {8, 16, 24, 27, 31, 32, 33, 40, 47, 48, 49, 52, 56, 64, 65, 72, 77, 78, 80, 81, 96, 128, 160}) {
            const size_t lds = (size_t)kb * 1024;
            hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int grid = 256 * 24;
            spin<<<grid, threads, lds>>>(out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            spin<<<grid, threads, lds>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // one workgroup alone takes t1 = iters * ~8 cycles (dependent fp64 fma) -> resident = grid/256 * t1 / ms
            printf("threads %3d lds %3d KiB: %8.3f ms  (24 rounds / this = %5.2f resident-equivalents x t1)\n", threads, kb, ms, 24.0 / ms);
        }
    }
    return 0;
}
