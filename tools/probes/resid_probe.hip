// Residency probe: which workgroups of a launch overlapped in time on one CU?  Every workgroup records the CU it ran on
// (XCC_ID, HW_ID) and its start / end time; the host counts the maximum overlap per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/resid_probe tools/probes/resid_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Rec { unsigned xcc, hw; unsigned long long t0, t1; };
template <int VG>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) spin(Rec *out, int iters)
{
    extern __shared__ double lds[];
    if (VG == 108) asm volatile("" ::: "v107");
    if (VG == 128) asm volatile("" ::: "v127");
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x * 1e-3;
    lds[threadIdx.x] = x;
    __syncthreads();
    for (int i = 0; i < iters; ++i) x = __builtin_fma(x, 0.999999, 1e-9);
    lds[threadIdx.x] += x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = {xcc, hw, t0, (unsigned long long)__builtin_amdgcn_s_memrealtime() + (lds[1] == 123.4 ? 1 : 0)};
    }
}
template <int VG>
void run(int threads, size_t lds)
{
    const int grid = 256 * 8;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * grid);
    hipFuncSetAttribute(reinterpret_cast<const void *>(spin<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    spin<VG><<<grid, threads, lds>>>(d, 20000);
    hipDeviceSynchronize();
    std::vector<Rec> h(grid);
    hipMemcpy(h.data(), d, sizeof(Rec) * grid, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
    for (auto &r : h) {
        const unsigned cu = ((r.xcc & 15) << 16) | (r.hw & 0xff00);   // xcc, se/sh/cu bits of HW_ID
        ev[cu].push_back({r.t0, +1});
        ev[cu].push_back({r.t1, -1});
    }
    int worst = 0, best = 1 << 30;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0, mx = 0;
        for (auto &e : kv.second) { cur += e.second; mx = std::max(mx, cur); }
        worst = std::max(worst, mx); best = std::min(best, mx);
    }
    printf("threads %3d vgpr %3d lds %6zu B: %zu CUs seen, max resident workgroups per CU: %d .. %d\n", threads, VG, lds, ev.size(), best, worst);
    hipFree(d);
}
int main()
{
    for (size_t lds : {(size_t)27 * 1024, (size_t)47904, (size_t)79264, (size_t)81920})
        for (int threads : {192, 320, 512}) {
            run<64>(threads, lds);
            run<108>(threads, lds);
            run<128>(threads, lds);
        }
    return 0;
}
