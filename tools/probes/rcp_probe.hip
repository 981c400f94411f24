#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *a, const double *b, double *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double bb = b[i], aa = a[i];
    double r0 = __builtin_amdgcn_rcp(bb);
    double e = __builtin_fma(-bb, r0, 1.0);
    double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-bb, r1, 1.0);
    double r2 = __builtin_fma(r1, e, r1);
    // fdiv with one NR + residual
    double q = aa * r1;
    double rem = __builtin_fma(-bb, q, aa);
    double q1 = __builtin_fma(rem, r1, q);
    // fdiv with two NR + residual
    double qq = aa * r2;
    rem = __builtin_fma(-bb, qq, aa);
    double q2 = __builtin_fma(rem, r2, qq);
    o[i * 5 + 0] = r0; o[i * 5 + 1] = r1; o[i * 5 + 2] = r2; o[i * 5 + 3] = q1; o[i * 5 + 4] = q2;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> a(n), b(n), o((size_t)n * 5);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30, 30), m(1.0, 2.0);
    for (int i = 0; i < n; ++i) { a[i] = m(g) * std::exp2(std::floor(u(g))); b[i] = m(g) * std::exp2(std::floor(u(g))); }
    double *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, (size_t)n * 40);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dout, n);
    hipMemcpy(o.data(), dout, (size_t)n * 40, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0; long d1 = 0, d2 = 0, x1 = 0, x2 = 0;
    for (int i = 0; i < n; ++i) {
        long double rr = 1.0L / (long double)b[i];
        e0 = fmax(e0, (double)fabsl(((long double)o[i * 5] - rr) / rr));
        e1 = fmax(e1, (double)fabsl(((long double)o[i * 5 + 1] - rr) / rr));
        e2 = fmax(e2, (double)fabsl(((long double)o[i * 5 + 2] - rr) / rr));
        double qe = a[i] / b[i];
        if (o[i * 5 + 3] != qe) { ++d1; if (fabs(o[i * 5 + 3] - qe) > 1.5 * fabs(qe) * 2.3e-16) ++x1; }
        if (o[i * 5 + 4] != qe) { ++d2; if (fabs(o[i * 5 + 4] - qe) > 1.5 * fabs(qe) * 2.3e-16) ++x2; }
    }
    printf("rcp rel err: raw %.3e (2^%.1f)  1NR %.3e  2NR %.3e\n", e0, log2(e0), e1, e2);
    printf("fdiv 1NR+res: %ld of %d differ from IEEE (%ld by more than 1 ulp); 2NR+res: %ld (%ld)\n", d1, n, x1, d2, x2);
    return 0;
}
