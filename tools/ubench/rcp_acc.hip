// rcp_acc.hip — empirical accuracy of v_rcp_f64 and of one / two Newton steps (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = x[i];
    double r = __builtin_amdgcn_rcp(s);
    r0[i] = r;
    double e = __builtin_fma(-s, r, 1.0); r = __builtin_fma(r, e, r);
    r1[i] = r;
    e = __builtin_fma(-s, r, 1.0); r = __builtin_fma(r, e, r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 24;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(42);
    for (int i = 0; i < n; ++i) { uint64_t m = g() & ((1ull << 52) - 1); uint64_t bits = (1023ull << 52) | m; memcpy(&x[i], &bits, 8); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0; long wrong1 = 0, wrong2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        double exact = (double)t;                      // correctly rounded (x87 80-bit then round: double rounding possible but rare)
        double ulp = std::ldexp(1.0, std::ilogb(exact) - 52);
        m0 = std::fmax(m0, std::fabs((double)((long double)a[i] - t)) / ulp);
        m1 = std::fmax(m1, std::fabs((double)((long double)b[i] - t)) / ulp);
        m2 = std::fmax(m2, std::fabs((double)((long double)c[i] - t)) / ulp);
        wrong1 += b[i] != exact; wrong2 += c[i] != exact;
    }
    printf("max error in ulp: rcp %.3g   after 1 NR %.3g (%ld of %d not correctly rounded)   after 2 NR %.3g (%ld)\n", m0, m1, wrong1, n, m2, wrong2);
    return 0;
}
