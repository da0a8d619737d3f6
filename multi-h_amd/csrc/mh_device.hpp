// mh_device.hpp — small __device__ helpers shared by the kernels (header-only,
// no relocatable device code needed).
#pragma once
#include <hip/hip_runtime.h>

namespace mh {

// Forward transfer error of one (point, model) pair; association order of
// M/MultiH.cpp:434-441.  Compiled with -ffp-contract=off.
__device__ __forceinline__ double fwd_d2(double h0, double h1, double h2, double h3, double h4,
                                         double h5, double h6, double h7, double h8, double x,
                                         double y, double x2, double y2)
{
    const double s = h6 * x + h7 * y + h8;
    const double u = (h0 * x + h1 * y + h2) / s;
    const double v = (h3 * x + h4 * y + h5) / s;
    const double dx = x2 - u;
    const double dy = y2 - v;
    return dx * dx + dy * dy;
}

// Two correctly rounded FP64 quotients n1/s, n2/s that share the denominator.
//
// hipcc lowers an IEEE f64 division to
//     v_div_scale x2, v_rcp_f64, 4 fma (two Newton steps), mul, fma, v_div_fmas, v_div_fixup
// and the three v_div_* helpers plus v_rcp_f64 issue at a fraction of the FMA rate, which made
// the residual sweep VALU-bound at half the HBM roofline.  When the denominator and both
// numerators are normal numbers with unbiased exponents in [-255, 256], v_div_scale returns its
// operands unscaled with VCC = 0, v_div_fmas is a plain fma and v_div_fixup passes the quotient
// through, so the SAME rounded operations can be issued directly — and the refined reciprocal,
// which depends on the denominator only, is computed once for both quotients.  The result is
// bit-identical to `n1 / s` and `n2 / s`; lanes outside the exponent window (zeros, denormals,
// infinities, NaNs, huge/tiny magnitudes) take the compiler's full IEEE division.
__device__ __forceinline__ void div2_shared(double n1, double n2, double s, double& u, double& v)
{
    const unsigned int hs = (unsigned int)__double2hiint(s);
    const unsigned int h1 = (unsigned int)__double2hiint(n1);
    const unsigned int h2 = (unsigned int)__double2hiint(n2);
    constexpr unsigned int LO = 768u << 21;          // biased exponent 768, sign shifted out
    // (h << 1) - LO < 2^30  <=>  biased exponent in [768, 1279]
    const unsigned int t = ((hs << 1) - LO) | ((h1 << 1) - LO) | ((h2 << 1) - LO);
    if (__builtin_expect(t < 0x40000000u, 1)) {
        double r = __builtin_amdgcn_rcp(s);
        double e = __builtin_fma(-s, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-s, r, 1.0);
        r = __builtin_fma(r, e, r);
        double q = n1 * r;
        double d = __builtin_fma(-s, q, n1);
        u = __builtin_fma(d, r, q);
        q = n2 * r;
        d = __builtin_fma(-s, q, n2);
        v = __builtin_fma(d, r, q);
    } else {
        // The empty volatile asm keeps hipcc from if-converting this branch into
        // "compute both paths and select", which costs more than the plain division.
        asm volatile("; div2_shared: IEEE path");
        u = n1 / s;
        v = n2 / s;
    }
}

// fwd_d2 with the shared-reciprocal division; bit-identical to fwd_d2.
__device__ __forceinline__ double fwd_d2_fast(double h0, double h1, double h2, double h3,
                                              double h4, double h5, double h6, double h7,
                                              double h8, double x, double y, double x2, double y2)
{
    const double s = h6 * x + h7 * y + h8;
    const double nx = h0 * x + h1 * y + h2;
    const double ny = h3 * x + h4 * y + h5;
    double u, v;
    div2_shared(nx, ny, s, u, v);
    const double dx = x2 - u;
    const double dy = y2 - v;
    return dx * dx + dy * dy;
}

// Cyclic Jacobi eigen-solver for a small symmetric matrix (n <= 4), run by a
// single thread.  Stands where the reference calls cv::eigen on 3x3 / 4x4
// matrices (M/MultiH.cpp:459, :973).  a: n*n row-major (destroyed);
// v: eigenvectors as columns; d: eigenvalues (unsorted).
// Sweep order (p<q lexicographic), the skip of exactly-zero off-diagonals and
// the stop test (off^2 <= 1e-30 * diag^2, <= 30 sweeps) define the result bits.
__device__ inline void jacobi_sym_dev(int n, double* a, double* v, double* d)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag = diag + a[i * n + i] * a[i * n + i];
            for (int j = i + 1; j < n; ++j) off = off + a[i * n + j] * a[i * n + j];
        }
        if (off <= 1e-30 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double sg = (theta >= 0.0) ? 1.0 : -1.0;
                const double t = sg / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) d[i] = a[i * n + i];
}

} // namespace mh
