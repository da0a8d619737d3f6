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
