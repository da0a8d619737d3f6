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

// ---------------------------------------------------------------------------
// Shared-reciprocal division for the residual sweep.
//
// hipcc lowers an IEEE f64 division n/s to
//     v_div_scale(s), v_div_scale(n), v_rcp_f64, 4 fma (two Newton steps on the reciprocal),
//     v_mul, v_fma (remainder), v_div_fmas, v_div_fixup
// The sweep needs two quotients with the SAME denominator per (point, model) pair and is
// FP64-VALU bound, so the reciprocal refinement is done once and the v_div_* helpers are
// dropped where they are provably no-ops:
//   * If s is a normal number with unbiased exponent in [-255, 256] and the numerator n is a
//     normal number with exponent in [-766, 767], v_div_scale returns both operands unscaled
//     with VCC = 0 (none of its triggers fires: exponent difference < 768, no denormal operand,
//     1/s and n/s normal, exponent(n) > 53), v_div_fmas is then a plain fma and v_div_fixup
//     passes the quotient through.  The sequence below issues exactly the remaining rounded
//     operations, so its result is bit-identical to `n / s`.
//   * n == +-0 gives +-0 on both paths (the sign of a zero quotient cannot reach d2).
// The preconditions are split so that the per-pair cost is ONE compare on s:
//   * |s| < 2^257 and |n| < 2^768, both finite     -> guaranteed by |h_i| < 2^120 for the MODEL
//                                                     (`model_pre`, once per workgroup) and
//                                                     |x|,|y| < 2^120 for the POINT (`point_pre`,
//                                                     once per tile): |s|, |n| < 3 * 2^240;
//   * |s| >= 2^-255 (and s not NaN)                -> the per-pair v_cmp_f64 with the |.| modifier;
//   * 0 < |n| < 2^-766 (incl. denormals)           -> |u| < 2^-510 on BOTH paths; x2 - u rounds to
//                                                     x2 on both unless |x2| < 2^-450, a per-POINT
//                                                     property folded into `point_pre` (which also
//                                                     rejects inf/NaN targets).
// Any lane failing a check recomputes the pair with the compiler's full IEEE division.
// ---------------------------------------------------------------------------

__device__ __forceinline__ unsigned int abs_hi(double v) { return (unsigned int)__double2hiint(v) & 0x7fffffffu; }

// |v| < 2^120 and not NaN/inf
__device__ __forceinline__ bool mag_bounded(double v) { return abs_hi(v) < ((1023u + 120u) << 20); }

// 2^-450 <= |v| < 2^120
__device__ __forceinline__ bool mag_mid(double v) { return (abs_hi(v) - ((1023u - 450u) << 20)) < (570u << 20); }

// per-point precondition: source (x, y) bounded, target (x2, y2) neither tiny nor huge
__device__ __forceinline__ bool point_pre(double x, double y, double x2, double y2)
{
    return mag_bounded(x) && mag_bounded(y) && mag_mid(x2) && mag_mid(y2);
}

// per-model precondition: all nine coefficients bounded
__device__ __forceinline__ bool model_pre(const double* h)
{
    bool ok = true;
    for (int i = 0; i < 9; ++i) ok = ok && mag_bounded(h[i]);
    return ok;
}

// Per-model bound on s over the bounding box of ALL source points: every operation of s = (h6*x + h7*y) + h8 is
// monotone in its operands and rounding is monotone, so the s the sweep computes for any point of the box lies in
// [lo, hi] as computed here with the same operations at the box's corners.  True when that interval keeps clear of
// (-2^-255, 2^-255): the model's horizon does not come near the data.  NaN / infinite bounds compare false.
__device__ __forceinline__ bool model_far(const double* h, double xmin, double xmax, double ymin, double ymax)
{
    const double a = h[6] * xmin, b = h[6] * xmax, c = h[7] * ymin, d = h[7] * ymax;
    const double lo = (fmin(a, b) + fmin(c, d)) + h[8];
    const double hi = (fmax(a, b) + fmax(c, d)) + h[8];
    const bool finite = (a == a) && (b == b) && (c == c) && (d == d);      // fmin/fmax would hide a NaN product
    return finite && (lo >= 0x1p-255 || hi <= -0x1p-255);
}

// SCHK = false: the caller has proved |s| >= 2^-255 for every point this model can meet (model_far below), so the
// per-pair compare disappears and the fallback is taken on the per-point / per-model precondition alone.
template <bool SCHK = true>
__device__ __forceinline__ double fwd_d2_fast(double h0, double h1, double h2, double h3,
                                              double h4, double h5, double h6, double h7,
                                              double h8, double x, double y, double x2, double y2,
                                              bool pre_ok)
{
    const double s = h6 * x + h7 * y + h8;
    const double nx = h0 * x + h1 * y + h2;
    const double ny = h3 * x + h4 * y + h5;
    double r = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = nx * r;
    double d = __builtin_fma(-s, q, nx);
    const double u = __builtin_fma(d, r, q);
    q = ny * r;
    d = __builtin_fma(-s, q, ny);
    const double v = __builtin_fma(d, r, q);
    const double dx = x2 - u;
    const double dy = y2 - v;
    double d2 = dx * dx + dy * dy;
    // bitwise, not short-circuit: two lane masks and-ed on the scalar unit, one branch
    const int ok = SCHK ? ((int)pre_ok & (int)(__builtin_fabs(s) >= 0x1p-255)) : (int)pre_ok;
    if (__builtin_expect(!ok, 0)) {
        // The empty volatile asm keeps hipcc from if-converting this branch into
        // "compute both and select", which would put the IEEE sequence back on the hot path.
        asm volatile("; fwd_d2_fast: IEEE path");
        const double ui = nx / s;
        const double vi = ny / s;
        const double dxi = x2 - ui;
        const double dyi = y2 - vi;
        d2 = dxi * dxi + dyi * dyi;
    }
    return d2;
}

// The fast path of fwd_d2_fast alone, for callers that have proved ALL its preconditions beforehand for every lane
// (model_pre and model_far for the model, point_pre for every point of the wave's tile): the same rounded operations in
// the same order, so the same bits as `/`, and nothing else.
// s_out: the denominator, for the caller's |s| >= 2^-255 test where the model is not provably `far`.
__device__ __forceinline__ double fwd_d2_lean(double h0, double h1, double h2, double h3, double h4, double h5,
                                              double h6, double h7, double h8, double x, double y, double x2, double y2,
                                              double& s_out)
{
    const double s = h6 * x + h7 * y + h8;
    s_out = s;
    const double nx = h0 * x + h1 * y + h2;
    const double ny = h3 * x + h4 * y + h5;
    double r = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = nx * r;
    double d = __builtin_fma(-s, q, nx);
    const double u = __builtin_fma(d, r, q);
    q = ny * r;
    d = __builtin_fma(-s, q, ny);
    const double v = __builtin_fma(d, r, q);
    const double dx = x2 - u;
    const double dy = y2 - v;
    return dx * dx + dy * dy;
}

// NOT the product arithmetic: the same residual with fused multiply-adds (20 FP64 operations per
// pair instead of 28).  It rounds differently from the reference (last-bit differences in d2, and
// therefore possibly different inlier decisions within an ulp of the threshold), so it exists only
// as tuning variant 10 of the residual kernel, to measure what the exact-rounding requirement costs
// (HISTORY.md section 7).
__device__ __forceinline__ double fwd_d2_contracted(double h0, double h1, double h2, double h3, double h4,
                                                    double h5, double h6, double h7, double h8, double x,
                                                    double y, double x2, double y2)
{
    const double s = __builtin_fma(h6, x, __builtin_fma(h7, y, h8));
    const double nx = __builtin_fma(h0, x, __builtin_fma(h1, y, h2));
    const double ny = __builtin_fma(h3, x, __builtin_fma(h4, y, h5));
    double r = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = nx * r;
    double d = __builtin_fma(-s, q, nx);
    const double u = __builtin_fma(d, r, q);
    q = ny * r;
    d = __builtin_fma(-s, q, ny);
    const double v = __builtin_fma(d, r, q);
    const double dx = x2 - u;
    const double dy = y2 - v;
    return __builtin_fma(dx, dx, dy * dy);
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
