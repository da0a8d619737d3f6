// compat.hip — the trial statistics of HomographyCompatibilityCheck (M/MultiH.cpp:128-196) on gfx950.
//
// The reference's post-filter runs, per cluster, 501 trials of {3 points -> 3-point homography -> squared forward
// transfer errors of the cluster's other points (:158-173) -> "median" of the sorted buffer (:175-176)}.  What a
// trial's result can depend on is small (host/merge_step.cpp, ClusterMedian): the order statistics of the N - 3 new
// values at ranks k-3 .. k+1 (k = (N-3)/2) and their three largest values; the host threads the three stale entries
// of the reference's buffer through the trials with those.  This file computes exactly these eight numbers per trial:
// one workgroup per (cluster, trial), the distances recomputed in every pass instead of stored —
//   * eight passes of a radix select on the 64-bit patterns of the (non-negative) distances, most significant byte
//     first: a 256-bin LDS histogram of the values that match the prefix found so far, one wavefront scans it;
//   * a ninth pass keeps, per thread, the four smallest values above the selected one and the three largest values,
//     merged pairwise through LDS.
// The distances are the formula of M/MultiH.cpp:162-170 (= fwd_d2, the engine's residual), NaN -> 1e300 as on the
// host, compiled with -ffp-contract=off: the statistics are bit-identical to std::nth_element / partial_sort on the
// host's values.  Integer/byte work apart from 30 FP64 operations per (point, trial, pass); latency bound at these
// sizes (501 x 6 workgroups of a few thousand points), reported as milliseconds per check.

#include "mh_kernels.hpp"
#include "mh_device.hpp"

namespace mh {

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ void insert_small4(u64 (&g)[4], u64 k)          // g ascending; keep the four smallest
{
    if (k >= g[3]) return;
    g[3] = k;
#pragma unroll
    for (int i = 3; i > 0; --i) if (g[i] < g[i - 1]) { const u64 t = g[i]; g[i] = g[i - 1]; g[i - 1] = t; }
}
__device__ __forceinline__ void insert_large3(u64 (&tp)[3], u64 k)         // tp ascending; keep the three largest
{
    if (k <= tp[0]) return;
    tp[0] = k;
#pragma unroll
    for (int i = 0; i < 2; ++i) if (tp[i] > tp[i + 1]) { const u64 t = tp[i]; tp[i] = tp[i + 1]; tp[i + 1] = t; }
}

} // namespace

// ---------------------------------------------------------------------------
// r06: the trials' 3-point homographies on the device too (until r05 the host fitted the 501 x clusters of them — 3.2 of the
// post-filter's 3.6 ms at configs[4]).  GetHomography3PT with do_numerical_refinement = false (M/MultiH.cpp:995-1050, called at
// :154): Hartley normalisation of the three source and the three destination points (Homography_Refine3PTCallback.h:161-196),
// Fn = T2^-T F T1^-1, the epipole of Fn from the eigenvector of Fn Fn^T with the smallest eigenvalue, the third row of the
// normalised H from the 6 x 3 least-squares system through its normal equations and an eigen-decomposition (eigenvalues within
// 2 eps sum |w| of zero dropped), the other two rows from it, H = T2^-1 Hn T1.  Operation for operation the host's
// Homography3PTLinear (host/merge_step.cpp: mat3_mul, jacobi3 = jacobi_sym_dev(3), sums from 0.0 in index order) — the fits are
// bit-identical, which tests/test_gpu_postfilter.py holds them to.  One thread per (cluster, trial).
// ---------------------------------------------------------------------------
namespace {

__device__ __forceinline__ void mat3_mul_dev(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s = s + a[3 * i + k] * b[3 * k + j];
            c[3 * i + j] = s;
        }
}

__device__ __forceinline__ void normalize3_dev(const double* pts, double* out, double* T)
{
    double cx = 0.0, cy = 0.0;
    for (int i = 0; i < 3; ++i) { cx = cx + pts[2 * i]; cy = cy + pts[2 * i + 1]; }
    const double invn = 1.0 / 3.0;
    cx = invn * cx; cy = invn * cy;
    double avg = 0.0;
    for (int i = 0; i < 3; ++i) {
        out[2 * i] = pts[2 * i] - cx;
        out[2 * i + 1] = pts[2 * i + 1] - cy;
        avg = avg + sqrt(out[2 * i] * out[2 * i] + out[2 * i + 1] * out[2 * i + 1]);
    }
    avg = avg / 3.0;
    const double ratio = sqrt(2.0) / avg;
    for (int i = 0; i < 6; ++i) out[i] = out[i] * ratio;
    T[0] = ratio; T[1] = 0; T[2] = -cx * ratio; T[3] = 0; T[4] = ratio; T[5] = -cy * ratio; T[6] = 0; T[7] = 0; T[8] = 1;
}

__device__ __forceinline__ void similarity_inverse_dev(const double* T, double* Ti)
{
    const double ir = 1.0 / T[0];
    Ti[0] = ir; Ti[1] = 0; Ti[2] = -T[2] * ir; Ti[3] = 0; Ti[4] = ir; Ti[5] = -T[5] * ir; Ti[6] = 0; Ti[7] = 0; Ti[8] = 1;
}

__device__ inline bool homography_3pt_linear_dev(const double* pts1, const double* pts2, const double* F, double* H)
{
    double p1[6], p2[6], T1[9], T2[9], T1i[9], T2i[9], T2it[9], tmp[9], Fn[9];
    normalize3_dev(pts1, p1, T1);
    normalize3_dev(pts2, p2, T2);
    similarity_inverse_dev(T1, T1i);
    similarity_inverse_dev(T2, T2i);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2it[3 * i + j] = T2i[3 * j + i];
    mat3_mul_dev(T2it, F, tmp);
    mat3_mul_dev(tmp, T1i, Fn);
    double FFt[9], Fnt[9], v[9], d[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fnt[3 * i + j] = Fn[3 * j + i];
    mat3_mul_dev(Fn, Fnt, FFt);
    jacobi_sym_dev(3, FFt, v, d);
    int jm = 0;
    for (int j = 1; j < 3; ++j) if (d[j] < d[jm]) jm = j;
    const double e0 = v[0 * 3 + jm] / v[2 * 3 + jm];
    const double e1 = v[1 * 3 + jm] / v[2 * 3 + jm];
    double A[18], rhs[6];
    for (int i = 0; i < 3; ++i) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        double* r = &A[6 * i];
        r[0] = e0 * x1 - x2 * x1; r[1] = e0 * y1 - x2 * y1; r[2] = e0 - x2;
        r[3] = e1 * x1 - y2 * x1; r[4] = e1 * y1 - y2 * y1; r[5] = e1 - y2;
        rhs[2 * i] = -(x1 * Fn[3] + y1 * Fn[4] + Fn[5]);
        rhs[2 * i + 1] = (x1 * Fn[0] + y1 * Fn[1] + Fn[2]);
    }
    double AtA[9], Atb[3];
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int i = 0; i < 6; ++i) s = s + A[3 * i + a] * A[3 * i + c];
            AtA[3 * a + c] = s;
        }
        double s = 0.0;
        for (int i = 0; i < 6; ++i) s = s + A[3 * i + a] * rhs[i];
        Atb[a] = s;
    }
    // sym_eig_solve3: x = pinv(AtA) Atb
    double w[3], h3[3] = { 0.0, 0.0, 0.0 };
    jacobi_sym_dev(3, AtA, v, w);
    double cut = 0.0;
    for (int k = 0; k < 3; ++k) cut = cut + fabs(w[k]);
    cut = cut * (2.0 * 2.220446049250313e-16);
    for (int k = 0; k < 3; ++k) {
        if (fabs(w[k]) <= cut) continue;
        double proj = 0.0;
        for (int i = 0; i < 3; ++i) proj = proj + v[3 * i + k] * Atb[i];
        proj = proj / w[k];
        for (int i = 0; i < 3; ++i) h3[i] = h3[i] + proj * v[3 * i + k];
    }
    double Hn[9];
    Hn[6] = h3[0]; Hn[7] = h3[1]; Hn[8] = h3[2];
    Hn[3] = e1 * h3[0] - Fn[0]; Hn[4] = e1 * h3[1] - Fn[1]; Hn[5] = e1 * h3[2] - Fn[2];
    Hn[0] = e0 * h3[0] + Fn[3]; Hn[1] = e0 * h3[1] + Fn[4]; Hn[2] = e0 * h3[2] + Fn[5];
    mat3_mul_dev(T2i, Hn, tmp);
    mat3_mul_dev(tmp, T1, H);
    bool ok = true;
    for (int i = 0; i < 9; ++i) if (!(fabs(H[i]) <= 1.7976931348623157e308)) ok = false;      // std::isfinite
    return ok;
}

} // namespace

struct Fund9 { double f[9]; };

__global__ void __launch_bounds__(128)
k_compat_fit(const double* __restrict__ pts, const int* __restrict__ begin, const int* __restrict__ tri, Fund9 F, int trials, int total,
             double* __restrict__ H, unsigned char* __restrict__ ok)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= total) return;
    const int b0 = begin[i / trials];
    double ms[6], md[6];
    for (int j = 0; j < 3; ++j) {
        const double* q = pts + 4 * (size_t)(b0 + tri[3 * (size_t)i + j]);
        ms[2 * j] = q[0]; ms[2 * j + 1] = q[1]; md[2 * j] = q[2]; md[2 * j + 1] = q[3];
    }
    double Hc[9];
    const bool good = homography_3pt_linear_dev(ms, md, F.f, Hc);
    for (int k = 0; k < 9; ++k) H[9 * (size_t)i + k] = good ? Hc[k] : 0.0;
    ok[i] = good ? 1 : 0;
}

hipError_t launch_compat_fit(const double* pts, const int* begin, int clusters, const int* tri, const double F[9], int trials,
                             double* H, unsigned char* ok, hipStream_t s)
{
    const int total = clusters * trials;
    if (total <= 0) return hipSuccess;
    Fund9 f;
    for (int i = 0; i < 9; ++i) f.f[i] = F[i];
    hipLaunchKernelGGL(k_compat_fit, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, pts, begin, tri, f, trials, total, H, ok);
    return hipGetLastError();
}

// pts: the clusters' points back to back, 4 doubles each (x1 y1 x2 y2); begin[c] .. begin[c+1]: cluster c;
// tri: per (cluster, trial) the positions inside the cluster of the three points the trial drew; H / ok: the trial's
// homography (row-major 9) and whether the 3-point fit succeeded; out: 8 doubles per (cluster, trial) —
// ranks k-3 .. k+1, then the three largest ascending.  Every cluster has at least 19 points.
__global__ void __launch_bounds__(256)
k_compat_select(const double* __restrict__ pts, const int* __restrict__ begin, const int* __restrict__ tri,
                const double* __restrict__ H, const unsigned char* __restrict__ ok, int trials, double* __restrict__ out)
{
    const int c = (int)blockIdx.x / trials;
    const int tid = (int)threadIdx.x;
    const int b0 = begin[c], nc = begin[c + 1] - b0;
    const int rest = nc - 3, lo = rest / 2 - 3;
    double* o = out + 8 * (size_t)blockIdx.x;
    if (!ok[blockIdx.x]) {                             // no homography: every distance is NaN -> 1e300 (:171-172 as the host keeps it)
        if (tid < 8) o[tid] = 1e300;
        return;
    }
    const double* h = H + 9 * (size_t)blockIdx.x;
    const double h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7], h8 = h[8];
    const int t0 = tri[3 * (size_t)blockIdx.x], t1 = tri[3 * (size_t)blockIdx.x + 1], t2 = tri[3 * (size_t)blockIdx.x + 2];
    const double* p = pts + 4 * (size_t)b0;
    auto key_at = [&](int i) -> u64 {
        const double2 a = *reinterpret_cast<const double2*>(p + 4 * (size_t)i);
        const double2 b = *reinterpret_cast<const double2*>(p + 4 * (size_t)i + 2);
        double d2 = fwd_d2(h0, h1, h2, h3, h4, h5, h6, h7, h8, a.x, a.y, b.x, b.y);
        if (d2 != d2) d2 = 1e300;
        return (u64)__double_as_longlong(d2);
    };

    __shared__ unsigned hist[256];
    __shared__ int s_bin, s_want;
    __shared__ unsigned s_eq;
    u64 prefix = 0, mask = 0;
    int want = lo;                                     // rank still to be found among the values that match the prefix
    for (int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < nc; i += 256) {
            if (i == t0 || i == t1 || i == t2) continue;
            const u64 k = key_at(i);
            if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                // one wavefront: four bins per lane, prefix over the lanes
            const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const unsigned s = c0 + c1 + c2 + c3;
            unsigned incl = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const unsigned y = __shfl_up(incl, d, 64); if (tid >= d) incl += y; }
            const unsigned excl = incl - s, w = (unsigned)want;
            if (excl <= w && w < incl) {               // exactly one lane: more values match the prefix than `want`
                unsigned r = w - excl;
                int b = 4 * tid;
                unsigned eq = c0;
                if (r >= c0) { r -= c0; ++b; eq = c1; if (r >= c1) { r -= c1; ++b; eq = c2; if (r >= c2) { r -= c2; ++b; eq = c3; } } }
                s_bin = b; s_want = (int)r; s_eq = eq;
            }
        }
        __syncthreads();
        want = s_want;
        prefix |= (u64)(unsigned)s_bin << shift;
        mask |= 0xffull << shift;
    }
    const u64 key_lo = prefix;                         // the value at rank `lo`; `want` = its position inside the run of equal values
    const int cnt_eq = (int)s_eq;

    u64 g[4] = { ~0ull, ~0ull, ~0ull, ~0ull }, tp[3] = { 0ull, 0ull, 0ull };
    for (int i = tid; i < nc; i += 256) {
        if (i == t0 || i == t1 || i == t2) continue;
        const u64 k = key_at(i);
        if (k > key_lo) insert_small4(g, k);
        insert_large3(tp, k);
    }
    __shared__ u64 s_g[256 * 4], s_t[256 * 3];
#pragma unroll
    for (int j = 0; j < 4; ++j) s_g[4 * tid + j] = g[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) s_t[3 * tid + j] = tp[j];
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (tid < stride) {
#pragma unroll
            for (int j = 0; j < 4; ++j) insert_small4(g, s_g[4 * (tid + stride) + j]);
#pragma unroll
            for (int j = 0; j < 3; ++j) insert_large3(tp, s_t[3 * (tid + stride) + j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) s_g[4 * tid + j] = g[j];
#pragma unroll
            for (int j = 0; j < 3; ++j) s_t[3 * tid + j] = tp[j];
        }
        __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int pos = want + j;                  // position counted from the start of the run of values equal to key_lo
            const u64 k = pos < cnt_eq ? key_lo : g[pos - cnt_eq];
            o[j] = __longlong_as_double((long long)k);
        }
        o[5] = __longlong_as_double((long long)tp[0]);
        o[6] = __longlong_as_double((long long)tp[1]);
        o[7] = __longlong_as_double((long long)tp[2]);
    }
}

hipError_t launch_compat_select(const double* pts, const int* begin, int clusters, const int* tri, const double* H,
                                const unsigned char* ok, int trials, double* out, hipStream_t s)
{
    if (clusters <= 0 || trials <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_compat_select, dim3((unsigned)clusters * (unsigned)trials), dim3(256), 0, s, pts, begin, tri, H, ok,
                       trials, out);
    return hipGetLastError();
}

} // namespace mh
