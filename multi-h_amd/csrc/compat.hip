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
