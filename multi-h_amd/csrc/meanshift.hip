// meanshift.hip — flat-kernel mean-shift climb on the GPU.
//
// MeanShiftClustering<T>::Cluster (M/moduls/mode_seeking/MeanShiftClustering.h:23-157) runs, for
// every seed, the loop :62-123: membership = { i : sum_j sqrt(d_ij^2) < bandwidth^2 } (an L1 ball,
// quirk A-8), new mean = (sum of members) * (1/count), every member gets a vote and is marked
// visited, stop when ||mean - old|| < 1e-3*bandwidth.  The reference allocates an N x D repmat per
// iteration; at N = 50k (EstablishStablePointSets, M/MultiH.cpp:604-694) that inner loop is the
// cost.  Here an iteration is two launches that never return to the host:
//   k_ms_partial  MS_GROUPS workgroups sweep the rows (global thread g takes rows g, g+T, g+2T, ...),
//                 members vote, each workgroup reduces its member sums with a binary tree;
//   k_ms_update   one wave adds the MS_GROUPS partials in order, forms the new mean, tests
//                 convergence and raises the `done` word that turns the rest of the batch into no-ops.
// The summation order (strided, tree inside a group, groups in sequence) is fixed — it does not
// depend on the device — and is mirrored by the oracle (mho_mean_shift), so modes and assignments
// are bit-identical.  The seed order, vote merging and final assignment stay on the host (they are
// sequential by definition, :52-56,:100-146).
// MS_BATCH climbs run side by side (blockIdx.y = climb): their seeds are drawn together from the rows that are
// unvisited when the batch starts, and the host applies the finished climbs in draw order, dropping a climb whose
// seed an earlier climb of the same batch has visited meanwhile (the reference never starts from a visited row).
// A climb depends only on the data and its seed, so the batch costs the host round trips of its longest climb
// instead of the sum.  The batch size is part of the definition (the oracle draws the same way).
#include "mh_kernels.hpp"

namespace mh {

constexpr int MS_MAXD = 16;
constexpr int MS_GROUPS = 64;            // workgroups per sweep; part of the numerical definition

// the slice of climb b
__device__ __forceinline__ MeanShiftWork ms_climb(const MeanShiftWork& a, int b)
{
    MeanShiftWork w = a;
    w.mean = a.mean + (size_t)b * MS_MAXD;
    w.votes = a.votes + (size_t)b * a.n;
    w.out = a.out + (size_t)b * 4;
    w.list = a.list + (size_t)b * 2 * a.n;
    w.partial = a.partial + (size_t)b * MS_GROUPS * MS_MAXD;
    w.partial_cnt = a.partial_cnt + (size_t)b * MS_GROUPS;
    return w;
}

__global__ void __launch_bounds__(256)
k_ms_partial(MeanShiftWork all, double band_sq)
{
    const MeanShiftWork w = ms_climb(all, blockIdx.y);
    if (w.out[1] || w.out[3]) return;                       // converged or dead end: rest of the batch idles
    const int t = threadIdx.x;
    const int D = w.d;
    const int T = MS_GROUPS * 256;
    __shared__ double sv[256][MS_MAXD];
    __shared__ int sc[256];
    double old[MS_MAXD], acc[MS_MAXD];
#pragma unroll
    for (int j = 0; j < MS_MAXD; ++j) { old[j] = j < D ? w.mean[j] : 0.0; acc[j] = 0.0; }
    int cnt = 0;
    for (int i = blockIdx.x * 256 + t; i < w.n; i += T) {
        const double* row = w.data + (size_t)i * D;
        double dist = 0.0;
        for (int j = 0; j < D; ++j) { const double r = old[j] - row[j]; dist += sqrt(r * r); }   // :78-83
        if (dist < band_sq) {                                                                    // :85
            for (int j = 0; j < D; ++j) acc[j] = acc[j] + row[j];
            ++cnt;
            w.votes[i] += 1;                               // row i belongs to this thread only
        }
    }
    for (int j = 0; j < MS_MAXD; ++j) sv[t][j] = acc[j];
    sc[t] = cnt;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) {
            for (int j = 0; j < D; ++j) sv[t][j] = sv[t][j] + sv[t + s][j];
            sc[t] += sc[t + s];
        }
        __syncthreads();
    }
    if (t < D) w.partial[(size_t)blockIdx.x * MS_MAXD + t] = sv[0][t];
    if (t == 0) w.partial_cnt[blockIdx.x] = sc[0];
}

__global__ void __launch_bounds__(64)
k_ms_update(MeanShiftWork all, double stop_thresh)
{
    const MeanShiftWork w = ms_climb(all, blockIdx.x);
    if (w.out[1] || w.out[3]) return;
    const int j = threadIdx.x;
    const int D = w.d;
    __shared__ double s_move[MS_MAXD];
    int in = 0;
    for (int b = 0; b < MS_GROUPS; ++b) in += w.partial_cnt[b];
    if (in == 0) { if (j == 0) w.out[3] = 1; return; }      // the reference would spin on a NaN mean
    double m = 0.0, dd = 0.0;
    if (j < D) {
        double s = 0.0;
        for (int b = 0; b < MS_GROUPS; ++b) s = s + w.partial[(size_t)b * MS_MAXD + j];
        m = s * (1.0 / (double)in);                         // cv::Mat / scalar scales by 1/s (:96)
        dd = m - w.mean[j];
        s_move[j] = dd * dd;
    }
    __syncthreads();
    if (j < D) w.mean[j] = m;
    if (j == 0) {
        double move = 0.0;
        for (int q = 0; q < D; ++q) move = move + s_move[q];
        w.out[0] += 1;
        if (sqrt(move) < stop_thresh) w.out[1] = 1;         // :98
    }
}

// (index, votes) of every row touched by the climb (the host sorts the short list); clears the votes.
__global__ void __launch_bounds__(256)
k_ms_collect(MeanShiftWork all)
{
    const MeanShiftWork w = ms_climb(all, blockIdx.y);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n) return;
    const int v = w.votes[i];
    if (v > 0) {
        const int pos = atomicAdd(&w.out[2], 1);
        w.list[2 * pos] = i;
        w.list[2 * pos + 1] = v;
        w.votes[i] = 0;
    }
}

// starts: the batch's seed rows (mapped pinned memory written by the host)
__global__ void __launch_bounds__(64)
k_ms_seed(MeanShiftWork all, const int* __restrict__ starts)
{
    const MeanShiftWork w = ms_climb(all, blockIdx.x);
    const int start = starts[blockIdx.x];
    const int j = threadIdx.x;
    if (j < w.d) w.mean[j] = w.data[(size_t)start * w.d + j];        // :58  myMean = data.row(stInd)
    if (j < 4) w.out[j] = 0;
}

// k_ms_collect, run only once the climb has ended (converged or dead end)
__global__ void __launch_bounds__(256)
k_ms_collect_if_done(MeanShiftWork all)
{
    const MeanShiftWork w = ms_climb(all, blockIdx.y);
    if (!(w.out[1] || w.out[3])) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n) return;
    const int v = w.votes[i];
    if (v > 0) {
        const int pos = atomicAdd(&w.out[2], 1);
        w.list[2 * pos] = i;
        w.list[2 * pos + 1] = v;
        w.votes[i] = 0;
    }
}

__global__ void __launch_bounds__(64)
k_ms_publish(MeanShiftWork all, MeanShiftResultBlock* results)
{
    const MeanShiftWork w = ms_climb(all, blockIdx.x);
    MeanShiftResultBlock* r = results + blockIdx.x;
    const int j = threadIdx.x;
    if (j < 4) r->out[j] = w.out[j];
    if (j < MS_MAXD) r->mean[j] = j < w.d ? w.mean[j] : 0.0;
}

hipError_t launch_ms_climb(const MeanShiftWork& w, int climbs, const int* starts_dev, double band_sq, double stop_thresh,
                           int iterations, MeanShiftResultBlock* result_dev, hipStream_t s)
{
    if (w.d > MS_MAXD || climbs < 1) return hipErrorInvalidValue;
    if (starts_dev) hipLaunchKernelGGL(k_ms_seed, dim3(climbs), dim3(64), 0, s, w, starts_dev);
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(k_ms_partial, dim3(MS_GROUPS, climbs), dim3(256), 0, s, w, band_sq);
        hipLaunchKernelGGL(k_ms_update, dim3(climbs), dim3(64), 0, s, w, stop_thresh);
    }
    hipLaunchKernelGGL(k_ms_collect_if_done, dim3((w.n + 255) / 256, climbs), dim3(256), 0, s, w);
    hipLaunchKernelGGL(k_ms_publish, dim3(climbs), dim3(64), 0, s, w, result_dev);
    return hipGetLastError();
}

hipError_t launch_ms_collect(const MeanShiftWork& w, int climbs, hipStream_t s)
{
    hipLaunchKernelGGL(k_ms_collect, dim3((w.n + 255) / 256, climbs), dim3(256), 0, s, w);
    return hipGetLastError();
}

} // namespace mh
