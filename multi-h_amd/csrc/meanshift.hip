// meanshift.hip — flat-kernel mean-shift climb on the GPU.
//
// MeanShiftClustering<T>::Cluster (M/moduls/mode_seeking/MeanShiftClustering.h:23-157) runs, for
// every seed, the loop :62-123: membership = { i : sum_j sqrt(d_ij^2) < bandwidth^2 } (an L1 ball,
// quirk A-8), new mean = (sum of members) * (1/count), every member gets a vote and is marked
// visited, stop when ||mean - old|| < 1e-3*bandwidth.  The reference allocates an N x D repmat per
// iteration; at N = 50k (EstablishStablePointSets, M/MultiH.cpp:604-694) that inner loop is the
// cost.  Here ONE workgroup runs a whole climb on the device: 256 threads sweep the rows strided,
// member sums are reduced in the engine's deterministic order (thread t adds rows t, t+256, ... then
// a binary tree), the convergence test is evaluated by every thread on the same LDS values, so no
// host round trip happens inside a climb.  The seed order, vote merging and final assignment stay on
// the host (they are sequential by definition, :52-56,:100-146).
#include "mh_kernels.hpp"

namespace mh {

constexpr int MS_MAXD = 16;

__global__ void __launch_bounds__(256)
k_ms_climb(MeanShiftWork w, double band_sq, double stop_thresh, int max_iters)
{
    const int t = threadIdx.x;
    const int D = w.d;
    __shared__ double s_mean[MS_MAXD];
    __shared__ double sv[256][MS_MAXD];
    __shared__ int sc[256];
    if (t < D) s_mean[t] = w.mean[t];
    __syncthreads();
    int it = 0, converged = 0;
    for (; it < max_iters; ++it) {
        double acc[MS_MAXD];
        double old[MS_MAXD];
#pragma unroll
        for (int j = 0; j < MS_MAXD; ++j) { acc[j] = 0.0; old[j] = j < D ? s_mean[j] : 0.0; }
        int cnt = 0;
        for (int i = t; i < w.n; i += 256) {
            const double* row = w.data + (size_t)i * D;
            double dist = 0.0;
            for (int j = 0; j < D; ++j) { const double r = old[j] - row[j]; dist += sqrt(r * r); }   // :78-83
            if (dist < band_sq) {                                                                    // :85
                for (int j = 0; j < D; ++j) acc[j] = acc[j] + row[j];
                ++cnt;
                w.votes[i] += 1;          // row i belongs to this thread only
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
        const int in = sc[0];
        if (in == 0) break;               // the reference would spin on a NaN mean; end the climb
        const double inv = 1.0 / (double)in;                          // cv::Mat / scalar (:96)
        double move = 0.0;
        for (int j = 0; j < D; ++j) { const double m = sv[0][j] * inv; const double dd = m - old[j]; move = move + dd * dd; }
        __syncthreads();
        if (t < D) s_mean[t] = sv[0][t] * inv;
        __syncthreads();
        if (sqrt(move) < stop_thresh) { converged = 1; ++it; break; }  // :98
    }
    if (t < D) w.mean[t] = s_mean[t];
    if (t == 0) { w.out[0] = it; w.out[1] = converged; }
}

// (index, votes) of every row touched by the climb, in index order per thread block scan is not
// needed: the host sorts the short list.  Clears the votes for the next climb.
__global__ void __launch_bounds__(256)
k_ms_collect(MeanShiftWork w)
{
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

hipError_t launch_ms_climb(const MeanShiftWork& w, double band_sq, double stop_thresh, int max_iters,
                           hipStream_t s)
{
    if (w.d > MS_MAXD) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ms_climb, dim3(1), dim3(256), 0, s, w, band_sq, stop_thresh, max_iters);
    return hipGetLastError();
}

hipError_t launch_ms_collect(const MeanShiftWork& w, hipStream_t s)
{
    hipLaunchKernelGGL(k_ms_collect, dim3((w.n + 255) / 256), dim3(256), 0, s, w);
    return hipGetLastError();
}

} // namespace mh
