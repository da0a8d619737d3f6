// knn.hip — exact k-nearest-neighbour hit list in the reference's 4-D float32
// space, gfx950.  SURVEY §8(f) row 1: replaces the FLANN radius search of
// ClusterMergingAndLabeling (M/MultiH.cpp:233-253), which is approximate,
// RNG-dependent and allocates N x N matrices inside OpenCV.  DEVIATION (stated
// in DESIGN.md): exact kNN instead of approximate radius search; the hit list
// has the same shape (directed hits per query) and feeds mh_set_neighbors_csr
// semantics unchanged.
//
// Point vectors are (float)x1, (float)y1, (float)x2, (float)y2 (:242-246).
// Distance = ((dx*dx + dy*dy) + dz*dz) + dw*dw in float32, rounding once per op;
// ties are broken by the smaller index, so the result is a pure function of
// the input.  One thread per query; candidates stream through LDS in tiles of
// 256 (16 B each, conflict-free broadcast reads); the running top-K is a
// sorted register array with a fully unrolled insertion.

#include "mh_kernels.hpp"

namespace mh {

template <int K>
__global__ void __launch_bounds__(256)
k_knn(const double* __restrict__ x1, const double* __restrict__ y1,
      const double* __restrict__ x2, const double* __restrict__ y2, int N, int k,
      int* __restrict__ out)
{
    __shared__ float4 tile[256];
    const int q = blockIdx.x * 256 + threadIdx.x;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < N) me = make_float4((float)x1[q], (float)y1[q], (float)x2[q], (float)y2[q]);
    float bd[K];
    int bi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = 0x7fffffff; }

    for (int base = 0; base < N; base += 256) {
        const int c = base + threadIdx.x;
        __syncthreads();
        if (c < N) tile[threadIdx.x] = make_float4((float)x1[c], (float)y1[c], (float)x2[c], (float)y2[c]);
        __syncthreads();
        const int lim = (N - base) < 256 ? (N - base) : 256;
        for (int t = 0; t < lim; ++t) {
            const int j = base + t;
            const float4 o = tile[t];
            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z, dw = me.w - o.w;
            const float d = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            if (j == q) continue;
            // (d, j) < (bd[K-1], bi[K-1]) lexicographically; j increases, so on equal d the
            // incumbent (smaller index) stays.
            if (d < bd[K - 1]) {
                float cd = d;
                int ci = j;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const bool lt = (cd < bd[i]) || (cd == bd[i] && ci < bi[i]);
                    const float td = bd[i];
                    const int ti = bi[i];
                    bd[i] = lt ? cd : td;
                    bi[i] = lt ? ci : ti;
                    cd = lt ? td : cd;
                    ci = lt ? ti : ci;
                }
            }
        }
    }
    if (q < N) {
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (i < k) out[(size_t)q * k + i] = bi[i];
    }
}

// Exact radius search — the reference's own neighbourhood rule (M/MultiH.cpp:252-253: radiusMatch with
// maxDistance = 1/locality in the same float32 4-D space; FLANN answers it approximately, this is the
// exact set).  Hit  <=>  ((dx^2+dy^2)+dz^2)+dw^2 <= r2 in float32, the query itself included (FLANN
// returns it at distance 0; LabelingStep skips it, :537).  Two passes over the same tiles: FILL =
// false counts the hits of each query, FILL = true writes them in increasing index order at
// rowptr[q] (the host does the prefix sum in between).
template <bool FILL>
__global__ void __launch_bounds__(256)
k_radius(const double* __restrict__ x1, const double* __restrict__ y1,
         const double* __restrict__ x2, const double* __restrict__ y2, int N, float r2,
         int* __restrict__ counts, const int* __restrict__ rowptr, int* __restrict__ col)
{
    __shared__ float4 tile[256];
    const int q = blockIdx.x * 256 + threadIdx.x;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < N) me = make_float4((float)x1[q], (float)y1[q], (float)x2[q], (float)y2[q]);
    int cnt = 0;
    int* dst = (FILL && q < N) ? col + rowptr[q] : nullptr;
    for (int base = 0; base < N; base += 256) {
        const int c = base + threadIdx.x;
        __syncthreads();
        if (c < N) tile[threadIdx.x] = make_float4((float)x1[c], (float)y1[c], (float)x2[c], (float)y2[c]);
        __syncthreads();
        const int lim = (N - base) < 256 ? (N - base) : 256;
        for (int t = 0; t < lim; ++t) {
            const float4 o = tile[t];
            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z, dw = me.w - o.w;
            const float d = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            if (d <= r2) {
                if (FILL && q < N) dst[cnt] = base + t;
                ++cnt;
            }
        }
    }
    if (!FILL && q < N) counts[q] = cnt;
}

hipError_t launch_radius_count(const Points& p, float r2, int* counts, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL((k_radius<false>), dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, r2,
                       counts, nullptr, nullptr);
    return hipGetLastError();
}

hipError_t launch_radius_fill(const Points& p, float r2, const int* rowptr, int* col, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL((k_radius<true>), dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, r2,
                       nullptr, rowptr, col);
    return hipGetLastError();
}

hipError_t launch_knn(const Points& p, int k, int* nbr_out, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    const dim3 grid((p.n + 255) / 256), blk(256);
    if (k <= 8) hipLaunchKernelGGL((k_knn<8>), grid, blk, 0, s, p.x1, p.y1, p.x2, p.y2, p.n, k, nbr_out);
    else if (k <= 16) hipLaunchKernelGGL((k_knn<16>), grid, blk, 0, s, p.x1, p.y1, p.x2, p.y2, p.n, k, nbr_out);
    else if (k <= 32) hipLaunchKernelGGL((k_knn<32>), grid, blk, 0, s, p.x1, p.y1, p.x2, p.y2, p.n, k, nbr_out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

} // namespace mh
