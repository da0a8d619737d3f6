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
// sorted register array with a fully unrolled insertion.  One thread per query
// alone leaves the chip empty (50k queries = 196 workgroups on 256 CUs, one
// wave per SIMD), so the candidate range is cut into `splits` slices
// (blockIdx.y), each slice keeps its own top-K per query, and k_knn_merge takes
// the K smallest (distance, index) pairs of the slices' lists: the same set in
// the same order as one pass over all candidates.

#include "mh_kernels.hpp"

namespace mh {

template <int K>
__global__ void __launch_bounds__(256)
k_knn(const double* __restrict__ x1, const double* __restrict__ y1,
      const double* __restrict__ x2, const double* __restrict__ y2, int N, int k, int slice,
      int* __restrict__ out, float* __restrict__ part_d, int* __restrict__ part_i)
{
    __shared__ float4 tile[256];
    const int q = blockIdx.x * 256 + threadIdx.x;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < N) me = make_float4((float)x1[q], (float)y1[q], (float)x2[q], (float)y2[q]);
    float bd[K];
    int bi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = 0x7fffffff; }

    const int first = blockIdx.y * slice;                       // my slice of the candidates (a multiple of 256 long)
    const int last = (first + slice) < N ? (first + slice) : N;
    for (int base = first; base < last; base += 256) {
        const int c = base + threadIdx.x;
        __syncthreads();
        if (c < N) tile[threadIdx.x] = make_float4((float)x1[c], (float)y1[c], (float)x2[c], (float)y2[c]);
        __syncthreads();
        const int lim = (last - base) < 256 ? (last - base) : 256;
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
    if (q >= N) return;
    if (gridDim.y == 1) {
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (i < k) out[(size_t)q * k + i] = bi[i];
    } else {
        const size_t o = ((size_t)blockIdx.y * N + q) * K;
#pragma unroll
        for (int i = 0; i < K; ++i) { part_d[o + i] = bd[i]; part_i[o + i] = bi[i]; }
    }
}

// The K smallest (distance, index) pairs of `splits` sorted lists per query: a K-step merge by list heads.
template <int K>
__global__ void __launch_bounds__(256)
k_knn_merge(int N, int k, int splits, const float* __restrict__ part_d, const int* __restrict__ part_i,
            int* __restrict__ out)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= N) return;
    int head[KNN_MAX_SPLITS];
    for (int s = 0; s < splits; ++s) head[s] = 0;
    for (int i = 0; i < k; ++i) {
        float bd = __builtin_inff();
        int bi = 0x7fffffff, bs = 0;
        for (int s = 0; s < splits; ++s) {
            if (head[s] >= K) continue;
            const size_t o = ((size_t)s * N + q) * K + head[s];
            const float d = part_d[o];
            const int j = part_i[o];
            if (d < bd || (d == bd && j < bi)) { bd = d; bi = j; bs = s; }
        }
        out[(size_t)q * k + i] = bi;
        ++head[bs];
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

template <int K>
static hipError_t launch_knn_k(const Points& p, int k, int* nbr_out, int splits, float* part_d, int* part_i, hipStream_t s)
{
    const int blocks = (p.n + 255) / 256;
    int slice = ((p.n + splits - 1) / splits + 255) / 256 * 256;
    if (slice < 256) slice = 256;
    hipLaunchKernelGGL((k_knn<K>), dim3(blocks, splits), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, k, slice, nbr_out,
                       part_d, part_i);
    if (splits > 1)
        hipLaunchKernelGGL((k_knn_merge<K>), dim3(blocks), dim3(256), 0, s, p.n, k, splits, part_d, part_i, nbr_out);
    return hipGetLastError();
}

// splits > 1 needs scratch for the slices' lists: splits x n x K floats and ints (K = 8, 16 or 32, the smallest >= k)
hipError_t launch_knn(const Points& p, int k, int* nbr_out, int splits, float* part_d, int* part_i, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    if (splits < 1 || splits > KNN_MAX_SPLITS || (splits > 1 && (!part_d || !part_i))) return hipErrorInvalidValue;
    if (k <= 8) return launch_knn_k<8>(p, k, nbr_out, splits, part_d, part_i, s);
    if (k <= 16) return launch_knn_k<16>(p, k, nbr_out, splits, part_d, part_i, s);
    if (k <= 32) return launch_knn_k<32>(p, k, nbr_out, splits, part_d, part_i, s);
    return hipErrorInvalidValue;
}

} // namespace mh
