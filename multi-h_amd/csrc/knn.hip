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
