// datacost.hip — PEARL data-cost matrix, gfx950.
//
// dataEnergy (M/MultiH.cpp:473-504, constants from EnergyDataStruct
// M/MultiH.h:33-46), evaluated once for every (site, label) instead of lazily
// inside GCO's callback:
//     lam = 100/lambda            T = thr^2 * 81/16
//     label 0 (outlier)         -> round(lam*T)
//     d2 < T                    -> round(lam*(1 - d2/T))     (quirk A-4: decreasing in d2)
//     otherwise                 -> 2*round(lam*T)
// cost is int32, site-major: cost[i*(Nh+1) + l]  (the layout GCO's dense
// data-cost array uses, GCoptimization.h setDataCost(EnergyTermType*)).
// One thread per (site, label) element; consecutive threads write consecutive
// ints (fully coalesced).  N*(Nh+1) is tiny (2.2 MB at 50k x 11): latency-bound.

#include "mh_kernels.hpp"
#include "mh_device.hpp"

namespace mh {

__global__ void __launch_bounds__(256)
k_data_cost(const double* __restrict__ x1, const double* __restrict__ y1,
            const double* __restrict__ x2, const double* __restrict__ y2, int N,
            const double* __restrict__ H, int L, double lam, double T, int* __restrict__ cost)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)N * L) return;
    const int i = (int)(e / L);
    const int l = (int)(e - (long long)i * L);
    int c;
    if (l == 0) {
        c = (int)round(lam * T);
    } else {
        const double* h = H + 9 * (size_t)(l - 1);
        const double d2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], x1[i],
                                 y1[i], x2[i], y2[i]);
        if (d2 < T) c = (int)round(lam * (1.0 - (d2 / T)));
        else c = 2 * (int)round(lam * T);
    }
    cost[e] = c;
}

// ---------------------------------------------------------------------------
// k_cost_matrix — the data cost of EVERY hypothesis of a batch against every point, materialised: the s = 4 variant of
// the residual-matrix roofline run ("also report s = 4 (int32 cost) = 20.0 GB").  C[m*ldc + i] =
// dataEnergy(point i, label m+1) as above, int32, model-major, rows 128-B aligned; inlier counts (d2 < thr2) fused as in
// k_residual.  Work split as k_residual: a 256-thread workgroup owns MC consecutive models (coefficients staged in
// LDS) and sweeps a slice of the points; a lane holds four consecutive points, so a wave-instruction stores 1 KiB of a
// row.  Arithmetic: the shared-reciprocal division of the residual sweep for d2 (bit-identical to `/`), the
// compiler's IEEE division for d2 / T, C round() — per pair about twice the FP64 work of the residual kernel for half
// the bytes, so this kernel is FP64-issue bound, not HBM bound; its HBM fraction is reported for completeness.
// (mh_cost_matrix runs k_cost32 of score32.hip instead — the same matrix behind an FP32 pre-test — wherever that kernel's
// preconditions hold; this one remains for the other inputs and as its A/B partner, mh_set_tuning key 15.)
// ---------------------------------------------------------------------------
template <int MC>
__global__ void __launch_bounds__(256)
k_cost_matrix(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
              const double* __restrict__ y2, int N, const double* __restrict__ H, int M, double lam, double T, double thr2,
              int* __restrict__ C, long long ldc, int* __restrict__ counts, int psplit)
{
    constexpr int TILE = 1024;                   // 4 waves x 64 lanes x 4 points
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * MC;
    __shared__ double s_h[MC * 9];
    __shared__ int s_hok[MC], s_cnt[MC];
    for (int i = threadIdx.x; i < MC * 9; i += 256) {
        const size_t g = (size_t)m0 * 9 + i;
        s_h[i] = (g < (size_t)M * 9) ? H[g] : 0.0;
    }
    if (threadIdx.x < MC) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < MC) s_hok[threadIdx.x] = model_pre(s_h + 9 * threadIdx.x);
    __syncthreads();
    const int beyond = 2 * (int)round(lam * T);
    for (int base = blockIdx.y * TILE; base < N; base += psplit * TILE) {
        const int n = base + wave * 256 + lane * 4;
        double px[4], py[4], qx[4], qy[4];
        bool ok[4], pok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ok[q] = n + q < N;
            px[q] = ok[q] ? x1[n + q] : 1.0; py[q] = ok[q] ? y1[n + q] : 1.0;
            qx[q] = ok[q] ? x2[n + q] : 1.0; qy[q] = ok[q] ? y2[n + q] : 1.0;
            pok[q] = point_pre(px[q], py[q], qx[q], qy[q]);
        }
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            const int m = m0 + mi;
            if (m >= M) break;
            const double* h = s_h + 9 * mi;
            const double h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7], h8 = h[8];
            const bool hok = __builtin_amdgcn_readfirstlane(s_hok[mi]) != 0;
            int c[4], inl = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double d2 = fwd_d2_fast<true>(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[q], py[q], qx[q], qy[q], pok[q] && hok);
                c[q] = d2 < T ? (int)round(lam * (1.0 - (d2 / T))) : beyond;
                inl += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok[q] && d2 < thr2));
            }
            int* dst = C + (size_t)m * ldc + n;
            if (n + 3 < N) *reinterpret_cast<int4*>(dst) = make_int4(c[0], c[1], c[2], c[3]);
            else
                for (int q = 0; q < 4; ++q) if (ok[q]) dst[q] = c[q];
            if (lane == 0 && inl) atomicAdd(&s_cnt[mi], inl);
        }
    }
    __syncthreads();
    if (threadIdx.x < MC && m0 + (int)threadIdx.x < M) {
        if (psplit == 1) counts[m0 + threadIdx.x] = s_cnt[threadIdx.x];
        else atomicAdd(&counts[m0 + threadIdx.x], s_cnt[threadIdx.x]);
    }
}

hipError_t launch_cost_matrix(const Points& p, const double* H, int M, double lambda, double thr2, int* C, long long ldc,
                              int* counts, hipStream_t s)
{
    if (M <= 0 || p.n <= 0) return hipSuccess;
    constexpr int MC = 16;
    const int gx = (M + MC - 1) / MC, ntiles = (p.n + 1023) / 1024;
    int psplit = gx < 1024 ? (2048 + gx - 1) / gx : (ntiles >= 32 ? 4 : 1);
    if (psplit > ntiles) psplit = ntiles;
    if (psplit < 1) psplit = 1;
    if (psplit > 1) {
        hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)M, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_cost_matrix<MC>, dim3(gx, psplit), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, M, 100.0 / lambda,
                       thr2 * 81.0 / 16.0, thr2, C, ldc, counts, psplit);
    return hipGetLastError();
}

hipError_t launch_data_cost(const Points& p, const double* H, int Nh, double lambda, double thr2,
                            int* cost, hipStream_t s)
{
    const int L = Nh + 1;
    const double lam = 100.0 / lambda;        // one_per_energy_lambda, M/MultiH.h:42
    const double T = thr2 * 81.0 / 16.0;      // truncated_sqr_threshold, M/MultiH.h:44
    const long long total = (long long)p.n * L;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_data_cost, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p.x1,
                       p.y1, p.x2, p.y2, p.n, H, L, lam, T, cost);
    return hipGetLastError();
}

} // namespace mh
