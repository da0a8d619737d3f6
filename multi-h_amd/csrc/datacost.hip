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
