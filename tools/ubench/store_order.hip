// store_order.hip — does the ORDER in which the residual sweep's workgroups write R matter to HBM?  (diagnostic, r04)
// R is row-major M x N doubles (fixed by the interface); a workgroup owns MC = 16 rows and walks 1 024-point tiles of a
// point slice, as k_residual_resident does.  Variants change which (model block, slice) an item is and where in its slice
// a row starts.  No arithmetic: the question is the write stream alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// order: 0 = model block fastest (the product), 1 = slice fastest, 2 = model block fastest with the XCD's share of blocks
// contiguous (item -> block permuted so that the 8 XCDs write 8 separate regions of R)
// stagger: row mi starts its walk at tile (mi * stagger) of the slice (mod tiles in the slice)
template <int MC, int PPL, bool NT>
__global__ void __launch_bounds__(256) k_rows_resident(double* R, int N, long long ld, int M, int psplit, int gx, int nitems,
                                                        int* ctl, int order, int stagger, double v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVE_PTS = 64 * PPL, TILE = 4 * WAVE_PTS;
    const int ntiles = (N + TILE - 1) / TILE;
    __shared__ int s_item;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&ctl[0], 1);
        __syncthreads();
        const int item = s_item;
        if (item >= nitems) break;
        int bx, by;
        if (order == 1) { bx = item / psplit; by = item - bx * psplit; }
        else { by = item / gx; bx = item - by * gx; }
        if (order == 2) { const int per = (gx + 7) / 8; const int x = bx % 8, k = bx / 8; bx = x * per + k; if (bx >= gx) { __syncthreads(); continue; } }
        const int m0 = bx * MC;
        // tiles by, by + psplit, ... of the row
        const int mine = (ntiles - by + psplit - 1) / psplit;
        for (int step = 0; step < mine; ++step) {
#pragma unroll 1
            for (int mi = 0; mi < MC; ++mi) {
                const int m = m0 + mi;
                if (m >= M) break;
                const int tstep = stagger ? (step + mi * stagger) % mine : step;
                const int base = (by + tstep * psplit) * TILE;
#pragma unroll
                for (int c = 0; c < PPL / 2; ++c) {
                    const int n = base + wave * WAVE_PTS + c * 128 + lane * 2;
                    if (n + 1 < N) {
                        double* d = R + (size_t)m * ld + n;
                        if (NT) { __builtin_nontemporal_store(v, d); __builtin_nontemporal_store(v + mi, d + 1); }
                        else *reinterpret_cast<double2*>(d) = make_double2(v, v + mi);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && atomicAdd(&ctl[1], 1) == (int)gridDim.x - 1) { ctl[1] = 0; __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

// contiguous slices instead of interleaved tiles: slice by owns tiles [by * per, (by + 1) * per)
template <int MC, int PPL, bool NT>
__global__ void __launch_bounds__(256) k_rows_contig(double* R, int N, long long ld, int M, int psplit, int gx, int nitems, int* ctl, double v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVE_PTS = 64 * PPL, TILE = 4 * WAVE_PTS;
    const int ntiles = (N + TILE - 1) / TILE, per = (ntiles + psplit - 1) / psplit;
    __shared__ int s_item;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&ctl[0], 1);
        __syncthreads();
        const int item = s_item;
        if (item >= nitems) break;
        const int by = item / gx, bx = item - by * gx, m0 = bx * MC;
        for (int t = by * per; t < (by + 1) * per && t < ntiles; ++t) {
#pragma unroll 1
            for (int mi = 0; mi < MC; ++mi) {
                const int m = m0 + mi;
                if (m >= M) break;
#pragma unroll
                for (int c = 0; c < PPL / 2; ++c) {
                    const int n = t * TILE + wave * WAVE_PTS + c * 128 + lane * 2;
                    if (n + 1 < N) {
                        double* d = R + (size_t)m * ld + n;
                        if (NT) { __builtin_nontemporal_store(v, d); __builtin_nontemporal_store(v + mi, d + 1); }
                        else *reinterpret_cast<double2*>(d) = make_double2(v, v + mi);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && atomicAdd(&ctl[1], 1) == (int)gridDim.x - 1) { ctl[1] = 0; __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

template <bool NT> __global__ void __launch_bounds__(256) k_seq(double2* p, size_t n2, double v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        if (NT) { __builtin_nontemporal_store(v, &p[i].x); __builtin_nontemporal_store(v + 1, &p[i].y); }
        else p[i] = make_double2(v, v + 1);
    }
}

int main()
{
    const int N = 50000, M = 100000;
    const long long ld = 50000;
    const size_t bytes = (size_t)M * ld * 8;
    double* R; int* ctl;
    CK(hipMalloc(&R, bytes)); CK(hipMalloc(&ctl, 64)); CK(hipMemset(ctl, 0, 64));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int r = 0; r < 8; ++r) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); tot += ms; if (ms < best) best = ms;
        }
        printf("%-64s avg %7.3f ms  best %7.3f ms  -> %7.1f GB/s\n", name, tot / 8, best, bytes / (tot / 8) / 1e6);
    };
    time("hipMemsetAsync", [&] { hipMemsetAsync(R, 0, bytes, 0); });
    time("seq grid=16384", [&] { hipLaunchKernelGGL(k_seq<false>, dim3(16384), dim3(256), 0, 0, (double2*)R, bytes / 16, 1.0); });
    const int gx = M / 16;
    for (int grid : {1280}) for (int ps : {6}) for (int order : {0, 1, 2}) for (int stagger : {0}) {
        char nm[96];
        snprintf(nm, 96, "rows nt resident grid=%d slices=%d order=%d stagger=%d", grid, ps, order, stagger);
        time(nm, [&] { hipLaunchKernelGGL((k_rows_resident<16, 4, true>), dim3(grid), dim3(256), 0, 0, R, N, ld, M, ps, gx, gx * ps, ctl, order, stagger, 1.0); });
    }
    for (int ps : {8, 12, 16, 24, 49}) for (int order : {0, 1}) {
        char nm[96];
        snprintf(nm, 96, "rows nt    resident grid=1280 slices=%d order=%d", ps, order);
        time(nm, [&] { hipLaunchKernelGGL((k_rows_resident<16, 4, true>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, ps, gx, gx * ps, ctl, order, 0, 1.0); });
        snprintf(nm, 96, "rows plain resident grid=1280 slices=%d order=%d", ps, order);
        time(nm, [&] { hipLaunchKernelGGL((k_rows_resident<16, 4, false>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, ps, gx, gx * ps, ctl, order, 0, 1.0); });
    }
    for (int ps : {6, 12}) {
        char nm[96];
        snprintf(nm, 96, "rows nt resident grid=1280 CONTIGUOUS slices=%d", ps);
        time(nm, [&] { hipLaunchKernelGGL((k_rows_contig<16, 4, true>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, ps, gx, gx * ps, ctl, 1.0); });
    }
    time("rows plain resident grid=1280 slices=6 order=0", [&] { hipLaunchKernelGGL((k_rows_resident<16, 4, false>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, 6, gx, gx * 6, ctl, 0, 0, 1.0); });
    time("rows nt MC32 resident grid=1280 slices=12", [&] { hipLaunchKernelGGL((k_rows_resident<32, 4, true>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, 12, M / 32, M / 32 * 12, ctl, 0, 0, 1.0); });
    time("rows nt MC8 resident grid=1280 slices=3", [&] { hipLaunchKernelGGL((k_rows_resident<8, 4, true>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, 3, M / 8, M / 8 * 3, ctl, 0, 0, 1.0); });
    time("rows nt MC4 resident grid=1280 slices=2", [&] { hipLaunchKernelGGL((k_rows_resident<4, 4, true>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, 2, M / 4, M / 4 * 2, ctl, 0, 0, 1.0); });
    time("rows nt MC1 resident grid=1280 slices=1", [&] { hipLaunchKernelGGL((k_rows_resident<1, 4, true>), dim3(1280), dim3(256), 0, 0, R, N, ld, M, 1, M, M, ctl, 0, 0, 1.0); });
    hipFree(R);
    return 0;
}
