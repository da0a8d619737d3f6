// store_bw.hip — write-bandwidth calibration for the residual kernel's store stream (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// pattern A: grid-stride fully sequential, 16 B per lane
template <bool NT> __global__ void __launch_bounds__(256) k_seq(double2* p, size_t n2, double v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        if (NT) { __builtin_nontemporal_store(v, &p[i].x); __builtin_nontemporal_store(v + 1, &p[i].y); }
        else p[i] = make_double2(v, v + 1);
    }
}
// pattern B: each WG owns a contiguous slab (bytes/gridDim) and streams through it
template <bool NT> __global__ void __launch_bounds__(256) k_slab(double2* p, size_t n2, double v)
{
    const size_t per = n2 / gridDim.x;
    double2* q = p + per * blockIdx.x;
    for (size_t i = threadIdx.x; i < per; i += 256) {
        if (NT) { __builtin_nontemporal_store(v, &q[i].x); __builtin_nontemporal_store(v + 1, &q[i].y); }
        else q[i] = make_double2(v, v + 1);
    }
}
// pattern C: the residual kernel's pattern: WG = MC rows, tile of TILE points, rows strided by ld
template <int MC, int PPL, bool NT> __global__ void __launch_bounds__(256) k_rows(double* R, int N, long long ld, int M, double v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * MC;
    constexpr int WAVE_PTS = 64 * PPL, TILE = 4 * WAVE_PTS;
    for (int base = 0; base < N; base += TILE) {
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            const int m = m0 + mi;
            if (m >= M) break;
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
}

int main()
{
    const int N = 50000, M = 100000;
    const long long ld = 50000;
    const size_t bytes = (size_t)M * ld * 8;
    double* R;
    CK(hipMalloc(&R, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char* name, auto launch) {
        launch();
        hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); tot += ms; if (ms < best) best = ms;
        }
        printf("%-44s avg %7.3f ms  best %7.3f ms  -> %7.1f GB/s (best %7.1f)\n", name, tot / 6, best, bytes / (tot / 6) / 1e6, bytes / best / 1e6);
    };
    const size_t n2 = bytes / 16;
    time("hipMemsetAsync", [&] { hipMemsetAsync(R, 0, bytes, 0); });
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "seq grid=%d", g); time(nm, [&] { hipLaunchKernelGGL(k_seq<false>, dim3(g), dim3(256), 0, 0, (double2*)R, n2, 1.0); });
        snprintf(nm, 64, "seq nt grid=%d", g); time(nm, [&] { hipLaunchKernelGGL(k_seq<true>, dim3(g), dim3(256), 0, 0, (double2*)R, n2, 1.0); });
    }
    for (int g : {2048, 6250, 25000}) {
        char nm[64];
        snprintf(nm, 64, "slab grid=%d", g); time(nm, [&] { hipLaunchKernelGGL(k_slab<false>, dim3(g), dim3(256), 0, 0, (double2*)R, n2, 1.0); });
        snprintf(nm, 64, "slab nt grid=%d", g); time(nm, [&] { hipLaunchKernelGGL(k_slab<true>, dim3(g), dim3(256), 0, 0, (double2*)R, n2, 1.0); });
    }
    time("rows MC16 PPL2", [&] { hipLaunchKernelGGL((k_rows<16, 2, false>), dim3(M / 16), dim3(256), 0, 0, R, N, ld, M, 1.0); });
    time("rows MC16 PPL2 nt", [&] { hipLaunchKernelGGL((k_rows<16, 2, true>), dim3(M / 16), dim3(256), 0, 0, R, N, ld, M, 1.0); });
    time("rows MC16 PPL4", [&] { hipLaunchKernelGGL((k_rows<16, 4, false>), dim3(M / 16), dim3(256), 0, 0, R, N, ld, M, 1.0); });
    time("rows MC16 PPL4 nt", [&] { hipLaunchKernelGGL((k_rows<16, 4, true>), dim3(M / 16), dim3(256), 0, 0, R, N, ld, M, 1.0); });
    time("rows MC1 PPL4", [&] { hipLaunchKernelGGL((k_rows<1, 4, false>), dim3(M), dim3(256), 0, 0, R, N, ld, M, 1.0); });
    time("rows MC4 PPL8", [&] { hipLaunchKernelGGL((k_rows<4, 8, false>), dim3(M / 4), dim3(256), 0, 0, R, N, ld, M, 1.0); });
    time("rows MC64 PPL2", [&] { hipLaunchKernelGGL((k_rows<64, 2, false>), dim3((M + 63) / 64), dim3(256), 0, 0, R, N, ld, M, 1.0); });
    hipFree(R);
    return 0;
}
