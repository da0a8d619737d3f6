// xcd_affinity.hip — does it matter WHICH XCD writes which part of a buffer?  (diagnostic, r03)
//
// Hypothesis under test: memory is interleaved over the eight HBM stacks in fixed-size granules, and an XCD's stores to
// "its own" stack travel a shorter way than stores to the others.  If so, a store stream in which XCD x only writes
// granules with (granule index + shift) % 8 == x runs at a different rate for some shift than for the others, and a
// kernel that owns its output layout (the residual matrix) could exploit it.
// Method: every workgroup reads its XCD from the hardware (HW_REG_XCC_ID), takes tickets from that XCD's own counter
// and writes, per ticket, RUN granules of G bytes that all have the residue (xcc + shift) % 8.  Sweeps G and shift;
// "mixed" = the same loop with the residue taken from the ticket instead (every XCD writes every residue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256)
k_affine(double2* __restrict__ buf, size_t granules, int g_bytes, int run, int shift, int mixed, unsigned int* __restrict__ tickets)
{
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);      // HW_REG_XCC_ID, bits 3:0
    const size_t per_res = granules / 8;                    // granules of one residue class
    const int per_g = g_bytes / 16;                         // double2 per granule
    __shared__ unsigned int s_t;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(&tickets[xcc * 32], 1u);
        __syncthreads();
        const unsigned int t = s_t;
        __syncthreads();
        const size_t first = (size_t)t * run;               // index inside the residue class
        if (first >= per_res) return;
        const int res = mixed ? (int)(t % 8u) : ((xcc + shift) & 7);
        for (int r = 0; r < run && first + r < per_res; ++r) {
            const size_t gi = (first + r) * 8 + res;        // granule index in the buffer
            double2* p = buf + gi * per_g;
            for (int i = threadIdx.x; i < per_g; i += 256) p[i] = make_double2((double)t, (double)r);
        }
    }
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 16.0) * (1ull << 30);
    double2* buf;
    unsigned int* tickets;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&tickets, 8 * 32 * sizeof(unsigned int)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8;
    auto run_one = [&](int g_bytes, int run, int shift, int mixed) -> float {
        const size_t granules = bytes / g_bytes / 8 * 8;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemsetAsync(tickets, 0, 8 * 32 * sizeof(unsigned int), 0);
            (void)hipEventRecord(a, 0);
            hipLaunchKernelGGL(k_affine, dim3(grid), dim3(256), 0, 0, buf, granules, g_bytes, run, shift, mixed, tickets);
            (void)hipEventRecord(b, 0);
            (void)hipEventSynchronize(b);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        return (float)((double)(granules * (size_t)g_bytes) / (best * 1e-3) / 1e9);
    };
    printf("# buffer %.1f GiB, grid %d x 256, GB/s (best of 3)\n", bytes / 1073741824.0, grid);
    for (int g_bytes : { 256, 1024, 4096, 16384, 65536, 1 << 20 }) {
        const int run = g_bytes >= 65536 ? 1 : 65536 / g_bytes;       // >= 64 KiB per ticket
        printf("granule %7d B  mixed %7.0f |", g_bytes, run_one(g_bytes, run, 0, 1));
        for (int shift = 0; shift < 8; ++shift) printf(" s%d %6.0f", shift, run_one(g_bytes, run, shift, 0));
        printf("\n");
        fflush(stdout);
    }
    (void)hipFree(buf); (void)hipFree(tickets);
    return 0;
}
