// xcd_pingpong.hip — what does a hand-off between two workgroups cost, same XCD vs different XCDs, by flavour of the
// writing operation?  (diagnostic for the alpha-expansion solver's hops, r03)
//   writer flavours: agent-scope atomic add / workgroup-scope atomic add / agent-scope (sc1) store / plain store
//   reader: always an agent-scope (sc1) load poll, which bypasses L1 and is served by the XCD's L2 when the line is there
// Two workgroups bounce a counter ROUNDS times; reported: microseconds per one-way hop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int FLAVOUR>
__device__ __forceinline__ void bump(int* p, int v)
{
    if (FLAVOUR == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (FLAVOUR == 1) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (FLAVOUR == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ctl[0..7]: arrivals per XCD; ctl[8]: pairing done; word = the bounced counter.  want_same: both players on one XCD.
template <int FLAVOUR>
__global__ void __launch_bounds__(64)
k_pingpong(int* ctl, int* word, int rounds, int want_same, unsigned long long* ticks_out)
{
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);
    if (threadIdx.x != 0) return;
    // roles: player 0 = first arrival on XCD A; player 1 = second arrival on XCD A (same) or first arrival on another XCD
    const int t = __hip_atomic_fetch_add(&ctl[xcc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int role = -1;
    if (t == 0 && __hip_atomic_fetch_add(&ctl[8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        role = 0;
        __hip_atomic_store(&ctl[9], xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // player 0's XCD
    }
    if (role < 0) {
        int a;
        while ((a = __hip_atomic_load(&ctl[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(2);
        const bool same = (a - 1) == xcc;
        if (same == (want_same != 0) && __hip_atomic_fetch_add(&ctl[10], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) role = 1;
    }
    if (role < 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        const int my_turn = 2 * r + role;                  // the counter value at which I write
        int spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_turn) {
            if (++spins > 2000000) {                        // the partner's write never became visible (expected across XCDs for L2-local flavours)
                if (role == 0) *ticks_out = ~0ull;
                return;
            }
        }
        bump<FLAVOUR>(word, my_turn + 1);
    }
    if (role == 0) *ticks_out = __builtin_amdgcn_s_memrealtime() - t0;       // 100 MHz ticks
}

int main()
{
    int *ctl, *word;
    unsigned long long* ticks;
    CK(hipMalloc(&ctl, 64 * sizeof(int)));
    CK(hipMalloc(&word, 256));
    CK(hipHostMalloc(&ticks, sizeof(unsigned long long)));
    const int rounds = 20000;
    const char* names[4] = { "agent-scope atomic add", "workgroup-scope atomic add", "agent-scope store (sc1)", "plain store" };
    for (int same = 1; same >= 0; --same)
        for (int f = 0; f < 4; ++f) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(ctl, 0, 64 * sizeof(int)));
                CK(hipMemset(word, 0, 256));
                *ticks = 0;
                void (*k)(int*, int*, int, int, unsigned long long*) = f == 0 ? k_pingpong<0> : f == 1 ? k_pingpong<1> : f == 2 ? k_pingpong<2> : k_pingpong<3>;
                hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, ctl, word, rounds, same, ticks);
                CK(hipDeviceSynchronize());
                if (*ticks == ~0ull) { best = -1.0; break; }
                const double us = (double)*ticks * 0.01 / (2.0 * rounds);
                if (us > 0 && us < best) best = us;
            }
            if (best < 0) printf("%-14s %-28s not visible to the partner (spin limit hit)\n", same ? "same XCD" : "other XCD", names[f]);
            else printf("%-14s %-28s %.3f us per hop\n", same ? "same XCD" : "other XCD", names[f], best);
            fflush(stdout);
        }
    return 0;
}
