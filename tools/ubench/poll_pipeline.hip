// poll_pipeline.hip — does a hand-off get cheaper when the receiver keeps several polls of the word in flight?  (r03)
// A hop of the alpha-expansion solver = the writer's memory-side atomic + the wait until the receiver's next poll passes
// the word + that poll's way back.  With one load in flight the middle term is half a load round trip on average; with
// Q + 1 loads issued a fraction of a round trip apart it shrinks accordingly.  Two single-lane workgroups bounce a counter
// (agent-scope atomic add, sc1 load polls); the whole receive loop is one asm block so that no register with a load in
// flight is ever touched by the compiler (every block ends with s_waitcnt vmcnt(0), behind the atomic that hands over).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// returns 0 when the word reached `turn` (and was bumped), 1 on spin limit
template <int Q, int GAP>
__device__ __forceinline__ int wait_and_bump(int* word, int turn)
{
    int q0, q1, q2, q3, fail = 0, spins = 4000000, one = 1;
    if (Q == 0) {
        asm volatile(
            "1:\n"
            "global_load_dword %[q0], %[p], off sc1\n"
            "s_waitcnt vmcnt(0)\n"
            "v_cmp_eq_u32 vcc, %[q0], %[t]\n"
            "s_cbranch_vccnz 2f\n"
            "s_sub_u32 %[n], %[n], 1\n"
            "s_cmp_eq_u32 %[n], 0\n"
            "s_cbranch_scc0 1b\n"
            "v_mov_b32 %[f], 1\n"
            "s_branch 3f\n"
            "2:\n"
            "global_atomic_add %[p], %[o], off sc1\n"
            "3:\n"
            "s_waitcnt vmcnt(0)\n"
            : [q0] "=&v"(q0), [f] "+v"(fail), [n] "+s"(spins)
            : [p] "v"(word), [t] "v"(turn), [o] "v"(one)
            : "vcc", "scc", "memory");
        return fail;
    }
    if (Q == 1) {
        asm volatile(
            "global_load_dword %[q0], %[p], off sc1\n"
            "s_sleep %[gap]\n"
            "global_load_dword %[q1], %[p], off sc1\n"
            "1:\n"
            "s_waitcnt vmcnt(1)\n"
            "v_cmp_eq_u32 vcc, %[q0], %[t]\n"
            "s_cbranch_vccnz 2f\n"
            "global_load_dword %[q0], %[p], off sc1\n"
            "s_waitcnt vmcnt(1)\n"
            "v_cmp_eq_u32 vcc, %[q1], %[t]\n"
            "s_cbranch_vccnz 2f\n"
            "global_load_dword %[q1], %[p], off sc1\n"
            "s_sub_u32 %[n], %[n], 1\n"
            "s_cmp_eq_u32 %[n], 0\n"
            "s_cbranch_scc0 1b\n"
            "v_mov_b32 %[f], 1\n"
            "s_branch 3f\n"
            "2:\n"
            "global_atomic_add %[p], %[o], off sc1\n"
            "3:\n"
            "s_waitcnt vmcnt(0)\n"
            : [q0] "=&v"(q0), [q1] "=&v"(q1), [f] "+v"(fail), [n] "+s"(spins)
            : [p] "v"(word), [t] "v"(turn), [o] "v"(one), [gap] "n"(GAP)
            : "vcc", "scc", "memory");
        return fail;
    }
    asm volatile(
        "global_load_dword %[q0], %[p], off sc1\n"
        "s_sleep %[gap]\n"
        "global_load_dword %[q1], %[p], off sc1\n"
        "s_sleep %[gap]\n"
        "global_load_dword %[q2], %[p], off sc1\n"
        "s_sleep %[gap]\n"
        "global_load_dword %[q3], %[p], off sc1\n"
        "1:\n"
        "s_waitcnt vmcnt(3)\n"
        "v_cmp_eq_u32 vcc, %[q0], %[t]\n"
        "s_cbranch_vccnz 2f\n"
        "global_load_dword %[q0], %[p], off sc1\n"
        "s_waitcnt vmcnt(3)\n"
        "v_cmp_eq_u32 vcc, %[q1], %[t]\n"
        "s_cbranch_vccnz 2f\n"
        "global_load_dword %[q1], %[p], off sc1\n"
        "s_waitcnt vmcnt(3)\n"
        "v_cmp_eq_u32 vcc, %[q2], %[t]\n"
        "s_cbranch_vccnz 2f\n"
        "global_load_dword %[q2], %[p], off sc1\n"
        "s_waitcnt vmcnt(3)\n"
        "v_cmp_eq_u32 vcc, %[q3], %[t]\n"
        "s_cbranch_vccnz 2f\n"
        "global_load_dword %[q3], %[p], off sc1\n"
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_eq_u32 %[n], 0\n"
        "s_cbranch_scc0 1b\n"
        "v_mov_b32 %[f], 1\n"
        "s_branch 3f\n"
        "2:\n"
        "global_atomic_add %[p], %[o], off sc1\n"
        "3:\n"
        "s_waitcnt vmcnt(0)\n"
        : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3), [f] "+v"(fail), [n] "+s"(spins)
        : [p] "v"(word), [t] "v"(turn), [o] "v"(one), [gap] "n"(GAP)
        : "vcc", "scc", "memory");
    return fail;
}

template <int Q, int GAP>
__global__ void __launch_bounds__(64)
k_pingpong(int* ctl, int* word, int rounds, int want_same, unsigned long long* ticks_out)
{
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);
    if (threadIdx.x != 0) return;
    const int t = __hip_atomic_fetch_add(&ctl[xcc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int role = -1;
    if (t == 0 && __hip_atomic_fetch_add(&ctl[8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        role = 0;
        __hip_atomic_store(&ctl[9], xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (role < 0) {
        int a;
        while ((a = __hip_atomic_load(&ctl[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(2);
        const bool same = (a - 1) == xcc;
        if (same == (want_same != 0) && __hip_atomic_fetch_add(&ctl[10], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) role = 1;
    }
    if (role < 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r)
        if (wait_and_bump<Q, GAP>(word, 2 * r + role)) { if (role == 0) *ticks_out = ~0ull; return; }
    if (role == 0) *ticks_out = __builtin_amdgcn_s_memrealtime() - t0;
}

template <int Q, int GAP>
static int run(int* ctl, int* word, unsigned long long* ticks)
{
    const int rounds = 20000;
    for (int same = 1; same >= 0; --same) {
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctl, 0, 64 * sizeof(int)));
            CK(hipMemset(word, 0, 256));
            *ticks = 0;
            hipLaunchKernelGGL((k_pingpong<Q, GAP>), dim3(64), dim3(64), 0, 0, ctl, word, rounds, same, ticks);
            CK(hipDeviceSynchronize());
            if (*ticks == ~0ull) { best = -1.0; break; }
            const double us = (double)*ticks * 0.01 / (2.0 * rounds);
            if (us > 0 && us < best) best = us;
        }
        if (best < 0) printf("%-10s polls in flight %d, gap s_sleep %d: spin limit hit\n", same ? "same XCD" : "other XCD", Q + 1, GAP);
        else printf("%-10s polls in flight %d, gap s_sleep %d: %.3f us per hop\n", same ? "same XCD" : "other XCD", Q + 1, GAP, best);
        fflush(stdout);
    }
    return 0;
}

int main()
{
    int *ctl, *word;
    unsigned long long* ticks;
    CK(hipMalloc(&ctl, 64 * sizeof(int)));
    CK(hipMalloc(&word, 256));
    CK(hipHostMalloc(&ticks, sizeof(unsigned long long)));
    if (run<0, 0>(ctl, word, ticks)) return 1;
    if (run<1, 1>(ctl, word, ticks)) return 1;
    if (run<1, 2>(ctl, word, ticks)) return 1;
    if (run<1, 4>(ctl, word, ticks)) return 1;
    if (run<3, 1>(ctl, word, ticks)) return 1;
    if (run<3, 2>(ctl, word, ticks)) return 1;
    if (run<3, 3>(ctl, word, ticks)) return 1;
    return 0;
}
