// meanshift.hip — flat-kernel mean-shift climb on the GPU.
//
// MeanShiftClustering<T>::Cluster (M/moduls/mode_seeking/MeanShiftClustering.h:23-157) runs, for
// every seed, the loop :62-123: membership = { i : sum_j sqrt(d_ij^2) < bandwidth^2 } (an L1 ball,
// quirk A-8), new mean = (sum of members) * (1/count), every member gets a vote and is marked
// visited, stop when ||mean - old|| < 1e-3*bandwidth.  The reference allocates an N x D repmat per
// iteration; at N = 50k (EstablishStablePointSets, M/MultiH.cpp:604-694) that inner loop is the
// cost.  Here an iteration is ONE launch that never returns to the host (k_ms_iterate):
//   up to MS_GROUPS workgroups per climb sweep the rows (global thread g takes rows g, g+T, g+2T, ...; workgroups beyond
//   the rows are not launched), members vote, each workgroup reduces its member sums with a binary tree (skipped when it
//   has no member: the tree would add zeros); the LAST workgroup of the climb to finish (a ticket) adds the partials in
//   group order, forms the new mean, tests convergence and raises the `done` word that turns the rest of the batch into
//   no-ops.  (r01-r04: two launches per iteration, k_ms_partial + k_ms_update.)
// The summation order (strided, tree inside a group, groups in sequence) is fixed — it does not
// depend on the device — and is mirrored by the oracle (mho_mean_shift), so modes and assignments
// are bit-identical.  The seed order, vote merging and final assignment stay on the host (they are
// sequential by definition, :52-56,:100-146).
// MS_BATCH climbs run side by side (blockIdx.y = climb): their seeds are drawn together from the rows that are
// unvisited when the batch starts, and the host applies the finished climbs in draw order, dropping a climb whose
// seed an earlier climb of the same batch has visited meanwhile (the reference never starts from a visited row).
// A climb depends only on the data and its seed, so the batch costs the host round trips of its longest climb
// instead of the sum.  The batch size is part of the definition (the oracle draws the same way).
#include "mh_kernels.hpp"

#include <algorithm>

namespace mh {

constexpr int MS_MAXD = 16;
constexpr int MS_GROUPS = 64;            // workgroups per sweep; part of the numerical definition

// the slice of climb b
__device__ __forceinline__ MeanShiftWork ms_climb(const MeanShiftWork& a, int b)
{
    MeanShiftWork w = a;
    w.mean = a.mean + (size_t)b * MS_MAXD;
    w.votes = a.votes + (size_t)b * a.n;
    w.out = a.out + (size_t)b * 4;
    w.list = a.list + (size_t)b * 2 * a.n;
    w.partial = a.partial + (size_t)b * MS_GROUPS * MS_MAXD;
    w.partial_cnt = a.partial_cnt + (size_t)b * MS_GROUPS;
    return w;
}

// One climb iteration in ONE launch: `groups` workgroups per climb form the partial sums, and the last of them to
// finish (a ticket per climb) adds the partials in group order, forms the new mean and tests convergence — the work of
// the former k_ms_update, without a launch of its own.  tickets: one int per climb, zero between launches.
__global__ void __launch_bounds__(256)
k_ms_iterate(MeanShiftWork all, MeanShiftActive active, int groups, double band_sq, double stop_thresh, int* __restrict__ tickets)
{
    const int climb = active.climb[blockIdx.y];
    const MeanShiftWork w = ms_climb(all, climb);
    // converged or dead end: rest of the batch idles.  (The words are written by the climb's LAST workgroup of a launch, and
    // every workgroup of the climb has read them before that one can know it is the last.)
    if (w.out[1] || w.out[3]) return;
    const int t = threadIdx.x;
    const int D = w.d;
    const int T = MS_GROUPS * 256;
    __shared__ double sv[MS_MAXD][256];                     // [component][thread]: a wave's lanes read consecutive words (the
                                                            // [thread][component] layout put all 64 lanes on one bank)
    __shared__ int sc[256];
    double old[MS_MAXD], acc[MS_MAXD];
#pragma unroll
    for (int j = 0; j < MS_MAXD; ++j) { old[j] = j < D ? w.mean[j] : 0.0; acc[j] = 0.0; }
    int cnt = 0;
    for (int i = blockIdx.x * 256 + t; i < w.n; i += T) {
        const double* row = w.data + (size_t)i * D;
        double dist = 0.0;
        // :78-83 takes sqrt(r * r) per component.  In binary floating point with correctly rounded operations that IS |r|
        // whenever r * r neither overflows nor underflows (radix-2 property; tests/test_oracle_cpu.py checks it on 10^7 values
        // and on the neighbours of every power of two), so the square root is only formed outside 2^-500 <= |r| <= 2^500.
        bool plain = true;
        double a[MS_MAXD];
#pragma unroll
        for (int j = 0; j < MS_MAXD; ++j) {
            a[j] = j < D ? fabs(old[j] - row[j]) : 0.0;
            plain = plain && (a[j] <= 0x1p500) && (a[j] >= 0x1p-500 || a[j] == 0.0);
        }
        if (__builtin_expect(plain, 1)) {
#pragma unroll
            for (int j = 0; j < MS_MAXD; ++j) if (j < D) dist += a[j];
        } else {
            asm volatile("; mean shift: sqrt path");         // (keeps the compiler from computing both and selecting)
            for (int j = 0; j < D; ++j) { const double r = old[j] - row[j]; dist += sqrt(r * r); }
        }
        if (dist < band_sq) {                                                                    // :85
            for (int j = 0; j < D; ++j) acc[j] = acc[j] + row[j];
            ++cnt;
            w.votes[i] += 1;                               // row i belongs to this thread only
        }
    }
    // Partials are written with device-scope stores and drained (s_waitcnt) before the ticket, and read back with
    // device-scope loads: no cache maintenance (a __threadfence here writes the XCD's L2 back, per workgroup: 3 x slower).
    auto put_sum = [&](int j, double v) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(w.partial) + (size_t)blockIdx.x * MS_MAXD + j,
                           (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto put_cnt = [&](int c) { __hip_atomic_store(w.partial_cnt + blockIdx.x, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (__syncthreads_or(cnt) == 0) {
        // no member among this workgroup's rows (the usual case: a ball holds a few rows): the tree below would add zeros
        if (t < D) put_sum(t, 0.0);
        if (t == 0) put_cnt(0);
    } else {
        for (int j = 0; j < MS_MAXD; ++j) sv[j][t] = acc[j];
        sc[t] = cnt;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (t < s) {
                for (int j = 0; j < D; ++j) sv[j][t] = sv[j][t] + sv[j][t + s];
                sc[t] += sc[t + s];
            }
            __syncthreads();
        }
        if (t < D) put_sum(t, sv[t][0]);
        if (t == 0) put_cnt(sc[0]);
    }

    // ---- the climb's last workgroup forms the new mean -------------------------------------------------------------
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my partials have landed before my ticket
    __syncthreads();
    if (t == 0) {
        const int k = atomicAdd(&tickets[climb], 1);
        s_last = (k == groups - 1);
        if (s_last) __hip_atomic_store(&tickets[climb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nobody else touches it before the next launch)
    }
    __syncthreads();
    if (!s_last) return;
    // the other workgroups' partials come from other compute units, other XCDs: device-scope loads (relaxed: all of them
    // in flight at once), not whatever this XCD's L2 holds of those lines
    auto cnt_of = [&](int b) { return __hip_atomic_load(w.partial_cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto sum_of = [&](int b, int j) {
        return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(w.partial) + (size_t)b * MS_MAXD + j,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    __shared__ double s_move[MS_MAXD];
    __shared__ int s_in;
    if (t < 64) {
        int in = 0;
#pragma unroll
        for (int b = 0; b < MS_GROUPS; ++b) { const int c = b < groups ? cnt_of(b) : 0; in += c; }
        if (t == 0) s_in = in;
        if (t < D && in != 0) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < MS_GROUPS; ++b) {           // groups in sequence (a group that was not launched: +0, and s is never -0)
                const double v = b < groups ? sum_of(b, t) : 0.0;
                s = s + v;
            }
            const double m = s * (1.0 / (double)in);        // cv::Mat / scalar scales by 1/s (:96)
            const double dd = m - w.mean[t];                // (still the mean this iteration started from: this is its only writer)
            s_move[t] = dd * dd;
            w.mean[t] = m;
        }
    }
    __syncthreads();
    if (t == 0) {
        if (s_in == 0) { w.out[3] = 1; return; }            // the reference would spin on a NaN mean
        double move = 0.0;
        for (int q = 0; q < D; ++q) move = move + s_move[q];
        w.out[0] += 1;
        if (sqrt(move) < stop_thresh) w.out[1] = 1;         // :98
    }
}

// ---- the tail of a batch without a launch per iteration (r05) ------------------------------------------------------
// After a round or two of k_ms_iterate most of a batch's 256 climbs have ended; the handful that have not take dozens to
// hundreds of iterations, and at one launch per iteration (about 29 us each with the host round trip every few: four
// dependent trips to memory for a thread's rows, the tree, the ticket, the launch) they were most of mh_mean_shift's time
// (DESIGN.md 3.5).  k_ms_persist runs those climbs to their end in ONE launch:
//  * one workgroup per GROUP of the definition (64 per climb), so the strided sums and the tree are the same operations;
//  * a thread's rows never change from iteration to iteration (rows g, g + T, ... of the 64 x 256 global threads), so it
//    LOADS THEM ONCE into registers (at most MS_CACHED_ROWS = 4 rows: n <= 65 536; larger inputs keep the launched form)
//    and an iteration reads nothing but the 64 partials; votes are counted in registers and added at the end;
//  * a barrier per climb and iteration on a counter of the climb's own; EVERY workgroup then adds the partials in group
//    order and forms the new mean for itself — the ticket holder's arithmetic of k_ms_iterate, operation for operation —
//    so nothing has to be handed back before the next sweep.  Partials are double-buffered by iteration parity (a
//    workgroup can be one barrier ahead of another, never two).
// Residency: the barrier needs all workgroups of a climb on the chip.  The first thing a workgroup does — before it
// touches any state — is to arrive on the climb's gate word and wait there; a workgroup that waits longer than
// `gate_timeout` closes the gate (one atomic on the same word decides between "everybody was there" and "closed"),
// every workgroup of the climb leaves, and the host goes on with launched iterations (a GPU shared with other work).
struct MeanShiftPersist {
    int* gate;               // [climb] arrivals | closed bit; zero on entry
    int* arrive;             // [climb] monotonic arrival counter of the iteration barriers; zero on entry
    double* partial2;        // [climb][2][MS_GROUPS][16]
    int* partial_cnt2;       // [climb][2][MS_GROUPS]
    int* fell_back;          // [climb] the gate's verdict: 1 = closed before everybody was there (the climb's state is untouched), 2 = open
    unsigned long long* ticks;   // nullable diagnostic: 100 MHz ticks of the first climb's first workgroup in {gate + row load, sweep + tree, barrier, new mean}
};
constexpr int MS_GATE_CLOSED = 1 << 30;
constexpr int MS_CACHED_ROWS = 4;

// (three waves per SIMD: 168 registers — the 10-D form would take 203 and leave room for two; it spills 15 of them into 60 bytes)
template <int D, bool TIMED>
__global__ void __launch_bounds__(256, 3)
k_ms_persist(MeanShiftWork all, MeanShiftActive active, MeanShiftPersist ps, int groups, double band_sq, double stop_thresh,
             int max_iters, unsigned long long gate_timeout)
{
    const int climb = active.climb[blockIdx.y];
    const MeanShiftWork w = ms_climb(all, climb);
    if (w.out[1] || w.out[3]) return;                       // (every workgroup of the climb reads the same words)
    const int t = threadIdx.x;
    const int b = blockIdx.x;                               // my group
    const int G = groups;
    const int T = MS_GROUPS * 256;
    __shared__ double sv[D][256];
    __shared__ int sc[256];
    __shared__ int s_flag, s_in, s_conv;
    __shared__ double s_mean[MS_MAXD], s_move[MS_MAXD];

    // TIMED (MULTIH_MS_STATS): phase ticks of the first climb's first workgroup; compiled out of the product instantiation
    const bool timed = TIMED && ps.ticks && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    unsigned long long tk[TIMED ? 5 : 1] = {}, tk0 = TIMED ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto lap = [&](int which) { if (TIMED) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); tk[TIMED ? which : 0] += now - tk0; tk0 = now; } };
    // ---- the gate ----
    // gate word: arrivals in the low bits (one atomicAdd per workgroup — a compare-and-swap loop of 64 workgroups on one
    // word cost 100 us per launch), MS_GATE_CLOSED on top; verdict word: written once by whoever closes the gate.  Open for
    // all: G arrivals seen with the bit clear, or the closer found G arrivals in the word it closed (nobody can be missing
    // then).  Closed for all: the closer found fewer — workgroups that arrive later find the bit in what their atomicAdd
    // returns, waiters find it in what they poll, and all of them follow the verdict.
    if (t == 0) {
        int* gate = ps.gate + climb;
        int* verdict = ps.fell_back + climb;                // 0 = none yet, 1 = closed (fall back), 2 = open after all
        int v = atomicAdd(gate, 1) + 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
        int open = -1;
        while (open < 0) {
            if (v & MS_GATE_CLOSED) {
                int d = 0;
                while ((d = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(2);
                open = d == 2 ? 1 : 0;
            } else if ((v & (MS_GATE_CLOSED - 1)) >= G) {
                open = 1;
            } else if (__builtin_amdgcn_s_memrealtime() - t0 > gate_timeout) {
                const int old = atomicOr(gate, MS_GATE_CLOSED);
                if (old & MS_GATE_CLOSED) { v = old; continue; }            // somebody else closed it: follow their verdict
                open = (old & (MS_GATE_CLOSED - 1)) >= G ? 1 : 0;
                __hip_atomic_store(verdict, open ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        s_flag = open;
    }
    __syncthreads();
    if (!s_flag) return;
    lap(4);

    // ---- my rows, once ----
    double r[MS_CACHED_ROWS][D];
    int vote[MS_CACHED_ROWS];
    int nr = 0;
#pragma unroll
    for (int k = 0; k < MS_CACHED_ROWS; ++k) {
        const int i = b * 256 + t + k * T;
        vote[k] = 0;
        if (i < w.n) {
            nr = k + 1;
            const double* row = w.data + (size_t)i * D;
#pragma unroll
            for (int j = 0; j < D; ++j) r[k][j] = row[j];
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) r[k][j] = 0.0;
        }
    }
    if (t < MS_MAXD) s_mean[t] = t < D ? w.mean[t] : 0.0;
    __syncthreads();
    lap(0);

    int* arrive = ps.arrive + climb;
    int iters = 0, converged = 0, dead = 0;
    for (int it = 0; it < max_iters; ++it) {
        double* part = ps.partial2 + ((size_t)climb * 2 + (it & 1)) * MS_GROUPS * MS_MAXD;
        int* pcnt = ps.partial_cnt2 + ((size_t)climb * 2 + (it & 1)) * MS_GROUPS;
        double acc[D];
#pragma unroll
        for (int j = 0; j < D; ++j) acc[j] = 0.0;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < MS_CACHED_ROWS; ++k) {
            if (k < nr) {
                // (the membership test of k_ms_iterate, operation for operation: |r| for sqrt(r * r) where that is exact; the
                // mean comes from LDS — broadcast reads — so that it does not occupy twenty registers beside the rows)
                double dist = 0.0;
                bool plain = true;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const double a = fabs(s_mean[j] - r[k][j]);
                    plain = plain && (a <= 0x1p500) && (a >= 0x1p-500 || a == 0.0);
                    dist += a;
                }
                if (__builtin_expect(!plain, 0)) {
                    asm volatile("; mean shift: sqrt path");
                    dist = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) { const double q = s_mean[j] - r[k][j]; dist += sqrt(q * q); }
                }
                if (dist < band_sq) {
#pragma unroll
                    for (int j = 0; j < D; ++j) acc[j] = acc[j] + r[k][j];
                    ++cnt;
                    ++vote[k];
                }
            }
        }
        auto put_sum = [&](int j, double v) {
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(part) + (size_t)b * MS_MAXD + j,
                               (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (__syncthreads_or(cnt) == 0) {
            if (t < D) put_sum(t, 0.0);
            if (t == 0) __hip_atomic_store(pcnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) sv[j][t] = acc[j];
            sc[t] = cnt;
            __syncthreads();
            // the tree of k_ms_iterate: levels 128 and 64 pair lanes of different waves (through LDS), levels 32 .. 1 pair
            // lane l with lane l + s of ONE wave — the same two operands, added once (a + b = b + a), fetched with a lane
            // shuffle instead of an LDS round trip and a workgroup barrier per level
            if (t < 128) {
#pragma unroll
                for (int j = 0; j < D; ++j) sv[j][t] = sv[j][t] + sv[j][t + 128];
                sc[t] += sc[t + 128];
            }
            __syncthreads();
            if (t < 64) {
                double v[D];
#pragma unroll
                for (int j = 0; j < D; ++j) v[j] = sv[j][t] + sv[j][t + 64];
                int c = sc[t] + sc[t + 64];
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
                    for (int j = 0; j < D; ++j) v[j] = v[j] + __shfl_down(v[j], s, 64);     // (lanes >= s compute values nobody reads)
                    c += __shfl_down(c, s, 64);
                }
                if (t == 0) {
#pragma unroll
                    for (int j = 0; j < D; ++j) put_sum(j, v[j]);
                    __hip_atomic_store(pcnt + b, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ---- the climb's barrier: my partials have landed, then everybody's ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        lap(1);
        if (t == 0) {
            atomicAdd(arrive, 1);
            const int target = G * (it + 1);
            while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        lap(2);
        // ---- every workgroup forms the new mean for itself (the ticket holder's arithmetic) ----
        // the climb's G x 16 partial sums come through LDS (all 256 threads fetch, device-scope loads all in flight at once);
        // 64 of them held in registers by each of D lanes cost the kernel a wave per SIMD
        double* flat = &sv[0][0];                           // (D x 256 doubles >= 64 x 16; the tree is done with it)
        for (int idx = t; idx < G * MS_MAXD; idx += 256)
            if ((idx & (MS_MAXD - 1)) < D) flat[idx] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(part) + idx,
                                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        int in = 0;
        if (t < 64) {
            in = t < G ? __hip_atomic_load(pcnt + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) in += __shfl_xor(in, m, 64);          // (integers: any order)
            if (t == 0) s_in = in;
        }
        __syncthreads();
        if (t < D && in != 0) {
            double s = 0.0;
            for (int q = 0; q < G; ++q) s = s + flat[q * MS_MAXD + t];               // groups in sequence (s is never -0)
            const double m = s * (1.0 / (double)in);
            const double dd = m - s_mean[t];                // (the mean this iteration started from: thread t is its only writer)
            s_move[t] = dd * dd;
            s_mean[t] = m;
        }
        __syncthreads();
        if (s_in == 0) { dead = 1; break; }
        if (t == 0) {
            double move = 0.0;
            for (int q = 0; q < D; ++q) move = move + s_move[q];
            s_conv = sqrt(move) < stop_thresh ? 1 : 0;
        }
        __syncthreads();
        ++iters;
        converged = s_conv;
        lap(3);
        if (converged) break;
        // (s_in / s_conv are rewritten only after the next iteration's barriers; s_mean is read at its top by everybody
        // before thread t < D of wave 0 can write it again — the tree's or the barrier's __syncthreads lie between)
    }
    // votes of my rows (this thread is their only writer)
#pragma unroll
    for (int k = 0; k < MS_CACHED_ROWS; ++k)
        if (k < nr && vote[k]) w.votes[b * 256 + t + k * T] += vote[k];
    if (TIMED && timed) for (int q = 0; q < 5; ++q) atomicAdd(ps.ticks + q, tk[TIMED ? q : 0]);
    if (blockIdx.x == 0) {                                  // one workgroup writes the climb's state back
        if (t < D) w.mean[t] = s_mean[t];
        if (t == 0) {
            w.out[0] += iters;
            if (converged) w.out[1] = 1;
            if (dead) w.out[3] = 1;
        }
    }
}

// (index, votes) of every row touched by the climb (the host sorts the short list); clears the votes.
__global__ void __launch_bounds__(256)
k_ms_collect(MeanShiftWork all)
{
    const MeanShiftWork w = ms_climb(all, blockIdx.y);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n) return;
    const int v = w.votes[i];
    if (v > 0) {
        const int pos = atomicAdd(&w.out[2], 1);
        w.list[2 * pos] = i;
        w.list[2 * pos + 1] = v;
        w.votes[i] = 0;
    }
}

// starts: the batch's seed rows (mapped pinned memory written by the host)
__global__ void __launch_bounds__(64)
k_ms_seed(MeanShiftWork all, MeanShiftActive active, const int* __restrict__ starts)
{
    const int b = active.climb[blockIdx.x];
    const MeanShiftWork w = ms_climb(all, b);
    const int start = starts[b];
    const int j = threadIdx.x;
    if (j < w.d) w.mean[j] = w.data[(size_t)start * w.d + j];        // :58  myMean = data.row(stInd)
    if (j < 4) w.out[j] = 0;
}

// k_ms_collect, run only once the climb has ended (converged or dead end); the head of the list also goes into a staging
// array laid out [position][climb], so that the first k pairs of ALL climbs are one contiguous range: the host fetches
// them with ONE copy per batch (k = the longest head) instead of one per climb
__global__ void __launch_bounds__(256)
k_ms_collect_if_done(MeanShiftWork all, MeanShiftActive active, int* __restrict__ heads, int prefix)
{
    const int b = active.climb[blockIdx.y];
    const MeanShiftWork w = ms_climb(all, b);
    if (!(w.out[1] || w.out[3])) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n) return;
    const int v = w.votes[i];
    if (v > 0) {
        const int pos = atomicAdd(&w.out[2], 1);
        w.list[2 * pos] = i;
        w.list[2 * pos + 1] = v;
        if (pos < prefix) {
            int* h = heads + ((size_t)pos * MS_BATCH + b) * 2;
            h[0] = i;
            h[1] = v;
        }
        w.votes[i] = 0;
    }
}

__global__ void __launch_bounds__(64)
k_ms_publish(MeanShiftWork all, MeanShiftActive active, MeanShiftResultBlock* results)
{
    const int b = active.climb[blockIdx.x];
    const MeanShiftWork w = ms_climb(all, b);
    MeanShiftResultBlock* r = results + b;
    const int j = threadIdx.x;
    if (j < 4) r->out[j] = w.out[j];
    if (j < MS_MAXD) r->mean[j] = j < w.d ? w.mean[j] : 0.0;
}

hipError_t launch_ms_climb(const MeanShiftWork& w, const MeanShiftActive& active, int n_active, const int* starts_dev, double band_sq,
                           double stop_thresh, int iterations, MeanShiftResultBlock* result_dev, int* heads_dev, int list_prefix,
                           int* tickets, hipStream_t s)
{
    if (w.d > MS_MAXD || n_active < 1 || n_active > MS_BATCH) return hipErrorInvalidValue;
    const int groups = std::min(MS_GROUPS, (w.n + 255) / 256);
    if (starts_dev) hipLaunchKernelGGL(k_ms_seed, dim3(n_active), dim3(64), 0, s, w, active, starts_dev);
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(k_ms_iterate, dim3(groups, n_active), dim3(256), 0, s, w, active, groups, band_sq, stop_thresh, tickets);
    }
    hipLaunchKernelGGL(k_ms_collect_if_done, dim3((w.n + 255) / 256, n_active), dim3(256), 0, s, w, active, heads_dev, list_prefix);
    hipLaunchKernelGGL(k_ms_publish, dim3(n_active), dim3(64), 0, s, w, active, result_dev);
    return hipGetLastError();
}

// The climbs active[0..n_active) to their end (or `max_iters` iterations) in one launch, one workgroup per group of the
// definition; then compact / publish as launch_ms_climb does.  ctl: 3 x MS_BATCH ints (gate, arrive, fell_back), cleared
// here.  hipErrorNotSupported: no persistent form for this input (d other than 6 / 10, more than 4 rows per thread).
hipError_t launch_ms_persist(const MeanShiftWork& w, const MeanShiftActive& active, int n_active, double band_sq, double stop_thresh,
                             int max_iters, int* ctl, double* partial2, int* partial_cnt2, MeanShiftResultBlock* result_dev,
                             int* heads_dev, int list_prefix, hipStream_t s, unsigned long long* ticks)
{
    if (w.d > MS_MAXD || n_active < 1 || n_active > MS_BATCH) return hipErrorInvalidValue;
    if (!ms_persist_supported(w.n, w.d)) return hipErrorNotSupported;
    const int groups = std::min(MS_GROUPS, (w.n + 255) / 256);
    hipError_t he = hipMemsetAsync(ctl, 0, sizeof(int) * 3 * MS_BATCH, s);
    if (he != hipSuccess) return he;
    MeanShiftPersist ps{ ctl, ctl + MS_BATCH, partial2, partial_cnt2, ctl + 2 * MS_BATCH, ticks };
    const unsigned long long gate_timeout = 25000000ull;    // 250 ms at 100 MHz
    const dim3 grid(groups, n_active);
    if (w.d == 10 && ticks) hipLaunchKernelGGL((k_ms_persist<10, true>), grid, dim3(256), 0, s, w, active, ps, groups, band_sq, stop_thresh, max_iters, gate_timeout);
    else if (w.d == 10) hipLaunchKernelGGL((k_ms_persist<10, false>), grid, dim3(256), 0, s, w, active, ps, groups, band_sq, stop_thresh, max_iters, gate_timeout);
    else if (ticks) hipLaunchKernelGGL((k_ms_persist<6, true>), grid, dim3(256), 0, s, w, active, ps, groups, band_sq, stop_thresh, max_iters, gate_timeout);
    else hipLaunchKernelGGL((k_ms_persist<6, false>), grid, dim3(256), 0, s, w, active, ps, groups, band_sq, stop_thresh, max_iters, gate_timeout);
    hipLaunchKernelGGL(k_ms_collect_if_done, dim3((w.n + 255) / 256, n_active), dim3(256), 0, s, w, active, heads_dev, list_prefix);
    hipLaunchKernelGGL(k_ms_publish, dim3(n_active), dim3(64), 0, s, w, active, result_dev);
    return hipGetLastError();
}

// the two feature spaces of the path: 10-D (EstablishStablePointSets) and 6-D (MergingStep); rows cached in registers
bool ms_persist_supported(int n, int d) { return (d == 10 || d == 6) && n <= MS_CACHED_ROWS * MS_GROUPS * 256; }

// workgroups of k_ms_persist a compute unit holds (0: the query failed — not cached by the caller)
int ms_persist_occupancy(int d)
{
    int per_cu = 0;
    const hipError_t he = d == 10 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_ms_persist<10, false>, 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_ms_persist<6, false>, 256, 0);
    return he == hipSuccess ? per_cu : 0;
}

hipError_t launch_ms_collect(const MeanShiftWork& w, int climbs, hipStream_t s)
{
    hipLaunchKernelGGL(k_ms_collect, dim3((w.n + 255) / 256, climbs), dim3(256), 0, s, w);
    return hipGetLastError();
}

} // namespace mh
