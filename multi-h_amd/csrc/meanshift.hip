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

// ---- r05: a workgroup per climb, the rows through an index on one coordinate ------------------------------------------
// What a climb costs is not its arithmetic but its trips: a launch (or a barrier across 64 workgroups) per iteration, and
// every iteration reading ALL n rows although a ball holds a handful to a few thousand.  k_ms_indexed gives a climb ONE
// workgroup of 1 024 threads that runs it from its seed to its end, collects its votes and publishes its result: a batch
// is one launch and one host synchronisation.
//  * Candidates.  The rows are binned once per call on ONE coordinate c (the host picks the widest) in cells of width
//    w = bandWidth^2 * (1 + 2^-20) — counting sort, component-major copy `rs` in cell order, `order` = position -> row.
//    A member has sum_j |mean_j - row_j| < bandWidth^2 in the definition's rounded arithmetic; every term is non-negative
//    and rounding is monotone, so |mean_c - row_c| < bandWidth^2 (1 + 2^-53) in exact terms, and the row lies in the
//    mean's cell or a neighbour (msx_cell is monotone, and two values less than w apart land at most one cell apart: the
//    2^-20 of slack dwarfs the three roundings of msx_cell, 65 536 cells at most).  Only those three cells are swept.
//  * Membership is decided by the definition's own expression, operation for operation (k_ms_iterate); a candidate is
//    dropped after four components when all of them are `plain` and their partial sum has reached bandWidth^2 — the
//    remaining terms are non-negative (or the sum ends as NaN / infinity): it could not come back.
//  * The sums keep the definition's shape — global thread g = row mod 16 384 adds its rows in ascending order, a binary
//    tree over the 256 threads of a group, the 64 groups in sequence — evaluated where there are members: the sweep marks
//    them in a bitmap over the ROWS (LDS), a wave takes one non-empty group at a time, each lane gathers the (at most
//    4 x ceil(n / 16 384)) rows of its four slots that are marked, adds them in ascending order, and the tree runs as in
//    k_ms_persist (levels 128 and 64 inside the lane, 32 .. 1 by lane shuffles).  An empty slot is +0 and x + 0 = x for
//    every x the sums can hold (never -0: they start at +0), an empty group is skipped for the same reason.
//  * The final compaction scans only the span of positions the climb's windows covered.
//  * A DENSE climb (more than `dense_limit` members in an iteration) is slow here — 64 group sums on ONE compute unit,
//    about 29 us an iteration against 10 with the 64 workgroups of k_ms_persist — but that kernel holds ten climbs at a
//    time.  So the batch drains here until the climbs still running (a counter every climb leaves through) are at most
//    `keep`; then the dense ones among them leave, not ended, and the host hands them on (the state — mean, iteration
//    count, votes by row — is what k_ms_persist expects).  keep = 0: every climb ends here.
constexpr int MSX_MAX_ROWS = 131072;          // bitmap: 16 KiB of LDS; at most 8 rows per slot, 64 words of marks per group (one per lane)
constexpr int MSX_MAX_CELLS = 65536;
constexpr int MSX_THREADS = 1024;

__device__ __forceinline__ int msx_cell(double x, double lo, double inv_w, int cells)
{
    const double v = (x - lo) * inv_w;
    if (!(v >= 0.0)) return 0;                              // below the first cell, or NaN (a NaN coordinate is never a member)
    if (v >= (double)(cells - 1)) return cells - 1;
    return (int)v;                                          // floor, v >= 0
}

__global__ void __launch_bounds__(256)
k_msx_count(const double* __restrict__ data, int n, int d, MeanShiftIndex ix, int* __restrict__ count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&count[msx_cell(data[(size_t)i * d + ix.coord], ix.lo, ix.inv_w, ix.cells)], 1);
}

// start[c] = rows in cells below c (start[cells] = n); cursor = a copy for the scatter.  One workgroup.
__global__ void __launch_bounds__(1024)
k_msx_scan(const int* __restrict__ count, int cells, int* __restrict__ start, int* __restrict__ cursor)
{
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int per = (cells + 1023) / 1024;
    const int c0 = t * per, c1 = min(cells, c0 + per);
    int sum = 0;
    for (int c = c0; c < c1; ++c) sum += count[c];
    s_part[t] = sum;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {                    // inclusive scan of the 1 024 chunk sums
        const int v = t >= s ? s_part[t - s] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;
    for (int c = c0; c < c1; ++c) { start[c] = run; cursor[c] = run; run += count[c]; }
    if (t == 1023) start[cells] = s_part[1023];
}

__global__ void __launch_bounds__(256)
k_msx_scatter(const double* __restrict__ data, int n, int d, MeanShiftIndex ix, int* __restrict__ cursor,
              int* __restrict__ order, double* __restrict__ rs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* row = data + (size_t)i * d;
    const int p = atomicAdd(&cursor[msx_cell(row[ix.coord], ix.lo, ix.inv_w, ix.cells)], 1);    // (the order inside a cell is immaterial)
    order[p] = i;
    for (int j = 0; j < d; ++j) rs[(size_t)j * n + p] = row[j];
}

// The sum of one group (or of its even / odd half) over its `cnt` <= 64 members, in the shape of the definition's tree.
// ml[base + l] = word << 5 | bit of the l-th mark in ascending order, i.e. ascending beta = rev8(t) * K + k: the slots in
// BIT-REVERSED order, a slot's rows in ascending order.  The definition's tree over the 256 slots pairs t with t + 128
// first and even with odd slots last; over bit-reversed slot numbers that is a binary trie merged from its deepest level
// up, in which the two children of a node are NEIGHBOURS among the elements still alive.  A child that is not there is
// +0, and x + 0 = x (x is never -0: every slot sum starts as 0.0 + row): a lone child moves up unchanged.  Lane l holds
// member l; the result is lane 0's.
template <int D>
__device__ __forceinline__ void sparse_tree(const double* __restrict__ data, const unsigned short* ml, int base, int cnt, int g, int K,
                                            int lane, double (&v)[D])
{
    constexpr int T = MS_GROUPS * 256;
    bool alive = lane < cnt;
    int node = 0x100 + lane;                                // (dead lanes: distinct keys above every slot)
    int slot_key = -1 - lane;
    double x[D];
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = 0.0;
    if (alive) {
        const int e = ml[base + lane];
        const int beta = e;                                 // (word << 5 | bit) IS beta: marks are stored at bit beta of the group
        const int rt = beta / K, k = beta - rt * K;
        const int t = (int)(__builtin_bitreverse32((unsigned)rt) >> 24);
        const double* row = data + (size_t)(k * T + g * 256 + t) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = row[j];
        node = rt;
        slot_key = rt;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) v[j] = 0.0 + x[j];          // the thread's accumulator of the definition starts at +0
    // a slot's rows (neighbours, ascending k) into its first lane, one after the other
    {
        const int prev = __shfl_up(slot_key, 1, 64);
        const bool follower = alive && lane > 0 && prev == slot_key;
        if (__ballot(follower) != 0ull) {
            const bool head = alive && !follower;
            for (int d = 1; d < K; ++d) {
                const int os = __shfl_down(slot_key, d, 64);
                const bool take = head && lane + d < 64 && os == slot_key;
#pragma unroll
                for (int j = 0; j < D; ++j) { const double o = __shfl_down(x[j], d, 64); if (take) v[j] = v[j] + o; }
            }
            alive = head;
        }
    }
#pragma unroll 1
    for (int L = 0; L < 8; ++L) {
        const unsigned long long mask = __ballot(alive);
        const unsigned long long above = lane < 63 ? (mask >> (lane + 1)) << (lane + 1) : 0ull;
        const unsigned long long below = mask & ((1ull << lane) - 1ull);
        const int nx = above ? (int)__builtin_ctzll(above) : 64;
        const int pv = below ? 63 - (int)__builtin_clzll(below) : -1;
        const int nk = __shfl(node, nx & 63, 64), pk = __shfl(node, pv & 63, 64);
        const bool left = alive && nx < 64 && (nk >> (L + 1)) == (node >> (L + 1));       // my sibling is the next alive lane
        const bool right = alive && pv >= 0 && (pk >> (L + 1)) == (node >> (L + 1));      // I am the sibling of the previous one
        if (__ballot(left) != 0ull) {
#pragma unroll
            for (int j = 0; j < D; ++j) { const double o = __shfl(v[j], nx & 63, 64); if (left) v[j] = v[j] + o; }
        }
        alive = alive && !right;
    }
}

template <int D, bool TIMED>
__global__ void __launch_bounds__(MSX_THREADS)
k_ms_indexed(MeanShiftWork all, MeanShiftActive active, const int* __restrict__ starts, MeanShiftIndex ix, double band_sq,
             double stop_thresh, int max_iters, int dense_limit, int keep, int* running, MeanShiftResultBlock* results,
             unsigned long long* ticks)
{
    const int climb = active.climb[blockIdx.x];
    const MeanShiftWork w = ms_climb(all, climb);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = w.n;
    constexpr int T = MS_GROUPS * 256;
    const int G = min(MS_GROUPS, (n + 255) / 256);
    const int K = (n + T - 1) / T;                          // rows a slot can hold
    __shared__ unsigned s_bits[MSX_MAX_ROWS / 32];          // members of the running iteration: [group][rev8(slot) * K + k]
    __shared__ unsigned short s_mlist[MSX_THREADS / 64][128];   // per wave: the marks of the group in hand, in order
    __shared__ double s_gsum[MS_GROUPS][D];
    __shared__ double s_mean[D], s_move[D];
    __shared__ unsigned s_gmask[2];
    __shared__ int s_in, s_len, s_run;

    const int WPG = 8 * K;                                  // words of marks per group: 256 slots x K rows
    for (int q = t; q < MS_GROUPS * WPG; q += MSX_THREADS) s_bits[q] = 0;
    if (t < D) s_mean[t] = starts ? w.data[(size_t)starts[climb] * D + t] : w.mean[t];     // :58  myMean = data.row(stInd)
    if (t == 0) { s_gmask[0] = 0; s_gmask[1] = 0; s_in = 0; s_len = 0; }
    __syncthreads();

    int iters = 0, converged = 0, dead = 0;
    int pmin = starts ? n : 0, pmax = starts ? 0 : n;       // (a continued climb has votes from windows this launch did not see)
    for (int it = 0; it < max_iters; ++it) {
        const unsigned long long tkA = TIMED ? __builtin_amdgcn_s_memrealtime() : 0ull;
        double mean[D];
#pragma unroll
        for (int j = 0; j < D; ++j) mean[j] = s_mean[j];
        const int c = msx_cell(s_mean[ix.coord], ix.lo, ix.inv_w, ix.cells);
        const int p0 = ix.cell_start[max(c - 1, 0)], p1 = ix.cell_start[min(c + 1, ix.cells - 1) + 1];
        pmin = min(pmin, p0);
        pmax = max(pmax, p1);
        int cnt = 0;
        unsigned gm0 = 0, gm1 = 0;
        // the membership test of k_ms_iterate: |r| for sqrt(r * r) where that is exact (`plain`).  Four candidates per thread
        // and pass: their first components are all in flight before any is looked at (a thread's candidates one after the
        // other cost a trip to the L2 each).
        constexpr int HEAD = D < 4 ? D : 4, U = 4;
        for (int pb = p0 + t; pb < p1; pb += U * MSX_THREADS) {
            double xh[U][HEAD];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = pb + u * MSX_THREADS;
#pragma unroll
                for (int j = 0; j < HEAD; ++j) xh[u][j] = p < p1 ? ix.rs[(size_t)j * n + p] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = pb + u * MSX_THREADS;
                if (p >= p1) continue;
                double dist = 0.0;
                bool plain = true;
#pragma unroll
                for (int j = 0; j < HEAD; ++j) {
                    const double a = fabs(mean[j] - xh[u][j]);
                    plain = plain && (a <= 0x1p500) && (a >= 0x1p-500 || a == 0.0);
                    dist += a;
                }
                if (plain && dist >= band_sq) continue;     // cannot come back below
                double x[D];
#pragma unroll
                for (int j = 0; j < HEAD; ++j) x[j] = xh[u][j];
#pragma unroll
                for (int j = HEAD; j < D; ++j) x[j] = ix.rs[(size_t)j * n + p];
#pragma unroll
                for (int j = HEAD; j < D; ++j) {
                    const double a = fabs(mean[j] - x[j]);
                    plain = plain && (a <= 0x1p500) && (a >= 0x1p-500 || a == 0.0);
                    dist += a;
                }
                if (__builtin_expect(!plain, 0)) {
                    asm volatile("; mean shift: sqrt path");
                    dist = 0.0;
#pragma unroll
                    for (int j = 0; j < D; ++j) { const double q = mean[j] - x[j]; dist += sqrt(q * q); }
                }
                if (dist < band_sq) {                                                                // :85
                    const int i = ix.order[p];
                    const int g = (i & (T - 1)) >> 8;
                    const int beta = (int)(__builtin_bitreverse32((unsigned)(i & 255)) >> 24) * K + i / T;   // place in the group's sparse tree
                    atomicOr(&s_bits[g * WPG + (beta >> 5)], 1u << (beta & 31));
                    atomicAdd(&w.votes[i], 1);
                    ++cnt;
                    if (g < 32) gm0 |= 1u << g; else gm1 |= 1u << (g - 32);
                }
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {                 // (integers: any order)
            cnt += __shfl_xor(cnt, m, 64);
            gm0 |= (unsigned)__shfl_xor((int)gm0, m, 64);
            gm1 |= (unsigned)__shfl_xor((int)gm1, m, 64);
        }
        if (lane == 0 && cnt) { atomicAdd(&s_in, cnt); atomicOr(&s_gmask[0], gm0); atomicOr(&s_gmask[1], gm1); }
        __syncthreads();
        const int in = s_in;
        const unsigned long long gmask = (unsigned long long)s_gmask[0] | ((unsigned long long)s_gmask[1] << 32);
        if (in == 0) { dead = 1; break; }                   // the reference would spin on a NaN mean
        const unsigned long long tkB = TIMED ? __builtin_amdgcn_s_memrealtime() : 0ull;

        // ---- the sums of the non-empty groups, one wave per group ----
        for (int g = wave; g < G; g += MSX_THREADS / 64) {
            if (!((gmask >> g) & 1ull)) continue;           // wave-uniform
            // the group's marks, one word per lane, in the order of the sparse tree: bit (rev8(t) * K + k)
            const unsigned wd = lane < WPG ? s_bits[g * WPG + lane] : 0u;
            if (lane < WPG) s_bits[g * WPG + lane] = 0u;
            const int pc = __builtin_popcount(wd);
            int incl = pc;
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) { const int o = __shfl_up(incl, sft, 64); if (lane >= sft) incl += o; }
            const int m = __shfl(incl, 63, 64);             // members of the group
            const int m_left = __shfl(incl - pc, WPG / 2, 64);      // ... with rev8(t) < 128 (the even slots: the root's left child)
            double v[D];
            if (m <= 64 || (m_left <= 64 && m - m_left <= 64)) {
                // the members' marks in order -> a list; lane l takes the l-th
                unsigned short* ml = s_mlist[wave];
                {
                    unsigned rem = wd;
                    int pos = incl - pc;
                    while (rem) { const int bit = __builtin_ctz(rem); rem &= rem - 1u; ml[pos++] = (unsigned short)((lane << 5) | bit); }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (m <= 64) {
                    sparse_tree<D>(w.data, ml, 0, m, g, K, lane, v);
                } else {
                    double vr[D];
                    sparse_tree<D>(w.data, ml, 0, m_left, g, K, lane, v);
                    sparse_tree<D>(w.data, ml, m_left, m - m_left, g, K, lane, vr);
#pragma unroll
                    for (int j = 0; j < D; ++j) v[j] = v[j] + vr[j];                // the root: even slots + odd slots (both non-empty here)
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                // more members than a wave holds one (or two halves) of: the dense walk — lane l sums its four slots
                // l, l + 64, l + 128, l + 192 row by row, then the tree as in k_ms_persist
                double a01[2][D];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    double lo_v[D], hi_v[D];                // slots lane + 64 * half and lane + 64 * (half + 2)
#pragma unroll
                    for (int j = 0; j < D; ++j) { lo_v[j] = 0.0; hi_v[j] = 0.0; }
                    const int t_lo = lane + 64 * half, t_hi = t_lo + 128;
                    const int r_lo = (int)(__builtin_bitreverse32((unsigned)t_lo) >> 24), r_hi = (int)(__builtin_bitreverse32((unsigned)t_hi) >> 24);
                    for (int k = 0; k < K; ++k) {           // a slot's rows in ascending order
                        const int b_lo = r_lo * K + k, b_hi = r_hi * K + k;
                        const unsigned w_lo = (unsigned)__shfl((int)wd, b_lo >> 5, 64), w_hi = (unsigned)__shfl((int)wd, b_hi >> 5, 64);
                        if ((w_lo >> (b_lo & 31)) & 1u) {
                            const double* row = w.data + (size_t)(k * T + g * 256 + t_lo) * D;
#pragma unroll
                            for (int j = 0; j < D; ++j) lo_v[j] = lo_v[j] + row[j];
                        }
                        if ((w_hi >> (b_hi & 31)) & 1u) {
                            const double* row = w.data + (size_t)(k * T + g * 256 + t_hi) * D;
#pragma unroll
                            for (int j = 0; j < D; ++j) hi_v[j] = hi_v[j] + row[j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < D; ++j) a01[half][j] = lo_v[j] + hi_v[j];   // level 128: slot t += slot t + 128
                }
#pragma unroll
                for (int j = 0; j < D; ++j) v[j] = a01[0][j] + a01[1][j];           // level 64
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) {
#pragma unroll
                    for (int j = 0; j < D; ++j) v[j] = v[j] + __shfl_down(v[j], sft, 64);   // (lanes >= sft compute values nobody reads)
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < D; ++j) s_gsum[g][j] = v[j];
            }
        }
        __syncthreads();
        const unsigned long long tkC = TIMED ? __builtin_amdgcn_s_memrealtime() : 0ull;
        if (t < D) {
            double s = 0.0;
            for (int g = 0; g < G; ++g)                     // groups in sequence (an empty one is +0, and s is never -0)
                if ((gmask >> g) & 1ull) s = s + s_gsum[g][t];
            const double m = s * (1.0 / (double)in);        // cv::Mat / scalar scales by 1/s (:96)
            const double dd = m - s_mean[t];
            s_move[t] = dd * dd;
            s_mean[t] = m;
        }
        if (t == 0) {
            s_in = 0; s_gmask[0] = 0; s_gmask[1] = 0;
            s_run = in > dense_limit ? __hip_atomic_load(running, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1 << 30);
        }
        __syncthreads();
        double move = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) move = move + s_move[q];
        ++iters;
        if (TIMED && ticks && t == 0) {                     // MULTIH_MS_STATS: where an iteration's time goes, sparse and dense apart
            const unsigned long long tkD = __builtin_amdgcn_s_memrealtime();
            const int o = in > 64 ? 4 : 0;
            atomicAdd(ticks + o, tkB - tkA); atomicAdd(ticks + o + 1, tkC - tkB); atomicAdd(ticks + o + 2, tkD - tkC); atomicAdd(ticks + o + 3, 1ull);
        }
        if (sqrt(move) < stop_thresh) { converged = 1; break; }                         // :98
        // a dense climb's iterations are cheaper with a workgroup per GROUP (k_ms_persist) — once the batch has drained so
        // far that the climbs still running fit that kernel's resident grid (`keep`), it leaves, not ended, to go on there
        if (in > dense_limit && s_run <= keep) break;
    }
    __syncthreads();

    // ---- the climb's list, its votes cleared; only when it has ended ----
    const int ended = converged | dead;
    if (ended && t == 0) atomicSub(running, 1);
    if (ended) {
        for (int p = pmin + t; p < pmax; p += MSX_THREADS) {
            const int i = ix.order[p];
            const int v = __hip_atomic_load(&w.votes[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v > 0) {
                const int pos = atomicAdd(&s_len, 1);
                w.list[2 * pos] = i;
                w.list[2 * pos + 1] = v;
                w.votes[i] = 0;
            }
        }
        __syncthreads();
    }
    const int base_iters = starts ? 0 : w.out[0];
    if (t < D) { w.mean[t] = s_mean[t]; results[climb].mean[t] = s_mean[t]; }
    else if (t < MS_MAXD) results[climb].mean[t] = 0.0;
    if (t == 0) {
        const int o0 = base_iters + iters, o2 = ended ? s_len : 0;
        w.out[0] = o0; w.out[1] = converged; w.out[2] = o2; w.out[3] = dead;
        results[climb].out[0] = o0; results[climb].out[1] = converged; results[climb].out[2] = o2; results[climb].out[3] = dead;
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

// k_ms_collect, run only once the climb has ended (converged or dead end)
__global__ void __launch_bounds__(256)
k_ms_collect_if_done(MeanShiftWork all, MeanShiftActive active)
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
                           double stop_thresh, int iterations, MeanShiftResultBlock* result_dev, int* tickets, hipStream_t s)
{
    if (w.d > MS_MAXD || n_active < 1 || n_active > MS_BATCH) return hipErrorInvalidValue;
    const int groups = std::min(MS_GROUPS, (w.n + 255) / 256);
    if (starts_dev) hipLaunchKernelGGL(k_ms_seed, dim3(n_active), dim3(64), 0, s, w, active, starts_dev);
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(k_ms_iterate, dim3(groups, n_active), dim3(256), 0, s, w, active, groups, band_sq, stop_thresh, tickets);
    }
    hipLaunchKernelGGL(k_ms_collect_if_done, dim3((w.n + 255) / 256, n_active), dim3(256), 0, s, w, active);
    hipLaunchKernelGGL(k_ms_publish, dim3(n_active), dim3(64), 0, s, w, active, result_dev);
    return hipGetLastError();
}

// The climbs active[0..n_active) to their end (or `max_iters` iterations) in one launch, one workgroup per group of the
// definition; then compact / publish as launch_ms_climb does.  ctl: 3 x MS_BATCH ints (gate, arrive, fell_back), cleared
// here.  hipErrorNotSupported: no persistent form for this input (d other than 6 / 10, more than 4 rows per thread).
hipError_t launch_ms_persist(const MeanShiftWork& w, const MeanShiftActive& active, int n_active, double band_sq, double stop_thresh,
                             int max_iters, int* ctl, double* partial2, int* partial_cnt2, MeanShiftResultBlock* result_dev,
                             hipStream_t s, unsigned long long* ticks)
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
    hipLaunchKernelGGL(k_ms_collect_if_done, dim3((w.n + 255) / 256, n_active), dim3(256), 0, s, w, active);
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

bool ms_indexed_supported(int n, int d) { return (d == 10 || d == 6) && n <= MSX_MAX_ROWS; }
int ms_index_max_cells() { return MSX_MAX_CELLS; }

// the index of one call: counting sort of the rows on ix.coord (ix.lo, ix.inv_w, ix.cells set by the caller; count: ix.cells ints,
// cursor: ix.cells ints, cell_start: ix.cells + 1 ints, order: n ints, rs: d x n doubles)
hipError_t launch_ms_index_build(const double* data, int n, int d, const MeanShiftIndex& ix, int* count, int* cursor, int* cell_start,
                                 int* order, double* rs, hipStream_t s)
{
    if (ix.cells < 1 || ix.cells > MSX_MAX_CELLS || ix.coord < 0 || ix.coord >= d) return hipErrorInvalidValue;
    hipError_t he = hipMemsetAsync(count, 0, sizeof(int) * (size_t)ix.cells, s);
    if (he != hipSuccess) return he;
    hipLaunchKernelGGL(k_msx_count, dim3((n + 255) / 256), dim3(256), 0, s, data, n, d, ix, count);
    hipLaunchKernelGGL(k_msx_scan, dim3(1), dim3(1024), 0, s, count, ix.cells, cell_start, cursor);
    hipLaunchKernelGGL(k_msx_scatter, dim3((n + 255) / 256), dim3(256), 0, s, data, n, d, ix, cursor, order, rs);
    return hipGetLastError();
}

// The climbs active[0..n_active) from their seeds (starts_dev; null: continue) to their end or max_iters iterations, their
// lists compacted and their results published: one launch, a workgroup per climb.  Once at most `keep` climbs are still
// running (*running: one int of device memory, set here), those that meet more than dense_limit members in an iteration
// come back running.
hipError_t launch_ms_indexed(const MeanShiftWork& w, const MeanShiftActive& active, int n_active, const int* starts_dev,
                             const MeanShiftIndex& ix, double band_sq, double stop_thresh, int max_iters, int dense_limit,
                             int keep, int* running, MeanShiftResultBlock* result_dev, hipStream_t s, unsigned long long* ticks)
{
    if (n_active < 1 || n_active > MS_BATCH || !running) return hipErrorInvalidValue;
    if (!ms_indexed_supported(w.n, w.d)) return hipErrorNotSupported;
    const hipError_t he = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(running), n_active, 1, s);    // the climbs still running
    if (he != hipSuccess) return he;
    const dim3 grid(n_active), block(MSX_THREADS);
    if (w.d == 10 && ticks) hipLaunchKernelGGL((k_ms_indexed<10, true>), grid, block, 0, s, w, active, starts_dev, ix, band_sq, stop_thresh, max_iters, dense_limit, keep, running, result_dev, ticks);
    else if (w.d == 10) hipLaunchKernelGGL((k_ms_indexed<10, false>), grid, block, 0, s, w, active, starts_dev, ix, band_sq, stop_thresh, max_iters, dense_limit, keep, running, result_dev, ticks);
    else if (ticks) hipLaunchKernelGGL((k_ms_indexed<6, true>), grid, block, 0, s, w, active, starts_dev, ix, band_sq, stop_thresh, max_iters, dense_limit, keep, running, result_dev, ticks);
    else hipLaunchKernelGGL((k_ms_indexed<6, false>), grid, block, 0, s, w, active, starts_dev, ix, band_sq, stop_thresh, max_iters, dense_limit, keep, running, result_dev, ticks);
    return hipGetLastError();
}

// The lists of a batch's climbs, one behind the other: pair k of climb b -> packed[offsets[b] + k].  The host, which knows
// the lengths from the published results, sets the offsets and fetches exactly the pairs there are with one copy (a
// staging array of the lists' HEADS, [position][climb], made that copy 4 MB whenever one climb of the batch was long,
// and every longer list a copy of its own).
__global__ void __launch_bounds__(256)
k_ms_pack(MeanShiftWork all, const int* __restrict__ offsets, const int* __restrict__ lengths, int* __restrict__ packed)
{
    const int b = blockIdx.y;
    const int len = lengths[b];
    const int* list = all.list + (size_t)b * 2 * all.n;
    int* dst = packed + 2 * (size_t)offsets[b];
    for (int k = blockIdx.x * 256 + threadIdx.x; k < 2 * len; k += gridDim.x * 256) dst[k] = list[k];
}

hipError_t launch_ms_pack(const MeanShiftWork& w, int climbs, int longest, const int* offsets_dev, const int* lengths_dev, int* packed,
                          hipStream_t s)
{
    if (climbs < 1 || longest < 1) return hipSuccess;
    const int gx = std::max(1, std::min(64, (2 * longest + 255) / 256));
    hipLaunchKernelGGL(k_ms_pack, dim3(gx, climbs), dim3(256), 0, s, w, offsets_dev, lengths_dev, packed);
    return hipGetLastError();
}

hipError_t launch_ms_collect(const MeanShiftWork& w, int climbs, hipStream_t s)
{
    hipLaunchKernelGGL(k_ms_collect, dim3((w.n + 255) / 256, climbs), dim3(256), 0, s, w);
    return hipGetLastError();
}

} // namespace mh
