// expand.hip — alpha-expansion relabel sweep on gfx950.
//
// Replaces GCoptimization::expansion's standard-cycle branch
// (GCoptimization.cpp:1032-1049), oneExpansionIteration (:1278-1289) and
// alpha_expansion (:1212-1274) together with the Energy/Graph/BK max-flow
// stack underneath (energy.h:204-253,324-328; maxflow.cpp:472-604;
// graph.h:478-488).  Energies are int32 like the reference's
// (GCoptimization.h:166-170); totals are accumulated in int64 and checked.
//
// Per move (label alpha, fixed order 0..L-1, :1285-1286):
//   k_move_setup   builds the binary energy's s-t graph on the PERSISTENT
//                  symmetric CSR (only capacities change per move):
//                    t-links  : add_term1(i, E0=cost(i,alpha), E1=cost(i,cur))   (:336-342)
//                               + Potts terms against neighbours already at alpha (:360-362)
//                               + the D part of add_term2 for j<i with different labels (energy.h:220)
//                    n-links  : for active pair i>j: cap(i->j)=w*potts,
//                               cap(j->i)= (l_i==l_j) ? w*potts : 0              (energy.h:221-252)
//                  and turns t-links into excess / sink capacity (Graph::add_tweights keeps
//                  only the difference).
//   k_reduce       dominance reduction (exact): a site whose net source surplus exceeds the total
//                  capacity of its outgoing n-links is on the source side of EVERY minimum cut; one
//                  whose net sink surplus exceeds its incoming capacity can always reach the sink.
//                  Such sites are decided, their n-links are folded into the neighbours' t-links,
//                  and the test cascades to a fixed point.  At 50k sites / 11 labels it settles
//                  70-95 % of the sites of a move before any flow is pushed (the rest — points that
//                  are inliers of both the current and the candidate plane — go to push-relabel).
//   push-relabel   lock-free preflow push (one thread per site, agent-scope atomics on
//                  excess and residual capacities, heights written only by their owner),
//                  interleaved with exact global relabelling (chaotic min-relaxation from
//                  the sink to a fixed point).  Finished when, right after a global
//                  relabel, no site with positive excess can still reach the sink.
//   cut read-out   the sites that cannot reach the sink in the residual graph
//                  (height == n after the final relabel) are the SOURCE side and take
//                  alpha (var 0, :429-433).  This is BK's what_segment rule (free nodes
//                  default to SOURCE, graph.h:478-488) = the unique minimal sink side of
//                  any maximum (pre)flow, so labels are solver-independent (SURVEY A-1).
//   k_delta/apply  accept iff the total energy strictly decreases (:1259,1273),
//                  decided on the device from the int64 energy difference.
// A cycle ends with k_energy; the loop stops when the energy is unchanged (:1045).
//
// Roofline: irregular, latency/atomic bound (no dense tile anywhere); reported as
// moves/s and launches per move, not as a bandwidth fraction.

#include "mh_kernels.hpp"

namespace mh {

enum { F_ACTIVE = 0, F_CHANGED = 1, F_EXCESS_NODES = 2, F_ACCEPTED = 3, F_OVERFLOW = 4, F_UNDECIDED_EXCESS = 5,
       F_SCRATCH = 7 /* write-only sink for launches whose 'changed' flag is not looked at */, F_COUNT = 8 };
enum { A_DELTA = 0, A_ENERGY = 1, A_EXCESS_SUM = 2, A_COUNT = 4 };

#define LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

__global__ void __launch_bounds__(256)
k_init_labeling(const int* __restrict__ cost, int L, int n, const int* __restrict__ init,
                int* __restrict__ label, int* __restrict__ cur_cost)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = init ? init[i] : 0;
    label[i] = l;
    cur_cost[i] = cost[(size_t)i * L + l];        // updateLabelingDataCosts, :445-451
}

// solveSpecialCases (:470-491): no neighbours at all -> per-site argmin, first minimum wins.
__global__ void __launch_bounds__(256)
k_argmin_labels(const int* __restrict__ cost, int L, int n, int* __restrict__ label,
                long long* __restrict__ acc)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    long long mine = 0;
    if (i < n) {
        int best = 0, bc = cost[(size_t)i * L];
        for (int l = 1; l < L; ++l) {
            const int c = cost[(size_t)i * L + l];
            if (c < bc) { bc = c; best = l; }
        }
        label[i] = best;
        mine = bc;
    }
    __shared__ long long s[256];
    s[threadIdx.x] = mine;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)&acc[A_ENERGY], (unsigned long long)s[0]);
}

// ---------------------------------------------------------------------------
// Site-parallel kernels.  A site's arcs are contiguous in the CSR, so LPN = 16 consecutive lanes
// (one DPP row; four sites per wavefront) scan them together: coalesced 64-B reads of col/cap,
// one or two dependent load latencies per pass instead of one per arc, and width-16 shuffles for
// the reductions.  With one thread per site these kernels took ~40 us each at N = 50k (eighteen
// serialised uncached loads per thread at <1 wave per SIMD); the lane-parallel form is what
// makes a move cost milliseconds.  All lanes of a site keep identical copies of the site's
// scalars (height, excess, sink capacity); lane 0 alone writes.
// ---------------------------------------------------------------------------
constexpr int LPN = 16;
constexpr int SITES_PER_BLOCK = 256 / LPN;
constexpr int SLOTS = 3;              // arcs per lane kept in registers by the relax / push kernels

__device__ __forceinline__ int row_min(int v)
{
#pragma unroll
    for (int m = LPN / 2; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m, LPN); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ long long row_min64(long long v)
{
#pragma unroll
    for (int m = LPN / 2; m >= 1; m >>= 1) { const long long o = __shfl_xor(v, m, LPN); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ long long row_sum64(long long v)
{
#pragma unroll
    for (int m = LPN / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LPN);
    return v;
}

__global__ void __launch_bounds__(256)
k_move_setup(Graph g, const int* __restrict__ cost, int L, int potts, int alpha,
             const int* __restrict__ label, const int* __restrict__ cur_cost,
             int* __restrict__ cap, int* __restrict__ excess, int* __restrict__ sink_cap,
             int* __restrict__ decided, int* __restrict__ flags, long long* __restrict__ acc)
{
    const int i = blockIdx.x * SITES_PER_BLOCK + threadIdx.x / LPN;
    const int sub = threadIdx.x % LPN;
    if (i >= g.n) return;
    const int li = label[i];
    const int k0 = g.rowptr[i], k1 = g.rowptr[i + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[A_DELTA] = 0;     // k_delta of this move accumulates into it later
    if (sub == 0) decided[i] = (li == alpha) ? 3 : 0;
    if (li == alpha) {
        if (sub == 0) { excess[i] = 0; sink_cap[i] = 0; }
        for (int k = k0 + sub; k < k1; k += LPN) cap[k] = 0;
        return;
    }
    long long S = 0;
    for (int k = k0 + sub; k < k1; k += LPN) {
        const int j = g.col[k];
        const int wk = g.w[k] * potts;
        const int lj = label[j];
        if (lj == alpha) { S += wk; cap[k] = 0; }
        else if (j < i) { if (li != lj) S += wk; cap[k] = wk; }
        else { cap[k] = (li == lj) ? wk : 0; }
    }
    S = row_sum64(S) + cur_cost[i];
    if (sub != 0) return;
    const long long K = cost[(size_t)i * L + alpha];
    const long long tr = S - K;
    if (tr > 0x7fffffffll || -tr > 0x7fffffffll) atomicExch(&flags[F_OVERFLOW], 1);
    const int ex = tr > 0 ? (int)tr : 0;
    excess[i] = ex;
    sink_cap[i] = tr < 0 ? (int)(-tr) : 0;
    if (ex > 0) {
        atomicAdd(&flags[F_EXCESS_NODES], 1);
        atomicAdd((unsigned long long*)&acc[A_EXCESS_SUM], (unsigned long long)ex);
    }
}

__global__ void __launch_bounds__(256)
k_bfs_init(int n, const int* __restrict__ decided, const int* __restrict__ sink_cap,
           int* __restrict__ height)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int d = decided[i];
    // decided sites keep their verdict: sink side -> 1 (< n, keeps its label), source side -> n
    height[i] = (d == 2 || (d == 0 && sink_cap[i] > 0)) ? 1 : n;
}

// Dominance reduction to a fixed point (see the header).  net(u) = excess - sink_cap.  An undecided
// site first folds its decided neighbours into its own t-link — a source-side neighbour v delivers
// cap(v->u), a sink-side neighbour absorbs cap(u->v) — and clears both arcs, then tests
//     net >  sum of cap(u->w) over undecided w   ->  source side in every minimum cut
//    -net >  sum of cap(w->u) over undecided w   ->  residual sink capacity survives every max flow
// Strict inequalities: ties stay undecided and go to push-relabel, so the minimal sink side (BK's
// read-out, SURVEY A-1) is untouched.  Arcs between two undecided sites are never written here and
// a decided site never writes again, so rounds may run chaotically: every verdict holds given any
// subset of earlier verdicts (a stale "undecided" view only makes the tests stricter).
__global__ void __launch_bounds__(256)
k_reduce(Graph g, int* cap, int* excess, int* sink_cap, int* decided, int* __restrict__ flags,
         int* __restrict__ changed_flag, int ROUNDS)
{
    const int u = blockIdx.x * SITES_PER_BLOCK + threadIdx.x / LPN;
    const int sub = threadIdx.x % LPN;
    if (u >= g.n) return;
    const int du = LD(&decided[u]);                 // first-level loads issued together
    const int k0 = g.rowptr[u], k1 = g.rowptr[u + 1];
    long long net = (long long)excess[u] - sink_cap[u];
    if (du != 0) return;
    bool dirty = false, changed = false;
    int verdict = 0;
    for (int r = 0; r < ROUNDS; ++r) {
        long long add = 0, out = 0, in = 0;
        for (int k = k0 + sub; k < k1; k += LPN) {
            const int kr = g.rev[k];
            const int co = cap[k], ci = cap[kr];
            if ((co | ci) == 0) continue;
            const int dv = LD(&decided[g.col[k]]);
            if (dv == 1) { add += ci; cap[k] = 0; cap[kr] = 0; }
            else if (dv == 2) { add -= co; cap[k] = 0; cap[kr] = 0; }
            else { out += co; in += ci; }
        }
        add = row_sum64(add); out = row_sum64(out); in = row_sum64(in);
        if (add != 0) { net += add; dirty = true; changed = true; }
        if (net > out) verdict = 1;
        else if (-net > in) verdict = 2;
        if (verdict) break;
    }
    if (sub != 0) return;
    if (dirty) {
        if (net > 0x7fffffffll || -net > 0x7fffffffll) atomicExch(&flags[F_OVERFLOW], 1);
        excess[u] = net > 0 ? (int)net : 0;
        sink_cap[u] = net < 0 ? (int)(-net) : 0;
    }
    if (verdict) { ST(&decided[u], verdict); changed = true; }
    if (changed) *changed_flag = 1;
}

// Undecided sites that still hold excess: zero means the cut is already known.
__global__ void __launch_bounds__(256)
k_count_undecided(int n, const int* __restrict__ excess, const int* __restrict__ decided, int* __restrict__ flags)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = (i < n) && decided[i] == 0 && excess[i] > 0;
    const unsigned long long b = __ballot(act);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&flags[F_UNDECIDED_EXCESS], __popcll(b));

}

// Chaotic min-relaxation towards exact residual distances to the sink.  Values are
// always upper bounds realised by residual paths and only decrease, so any schedule
// converges to the BFS distances; `changed` is raised when a launch lowered anything.
// Residual capacities are constant while relaxation kernels run (plain loads); heights move
// (agent-scope loads/stores, which bypass the per-CU L1).
__global__ void __launch_bounds__(256)
k_bfs_relax(Graph g, const int* __restrict__ decided, const int* __restrict__ cap,
            int* height, int* __restrict__ changed_flag, int ROUNDS)
{
    const int u = blockIdx.x * SITES_PER_BLOCK + threadIdx.x / LPN;
    const int sub = threadIdx.x % LPN;
    if (u >= g.n) return;
    // all first-level loads issued together (the early return must not serialise them)
    const int du = decided[u];
    const int k0 = g.rowptr[u], k1 = g.rowptr[u + 1];
    int hu = LD(&height[u]);
    if (du != 0) return;
    bool any = false;
    if (k1 - k0 <= SLOTS * LPN) {
        // The usual case (degree <= 48): each lane keeps its <= 3 residual arcs' heads in registers
        // (capacities do not change while relaxation runs), so a round is ONE level of independent
        // agent-scope loads instead of a cap -> col -> height chain.
        int nb[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int k = k0 + sub + q * LPN;
            nb[q] = (k < k1 && cap[k] > 0) ? g.col[k] : -1;
        }
        for (int r = 0; r < ROUNDS; ++r) {
            if (hu <= 1) break;
            int best = hu;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                if (nb[q] >= 0) {
                    const int hv = LD(&height[nb[q]]) + 1;
                    if (hv < best) best = hv;
                }
            }
            best = row_min(best);
            if (best < hu) {
                hu = best;
                if (sub == 0) ST(&height[u], hu);
                any = true;
            }
        }
    } else {
        for (int r = 0; r < ROUNDS; ++r) {
            if (hu <= 1) break;
            int best = hu;
            for (int k = k0 + sub; k < k1; k += LPN) {
                if (cap[k] > 0) {
                    const int hv = LD(&height[g.col[k]]) + 1;
                    if (hv < best) best = hv;
                }
            }
            best = row_min(best);
            if (best < hu) {                             // keep polling otherwise: a neighbour may still
                hu = best;                               // drop inside this launch
                if (sub == 0) ST(&height[u], hu);
                any = true;
            }
        }
    }
    if (any && sub == 0) *changed_flag = 1;
}

__global__ void __launch_bounds__(256)
k_count_active(int n, const int* __restrict__ excess, const int* __restrict__ height,
               const int* __restrict__ decided, int* __restrict__ flags)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = (i < n) && decided[i] == 0 && excess[i] > 0 && height[i] < n;
    const unsigned long long b = __ballot(act);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&flags[F_ACTIVE], __popcll(b));
}

// Lock-free push-relabel (Hong's formulation): the owner (lane 0 of the site's row) is the only
// one that lowers excess[u], lowers cap[u->*], touches sink_cap[u] or writes height[u]; other
// sites only ADD to excess[u] and to cap[u->*] (reverse arcs of their pushes).
__global__ void __launch_bounds__(256)
k_push_relabel(Graph g, const int* __restrict__ decided, int* cap, int* excess,
               int* __restrict__ sink_cap, int* height, int CYCLES)
{
    const int u = blockIdx.x * SITES_PER_BLOCK + threadIdx.x / LPN;
    const int sub = threadIdx.x % LPN;
    if (u >= g.n) return;
    const int n = g.n;
    const int du = decided[u];                      // first-level loads issued together
    const int k0 = g.rowptr[u], k1 = g.rowptr[u + 1];
    int hu = LD(&height[u]);
    int sc = sink_cap[u];
    if (du != 0) return;
    if ((k1 - k0) <= SLOTS * LPN) {
        // The usual case (degree <= 48).  Each lane keeps its <= 3 arcs (head, reverse arc) in registers
        // and, while the site is active, fetches the next cycle's excess, capacities and neighbour
        // heights TOGETHER at the end of the current cycle: one memory round trip per active cycle
        // instead of three (excess -> capacity/height -> capacity again).  The lane that owns the
        // chosen arc issues the capacity updates itself; lane 0 remains the only one that lowers
        // excess[u], so its next load is ordered behind its own atomic.  Values read one cycle early
        // are only ever conservative: cap(u->*) is lowered by this site alone, heights only guide.
        int nb[SLOTS], rv[SLOTS], cq[SLOTS], hq[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int k = k0 + sub + q * LPN;
            nb[q] = k < k1 ? g.col[k] : -1;
            rv[q] = k < k1 ? g.rev[k] : 0;
            cq[q] = 0; hq[q] = 0;
        }
        bool arcs_valid = false;
        int e_next = (sub == 0) ? LD(&excess[u]) : 0;
        for (int cyc = 0; cyc < CYCLES; ++cyc) {
            if (hu >= n) break;
            int e = __shfl(e_next, 0, LPN);
            if (e > 0 && sc > 0) {                      // t-link: h(t) = 0, h(u) = 1
                const int d = e < sc ? e : sc;
                sc -= d;
                if (sub == 0) { sink_cap[u] = sc; atomicSub(&excess[u], d); }
                e -= d;
            }
            if (e <= 0) {                               // idle (or drained into the sink): just poll
                e_next = (sub == 0) ? LD(&excess[u]) : 0;
                arcs_valid = false;
                continue;
            }
            if (!arcs_valid) {
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) {
                    cq[q] = nb[q] >= 0 ? LD(&cap[k0 + sub + q * LPN]) : 0;
                    hq[q] = nb[q] >= 0 ? LD(&height[nb[q]]) : 0;
                }
            }
            long long key = 0x7fffffffffffffffll;       // (height << 32) | arc
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                if (cq[q] > 0) {
                    const long long cand = ((long long)hq[q] << 32) | (unsigned int)(k0 + sub + q * LPN);
                    if (cand < key) key = cand;
                }
            }
            key = row_min64(key);
            if (key == 0x7fffffffffffffffll) {          // no way out at all
                hu = n;
                if (sub == 0) ST(&height[u], hu);
                break;
            }
            const int hmin = (int)(key >> 32), kmin = (int)(key & 0xffffffffll);
            if (hu > hmin) {
                const int off = kmin - k0, owner = off % LPN, slot = off / LPN;
                int c_mine = cq[0], v_mine = nb[0], r_mine = rv[0];
#pragma unroll
                for (int q = 1; q < SLOTS; ++q)
                    if (slot == q) { c_mine = cq[q]; v_mine = nb[q]; r_mine = rv[q]; }
                const int c = __shfl(c_mine, owner, LPN);
                const int d = e < c ? e : c;
                if (sub == owner) {
                    atomicSub(&cap[kmin], d);
                    atomicAdd(&cap[r_mine], d);
                    atomicAdd(&excess[v_mine], d);
                }
                if (sub == 0) atomicSub(&excess[u], d);
            } else {
                hu = hmin + 1;
                if (hu > n) hu = n;
                if (sub == 0) ST(&height[u], hu);
            }
            // next cycle's inputs, all in flight together
            e_next = (sub == 0) ? LD(&excess[u]) : 0;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                cq[q] = nb[q] >= 0 ? LD(&cap[k0 + sub + q * LPN]) : 0;
                hq[q] = nb[q] >= 0 ? LD(&height[nb[q]]) : 0;
            }
            arcs_valid = true;
        }
        return;
    }
    for (int cyc = 0; cyc < CYCLES; ++cyc) {            // generic path: any degree
        if (hu >= n) break;
        // lane 0 reads the excess and broadcasts it: all lanes of the row act on ONE value
        int e = (sub == 0) ? LD(&excess[u]) : 0;
        e = __shfl(e, 0, LPN);
        if (e <= 0) continue;
        if (sc > 0) {                               // t-link: h(t) = 0, h(u) = 1
            const int d = e < sc ? e : sc;
            sc -= d;
            if (sub == 0) { sink_cap[u] = sc; atomicSub(&excess[u], d); }
            e -= d;
            if (e == 0) continue;
        }
        long long key = 0x7fffffffffffffffll;       // (height << 32) | arc
        for (int k = k0 + sub; k < k1; k += LPN) {
            if (LD(&cap[k]) > 0) {
                const long long cand = ((long long)LD(&height[g.col[k]]) << 32) | (unsigned int)k;
                if (cand < key) key = cand;
            }
        }
        key = row_min64(key);
        if (key == 0x7fffffffffffffffll) {          // no way out at all
            hu = n;
            if (sub == 0) ST(&height[u], hu);
            break;
        }
        const int hmin = (int)(key >> 32), kmin = (int)(key & 0xffffffffll);
        if (hu > hmin) {
            if (sub == 0) {
                const int c = LD(&cap[kmin]);
                const int d = e < c ? e : c;
                atomicSub(&cap[kmin], d);
                atomicAdd(&cap[g.rev[kmin]], d);
                atomicSub(&excess[u], d);
                atomicAdd(&excess[g.col[kmin]], d);
            }
        } else {
            hu = hmin + 1;
            if (hu > n) hu = n;
            if (sub == 0) ST(&height[u], hu);
        }
    }
}

// Energy difference of the candidate labeling (sites with height == n take alpha).
__global__ void __launch_bounds__(256)
k_delta(Graph g, const int* __restrict__ cost, int L, int potts, int alpha,
        const int* __restrict__ label, const int* __restrict__ cur_cost,
        const int* __restrict__ height, long long* __restrict__ acc)
{
    const int i = blockIdx.x * SITES_PER_BLOCK + threadIdx.x / LPN;
    const int sub = threadIdx.x % LPN;
    long long mine = 0;
    if (i < g.n) {
        const int oi = label[i];
        const int ni = (oi != alpha && height[i] >= g.n) ? alpha : oi;
        if (sub == 0 && ni != oi) mine += (long long)cost[(size_t)i * L + alpha] - cur_cost[i];
        for (int k = g.rowptr[i] + sub; k < g.rowptr[i + 1]; k += LPN) {
            const int j = g.col[k];
            if (j < i) {
                const int oj = label[j];
                const int nj = (oj != alpha && height[j] >= g.n) ? alpha : oj;
                const int dn = (ni != nj) - (oi != oj);
                mine += (long long)dn * g.w[k] * potts;
            }
        }
    }
    __shared__ long long s[256];
    s[threadIdx.x] = mine;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && s[0] != 0)
        atomicAdd((unsigned long long*)&acc[A_DELTA], (unsigned long long)s[0]);
}

__global__ void __launch_bounds__(256)
k_apply(int n, const int* __restrict__ cost, int L, int alpha, int* __restrict__ label,
        int* __restrict__ cur_cost, const int* __restrict__ height,
        const long long* __restrict__ acc, int* __restrict__ flags)
{
    if (acc[A_DELTA] >= 0) return;                   // strict decrease only (:1259)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) flags[F_ACCEPTED] += 1;
    if (i >= n) return;
    if (label[i] != alpha && height[i] >= n) {       // applyNewLabeling, :423-441
        label[i] = alpha;
        cur_cost[i] = cost[(size_t)i * L + alpha];
    }
}

// compute_energy = data + smooth (:953-956; giveSmoothEnergyInternal :267-286)
__global__ void __launch_bounds__(256)
k_energy(Graph g, int potts, const int* __restrict__ label, const int* __restrict__ cur_cost,
         long long* __restrict__ acc)
{
    const int i = blockIdx.x * SITES_PER_BLOCK + threadIdx.x / LPN;
    const int sub = threadIdx.x % LPN;
    long long mine = 0;
    if (i < g.n) {
        if (sub == 0) mine = cur_cost[i];
        const int li = label[i];
        for (int k = g.rowptr[i] + sub; k < g.rowptr[i + 1]; k += LPN) {
            const int j = g.col[k];
            if (j < i && label[j] != li) mine += (long long)g.w[k] * potts;
        }
    }
    __shared__ long long s[256];
    s[threadIdx.x] = mine;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)&acc[A_ENERGY], (unsigned long long)s[0]);
}

hipError_t launch_init_labeling(const int* cost, int L, int n, const int* init, int* label,
                                int* cur_cost, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_labeling, dim3((n + 255) / 256), dim3(256), 0, s, cost, L, n, init,
                       label, cur_cost);
    return hipGetLastError();
}

hipError_t launch_argmin_labels(const int* cost, int L, int n, int* label, long long* acc,
                                hipStream_t s)
{
    hipLaunchKernelGGL(k_argmin_labels, dim3((n + 255) / 256), dim3(256), 0, s, cost, L, n, label,
                       acc);
    return hipGetLastError();
}

#define RET_IF(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)


// Control words travel to the host through device-mapped pinned memory: one 1-wave kernel copies
// them (no hipMemcpy calls, which cost tens of microseconds each for 32 bytes), then the host
// waits for the stream.
__global__ void k_publish(int* __restrict__ flags, long long* __restrict__ acc,
                          int* __restrict__ h_flags, long long* __restrict__ h_acc)
{
    const int t = threadIdx.x;
    if (t < F_COUNT) h_flags[t] = flags[t];
    if (t < A_COUNT) h_acc[t] = acc[t];
    // the per-check words start the next batch from zero (saves a 4-byte memset launch per word)
    if (t == F_ACTIVE || t == F_CHANGED || t == F_EXCESS_NODES || t == F_UNDECIDED_EXCESS) flags[t] = 0;
    if (t == A_EXCESS_SUM) acc[t] = 0;
}

static hipError_t fetch(ExpandWork& w, hipStream_t s)
{
    ++w.host_syncs;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, w.flags, w.acc, w.h_flags_dev, w.h_acc_dev);
    RET_IF(hipGetLastError());
    return hipStreamSynchronize(s);
}

static hipError_t total_energy(const Graph& g, int potts, ExpandWork& w, long long* out,
                               hipStream_t s)
{
    const dim3 grid((g.n + SITES_PER_BLOCK - 1) / SITES_PER_BLOCK), blk(256);
    RET_IF(hipMemsetAsync(&w.acc[A_ENERGY], 0, sizeof(long long), s));
    hipLaunchKernelGGL(k_energy, grid, blk, 0, s, g, potts, w.label, w.cur_cost, w.acc);
    RET_IF(hipGetLastError());
    RET_IF(fetch(w, s));
    *out = w.h_acc[A_ENERGY];
    return hipSuccess;
}

hipError_t run_expansion(const Graph& g, const int* cost, int L, int potts, ExpandWork& w,
                         int max_cycles, ExpandStats* st, hipStream_t s)
{
    const dim3 grid1((g.n + 255) / 256), blk(256);                          // one thread per site
    const dim3 grid((g.n + SITES_PER_BLOCK - 1) / SITES_PER_BLOCK);        // LPN lanes per site
    ExpandStats stats = {};
    w.host_syncs = 0;
    RET_IF(hipMemsetAsync(w.flags, 0, sizeof(int) * F_COUNT, s));
    RET_IF(hipMemsetAsync(w.acc, 0, sizeof(long long) * A_COUNT, s));

    if (g.nnz == 0) {                                  // solveSpecialCases, :470-491
        RET_IF(launch_argmin_labels(cost, L, g.n, w.label, w.acc, s));
        RET_IF(fetch(w, s));
        stats.energy = w.h_acc[A_ENERGY];
        if (st) *st = stats;
        return hipSuccess;
    }

    long long energy = 0, old_energy = 0;
    RET_IF(total_energy(g, potts, w, &energy, s));     // :1036
    int bfs_need = w.bfs_batch;                        // relax launches before the first check of a relabel

    for (int cycle = 1; cycle <= max_cycles; ++cycle) {
        old_energy = energy;
        for (int alpha = 0; alpha < L; ++alpha) {
            ++stats.moves;
            // (ACTIVE, CHANGED, EXCESS_NODES, EXCESS_SUM were zeroed by the last k_publish; DELTA by k_move_setup)
            hipLaunchKernelGGL(k_move_setup, grid, blk, 0, s, g, cost, L, potts, alpha, w.label,
                               w.cur_cost, w.cap, w.excess, w.sink_cap, w.decided, w.flags, w.acc);
            RET_IF(hipGetLastError());
            // The first host check of the move looks at the set-up counters (overflow, number of sites
            // with excess) and, in the same round trip, at the first batch of the dominance reduction.
            bool flow_needed = true, first_check = true, skip_move = false;
            for (;;) {
                if (w.reduce_rounds > 0) {
                    // four launches per host check; only the last one's flag decides (a launch that
                    // changed nothing is the fixed point)
                    for (int b = 0; b < 4; ++b)
                        hipLaunchKernelGGL(k_reduce, grid, blk, 0, s, g, w.cap, w.excess, w.sink_cap, w.decided,
                                           w.flags, &w.flags[b == 3 ? F_CHANGED : F_SCRATCH], w.reduce_rounds);
                    stats.reduce_launches += 4;
                    hipLaunchKernelGGL(k_count_undecided, grid1, blk, 0, s, g.n, w.excess, w.decided, w.flags);
                    RET_IF(hipGetLastError());
                }
                RET_IF(fetch(w, s));
                if (w.h_flags[F_OVERFLOW] || (first_check && w.h_acc[A_EXCESS_SUM] > 0x7fffffffll)) {
                    if (st) { stats.energy = -1; *st = stats; }
                    return hipErrorInvalidValue;       // int32 energy terms would overflow
                }
                // No excess anywhere: max-flow is 0, after == before, the move is rejected (:1259).
                if (first_check && w.h_flags[F_EXCESS_NODES] == 0) { skip_move = true; break; }
                first_check = false;
                if (w.reduce_rounds <= 0) break;
                // no undecided site holds excess: nothing can flow any more, the residual graph is final
                flow_needed = w.h_flags[F_UNDECIDED_EXCESS] != 0;
                if (!w.h_flags[F_CHANGED]) break;
            }
            if (skip_move) continue;
            if (flow_needed) ++stats.flow_moves;

            for (int round = 0; round < 100000; ++round) {
                // exact global relabel
                hipLaunchKernelGGL(k_bfs_init, grid1, blk, 0, s, g.n, w.decided, w.sink_cap,
                                   w.height);
                // Batches of relaxation launches; the flag of the LAST launch of a batch decides
                // (a launch that lowered nothing is a fixed point).  The active count is taken
                // in the same batch: it is only trusted when that last launch changed nothing.
                // The first batch is as long as the previous relabel needed (relabels of one move, and of
                // neighbouring moves, converge in similar numbers of launches): usually one host check.
                int launched = 0;
                for (int batch = bfs_need;; batch = w.bfs_batch) {
                    for (int b = 0; b < batch; ++b) {
                        hipLaunchKernelGGL(k_bfs_relax, grid, blk, 0, s, g, w.decided, w.cap, w.height,
                                           &w.flags[b == batch - 1 ? F_CHANGED : F_SCRATCH], w.bfs_rounds);
                        ++stats.bfs_launches;
                    }
                    launched += batch;
                    hipLaunchKernelGGL(k_count_active, grid1, blk, 0, s, g.n, w.excess, w.height, w.decided, w.flags);
                    RET_IF(hipGetLastError());
                    RET_IF(fetch(w, s));
                    if (!w.h_flags[F_CHANGED]) break;
                }
                bfs_need = launched > bfs_need ? launched : bfs_need - 1;
                if (bfs_need < w.bfs_batch) bfs_need = w.bfs_batch;
                if (bfs_need > 12) bfs_need = 12;
                if (w.h_flags[F_ACTIVE] == 0) break;
                for (int b = 0; b < w.pr_batch; ++b) {
                    hipLaunchKernelGGL(k_push_relabel, grid, blk, 0, s, g, w.decided,
                                       w.cap, w.excess, w.sink_cap, w.height, w.pr_cycles);
                    ++stats.pr_launches;
                }
                RET_IF(hipGetLastError());
            }
            hipLaunchKernelGGL(k_delta, grid, blk, 0, s, g, cost, L, potts, alpha, w.label,
                               w.cur_cost, w.height, w.acc);
            hipLaunchKernelGGL(k_apply, grid1, blk, 0, s, g.n, cost, L, alpha, w.label, w.cur_cost,
                               w.height, w.acc, w.flags);
            RET_IF(hipGetLastError());
        }
        RET_IF(total_energy(g, potts, w, &energy, s));
        stats.cycles = cycle;
        if (energy == old_energy) break;               // :1045
    }
    stats.energy = energy;
    stats.host_syncs = w.host_syncs;
    stats.accepted = w.h_flags[F_ACCEPTED];
    if (st) *st = stats;
    return hipSuccess;
}

} // namespace mh
