// expand.hip — alpha-expansion relabel sweep on gfx950.
//
// Replaces GCoptimization::expansion's standard-cycle branch
// (GCoptimization.cpp:1032-1049), oneExpansionIteration (:1278-1289) and
// alpha_expansion (:1212-1274) together with the Energy/Graph/BK max-flow
// stack underneath (energy.h:204-253,324-328; maxflow.cpp:472-604;
// graph.h:478-488).  Energies are int32 like the reference's
// (GCoptimization.h:166-170); totals are accumulated in int64 and checked.
//
// The whole sweep is driven from the device: the host enqueues FOUR launches per move and looks at
// the control words once per cycle (the energy test of :1045).
//
// Per move (label alpha, fixed order 0..L-1, :1285-1286; t = running move index):
//   k_move_setup   applies the previous accepted move (labels are updated lazily, see below), then
//                  builds the binary energy's s-t graph on the PERSISTENT symmetric CSR (only
//                  capacities change per move):
//                    t-links  : add_term1(i, E0=cost(i,alpha), E1=cost(i,cur))   (:336-342)
//                               + Potts terms against neighbours already at alpha (:360-362)
//                               + the D part of add_term2 for j<i with different labels (energy.h:220)
//                    n-links  : for active pair i>j: cap(i->j)=w*potts,
//                               cap(j->i)= (l_i==l_j) ? w*potts : 0              (energy.h:221-252)
//                  turns t-links into excess / sink capacity (Graph::add_tweights keeps only the
//                  difference) and takes the first dominance verdict from the site's own numbers.
//   k_reduce       dominance reduction (exact; a second launch in front of it is a schedule option): a site whose net source surplus exceeds the total
//                  capacity of its outgoing n-links is on the source side of EVERY minimum cut; one
//                  whose net sink surplus exceeds its incoming capacity can always reach the sink.
//                  Such sites are decided, their n-links are folded into the neighbours' t-links,
//                  and the test cascades.  At 50k sites / 11 labels it settles 70-95 % of the sites
//                  of a move before any flow is pushed.  The second launch appends the sites that
//                  are still undecided to a compact list (the "core": points that are inliers of
//                  both the current and the candidate plane, a few thousand sites).
//   k_solve        ONE persistent launch over the core only (8 lanes per site, arcs in registers,
//                  phases separated by a grid barrier): finishes the reduction cascade, then
//                  alternates exact global relabelling (chaotic min-relaxation from the sink to a
//                  fixed point) with lock-free preflow pushes (agent-scope atomics on excess and
//                  residual capacities, heights written only by their owner) until, right after a
//                  global relabel, no site with positive excess can still reach the sink.
//                  Cut read-out: the sites that cannot reach the sink in the residual graph are the
//                  SOURCE side and take alpha (var 0, :429-433).  This is BK's what_segment rule
//                  (free nodes default to SOURCE, graph.h:478-488) = the unique minimal sink side of
//                  any maximum (pre)flow, so labels are solver-independent (SURVEY A-1).
//   k_delta        int64 energy difference of the candidate labeling; the last workgroup to finish
//                  accepts the move iff the energy strictly decreases (:1259,1273) and records it in
//                  the control words.  The labels themselves are rewritten by the NEXT launch that
//                  walks the sites (the next k_move_setup, or k_apply_pending at the end of a cycle).
//
// Idempotent moves are skipped on the device: the expansion on alpha of a labeling that has not
// changed since the previous expansion on alpha completed cannot lower the energy (every expansion
// of f' = move_alpha(f) is an expansion of f, and f' was the minimum over those), so the reference
// rejects it (:1259) and leaves the labels untouched.  Every launch of such a move returns at once.
// The last cycle of every expansion (the one that confirms convergence) consists of such moves.
//
// A cycle ends with k_energy; the loop stops when the energy is unchanged (:1045).
//
// Roofline: irregular, latency/atomic bound (no dense tile anywhere); reported as time per
// LabelingStep, launches and grid barriers, not as a bandwidth fraction.

#include "mh_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>

namespace mh {

// control words (int); the first EXPAND_HOST_WORDS are mirrored to the host
enum { C_TLAST = 0,        // index of the last accepted move, -1 before the first
       C_PEND = 1,         // alpha of an accepted move whose labels are not yet written, -1 none
       C_ERROR = 2,        // sticky: ERR_*
       C_ACCEPTED = 3,
       C_EXCESS_NODES = 4,
       C_ARRIVE = 5,       // arrivals at the first (flat) grid barrier of the running k_solve
       C_TICKET = 6,       // finished workgroups of the running k_delta
       C_FLOW_MOVES = 7,
       C_CORE = 8,         // 8 shard counters of the core list
       C_MOVES_SOLVED = 16, C_CORE_MAX = 17, C_MOVES_RUN = 18, C_XCD_USED = 19,
       C_TOOK_N = 20,      // sites in the context's took list (k_delta appends, k_commit walks; cleared by the move's first reduction launch)
       // grid barrier of k_solve, one 128-B line per XCD and one for the top level:
       //   line + 0 census (workgroups on this XCD), + 2 .. 9 four 64-bit {arrivals, changed, active} words, + 10 .. 13 four hmax words
       C_XCD = 64, C_LINE = 32, C_TOP = C_XCD + 8 * C_LINE,
       // k_delta's "last workgroup" ticket, two levels: 16 sub-tickets (one line each), then C_TICKET
       C_SUBTICKET = C_TOP + C_LINE, SUBTICKETS = 16,
       C_COUNT = EXPAND_FLAG_WORDS };
static_assert(C_SUBTICKET + SUBTICKETS * C_LINE <= C_COUNT, "control block too small");
// 64-bit accumulators
// (sums that every workgroup of a whole-graph launch contributes to are striped over STRIPES words, each on a
// 128-B line of its own, so that the adds of a launch do not queue behind ONE address: 3 000 adds to one word
// took 36 us; the reader adds the stripes up)
constexpr int STRIPES = 64, STRIPE_LL = 16;
enum { A_DELTA_S = 64, A_ENERGY_S = A_DELTA_S + STRIPES * STRIPE_LL, A_EXCESS_S = A_ENERGY_S + STRIPES * STRIPE_LL,
       A_TOTAL = A_EXCESS_S + STRIPES * STRIPE_LL };
static_assert(A_TOTAL <= EXPAND_ACC_WORDS, "accumulator block too small");
enum { A_DELTA = 0, A_ENERGY = 1, A_EXCESS_SUM = 2, A_CORE_SUM = 3, A_OUTER = 4, A_RELAX = 5, A_PUSH = 6,
       A_BARRIERS = 7, A_TICKS = 8, A_T_BAR = 9, A_T_RELAX = 10, A_T_PUSH = 11, A_T_TAIL = 12, A_TAIL_ROUNDS = 13, A_MAX_WAIT = 14, A_COUNT = 16 /* mirrored to the host */ };
enum { ERR_OVERFLOW = 1, ERR_BARRIER_TIMEOUT = 2, ERR_NO_CONVERGENCE = 3 };

// r06: the per-move state of ONE move in flight (a "context"), and a batch of them: every kernel of a move takes a MoveBatch
// and serves the context blockIdx.y selects, so K consecutive moves that are solved on the same labeling (see k_commit) cost
// the launches of one.  A move alone is a batch of one on context 0.
struct MoveCtx {
    int* cap; int* sent; int* excess; int* sink_cap; int* height; int* decided; unsigned char* took; int* core;
    int* took_list;        // the sites with took[] set, in any order (null: not kept — a move alone applies its labels lazily from took[])
    int* flags; long long* acc;
    int* saved_flow; int* saved_sink; int* trace; int* detail;
    int alpha, t, warm;
};
struct MoveBatch { MoveCtx c[EXPAND_MAX_CTX]; int count; };
// batch control words (device) and what k_commit publishes to the host
enum { B_TLAST0 = 0,       // index of the last accepted move when the batch began (what its moves' skip tests saw)
       B_SEQ = 1,
       B_TICKET = 2,       // workgroups of the running k_batch_commit that have finished
       B_BAD = 8,          // EXPAND_MAX_CTX words: bit k of word j = move k of the batch fails its test against what move j changes
       B_WORDS = 8 + EXPAND_MAX_CTX };
static_assert(EXPAND_MAX_CTX <= 32, "one bit per move of a batch");
enum { HB_FIRST_INVALID = 0, HB_TLAST = 1, HB_ERROR = 2, HB_ACCEPTED_HERE = 3, HB_DIRTY = 4, HB_SEQ = 5, HB_WORDS = 8 };

#define LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// Every launch of a move evaluates this on words that only EARLIER launches wrote.
__device__ __forceinline__ bool move_is_skipped(const int* flags, int t, int L)
{
    return flags[C_ERROR] != 0 || (t >= L && flags[C_TLAST] <= t - L);
}

__global__ void k_ctl_init(MoveBatch b)                 // one workgroup per context
{
    int* flags = b.c[blockIdx.x].flags;
    long long* acc = b.c[blockIdx.x].acc;
    for (int t = threadIdx.x; t < C_COUNT; t += blockDim.x) flags[t] = (t == C_TLAST || t == C_PEND) ? -1 : 0;
    for (int t = threadIdx.x; t < A_TOTAL; t += blockDim.x) acc[t] = 0;
}

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
    if (threadIdx.x == 0)
        atomicAdd((unsigned long long*)&acc[A_ENERGY_S + (blockIdx.x % STRIPES) * STRIPE_LL], (unsigned long long)s[0]);
}

// ---------------------------------------------------------------------------
// Whole-graph kernels.  A site's arcs are contiguous in the CSR, so LPN = 16 consecutive lanes
// (one DPP row; four sites per wavefront) scan them together: coalesced 64-B reads of col/cap and
// width-16 shuffles for the reductions.  All lanes of a site keep identical copies of the site's
// scalars; lane 0 alone writes.
// ---------------------------------------------------------------------------
constexpr int LPN = 16;
constexpr int SITES_PER_BLOCK = 256 / LPN;

// Reductions over the W = 8 or 16 consecutive lanes of a site, as DPP moves inside the VALU (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror): a butterfly through ds_bpermute costs an LDS round trip per step, and the solver's push
// cycle has four of them on its critical path.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ long long dpp_mov64(long long v)
{
    const int lo = dpp_mov<CTRL>((int)(unsigned)((unsigned long long)v & 0xffffffffull)), hi = dpp_mov<CTRL>((int)(v >> 32));
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
template <int W>
__device__ __forceinline__ int row_min(int v)
{
    static_assert(W == 8 || W == 16, "a DPP row");
    int o = dpp_mov<DPP_XOR1>(v); v = o < v ? o : v;
    o = dpp_mov<DPP_XOR2>(v); v = o < v ? o : v;
    o = dpp_mov<DPP_HALF_MIRROR>(v); v = o < v ? o : v;
    if (W == 16) { o = dpp_mov<DPP_MIRROR>(v); v = o < v ? o : v; }
    return v;
}
template <int W>
__device__ __forceinline__ long long row_min64(long long v)
{
    static_assert(W == 8 || W == 16, "a DPP row");
    long long o = dpp_mov64<DPP_XOR1>(v); v = o < v ? o : v;
    o = dpp_mov64<DPP_XOR2>(v); v = o < v ? o : v;
    o = dpp_mov64<DPP_HALF_MIRROR>(v); v = o < v ? o : v;
    if (W == 16) { o = dpp_mov64<DPP_MIRROR>(v); v = o < v ? o : v; }
    return v;
}
template <int W>
__device__ __forceinline__ long long row_sum64(long long v)
{
    static_assert(W == 8 || W == 16, "a DPP row");
    v += dpp_mov64<DPP_XOR1>(v);
    v += dpp_mov64<DPP_XOR2>(v);
    v += dpp_mov64<DPP_HALF_MIRROR>(v);
    if (W == 16) v += dpp_mov64<DPP_MIRROR>(v);
    return v;
}
template <int W>
__device__ __forceinline__ int row_sum(int v)
{
    static_assert(W == 8 || W == 16, "a DPP row");
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);
    if (W == 16) v += dpp_mov<DPP_MIRROR>(v);
    return v;
}

// Most sites of a move need no arc work at all — they are decided by their own numbers, or already carry alpha —
// so the whole-graph kernels of a move run in two steps per wavefront: one LANE per site looks at the site's scalars
// (64 sites per wave), then the sites that do need their arcs walked get LPN lanes each, four sites at a time.
// for_flagged calls body(m) on the 16 lanes of a group with m = the wave lane whose site the group serves; `mask`
// (the ballot of the lanes that asked for it) is wave-uniform, so every lane runs the loop and cross-lane reads of
// per-site scalars (__shfl(value, m)) may precede the call.
template <typename F>
__device__ __forceinline__ void for_flagged(unsigned long long mask, F body)
{
    const int grp = (int)(threadIdx.x & 63) / LPN;
    while (mask) {
        int b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b[q] = mask ? (int)__builtin_ctzll(mask) : -1;
            if (mask) mask &= mask - 1;
        }
        const int m = grp == 0 ? b[0] : grp == 1 ? b[1] : grp == 2 ? b[2] : b[3];
        body(m);
    }
}
// Sites per wavefront in step one.  64 would use every lane, but then a 50k-site graph is 782 waves — less than one
// per SIMD — and a wave with many flagged sites serves them one group of four after the other (k_reduce at 11 labels:
// 27 us against 16 us for the plain LPN-lanes-per-site mapping).  16 keeps 3 125 waves and at most four rounds.
// r06: with several moves per launch (context = blockIdx.y) the launch has that many times the waves, and 16 sites per wave
// leave it with several rounds of residency, each paying the kernel's chain of dependent loads: `spw` is a launch argument,
// 64 for a batch's k_delta (few sites move), 16 / 32 / 64 by the size of the launch for its setup and reduction (run_expansion).
constexpr int SPW = 16;                         // a move alone
constexpr int SPW_MAX = 64;
constexpr int SITES_PER_WBLOCK_MAX = 4 * SPW_MAX;      // per 256-thread workgroup

// `label` is read (neighbours) and written (own site, pending move) in the same launch: a neighbour
// j with took[j] reads as `pa` whether or not its row has already stored the new label.
__global__ void __launch_bounds__(256)
k_move_setup(Graph g, const int* __restrict__ cost, int L, int potts, int reduce_on,
             int* label, int* __restrict__ cur_cost, MoveBatch b, int spw)
{
    const MoveCtx& c = b.c[blockIdx.y];
    const int alpha = c.alpha, t = c.t;
    const unsigned char* __restrict__ took = c.took;
    int* __restrict__ cap = c.cap;
    int* __restrict__ excess = c.excess;
    int* __restrict__ sink_cap = c.sink_cap;
    int* __restrict__ decided = c.decided;
    int* __restrict__ flags = c.flags;
    long long* __restrict__ acc = c.acc;
    const int lane = threadIdx.x & 63;
    const int sub = threadIdx.x % LPN;
    const int s0 = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * spw;
    const int i = s0 + lane;
    const bool in = lane < spw && i < g.n;
    const int pa = flags[C_PEND];
    int li = 0, cc = 0;
    if (in) {
        li = label[i];
        cc = cur_cost[i];
        if (pa >= 0 && took[i]) {                    // applyNewLabeling of the previous move, :423-441
            li = pa;
            cc = cost[(size_t)i * L + pa];
            label[i] = pa; cur_cost[i] = cc;
        }
    }
    if (__shfl(move_is_skipped(flags, t, L) ? 1 : 0, 0)) return;      // one evaluation per wave: C_ERROR may be raised meanwhile
    // step one.  A site that carries alpha is not in the graph (its neighbours clear the arcs towards it, both
    // directions).  A site whose own costs already outweigh every n-link it has is on the sink side whatever its
    // neighbours are: S <= cc + potts * wsum and in <= potts * wsum, so K - cc > 2 * potts * wsum implies -tr > in below.
    long long K = 0;
    bool walk = false;
    if (in) {
        if (li == alpha) decided[i] = 3;
        else {
            K = cost[(size_t)i * L + alpha];
            if (reduce_on && K - cc > 2ll * potts * g.wsum[i]) decided[i] = 2;
            else walk = true;
        }
    }
    for_flagged(__ballot(walk), [&](int m) {
        const int mm = m < 0 ? 0 : m;
        const int li_m = __shfl(li, mm), cc_m = __shfl(cc, mm);
        const long long K_m = __shfl(K, mm);
        if (m < 0) return;
        const int im = s0 + m;
        const int k0 = g.rowptr[im], k1 = g.rowptr[im + 1];
        long long S = 0, out = 0, inn = 0;
        for (int k = k0 + sub; k < k1; k += LPN) {
            const int j = g.col[k];
            const int wk = g.w[k] * potts;
            int lj = label[j];
            if (pa >= 0 && took[j]) lj = pa;
            if (lj == alpha) { S += wk; cap[k] = 0; cap[g.rev[k]] = 0; }
            else if (j < im) { if (li_m != lj) S += wk; cap[k] = wk; out += wk; if (li_m == lj) inn += wk; }
            else { const int c = (li_m == lj) ? wk : 0; cap[k] = c; out += c; inn += wk; }
        }
        S = row_sum64<LPN>(S) + cc_m;
        out = row_sum64<LPN>(out);
        inn = row_sum64<LPN>(inn);
        if (sub != 0) return;
        const long long tr = S - K_m;
        if (tr > 0x7fffffffll || -tr > 0x7fffffffll) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
        const int ex = tr > 0 ? (int)tr : 0;
        excess[im] = ex;
        sink_cap[im] = tr < 0 ? (int)(-tr) : 0;
        // first dominance test (see k_reduce): nothing is decided yet, so there is nothing to fold
        int verdict = 0;
        if (reduce_on) {
            if (tr > out) verdict = 1;
            else if (-tr > inn) verdict = 2;
        }
        decided[im] = verdict;
        // total excess of the move (the reference's flow counter is an int: Graph<int,int,int>), striped
        if (ex > 0) atomicAdd((unsigned long long*)&acc[A_EXCESS_S + (blockIdx.x % STRIPES) * STRIPE_LL], (unsigned long long)ex);
    });
}

// Dominance reduction (see the header).  net(u) = excess - sink_cap.  An undecided site first folds its
// decided neighbours into its own t-link — a source-side neighbour v delivers cap(v->u), a sink-side
// neighbour absorbs cap(u->v) — and clears both arcs, then tests
//     net >  sum of cap(u->w) over undecided w   ->  source side in every minimum cut
//    -net >  sum of cap(w->u) over undecided w   ->  residual sink capacity survives every max flow
// Strict inequalities: ties stay undecided and go to push-relabel, so the minimal sink side (BK's
// read-out, SURVEY A-1) is untouched.  Arcs between two undecided sites are never written here and
// a decided site never writes again, so rounds may run chaotically: every verdict holds given any
// subset of earlier verdicts (a stale "undecided" view only makes the tests stricter).
// The fixed point is reached inside k_solve; these launches take the bulk of the cascade off it.
// COMPACT: sites still undecided when their row finishes are appended to the core list.
template <bool COMPACT>
__global__ void __launch_bounds__(256)
k_reduce(Graph g, int L, MoveBatch b, int ROUNDS, int housekeep, int spw)
{
    const MoveCtx& c = b.c[blockIdx.y];
    const int t = c.t;
    int* cap = c.cap;
    int* excess = c.excess;
    int* sink_cap = c.sink_cap;
    int* decided = c.decided;
    int* __restrict__ flags = c.flags;
    long long* __restrict__ acc = c.acc;
    int* __restrict__ core = c.core;
    // COMPACT walks the sites in a fixed pseudo-random order (g.order).  The solver gives 64 consecutive core sites to
    // one workgroup, and the sites that are busy in a move are neighbours in the image: were they also neighbours in
    // the list (input sorted along a scan line, a Z-curve ...) a few CUs would issue all the uncoalesced requests of
    // a phase.  Measured with the core in Z-curve order: 55-60 ms of solver time per LabelingStep instead of 20.
    const int lane = threadIdx.x & 63;
    const int sub = threadIdx.x % LPN;
    const int slot = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * spw + lane;
    const int u = (lane < spw && slot < g.n) ? (COMPACT ? g.order[slot] : slot) : -1;
    __shared__ int s_skip;                               // one evaluation per workgroup: C_ERROR may be raised meanwhile
    __shared__ int s_cnt, s_base;
    __shared__ int s_list[SITES_PER_WBLOCK_MAX];
    if (threadIdx.x == 0) { s_skip = move_is_skipped(flags, t, L) ? 1 : 0; s_cnt = 0; }
    __syncthreads();
    const bool skipped = s_skip != 0;
    if (housekeep && blockIdx.x == 0 && threadIdx.x == 0) {
        // housekeeping of the move, for the launches behind this one (the first reduction launch of a move does it;
        // the core counters and the solver's barrier words are cleared by k_delta, behind the solver launch that used them)
        flags[C_PEND] = -1;                              // k_move_setup has applied it
        flags[C_TICKET] = 0;
        flags[C_TOOK_N] = 0;
        for (int s = 0; s < SUBTICKETS; ++s) flags[C_SUBTICKET + s * C_LINE] = 0;
        long long ex = 0;
        for (int s = 0; s < STRIPES; ++s) { ex += acc[A_EXCESS_S + s * STRIPE_LL]; acc[A_EXCESS_S + s * STRIPE_LL] = 0; acc[A_DELTA_S + s * STRIPE_LL] = 0; }
        if (!skipped && ex > 0x7fffffffll) flags[C_ERROR] = ERR_OVERFLOW;
    }
    if (skipped) return;
    // step one: who is still undecided; step two: LPN lanes per such site
    const bool open = u >= 0 && LD(&decided[u]) == 0;
    for_flagged(__ballot(open), [&](int m) {
        const int um = __shfl(u, m < 0 ? 0 : m);
        if (m < 0) return;
        const int k0 = g.rowptr[um], k1 = g.rowptr[um + 1];
        long long net = (long long)excess[um] - sink_cap[um];
        bool dirty = false;
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
            add = row_sum64<LPN>(add); out = row_sum64<LPN>(out); in = row_sum64<LPN>(in);
            if (add != 0) { net += add; dirty = true; }
            if (net > out) verdict = 1;
            else if (-net > in) verdict = 2;
            if (verdict) break;
        }
        if (sub == 0) {
            if (dirty) {
                if (net > 0x7fffffffll || -net > 0x7fffffffll) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
                excess[um] = net > 0 ? (int)net : 0;
                sink_cap[um] = net < 0 ? (int)(-net) : 0;
            }
            if (verdict) ST(&decided[um], verdict);
            else if (COMPACT) s_list[atomicAdd(&s_cnt, 1)] = um;
        }
    });
    if (COMPACT) {
        __syncthreads();
        const int shard = blockIdx.x % EXPAND_CORE_SHARDS;
        if (threadIdx.x == 0 && s_cnt > 0) s_base = atomicAdd(&flags[C_CORE + shard], s_cnt);
        __syncthreads();
        if ((int)threadIdx.x < s_cnt) core[(size_t)shard * g.n + s_base + threadIdx.x] = s_list[threadIdx.x];
    }
}

// ---------------------------------------------------------------------------
// k_solve — the flow problem of one move on the compacted core, in ONE launch.
//
// SLPN = 8 lanes per site, SSLOTS = 6 arcs per lane in registers (degree <= 48; larger sites walk
// their arcs in memory), 512 threads = 64 sites per workgroup, P = ceil(K / 64) workgroups take part
// (the rest of the launch returns at once).  When K exceeds what the launch can hold one site per
// row, rows walk several sites and reload them at every visit.
//
// The cost of this kernel is the number of uncoalesced agent-scope requests a CU issues per round
// (about one 64-B request per clock and CU, whatever the payload) times the rounds a move needs, so
// both phases are written as "poll ONE word per site, act only when it changed":
//   * global relabel = frontier BFS from the sink: a site whose height word dropped since it last
//     told its neighbours lowers THEIR words with atomicMin (no return value) along the arcs that
//     are residual towards it.  Idle sites cost one load per round; every height drops a few times.
//   * push: what other sites have sent to u accumulates in excess[u] (atomicAdd without return, issued
//     by the pusher); what u itself has spent — sent on or drained into the sink — is a register of
//     its owner.  An idle site polls excess[u]; only a site that owns excess loads its neighbours'
//     heights and the counters of the reverse arcs.
// Arc state is SINGLE-WRITER: arc k = (u->v) carries the cumulative counter sent[k] that only u's
// row writes, and r(u->v) = cap[k] - sent[k] + sent[rev k].  A counter read late only under-states
// the residual capacity, so a row never pushes more than an arc holds; it never pushes more than it
// owns either, because only arrivals are asynchronous.  Per cycle a site with excess pushes along
// ALL its downhill residual arcs at once (lanes share the excess through a row prefix sum) or lifts
// itself just above its lowest residual neighbour — Hong's lock-free push-relabel rule per arc.
// (Earlier versions: atomics on shared excess/capacity words with the owner's atomic -> load chain,
// 35-57 ms of solver time per LabelingStep at 50k sites; every lane loading all its neighbours'
// words every cycle, 66 ms.)
//
// Everything another row may read is written with agent-scope (sc1) stores or atomics and read with
// agent-scope loads; before a barrier every wave drains its memory operations (s_waitcnt vmcnt(0)),
// so no cache maintenance is needed (MI355X_MICROARCH.md, valid hand-off forms).  The barrier is one
// monotonic arrival counter polled by one lane per workgroup; the reductions every phase needs ("did
// any row change anything", "how many rows are active", "largest finite height") travel through a
// ring of four slots.
// ---------------------------------------------------------------------------
constexpr int SLPN = 8;
constexpr int SSLOTS = 6;
constexpr int SOLVE_THREADS = 512;
constexpr int SOLVE_ROWS = SOLVE_THREADS / SLPN;
constexpr int H_SHIFT = 24, H_MASK = (1 << H_SHIFT) - 1, H_EPOCHS = 126;   // height word = (H_EPOCHS - epoch) << 24 | height

struct SolveParams { int reduce_rounds, relax_rounds, push_cycles, push_phases, push_mult, max_outer, cascade_iters; unsigned long long barrier_timeout, barrier_first_timeout; };

// Grid barrier with reductions, two levels: the workgroups of one XCD meet on a line of their own, the last
// arriver of each XCD carries the XCD's reductions to the top line and arrives there, everybody polls the top
// counter.  Which XCD a workgroup runs on is read from the hardware (HW_REG_XCC_ID); the census of the first
// barrier (a flat one) tells every workgroup how many partners its XCD has.  Placement only affects speed.
struct GridBarrier {
    int* flags;
    int* s_red;          // shared int[8]: wg changed, wg active, wg hmax, out changed, out active, out hmax, abort
    int P;               // participating workgroups
    int xcd, xcd_wgs, xcds;
    unsigned seq;        // barriers passed on the two-level counters
    unsigned long long ticks;
    unsigned long long timeout;      // 100 MHz ticks a poll may last before the launch gives up (a workgroup is not resident)
    unsigned long long timeout_first;   // ... at the FIRST barrier of a launch, which waits for every workgroup to become resident: on a
                                        // shared GPU a delayed dispatch is not a missing workgroup (r04 advisor finding)
    unsigned long long max_wait;     // longest poll of this workgroup's leader so far
};

__device__ __forceinline__ bool spin_until(const int* word, int target, int* flags, unsigned long long timeout)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (LD(word) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {               // a workgroup of this launch is not resident
            atomicExch(&flags[C_ERROR], ERR_BARRIER_TIMEOUT);
            return false;
        }
    }
    return true;
}

// First barrier of a launch: flat, doubles as the census.
__device__ __forceinline__ bool grid_sync_first(GridBarrier& b)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // a core of at most 64 sites: the workgroup's own barrier is the grid's.  Up to 32 workgroups: the barriers are flat (one
    // counter, no per-XCD level), so nobody needs the census, and the first barrier of the cascade orders what this one would
    if (b.P <= 32) { b.xcds = 1; b.xcd_wgs = 1; return true; }
    if (threadIdx.x == 0) {
        b.s_red[6] = 0;
        atomicAdd(&b.flags[C_XCD + b.xcd * C_LINE], 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&b.flags[C_ARRIVE], 1);
        if (!spin_until(&b.flags[C_ARRIVE], b.P, b.flags, b.timeout_first)) b.s_red[6] = 1;
        int used = 0;
        for (int x = 0; x < 8; ++x) used += LD(&b.flags[C_XCD + x * C_LINE]) > 0 ? 1 : 0;
        b.s_red[3] = used;
        b.s_red[4] = LD(&b.flags[C_XCD + b.xcd * C_LINE]);
    }
    __syncthreads();
    b.xcds = b.s_red[3];
    b.xcd_wgs = b.s_red[4];
    return b.s_red[6] == 0;
}

// c_changed: any lane of the workgroup; c_active: counted per lane (a site contributes through ONE lane);
// c_hmax: any lane (largest value wins; a scheduling hint only, so it travels outside the ordered path).
// Arrival and reductions share ONE 64-bit word per level and barrier (bits 0-15 arrivals, 16-31 workgroups that
// changed something, 32-63 active sites): a workgroup pays one returning atomic, the last of an XCD one more, and
// the load that sees the last arrival also carries the sums.
__device__ __forceinline__ bool grid_sync(GridBarrier& b, bool c_changed, bool c_active, int c_hmax,
                                          int& any_changed, int& n_active, int& hmax)
{
    typedef unsigned long long u64;
    const u64 t_in = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave: my stores and atomics have landed
    if (threadIdx.x == 0) { b.s_red[0] = 0; b.s_red[1] = 0; b.s_red[2] = 0; }
    __syncthreads();
    if (__ballot(c_changed) && (threadIdx.x & 63) == 0) b.s_red[0] = 1;
    { const u64 b1 = __ballot(c_active); if (b1 && (threadIdx.x & 63) == 0) atomicAdd(&b.s_red[1], __popcll(b1)); }
    if (c_hmax > 0) atomicMax(&b.s_red[2], c_hmax);
    __syncthreads();
    if (b.P == 1) {                                             // one workgroup: nothing leaves the CU
        any_changed = b.s_red[0];
        n_active = b.s_red[1];
        hmax = b.s_red[2];
        __syncthreads();                                        // everybody has read before the next barrier clears the words
        ++b.seq;
        b.ticks += __builtin_amdgcn_s_memrealtime() - t_in;
        return true;
    }
    if (threadIdx.x == 0) {
        const int slot = (int)(b.seq & 3u), next = (int)((b.seq + 2u) & 3u);
        int* mine = b.flags + C_XCD + b.xcd * C_LINE;
        int* top = b.flags + C_TOP;
        u64* xs = (u64*)(mine + 2);
        u64* ts = (u64*)(top + 2);
        if (b.s_red[2]) {
            atomicMax(&top[10 + slot], b.s_red[2]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const u64 me = 1ull | ((u64)(b.s_red[0] ? 1 : 0) << 16) | ((u64)(unsigned)b.s_red[1] << 32);
        const bool flat = b.P <= 32;                            // few workgroups: one level is a round trip shorter
        if (flat) {
            atomicAdd(&ts[slot], me);
        } else {
            const u64 old = atomicAdd(&xs[slot], me);
            if ((int)(old & 0xffffull) + 1 == b.xcd_wgs) {      // last of my XCD: carry its sums up
                atomicAdd(&ts[slot], ((old + me) & ~0xffffull) | 1ull);
                ST(&xs[next], 0ull);                            // my XCD's word of the barrier after next
            }
        }
        const int target = flat ? b.P : b.xcds;
        u64 v = 0;
        int hm = 0;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        const u64 limit = b.seq == 0 ? b.timeout_first : b.timeout;          // (launches of at most 32 workgroups meet here first)
        int abort = 0;
        for (;;) {
            v = LD(&ts[slot]);
            hm = LD(&top[10 + slot]);
            if ((int)(v & 0xffffull) >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > limit) {             // a workgroup of this launch is not resident
                atomicExch(&b.flags[C_ERROR], ERR_BARRIER_TIMEOUT);
                abort = 1;
                break;
            }
        }
        { const u64 waited = __builtin_amdgcn_s_memrealtime() - t0; if (b.seq != 0 && waited > b.max_wait) b.max_wait = waited; }    // (steady-state waits only)
        b.s_red[3] = (int)((v >> 16) & 0xffffull);
        b.s_red[4] = (int)(v >> 32);
        b.s_red[5] = hm;
        b.s_red[6] = abort;
        if (blockIdx.x == 0) { ST(&ts[next], 0ull); ST(&top[10 + next], 0); }
    }
    __syncthreads();
    any_changed = b.s_red[3];
    n_active = b.s_red[4];
    hmax = b.s_red[5];
    ++b.seq;
    b.ticks += __builtin_amdgcn_s_memrealtime() - t_in;
    return b.s_red[6] == 0;
}

// per-site state of the rows' sites, in LDS: [field][slot][row of the workgroup]
enum { F_SITE = 0, F_HPROP, F_SPENT, F_SC, F_HU, F_SINK0, F_FIELDS };

// The solver proper.  MULTI = false: the core fits the launch one site per row — the site stays in registers for the
// whole move and nothing below touches LDS.  MULTI = true: rows own several sites (see `cur` below).
template <bool MULTI>
__device__ __forceinline__ void
solve_body(const Graph& g, int t, int* cap, int* sent, int* excess, int* sink_cap, int* height, int* decided,
           const int* __restrict__ core, int* flags, long long* acc, int* __restrict__ trace, int* __restrict__ detail,
           int* __restrict__ saved_flow, int* __restrict__ saved_sink, int warm, int mslots, const SolveParams& sp,
           int* s_priv, int* s_red, const int (&pre)[EXPAND_CORE_SHARDS + 1], int K, int P)
{
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    const unsigned long long tick0 = __builtin_amdgcn_s_memrealtime();
    const int total_rows = P * SOLVE_ROWS;
    const int lr = threadIdx.x / SLPN;               // my row inside the workgroup
    const int row = blockIdx.x * SOLVE_ROWS + lr;
    const int sub = threadIdx.x % SLPN;
    // A row owns the core sites row, row + total_rows, ...: S of them ("slots"; one when the core fits the launch).
    const int S = row < K ? (MULTI ? (K - row + total_rows - 1) / total_rows : 1) : 0;
    if ((K + total_rows - 1) / total_rows > mslots) {            // (the host sizes mslots for a core of all n sites)
        if (threadIdx.x == 0) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
        return;
    }
    const int INF = K + 1;                           // no residual path in the core is longer than K
    // HW_REG_XCC_ID (id 20), bits 3:0: the XCD this workgroup runs on
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);
    GridBarrier bar{ flags, s_red, P, xcc, 1, 1, 0u, 0ull, sp.barrier_timeout, sp.barrier_first_timeout, 0ull };
    long long st_outer = 0, st_relax = 0, st_push = 0;
    unsigned long long tk_relax = 0, tk_push = 0, tk_tail = 0;
    long long st_tail = 0;                           // relabel/push rounds that began with fewer than 64 active rows
#define PRIV(field, slot) s_priv[((field) * mslots + (slot)) * SOLVE_ROWS + lr]
#define CUR(slot) (!MULTI || (slot) == cur)

    // ONE site at a time lives in registers (`cur` = its slot): head, then on demand neighbours, arc state and the
    // relabel mask, plus the site's scalars.  The scalars of the other slots wait in LDS; all lanes of a row hold
    // identical copies and write identical values.  Sites are polled through ONE word (height or excess) and
    // entered only when that word says there is something to do, so a row with several sites costs one load per
    // site and round like a row with one.  (The first version of the several-sites case reloaded every site at
    // every visit and gave every BFS level a grid barrier of its own: 203 ms for one 18 313-site move.)
    int cur = -1, u = -1, k0 = 0, k1 = 0, dec = 3;
    int hu = 0, sc = 0, spent = 0, hprop = 0, sink0 = 0;
    bool fast = false, nb_ok = false, arcs_ok = false, rin_ok = false, priv_live = false;
    int nb[SSLOTS], rv[SSLOTS], c0[SSLOTS], so[SSLOTS];
    unsigned rin = 0;                                // bit q: the arc nb[q] -> u is residual (relabel direction)

    auto enter = [&](int slot) {
        if (cur == slot) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the counters I stored for the site I leave are read back when I return
        if (cur >= 0 && priv_live) { PRIV(F_HPROP, cur) = hprop; PRIV(F_SPENT, cur) = spent; PRIV(F_SC, cur) = sc; PRIV(F_HU, cur) = hu; }
        cur = slot;
        const int c = row + slot * total_rows;
        int s = 0, first = 0;
#pragma unroll
        for (int q = 1; q < EXPAND_CORE_SHARDS; ++q) if (c >= pre[q]) { s = q; first = pre[q]; }
        u = core[(size_t)s * g.n + (c - first)];
        k0 = g.rowptr[u];
        k1 = g.rowptr[u + 1];
        fast = (k1 - k0) <= SLPN * SSLOTS;
        dec = LD(&decided[u]);
        nb_ok = arcs_ok = rin_ok = false;
        if (priv_live) { hprop = PRIV(F_HPROP, slot); spent = PRIV(F_SPENT, slot); sc = PRIV(F_SC, slot); hu = PRIV(F_HU, slot); sink0 = PRIV(F_SINK0, slot); }
    };
    auto need_nbrs = [&]() {
        if (nb_ok) return;
#pragma unroll
        for (int q = 0; q < SSLOTS; ++q) {
            const int k = k0 + sub + q * SLPN;
            const bool in = fast && k < k1;
            nb[q] = in ? g.col[k] : -1;
            rv[q] = in ? g.rev[k] : 0;
            c0[q] = 0; so[q] = 0;
        }
        nb_ok = true;
    };
    // capacities (final once the reduction has reached its fixed point) and the row's own counters; arcs
    // without capacity in either direction (decided neighbours) drop out
    auto need_arcs = [&]() {
        need_nbrs();
        if (arcs_ok) return;
        if (fast) {
#pragma unroll
            for (int q = 0; q < SSLOTS; ++q) {
                if (nb[q] >= 0) {
                    const int k = k0 + sub + q * SLPN;
                    c0[q] = LD(&cap[k]);
                    so[q] = LD(&sent[k]);
                    if ((c0[q] | LD(&cap[rv[q]])) == 0) nb[q] = -1;
                }
            }
        }
        arcs_ok = true;
    };
    auto need_rin = [&]() {                          // arcs residual TOWARDS me, constant between two push phases
        need_arcs();
        if (rin_ok) return;
        rin = 0;
        if (fast) {
#pragma unroll
            for (int q = 0; q < SSLOTS; ++q)
                if (nb[q] >= 0 && LD(&cap[rv[q]]) - LD(&sent[rv[q]]) + so[q] > 0) rin |= 1u << q;
        }
        rin_ok = true;
    };
#define FOR_MY_SLOTS(slot) for (int slot = 0; slot < S; ++slot)
    int site0 = -1;                                  // PRIV(F_SITE, 0): the undecided site of slot 0, or -1
#define SITE_OF(slot) ((!MULTI || (slot) == 0) ? site0 : PRIV(F_SITE, slot))

    // my arcs' flow counters start at zero (visible to everybody behind the barriers of phase 0)
    FOR_MY_SLOTS(slot) {
        enter(slot);
        if (!warm) for (int k = k0 + sub; k < k1; k += SLPN) ST(&sent[k], 0);
        if (sub == 0) ST(&height[u], 0x7fffffff);     // a word of no epoch
    }
    if (!grid_sync_first(bar)) return;

    // ---- phase 0: the reduction cascade to its fixed point (k_reduce's body on the core) -------------
    int cascade_pass = 0;
    for (;;) {
        // (a capped cascade ends with a pass that only folds: every verdict taken before it is then folded into its
        // undecided neighbours, which is what the flow phase relies on — decided sites are not part of its graph)
        const bool fold_only = sp.cascade_iters > 0 && cascade_pass + 1 >= sp.cascade_iters;
        bool changed = false;
        FOR_MY_SLOTS(slot) {
            enter(slot);
            if (MULTI) dec = LD(&decided[u]);          // (a row with one site keeps its verdict in the register)
            if (dec != 0) continue;
            long long net = (long long)LD(&excess[u]) - LD(&sink_cap[u]);
            bool dirty = false;
            int verdict = 0;
            for (int r = 0; r < sp.reduce_rounds; ++r) {
                long long add = 0, out = 0, in = 0;
                for (int k = k0 + sub; k < k1; k += SLPN) {
                    const int kr = g.rev[k];
                    const int co = LD(&cap[k]), ci = LD(&cap[kr]);
                    if ((co | ci) == 0) continue;
                    const int dv = LD(&decided[g.col[k]]);
                    if (dv == 1) { add += ci; ST(&cap[k], 0); ST(&cap[kr], 0); }
                    else if (dv == 2) { add -= co; ST(&cap[k], 0); ST(&cap[kr], 0); }
                    else { out += co; in += ci; }
                }
                add = row_sum64<SLPN>(add); out = row_sum64<SLPN>(out); in = row_sum64<SLPN>(in);
                if (add != 0) { net += add; dirty = true; changed = true; }
                if (fold_only) break;                 // the capped cascade's last pass: no new verdicts, so none is left unfolded
                if (net > out) verdict = 1;
                else if (-net > in) verdict = 2;
                if (verdict || add == 0) break;       // a round that folded nothing would repeat itself
            }
            if (sub == 0) {
                if (dirty) {
                    if (net > 0x7fffffffll || -net > 0x7fffffffll) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
                    ST(&excess[u], net > 0 ? (int)net : 0);
                    ST(&sink_cap[u], net < 0 ? (int)(-net) : 0);
                }
                if (verdict) ST(&decided[u], verdict);
            }
            if (verdict) { dec = verdict; changed = true; }
        }
        int any, nact, hm;
        if (!grid_sync(bar, changed && sub == 0, false, 0, any, nact, hm)) return;
        if (!any || sp.reduce_rounds <= 0) break;
        // The cascade is exact but optional: a site it would still decide stays in the flow problem, whose read-out gives it
        // the same side.  A cap trades its barrier-separated passes against a slightly larger flow problem.
        if (fold_only) break;
        ++cascade_pass;
    }

    // ---- flow recycling: start from what the previous expansion on this label left in the arcs ----------
    // saved_flow[k] = net flow the arc carried when that move ended, saved_sink[u] = what u drained into the sink.
    // Clipped to this move's capacities they are a feasible pseudo-flow: every arc counter lies in [0, cap], so
    // cap - sent + sent[rev] stays within the pair's total.  A site left owing flow (it sends on more than it owns)
    // gets the missing amount as source supply AND as sink capacity — the same constant on both t-links changes the
    // value of every cut alike (Kohli & Torr's reparametrisation) — so excess is never negative.  A maximum flow
    // from any feasible start has the same residual reachability, so the read-out (hence every label) is unchanged;
    // only the number of relabel/push rounds is, as later cycles re-solve almost the same problem.
    if (warm) {
        FOR_MY_SLOTS(slot) {
            enter(slot);
            if (MULTI) dec = LD(&decided[u]);
            if (dec != 0) continue;
            for (int k = k0 + sub; k < k1; k += SLPN) {
                const int f = saved_flow[k], ck = LD(&cap[k]);
                ST(&sent[k], f < ck ? f : ck);
            }
        }
        int a0, a1, a2;
        if (!grid_sync(bar, false, false, 0, a0, a1, a2)) return;
    }
    FOR_MY_SLOTS(slot) {
        enter(slot);
        if (MULTI) { dec = LD(&decided[u]); PRIV(F_SITE, slot) = dec == 0 ? u : -1; }
        else site0 = dec == 0 ? u : -1;
        if (dec != 0) continue;
        int sc_s = LD(&sink_cap[u]), spent_s = 0, sink0_s = sc_s;
        if (warm) {
            long long net = 0;                            // what my arcs carry away, net
            for (int k = k0 + sub; k < k1; k += SLPN) {
                const int kr = g.rev[k];
                if ((LD(&cap[k]) | LD(&cap[kr])) == 0) continue;      // decided neighbour: folded, and its counter is not maintained
                net += (long long)LD(&sent[k]) - LD(&sent[kr]);
            }
            net = row_sum64<SLPN>(net);
            const int sv = saved_sink[u];
            const int sf = sv < sc_s ? sv : sc_s;
            sc_s -= sf;
            long long own = (long long)sf + net;
            const long long e = (long long)LD(&excess[u]) - own;
            if (e < 0) {
                own += e;
                const long long s2 = (long long)sc_s - e;
                if (s2 > (1 << 30)) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
                sc_s = (int)s2;
            }
            spent_s = (int)own;
            sink0_s = sc_s + sf;
            if (sub == 0) ST(&sink_cap[u], sc_s);
        }
        if (MULTI) {
            PRIV(F_SC, slot) = sc_s; PRIV(F_SPENT, slot) = spent_s; PRIV(F_SINK0, slot) = sink0_s;
            PRIV(F_HPROP, slot) = INF; PRIV(F_HU, slot) = INF;
        } else {
            sc = sc_s; spent = spent_s; sink0 = sink0_s; hprop = INF; hu = INF;
        }
    }
    // from here on the scalars of the site in registers are live (written back when the row turns to another site)
    if (MULTI) {
        site0 = S > 0 ? PRIV(F_SITE, 0) : -1;
        if (cur >= 0) { hprop = PRIV(F_HPROP, cur); spent = PRIV(F_SPENT, cur); sc = PRIV(F_SC, cur); hu = PRIV(F_HU, cur); sink0 = PRIV(F_SINK0, cur); }
        priv_live = true;
    }
    arcs_ok = rin_ok = false;
    if (!MULTI && site0 >= 0) need_arcs();

    // ---- phase 1: global relabel <-> push until no site with excess can reach the sink ---------------
    int outer = 0, epoch = 0, hprev = 4;
    for (;; ++outer) {
        if (outer >= sp.max_outer) { if (leader) atomicExch(&flags[C_ERROR], ERR_NO_CONVERGENCE); break; }
        ++st_outer;
        const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
        // Global relabel.  Height words carry the relabel's epoch in their top bits, newer epochs SMALLER: a word of
        // an older epoch reads as "unreachable" and loses against any atomicMin of the running epoch, so a relabel
        // needs no pass that resets the words, and the sites next to the sink enter with an atomicMin of their own.
        if (++epoch > H_EPOCHS) {                      // out of epochs (never seen): start over behind two barriers
            int a0, a1, a2;
            if (!grid_sync(bar, false, false, 0, a0, a1, a2)) return;
            FOR_MY_SLOTS(slot) { const int us = SITE_OF(slot); if (us >= 0 && sub == 0) ST(&height[us], 0x7fffffff); }
            if (!grid_sync(bar, false, false, 0, a0, a1, a2)) return;
            epoch = 1;
        }
        const int ebits = (H_EPOCHS - epoch) << H_SHIFT;
        rin_ok = false;                                // the pushes since the last relabel moved the residual capacities
        FOR_MY_SLOTS(slot) {
            const int us = SITE_OF(slot);
            if (us < 0) continue;
            const int sc_s = CUR(slot) ? sc : PRIV(F_SC, slot);
            if (sc_s > 0 && sub == 0) atomicMin(&height[us], ebits | 1);
            if (CUR(slot)) hprop = INF; else PRIV(F_HPROP, slot) = INF;      // nothing told to the neighbours yet
        }
        // the site in registers has its relabel mask ready before the frontier arrives (a mask computed on arrival
        // puts a dozen dependent loads between hearing and telling: every BFS level took twice as long)
        if (MULTI ? (cur >= 0 && SITE_OF(cur) >= 0) : (site0 >= 0)) need_rin();
        int any, nact, hmax;
        // (the frontier advances one level per round: a few rounds more than the last relabel was deep)
        int relax_rounds = hprev + 12;
        if (relax_rounds > sp.relax_rounds) relax_rounds = sp.relax_rounds;
        // ... then the frontier runs: a site whose word dropped since it last spoke lowers its upstream
        // neighbours' words to its height + 1.  Words only decrease and every value is realised by a residual
        // path, so any schedule converges to the BFS distances; an interval without a single announcement is
        // the fixed point (all atomics of earlier intervals had landed before it began).
        for (;;) {
            ++st_relax;
            bool spoke = false, active = false;
            int my_h = 0;
            for (int r = 0; r < relax_rounds; ++r) {
                bool settled = true;                  // every site of mine next to the sink and announced: final
                FOR_MY_SLOTS(slot) {
                    const int us = SITE_OF(slot);
                    if (us < 0) continue;
                    const int w = LD(&height[us]);
                    const int h = (w >> H_SHIFT) == (ebits >> H_SHIFT) ? (w & H_MASK) : INF;
                    const int hp = CUR(slot) ? hprop : PRIV(F_HPROP, slot);
                    if (h < hp) {
                        if (MULTI) { enter(slot); need_rin(); }
                        if (fast) {
#pragma unroll
                            for (int q = 0; q < SSLOTS; ++q)
                                if (rin & (1u << q)) atomicMin(&height[nb[q]], ebits | (h + 1));
                        } else {
                            for (int k = k0 + sub; k < k1; k += SLPN) {
                                const int kr = g.rev[k];
                                const int ckr = LD(&cap[kr]);
                                if ((LD(&cap[k]) | ckr) == 0) continue;      // folded arc: the neighbour is decided, its counter is stale
                                if (ckr - LD(&sent[kr]) + LD(&sent[k]) > 0) atomicMin(&height[g.col[k]], ebits | (h + 1));
                            }
                        }
                        hprop = h;
                        spoke = true;
                        settled = false;
                    } else if (h > 1) settled = false;
                    if (CUR(slot)) hu = h; else PRIV(F_HU, slot) = h;
                }
                if (settled) break;
            }
            FOR_MY_SLOTS(slot) {
                const int us = SITE_OF(slot);
                if (us < 0) continue;
                const int h = CUR(slot) ? hu : PRIV(F_HU, slot);
                if (h < INF) {
                    if (h > my_h) my_h = h;
                    if (sub == 0 && LD(&excess[us]) - (CUR(slot) ? spent : PRIV(F_SPENT, slot)) > 0) active = true;
                }
            }
            if (!grid_sync(bar, spoke, active, my_h, any, nact, hmax)) return;
            // no announcement in a whole interval: the distances are exact.  Otherwise every finite height is still
            // realised by a residual path, which is all the pushes need: go on as soon as somebody can push.
            if (!any || nact) break;
        }
        hprev = hmax;
        tk_relax += __builtin_amdgcn_s_memrealtime() - tr0;
        if (detail && leader && outer < 2048) {
            int* d = detail + 4 * (size_t)outer;
            d[0] = nact; d[1] = hmax; d[2] = (int)st_relax; d[3] = (int)(__builtin_amdgcn_s_memrealtime() - tick0);
        }
        if (nact == 0) break;                          // (then exact) finished: nobody with excess reaches the sink
        const bool tail_round = nact < 64;
        if (outer == 0 && leader) atomicAdd(&flags[C_FLOW_MOVES], 1);

        // lock-free push-relabel (see the header of this kernel); a wave of pushes needs about as many cycles
        // as the longest residual path is long
        const unsigned long long tp0 = __builtin_amdgcn_s_memrealtime();
        int cycles = sp.push_mult * (hmax + 3);
        if (cycles > sp.push_cycles) cycles = sp.push_cycles;
        for (int phase = 0; phase < sp.push_phases; ++phase) {
            ++st_push;
            bool active = false;
            FOR_MY_SLOTS(slot) {                      // my words as the last announcements left them
                const int us = SITE_OF(slot);
                if (us < 0) continue;
                const int w = LD(&height[us]);
                const int h = (w >> H_SHIFT) == (ebits >> H_SHIFT) ? (w & H_MASK) : INF;
                if (CUR(slot)) hu = h; else PRIV(F_HU, slot) = h;
            }
            for (int cyc = 0; cyc < cycles; ++cyc) {
                bool alive = false;                   // somebody of mine can still reach the sink
                FOR_MY_SLOTS(slot) {
                    const int us = SITE_OF(slot);
                    if (us < 0) continue;
                    if ((CUR(slot) ? hu : PRIV(F_HU, slot)) >= INF) continue;
                    alive = true;
                    const int x = LD(&excess[us]);
                    if (x > (1 << 30)) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
                    if (x - (CUR(slot) ? spent : PRIV(F_SPENT, slot)) <= 0) continue;      // idle: poll again
                    if (MULTI) { enter(slot); need_arcs(); }
                    int e = x - spent;
                    if (sc > 0) {                                   // t-link: h(t) = 0, h(u) >= 1
                        const int d = e < sc ? e : sc;
                        sc -= d; spent += d; e -= d;
                        if (sub == 0) ST(&sink_cap[u], sc);
                    }
                    if (e <= 0) continue;
                    if (fast) {
                        int r[SSLOTS], hq[SSLOTS];
#pragma unroll
                        for (int q = 0; q < SSLOTS; ++q) {          // one level of independent loads
                            r[q] = nb[q] >= 0 ? LD(&sent[rv[q]]) : 0;
                            hq[q] = nb[q] >= 0 ? LD(&height[nb[q]]) : 0;
                        }
#pragma unroll
                        for (int q = 0; q < SSLOTS; ++q)
                            hq[q] = (hq[q] >> H_SHIFT) == (ebits >> H_SHIFT) ? (hq[q] & H_MASK) : INF;
                        int D = 0;                                  // what my downhill residual arcs can take
#pragma unroll
                        for (int q = 0; q < SSLOTS; ++q) {
                            r[q] = nb[q] >= 0 ? c0[q] - so[q] + r[q] : 0;
                            if (r[q] > 0 && hq[q] < hu) D += r[q];
                        }
                        static_assert(SLPN == 8, "the prefix below is row_shr 1, 2, 4: complete for rows of 8 lanes only (16 would need a row_shr 8 step)");
                        // (the lanes of a row are convergent here and at every row_min / row_sum: the DPP moves use bound_ctrl,
                        // a lane switched off by EXEC would read as 0)
                        int incl = D;                               // prefix over the row's lanes: row_shr 1, 2, 4
                        { const int y = dpp_mov<0x111>(incl); if (sub >= 1) incl += y; }
                        { const int y = dpp_mov<0x112>(incl); if (sub >= 2) incl += y; }
                        { const int y = dpp_mov<0x114>(incl); if (sub >= 4) incl += y; }
                        const int tot = row_sum<SLPN>(D);
                        if (tot > 0) {
                            int budget = e - (incl - D);            // the lanes in front of me are served first
                            if (budget > D) budget = D;
#pragma unroll
                            for (int q = 0; q < SSLOTS; ++q) {
                                if (budget > 0 && r[q] > 0 && hq[q] < hu) {
                                    const int d = r[q] < budget ? r[q] : budget;
                                    so[q] += d;
                                    budget -= d;
                                    ST(&sent[k0 + sub + q * SLPN], so[q]);
                                    atomicAdd(&excess[nb[q]], d);
                                    if (so[q] > (1 << 30)) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
                                }
                            }
                            spent += e < tot ? e : tot;
                        } else {                                    // nothing downhill: lift just above the lowest residual neighbour
                            int hmin = INF;
#pragma unroll
                            for (int q = 0; q < SSLOTS; ++q) if (r[q] > 0 && hq[q] < hmin) hmin = hq[q];
                            hmin = row_min<SLPN>(hmin);
                            hu = hmin >= INF ? INF : hmin + 1;
                            if (sub == 0) ST(&height[u], ebits | hu);
                        }
                    } else {                                        // any degree: one arc per cycle, arcs walked in memory
                        long long key = 0x7fffffffffffffffll;       // (height << 32) | arc
                        for (int k = k0 + sub; k < k1; k += SLPN) {
                            // An arc whose pair has no capacity left leads to a neighbour decided in this move: it is not in
                            // the graph, and neither its counter (last written in an earlier move) nor its height word
                            // (another move's epoch numbering) means anything here.
                            const int ck = LD(&cap[k]), kr = g.rev[k];
                            if ((ck | LD(&cap[kr])) == 0) continue;
                            if (ck - LD(&sent[k]) + LD(&sent[kr]) > 0) {
                                const int w = LD(&height[g.col[k]]);
                                const int hv = (w >> H_SHIFT) == (ebits >> H_SHIFT) ? (w & H_MASK) : INF;
                                const long long cand = ((long long)hv << 32) | (unsigned int)k;
                                if (cand < key) key = cand;
                            }
                        }
                        key = row_min64<SLPN>(key);
                        if (key == 0x7fffffffffffffffll) {          // no way out at all
                            hu = INF;
                            if (sub == 0) ST(&height[u], ebits | hu);
                            continue;
                        }
                        const int hmin = (int)(key >> 32), kmin = (int)(key & 0xffffffffll);
                        if (hu > hmin) {
                            const int a = LD(&sent[kmin]);
                            const int rr = LD(&cap[kmin]) - a + LD(&sent[g.rev[kmin]]);
                            const int d = e < rr ? e : rr;
                            if (sub == 0) {
                                ST(&sent[kmin], a + d);
                                atomicAdd(&excess[g.col[kmin]], d);
                                if (a + d > (1 << 30)) atomicExch(&flags[C_ERROR], ERR_OVERFLOW);
                            }
                            spent += d;
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next cycle reads the counter back
                        } else {
                            hu = hmin + 1;
                            if (hu > INF) hu = INF;
                            if (sub == 0) ST(&height[u], ebits | hu);
                        }
                    }
                }
                if (!alive) break;
            }
            FOR_MY_SLOTS(slot) {
                const int us = SITE_OF(slot);
                if (us < 0) continue;
                if (sub == 0 && (CUR(slot) ? hu : PRIV(F_HU, slot)) < INF &&
                    LD(&excess[us]) - (CUR(slot) ? spent : PRIV(F_SPENT, slot)) > 0) active = true;
            }
            // this barrier also separates the pushes from the next relabel (the counters are final behind it)
            if (!grid_sync(bar, false, active, 0, any, nact, hmax)) return;
            if (nact == 0) break;
        }
        tk_push += __builtin_amdgcn_s_memrealtime() - tp0;
        if (tail_round) { tk_tail += __builtin_amdgcn_s_memrealtime() - tr0; ++st_tail; }
    }

    // ---- read-out: whoever cannot reach the sink takes alpha ---------------------------------------
    FOR_MY_SLOTS(slot) {
        const int us = SITE_OF(slot);
        if (us < 0) continue;
        const int h = CUR(slot) ? hu : PRIV(F_HU, slot);
        if (sub == 0) ST(&decided[us], h >= INF ? 1 : 2);
        if (saved_flow) {                                 // for the next expansion on this label (the counters are final)
            enter(slot);
            for (int k = k0 + sub; k < k1; k += SLPN) {
                const int kr = g.rev[k];
                int f = 0;
                if ((LD(&cap[k]) | LD(&cap[kr])) != 0) f = LD(&sent[k]) - LD(&sent[kr]);
                saved_flow[k] = f > 0 ? f : 0;
            }
            if (sub == 0) saved_sink[u] = sink0 - sc;
        }
    }
#undef FOR_MY_SLOTS
#undef SITE_OF
#undef CUR
#undef PRIV
    if (leader) {
        atomicAdd(&flags[C_MOVES_SOLVED], 1);
        atomicMax(&flags[C_CORE_MAX], K);
        atomicAdd((unsigned long long*)&acc[A_CORE_SUM], (unsigned long long)K);
        atomicAdd((unsigned long long*)&acc[A_OUTER], (unsigned long long)st_outer);
        atomicAdd((unsigned long long*)&acc[A_RELAX], (unsigned long long)st_relax);
        atomicAdd((unsigned long long*)&acc[A_PUSH], (unsigned long long)st_push);
        atomicAdd((unsigned long long*)&acc[A_BARRIERS], (unsigned long long)bar.seq);
        atomicAdd((unsigned long long*)&acc[A_TICKS], __builtin_amdgcn_s_memrealtime() - tick0);
        atomicAdd((unsigned long long*)&acc[A_T_BAR], bar.ticks);
        atomicAdd((unsigned long long*)&acc[A_T_RELAX], tk_relax);
        atomicAdd((unsigned long long*)&acc[A_T_PUSH], tk_push);
        atomicAdd((unsigned long long*)&acc[A_T_TAIL], tk_tail);
        atomicAdd((unsigned long long*)&acc[A_TAIL_ROUNDS], (unsigned long long)st_tail);
        atomicMax((unsigned long long*)&acc[A_MAX_WAIT], bar.max_wait);
        atomicMax(&flags[C_XCD_USED], bar.xcds);
        if (trace) {
            int* tr = trace + 8 * (size_t)t;
            tr[0] = K; tr[1] = P; tr[2] = (int)st_outer; tr[3] = (int)st_relax; tr[4] = (int)st_push; tr[5] = (int)bar.seq;
            tr[6] = (int)(__builtin_amdgcn_s_memrealtime() - tick0); tr[7] = (int)bar.ticks;
        }
    }
}

// DIAGNOSTIC (mh_set_tuning key 21): connected components of a move's undecided core — the sites the
// dominance reduction left to the flow solver, joined where two of them are neighbours (their n-link has capacity; links
// to decided sites were folded into t-links).  One workgroup: min-label propagation with pointer jumping to the fixed
// point, then sizes.  out[16]: {core sites, components, largest, second largest, sites in components of <= 64 / 256 / 1024 /
// 2048 / 8192 sites (cumulative), components of <= 64 / 256 / 1024 / 2048 / 8192 sites (cumulative), propagation rounds, 0}.
// comp, csize: n ints of scratch each.  Never on the product path.
__global__ void __launch_bounds__(1024)
k_core_components(Graph g, int L, int t, const int* __restrict__ decided, const int* __restrict__ flags, int* __restrict__ comp,
                  int* __restrict__ csize, int* __restrict__ out)
{
    __shared__ int s_changed, s_acc[16];
    if (threadIdx.x < 16) { s_acc[threadIdx.x] = 0; out[threadIdx.x] = 0; }
    if (move_is_skipped(flags, t, L)) return;
    for (int i = threadIdx.x; i < g.n; i += 1024) { comp[i] = decided[i] == 0 ? i : -1; csize[i] = 0; }
    __syncthreads();
    int rounds = 0;
    for (;;) {
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < g.n; i += 1024) {
            int c = comp[i];
            if (c < 0) continue;
            int m = c;
            for (int k = g.rowptr[i]; k < g.rowptr[i + 1]; ++k) { const int cj = comp[g.col[k]]; if (cj >= 0 && cj < m) m = cj; }
            if (m < c) { atomicMin(&comp[i], m); atomicMin(&comp[c], m); s_changed = 1; }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < g.n; i += 1024) {          // pointer jumping
            int c = comp[i];
            if (c < 0) continue;
            while (comp[c] < c) c = comp[c];
            comp[i] = c;
        }
        __syncthreads();
        ++rounds;
        if (!s_changed || rounds > 4096) break;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < g.n; i += 1024) if (comp[i] >= 0) { atomicAdd(&csize[comp[i]], 1); atomicAdd(&s_acc[0], 1); }
    __syncthreads();
    for (int i = threadIdx.x; i < g.n; i += 1024) {
        const int sz = csize[i];
        if (sz <= 0) continue;
        atomicAdd(&s_acc[1], 1);
        const int old = atomicMax(&s_acc[2], sz);
        atomicMax(&s_acc[3], old < sz ? old : sz);              // (a lower bound of the second largest; exact when sizes arrive in any order but ties)
        if (sz <= 64) { atomicAdd(&s_acc[4], sz); atomicAdd(&s_acc[9], 1); }
        if (sz <= 256) { atomicAdd(&s_acc[5], sz); atomicAdd(&s_acc[10], 1); }
        if (sz <= 1024) { atomicAdd(&s_acc[6], sz); atomicAdd(&s_acc[11], 1); }
        if (sz <= 2048) { atomicAdd(&s_acc[7], sz); atomicAdd(&s_acc[12], 1); }
        if (sz <= 8192) { atomicAdd(&s_acc[8], sz); atomicAdd(&s_acc[13], 1); }
    }
    __syncthreads();
    if (threadIdx.x < 16) out[threadIdx.x] = threadIdx.x == 14 ? rounds : s_acc[threadIdx.x];
}

__global__ void __launch_bounds__(SOLVE_THREADS)
k_solve(Graph g, int L, MoveBatch b, int mslots, SolveParams sp)
{
    const MoveCtx& c = b.c[blockIdx.y];
    const int t = c.t, warm = c.warm;
    int* cap = c.cap;
    int* sent = c.sent;
    int* excess = c.excess;
    int* sink_cap = c.sink_cap;
    int* height = c.height;
    int* decided = c.decided;
    const int* __restrict__ core = c.core;
    int* flags = c.flags;
    long long* acc = c.acc;
    int* __restrict__ trace = c.trace;
    int* __restrict__ detail = c.detail;
    int* __restrict__ saved_flow = c.saved_flow;
    int* __restrict__ saved_sink = c.saved_sink;
    extern __shared__ int s_priv[];                  // F_FIELDS x mslots x SOLVE_ROWS
    __shared__ int s_red[8];
    __shared__ int s_skip;
    if (threadIdx.x == 0) s_skip = move_is_skipped(flags, t, L) ? 1 : 0;
    __syncthreads();
    if (s_skip) return;
    if (g.n >= H_MASK) { if (threadIdx.x == 0) atomicExch(&flags[C_ERROR], ERR_OVERFLOW); return; }   // heights carry 24 bits
    int pre[EXPAND_CORE_SHARDS + 1];
    pre[0] = 0;
#pragma unroll
    for (int s = 0; s < EXPAND_CORE_SHARDS; ++s) pre[s + 1] = pre[s] + flags[C_CORE + s];
    const int K = pre[EXPAND_CORE_SHARDS];
    if (K == 0) return;
    int P = (K + SOLVE_ROWS - 1) / SOLVE_ROWS;
    if (P > (int)gridDim.x) P = (int)gridDim.x;
    if ((int)blockIdx.x >= P) return;
    if (K <= P * SOLVE_ROWS) solve_body<false>(g, t, cap, sent, excess, sink_cap, height, decided, core, flags, acc, trace, detail,
                                                saved_flow, saved_sink, warm, mslots, sp, s_priv, s_red, pre, K, P);
    else solve_body<true>(g, t, cap, sent, excess, sink_cap, height, decided, core, flags, acc, trace, detail,
                          saved_flow, saved_sink, warm, mslots, sp, s_priv, s_red, pre, K, P);
}

// Energy difference of the candidate labeling (sites with decided == 1 take alpha) and, by the last
// workgroup to finish, the verdict of the move: accept iff the energy strictly decreases (:1259).
__global__ void __launch_bounds__(256)
k_delta(Graph g, const int* __restrict__ cost, int L, int potts,
        const int* __restrict__ label, const int* __restrict__ cur_cost, MoveBatch b, int spw)
{
    const MoveCtx& c = b.c[blockIdx.y];
    const int alpha = c.alpha, t = c.t;
    const int* __restrict__ decided = c.decided;
    unsigned char* __restrict__ took = c.took;
    int* __restrict__ flags = c.flags;
    long long* acc = c.acc;
    __shared__ long long s[256];
    __shared__ int s_last;
    if (threadIdx.x == 0) s_last = move_is_skipped(flags, t, L) ? 1 : 0;
    __syncthreads();
    if (s_last) return;
    // step one: every site records whether it moves; step two: the sites that move walk their arcs.  A pair of
    // neighbours changes its Potts term only if one of them moves: it is counted from the moving end, or — both
    // moving — from the end with the larger index.
    const int lane = threadIdx.x & 63;
    const int sub = threadIdx.x % LPN;
    const int s0 = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * spw;
    const int i = s0 + lane;
    long long mine = 0;
    int oi = 0;
    bool ti = false;
    if (lane < spw && i < g.n) {
        oi = label[i];
        ti = decided[i] == 1;
        took[i] = ti ? 1 : 0;
        if (ti) mine += (long long)cost[(size_t)i * L + alpha] - cur_cost[i];
    }
    if (c.took_list) {                                   // (wave-uniform) the moving sites as a list, for k_commit
        const unsigned long long tb = __ballot(ti);
        if (tb) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&flags[C_TOOK_N], (int)__popcll(tb));
            base = __shfl(base, 0);
            if (ti) c.took_list[base + (int)__popcll(tb & ((1ull << lane) - 1ull))] = i;
        }
    }
    for_flagged(__ballot(ti), [&](int m) {
        const int oi_m = __shfl(oi, m < 0 ? 0 : m);
        if (m < 0) return;
        const int im = s0 + m;
        for (int k = g.rowptr[im] + sub; k < g.rowptr[im + 1]; k += LPN) {
            const int j = g.col[k];
            const bool tj = decided[j] == 1;
            if (tj && j > im) continue;
            const int oj = label[j];
            const int nj = tj ? alpha : oj;
            const int dn = (alpha != nj) - (oi_m != oj);
            mine += (long long)dn * g.w[k] * potts;
        }
    });
    s[threadIdx.x] = mine;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (s[0] != 0) atomicAdd((unsigned long long*)&acc[A_DELTA_S + (blockIdx.x % STRIPES) * STRIPE_LL], (unsigned long long)s[0]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my sum has landed before my ticket
        // two-level ticket: the last of my sub-ticket's workgroups takes a top ticket, the last of those finishes the move
        const int st = (int)(blockIdx.x % SUBTICKETS);
        const int members = ((int)gridDim.x - st + SUBTICKETS - 1) / SUBTICKETS;
        int last = 0;
        if (atomicAdd(&flags[C_SUBTICKET + st * C_LINE], 1) == members - 1) {
            const int used = (int)gridDim.x < SUBTICKETS ? (int)gridDim.x : SUBTICKETS;
            last = atomicAdd(&flags[C_TICKET], 1) == used - 1;
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        long long d = (int)threadIdx.x < STRIPES ? LD(&acc[A_DELTA_S + threadIdx.x * STRIPE_LL]) : 0ll;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
        if (threadIdx.x == 0 && d < 0) {
            flags[C_TLAST] = t;
            flags[C_PEND] = alpha;
            flags[C_ACCEPTED] += 1;
        }
        if (threadIdx.x == 0) {
            flags[C_MOVES_RUN] += 1;
            // for the next move: the core list is empty again, the solver's barrier words are clear
            for (int s = 0; s < EXPAND_CORE_SHARDS; ++s) flags[C_CORE + s] = 0;
            for (int s = C_XCD; s < C_TOP + C_LINE; ++s) flags[s] = 0;
            flags[C_ARRIVE] = 0;
        }
    }
}

// The labels of a pending accepted move, when no k_move_setup follows (end of a cycle).
__global__ void __launch_bounds__(256)
k_apply_pending(int n, const int* __restrict__ cost, int L, int* __restrict__ label,
                int* __restrict__ cur_cost, const unsigned char* __restrict__ took,
                const int* __restrict__ flags, int /* in_front_of_a_batch */)
{
    const int pa = flags[C_PEND];
    if (pa < 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && took[i]) {                                      // applyNewLabeling, :423-441
        label[i] = pa;
        cur_cost[i] = cost[(size_t)i * L + pa];
    }
}


// ---------------------------------------------------------------------------
// r06: CONCURRENT ALPHA-MOVES.  oneExpansionIteration (GCoptimization.cpp:1278-1289) runs the expansions on alpha = 0, 1, ...
// one after the other, each on the labeling its predecessor left.  Here K consecutive moves are solved TOGETHER on the same
// labeling f (one launch of each move kernel, context = blockIdx.y), and k_batch_commit then walks them in the reference's order
// and keeps a move's result only when it is provably what the sequential order would have given:
//
//   Let the accepted predecessors of move beta in this batch have changed the sites S (f -> f').  Write W_p = potts * sum_q w_pq
//   and call p UNARY-KEPT for beta under a labeling g when  D_beta(p) - D_{g_p}(p) > W_p  (or g_p = beta: p is not a variable
//   of the move at all): switching p to beta costs more in its data term than all its n-links together can return, so p keeps
//   its label in EVERY minimum cut.  If every p in S u N(S) is unary-kept for beta under f AND under f', then in both problems
//   — beta on f, beta on f' — those sites stay, the sites outside S u N(S) see the same unary terms, the same n-links among
//   themselves and the same (unchanged) labels on the N(S) side of their other n-links: the two problems have the same set of
//   minimum cuts, hence the same canonical one (the read-out is BK's minimal sink side, SURVEY A-1), the same sites take beta,
//   and — every term that changes touches only sites whose labels agree in f and f' — the same energy gain, so the same verdict.
//   A move that fails the test is not kept: it becomes the first move of the next batch, solved on the labeling as it then stands.
//   The test for (j, k) reads the labeling the batch started from, j's list of moving sites and the two labels — not which other
//   moves are accepted — so k_batch_check evaluates all pairs in parallel and k_batch_commit only walks a K x K bit table.
//
// A move that was skipped as idempotent when the batch began (move_is_skipped: nothing accepted since its label's last
// expansion) stays skipped only while no predecessor in its batch is accepted; otherwise it is re-run too.  Labels, energies and
// cycle counts are therefore the sequential ones (every parity test, golden file and stress tool runs through this path).
// ---------------------------------------------------------------------------

// In front of a batch: the contexts beyond 0 see the control words context 0 — the global ones — holds now.
__global__ void k_batch_prep(MoveBatch b, int* __restrict__ bctl)
{
    int* G = b.c[0].flags;
    const int k = threadIdx.x;
    if (k == 0) { bctl[B_TLAST0] = G[C_TLAST]; G[C_PEND] = -1; }      // (whatever was pending has been applied: k_apply_pending, or k_batch_commit)
    if (k < EXPAND_MAX_CTX) bctl[B_BAD + k] = 0;
    if (k >= 1 && k < b.count) {
        int* f = b.c[k].flags;
        f[C_TLAST] = G[C_TLAST];
        f[C_PEND] = -1;
        f[C_ERROR] = G[C_ERROR];
    }
}

__device__ __forceinline__ bool batch_move_skipped(const MoveCtx& c, int tlast0, int L)
{
    return c.flags[C_ERROR] != 0 || (c.t >= L && tlast0 <= c.t - L);           // what every launch of this move evaluated
}

// Behind a batch's k_delta, in parallel over everything: for every move j of the batch that would be accepted and every LATER move
// k, does k pass the test above against what j changes?  The test looks only at the labeling the batch started from, at j's took
// list and at the two labels — nothing in it depends on which other moves end up accepted — so all pairs are checked at once:
// bit k of bctl[B_BAD + j] = "move k must not be kept if move j is".  Context blockIdx.y = j; 16 lanes per site of j's list walk
// the site and its neighbours.  (The labels of the batch are consecutive, so a site's costs for all later moves are contiguous.)
__global__ void __launch_bounds__(256)
k_batch_check(Graph g, const int* __restrict__ cost, int L, int potts, const int* __restrict__ label, const int* __restrict__ cur_cost,
              MoveBatch b, int* __restrict__ bctl)
{
    const int j = blockIdx.y;
    if (j + 1 >= b.count) return;                        // nothing comes behind the last move
    const MoveCtx& cj = b.c[j];
    const int tlast0 = bctl[B_TLAST0];
    if (batch_move_skipped(cj, tlast0, L) || cj.flags[C_PEND] < 0) return;       // skipped or rejected: it changes nothing
    const int nt = cj.flags[C_TOOK_N];
    const int sub = threadIdx.x % LPN;
    const int alpha_j = cj.alpha, later = b.count - 1 - j;
    unsigned bad = 0;
    for (int idx = blockIdx.x * SITES_PER_BLOCK + (int)threadIdx.x / LPN; idx < nt; idx += gridDim.x * SITES_PER_BLOCK) {
        const int p = cj.took_list[idx];
        const int k0 = g.rowptr[p], k1 = g.rowptr[p + 1];
        // lane `sub` serves the neighbours k0 + sub, k0 + sub + 16, ...; lane 0 also the site itself (slot -1)
        for (int a = k0 + sub - (sub == 0 ? 1 : 0); a < k1; a = (a < k0 ? k0 : a + LPN)) {
            const int s_ = a < k0 ? p : g.col[a];
            const long long W = (long long)potts * g.wsum[s_];
            const int lb = label[s_];
            const long long Dl = cur_cost[s_];
            const int* row = cost + (size_t)s_ * L + (alpha_j + 1);
            const long long Dj = a < k0 ? (long long)cost[(size_t)s_ * L + alpha_j] : 0;
            for (int q = 0; q < later; ++q) {
                const long long Db = row[q];
                bool ok = lb == alpha_j + 1 + q || Db - Dl > W;                    // under the labeling the batch started from
                if (ok && a < k0) ok = Db - Dj > W;                               // the changed site itself: also under its new label
                if (!ok) bad |= 1u << (j + 1 + q);
            }
        }
    }
    // (moves skipped when the batch began have no verdict to keep: k_batch_commit handles them)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) bad |= (unsigned)__shfl_xor((int)bad, m, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicOr((unsigned*)&bctl[B_BAD + j], bad);
}

// ... then the commit, in the reference's order (see above): every workgroup derives the same decisions from the same words —
// none of which is written before all workgroups have read them — and applies its share of the accepted moves' took lists
// (applyNewLabeling, :423-441); the LAST workgroup to finish records the outcome in the global control words, publishes it to the
// host and prepares the contexts for the next batch.
struct CtxFlags { int* p[EXPAND_MAX_CTX]; };       // the control words of ALL contexts of the engine (a batch may use fewer)
__global__ void __launch_bounds__(256)
k_batch_commit(const int* __restrict__ cost, int L, int* __restrict__ label, int* __restrict__ cur_cost, MoveBatch b,
               int* __restrict__ bctl, int* __restrict__ h_batch, CtxFlags all_flags, int n_ctx_all)
{
    __shared__ int s_first, s_err, s_last_t, s_n_acc;
    __shared__ unsigned s_acc;
    // the words the walk below looks at, fetched by one lane per move (the walk itself is sequential: on one lane's own loads it
    // cost a memory round trip per move and word, 12 us for sixteen moves)
    __shared__ int s_ek[EXPAND_MAX_CTX], s_pend[EXPAND_MAX_CTX], s_nt[EXPAND_MAX_CTX];
    __shared__ unsigned s_bad[EXPAND_MAX_CTX];
    __shared__ int s_g[5];                               // B_TLAST0, the global C_ERROR / C_TLAST / C_ACCEPTED, B_SEQ: written by this launch's LAST workgroup only
    if ((int)threadIdx.x < b.count) {
        const int* f = b.c[threadIdx.x].flags;
        s_ek[threadIdx.x] = f[C_ERROR];
        s_pend[threadIdx.x] = f[C_PEND];
        s_nt[threadIdx.x] = f[C_TOOK_N];
        s_bad[threadIdx.x] = (unsigned)bctl[B_BAD + threadIdx.x];
    } else if (threadIdx.x >= 64 && threadIdx.x < 69) {
        const int q = threadIdx.x - 64;
        const int* G = b.c[0].flags;
        s_g[q] = q == 0 ? bctl[B_TLAST0] : q == 1 ? G[C_ERROR] : q == 2 ? G[C_TLAST] : q == 3 ? G[C_ACCEPTED] : bctl[B_SEQ];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tlast0 = s_g[0];
        unsigned accepted = 0;
        int first_invalid = b.count, err = 0, last_t = -1, n_acc = 0;
        for (int k = 0; k < b.count; ++k) {
            const MoveCtx& c = b.c[k];
            const int ek = s_ek[k];
            if (ek) err = ek;
            const bool skipped = ek != 0 || (c.t >= L && tlast0 <= c.t - L);
            bool valid = true;
            if (accepted) {
                if (skipped) valid = false;                  // no longer idempotent: something in front of it changed the labeling
                else
                    for (int j = 0; j < k; ++j)
                        if ((accepted >> j & 1u) && (s_bad[j] >> k & 1u)) valid = false;
            }
            if (!valid) { first_invalid = k; break; }
            if (!skipped && s_pend[k] >= 0) { accepted |= 1u << k; last_t = c.t; ++n_acc; }
        }
        s_first = first_invalid; s_err = err; s_acc = accepted; s_last_t = last_t; s_n_acc = n_acc;
    }
    __syncthreads();
    const unsigned accepted = s_acc;
    for (int k = 0; k < b.count; ++k) {
        if (!(accepted >> k & 1u)) continue;
        const MoveCtx& c = b.c[k];
        const int nt = s_nt[k], a = c.alpha;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nt; i += gridDim.x * 256) {
            const int p = c.took_list[i];
            label[p] = a;
            cur_cost[p] = cost[(size_t)p * L + a];
        }
    }
    // The global control words, the publication and the next batch's preparation are the LAST workgroup's to write: every workgroup
    // has derived its decisions from those words and must have finished reading them.
    __shared__ int s_is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_is_last = atomicAdd(&bctl[B_TICKET], 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (s_is_last && threadIdx.x == 0) {
        bctl[B_TICKET] = 0;
        int* G = b.c[0].flags;
        // (the global words as they stood when the launch began — nobody but this thread writes them in it)
        const int g_err = s_g[1] ? s_g[1] : s_err;
        // (context 0's k_delta wrote the global words for its own move itself: count the others)
        const int g_tlast = s_last_t > s_g[2] ? s_last_t : s_g[2];
        const int seq = s_g[4] + 1;
        if (g_err != s_g[1]) G[C_ERROR] = g_err;
        if (g_tlast != s_g[2]) G[C_TLAST] = g_tlast;
        G[C_ACCEPTED] = s_g[3] + s_n_acc - (int)(accepted & 1u);
        bctl[B_SEQ] = seq;
        h_batch[HB_FIRST_INVALID] = s_first;
        h_batch[HB_TLAST] = g_tlast;
        h_batch[HB_ERROR] = g_err;
        h_batch[HB_ACCEPTED_HERE] = s_n_acc;
        h_batch[HB_DIRTY] = 0;
        // the host POLLS the sequence word (run_expansion): everything above — and every label the other workgroups wrote, which
        // their fences ordered before their tickets — is visible to it before the word changes
        __threadfence_system();
        __hip_atomic_store(&h_batch[HB_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // ... and what k_batch_prep would do in front of the NEXT batch, whatever that batch will hold (every context, not just the
        // ones it uses): a batch that follows a commit directly needs no launch of its own for it
        bctl[B_TLAST0] = g_tlast;
        G[C_PEND] = -1;
        for (int k = 0; k < EXPAND_MAX_CTX; ++k) bctl[B_BAD + k] = 0;
        for (int k = 1; k < n_ctx_all; ++k) {
            int* f = all_flags.p[k];
            f[C_TLAST] = g_tlast;
            f[C_PEND] = -1;
            f[C_ERROR] = g_err;
        }
    }
}

// The solver's statistics of the contexts beyond 0 into context 0's words (which k_publish mirrors to the host).  One thread
// per word, the contexts in sequence.
__global__ void k_stats_merge(MoveBatch b)
{
    const int t = threadIdx.x;
    int* G = b.c[0].flags;
    long long* A = b.c[0].acc;
    const int fw[5] = { C_FLOW_MOVES, C_MOVES_SOLVED, C_MOVES_RUN, C_CORE_MAX, C_XCD_USED };
    if (t < 5) {
        const int w = fw[t];
        int v = G[w];
        for (int k = 1; k < b.count; ++k) {
            int* f = b.c[k].flags;
            if (t < 3) v += f[w]; else if (f[w] > v) v = f[w];
            f[w] = 0;
        }
        G[w] = v;
    } else if (t >= 8 && t - 8 + A_CORE_SUM < A_COUNT) {
        const int w = t - 8 + A_CORE_SUM;
        long long v = A[w];
        for (int k = 1; k < b.count; ++k) {
            long long* a = b.c[k].acc;
            if (w == A_MAX_WAIT) { if (a[w] > v) v = a[w]; } else v += a[w];
            a[w] = 0;
        }
        A[w] = v;
    }
}

// compute_energy = data + smooth (:953-956; giveSmoothEnergyInternal :267-286)
__global__ void __launch_bounds__(256)
k_energy(Graph g, int potts, const int* __restrict__ label, const int* __restrict__ cur_cost,
         int* __restrict__ flags, long long* __restrict__ acc)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[C_PEND] = -1;     // k_apply_pending ran in front of this launch
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
    if (threadIdx.x == 0)
        atomicAdd((unsigned long long*)&acc[A_ENERGY_S + (blockIdx.x % STRIPES) * STRIPE_LL], (unsigned long long)s[0]);
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
// them (no hipMemcpy calls, which cost tens of microseconds each for a few bytes), then the host
// waits for the stream.
__global__ void k_publish(const int* __restrict__ flags, long long* __restrict__ acc,
                          int* __restrict__ h_flags, long long* __restrict__ h_acc)
{
    const int t = threadIdx.x;
    if (t < EXPAND_HOST_WORDS) h_flags[t] = flags[t];
    long long en = t < STRIPES ? acc[A_ENERGY_S + t * STRIPE_LL] : 0ll;      // k_energy's (or k_argmin_labels') stripes
    if (t < STRIPES) acc[A_ENERGY_S + t * STRIPE_LL] = 0;                     // the next k_energy accumulates from zero
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) en += __shfl_xor(en, m, 64);
    if (t < A_COUNT) h_acc[t] = t == A_ENERGY ? en : acc[t];
}

static hipError_t fetch(ExpandWork& w, ExpandStats& st, hipStream_t s)
{
    ++st.host_syncs;
    ++st.launches;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, w.flags, w.acc, w.h_flags_dev, w.h_acc_dev);
    RET_IF(hipGetLastError());
    return hipStreamSynchronize(s);
}

hipError_t solver_blocks_per_cu(int* blocks)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, (const void*)k_solve, SOLVE_THREADS, sizeof(int) * F_FIELDS * SOLVE_ROWS);
}

hipError_t run_expansion(const Graph& g, const int* cost, int L, int potts, ExpandWork& w,
                         int max_cycles, ExpandStats* st, hipStream_t s)
{
    const dim3 grid1((g.n + 255) / 256), blk(256);                          // one thread per site
    const dim3 grid((g.n + SITES_PER_BLOCK - 1) / SITES_PER_BLOCK);        // LPN lanes per site, 256 threads
    // whole-graph launches of a move: one lane per site first (spw sites per wave), then LPN lanes for the sites that need them
    // (setup and reduction: a wave with many sites to walk serves them four at a time, so more sites per wave pay only once the
    // launch would otherwise need several rounds of residency — 16 moves x 50 000 sites at 16 per wave are 50 000 waves, six
    // rounds: loop of the reference's route at configs[4] 0.363 -> 0.349 s with 64; at 20 000 sites no difference.  0 = by size)
    auto spw_for = [&](int count) {
        if (w.batch_spw == 16 || w.batch_spw == 32 || w.batch_spw == 64) return w.batch_spw;
        const long long site_moves = (long long)count * g.n;
        return site_moves > (1ll << 19) ? 64 : site_moves > (1ll << 18) ? 32 : 16;
    };
    ExpandStats stats = {};
    // the contexts: 0 = the work area's own buffers (and the global control words), k = w.ctx[k - 1]
    const int n_ctx = std::max(1, std::min(w.n_ctx, EXPAND_MAX_CTX));
    MoveBatch all{};
    all.count = n_ctx;
    for (int k = 0; k < n_ctx; ++k) {
        MoveCtx& c = all.c[k];
        if (k == 0) { c.cap = w.cap; c.sent = w.sent; c.excess = w.excess; c.sink_cap = w.sink_cap; c.height = w.height; c.decided = w.decided;
                      c.took = w.took; c.core = w.core; c.flags = w.flags; c.acc = w.acc; c.took_list = w.took_list0; }
        else { const ExpandWork::Ctx& x = w.ctx[k - 1];
               c.cap = x.cap; c.sent = x.sent; c.excess = x.excess; c.sink_cap = x.sink_cap; c.height = x.height; c.decided = x.decided;
               c.took = x.took; c.core = x.core; c.flags = x.flags; c.acc = x.acc; c.took_list = x.took_list; }
    }
    hipLaunchKernelGGL(k_ctl_init, dim3(n_ctx), dim3(64), 0, s, all);
    RET_IF(hipGetLastError());
    ++stats.launches;
    auto finish_cycle = [&]() -> hipError_t {             // the contexts' solver statistics, then the control words to the host
        if (n_ctx > 1) { hipLaunchKernelGGL(k_stats_merge, dim3(1), dim3(64), 0, s, all); ++stats.launches; }
        return fetch(w, stats, s);
    };

    if (g.nnz == 0) {                                  // solveSpecialCases, :470-491
        RET_IF(launch_argmin_labels(cost, L, g.n, w.label, w.acc, s));
        RET_IF(fetch(w, stats, s));
        stats.energy = w.h_acc[A_ENERGY];
        if (st) *st = stats;
        return hipSuccess;
    }

    long long energy = 0, old_energy = 0;
    hipLaunchKernelGGL(k_energy, grid, blk, 0, s, g, potts, w.label, w.cur_cost, w.flags, w.acc);     // :1036
    RET_IF(fetch(w, stats, s));
    stats.launches += 1;
    energy = w.h_acc[A_ENERGY];

    SolveParams sp{ w.reduce_rounds, w.relax_rounds, w.push_cycles, w.push_phases, w.push_mult > 0 ? w.push_mult : 1, 1 << 20, w.cascade_iters,
                    w.barrier_timeout_ticks > 0 ? (unsigned long long)w.barrier_timeout_ticks : 300000000ull,
                    w.barrier_first_timeout_ticks > 0 ? (unsigned long long)w.barrier_first_timeout_ticks : 300000000ull };
    int solve_grid = w.solve_grid > 0 ? w.solve_grid : 128;
    // slots per solver row for a core of all n sites, and the LDS that holds their scalars
    const int mslots = std::max(1, (g.n + solve_grid * SOLVE_ROWS - 1) / (solve_grid * SOLVE_ROWS));
    const size_t solve_lds = sizeof(int) * F_FIELDS * (size_t)mslots * SOLVE_ROWS;
    if (solve_lds > 128 * 1024) return hipErrorOutOfMemory;     // > 1.3 M sites at 256 workgroups
    if (solve_lds > 48 * 1024)
        RET_IF(hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds));

    // The launches of `count` consecutive moves (labels alpha0 ..., move indices t0 ...) solved on the same labeling, contexts
    // 0 .. count - 1.  count == 1: a move alone, the sequential form.
    CtxFlags all_flags{};
    for (int k = 0; k < n_ctx; ++k) all_flags.p[k] = all.c[k].flags;
    bool prepared = false;       // the contexts beyond 0 hold the global control words as they stand (nothing but batches has run since)
    auto enqueue_moves = [&](int alpha0, int t0, int count, int cycle) -> hipError_t {
        MoveBatch b = all;
        b.count = count;
        for (int k = 0; k < count; ++k) {
            MoveCtx& c = b.c[k];
            c.alpha = alpha0 + k;
            c.t = t0 + k;
            c.trace = (w.trace && c.t < w.trace_moves) ? w.trace : nullptr;
            c.detail = (w.trace && c.t == w.detail_move) ? w.trace + 8 * (size_t)w.trace_moves : nullptr;
            c.saved_flow = w.saved_flow ? w.saved_flow + (size_t)c.alpha * g.nnz : nullptr;
            c.saved_sink = w.saved_flow ? w.saved_sink + (size_t)c.alpha * g.n : nullptr;
            c.warm = (w.saved_flow && cycle > 1) ? 1 : 0;
            if (count == 1) c.took_list = nullptr;             // a move alone: nobody walks the list
        }
        const int spw = count > 1 ? spw_for(count) : SPW;
        const int spw_d = count > 1 ? SPW_MAX : SPW;                  // k_delta: few sites move, every lane looks at one
        const dim3 grid_w((unsigned)((g.n + 4 * spw - 1) / (4 * spw)), (unsigned)count);
        const dim3 grid_d((unsigned)((g.n + 4 * spw_d - 1) / (4 * spw_d)), (unsigned)count);
        // (a batch right behind another batch's commit finds the contexts prepared by that commit)
        if (count > 1 && !prepared) { hipLaunchKernelGGL(k_batch_prep, dim3(1), dim3(64), 0, s, b, w.bctl); ++stats.launches; }
        prepared = count > 1;
        hipLaunchKernelGGL(k_move_setup, grid_w, blk, 0, s, g, cost, L, potts, w.reduce_rounds > 0 ? 1 : 0, w.label, w.cur_cost, b, spw);
        if (w.reduce_launches > 1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reduce<false>), grid_w, blk, 0, s, g, L, b, w.reduce_rounds, 1, spw);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reduce<true>), grid_w, blk, 0, s, g, L, b, w.reduce_rounds, w.reduce_launches > 1 ? 0 : 1, spw);
        if (count == 1 && w.comp_out && t0 < w.comp_moves)             // diagnostic: components of the core the reduction left
            hipLaunchKernelGGL(k_core_components, dim3(1), dim3(1024), 0, s, g, L, t0, w.decided, w.flags, w.comp_scratch,
                               w.comp_scratch + g.n, w.comp_out + 16 * (size_t)t0);
        hipLaunchKernelGGL(k_solve, dim3((unsigned)solve_grid, (unsigned)count), dim3(SOLVE_THREADS), solve_lds, s, g, L, b, mslots, sp);
        hipLaunchKernelGGL(k_delta, grid_d, blk, 0, s, g, cost, L, potts, w.label, w.cur_cost, b, spw_d);
        if (count > 1) {
            hipLaunchKernelGGL(k_batch_check, dim3(32, (unsigned)count), blk, 0, s, g, cost, L, potts, w.label, w.cur_cost, b, w.bctl);
            hipLaunchKernelGGL(k_batch_commit, dim3(32), blk, 0, s, cost, L, w.label, w.cur_cost, b, w.bctl, w.h_batch_dev, all_flags, n_ctx);
            stats.launches += 2;
        }
        stats.launches += w.reduce_launches > 1 ? 5 : 4;
        if (w.reduce_rounds > 0) stats.reduce_launches += w.reduce_launches > 1 ? 2 : 1;
        return hipGetLastError();
    };

    // Batches: only with contexts to run them in, a label set worth it, and none of the per-move diagnostics that look at
    // context 0's state between the launches of a move.
    const bool can_batch = n_ctx > 1 && w.bctl && w.h_batch && L >= 2 && !w.comp_out;
    bool batching = can_batch && L >= w.batch_min_labels;       // ... from the first cycle on; smaller label sets from the second (see below)
    if (can_batch) {
        RET_IF(hipMemsetAsync(w.bctl, 0, sizeof(int) * B_WORDS, s));
        w.h_batch[HB_SEQ] = 0;                          // (no commit of an earlier expansion is in flight: each one ends behind a stream wait)
    }
    // What the host knows of C_TLAST (the index of the last accepted move): exact right behind a commit or a cycle's end,
    // unknown once a move has been enqueued on its own since.  A move t >= L with TLAST <= t - L is idempotent — the device
    // would return from each of its launches at once (move_is_skipped) — and is not launched at all when the host can tell.
    int host_tlast = -1;
    bool host_fresh = true;
    int bsize = n_ctx;                                  // moves of the next batch: halved when a batch keeps little, doubled when one is kept whole
    int batch_seq = 0;                                  // batches enqueued = the sequence number the next commit publishes (bctl[B_SEQ] starts at 0)
    int t = 0;
    for (int cycle = 1; cycle <= max_cycles; ++cycle) {
        old_energy = energy;
        int alpha = 0;
        // A first cycle from a cold start rewrites the labeling wholesale: with a handful of labels every accepted move
        // touches its successor's sites and nothing validates; from the second cycle on most moves change little or nothing.
        if (can_batch && cycle >= 2) batching = true;
        while (alpha < L) {
            if (host_fresh && t >= L && host_tlast <= t - L) {      // idempotent from here to the end of the cycle (nothing can be accepted in between)
                const int rest = L - alpha;
                stats.moves += rest; stats.host_skipped += rest;
                alpha += rest; t += rest;
                break;
            }
            int B = std::min(bsize, L - alpha);
            if (host_fresh && t + B > host_tlast + L && host_tlast + L > t && t >= L) B = host_tlast + L - t;   // stop in front of the first idempotent move
            if (!batching || B < 2) {
                RET_IF(enqueue_moves(alpha, t, 1, cycle));
                ++stats.moves; ++stats.solo_moves;
                ++alpha; ++t;
                host_fresh = false;
                continue;
            }
            if (!host_fresh) {
                // a move enqueued on its own may have left its labels pending (they are written lazily): the batch's moves all
                // read the labeling, so it is brought up to date first
                hipLaunchKernelGGL(k_apply_pending, grid1, blk, 0, s, g.n, cost, L, w.label, w.cur_cost, w.took, w.flags, 1);
                ++stats.launches;
            }
            RET_IF(enqueue_moves(alpha, t, B, cycle));
            ++stats.batches;
            ++stats.host_syncs;
            ++batch_seq;
            // The host needs four words of the commit before it can enqueue the next batch.  Waiting for the stream costs 15-20 us
            // a time (a thousand times per Process() on the reference's route); the commit writes its words to mapped host memory
            // and the sequence number last, behind a system-scope fence, so the host polls that word instead — for at most 2 ms
            // (a batch with large cores), then it waits for the stream as before.  Launches stay ordered by the stream either way.
            {
                volatile int* seq_word = w.h_batch + HB_SEQ;
                const auto t_poll = std::chrono::steady_clock::now();
                bool seen = false;
                for (int spin = 0;; ++spin) {
                    if (*seq_word == batch_seq) { seen = true; break; }
                    if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t_poll > std::chrono::milliseconds(2)) break;
                }
                if (seen) std::atomic_thread_fence(std::memory_order_acquire);
                else RET_IF(hipStreamSynchronize(s));
                if (w.h_batch[HB_SEQ] != batch_seq) return hipErrorUnknown;           // (cannot happen: the commit always publishes)
            }
            if (w.h_batch[HB_ERROR]) break;                           // the cycle's end reports it
            const int j = w.h_batch[HB_FIRST_INVALID];
            host_tlast = w.h_batch[HB_TLAST];
            host_fresh = true;
            stats.moves += j; stats.batch_committed += j;
            alpha += j; t += j;
            if (j < B) {
                // Move alpha failed its test: it heads the next batch — a batch's first move has no predecessor to be tested
                // against, so every batch keeps at least one move and no move is ever run twice in a row.  What came behind it
                // in this batch is solved again there.  A batch that kept little was mostly wasted work: the next is smaller.
                ++stats.batch_invalid;
                if (2 * j < B) bsize = std::max(2, bsize / 2);
            } else if (B == bsize) bsize = std::min(n_ctx, 2 * bsize);
        }
        hipLaunchKernelGGL(k_apply_pending, grid1, blk, 0, s, g.n, cost, L, w.label, w.cur_cost, w.took, w.flags, 0);
        hipLaunchKernelGGL(k_energy, grid, blk, 0, s, g, potts, w.label, w.cur_cost, w.flags, w.acc);
        stats.launches += 2;
        RET_IF(finish_cycle());
        stats.cycles = cycle;
        if (w.h_flags[C_ERROR]) {
            stats.energy = -(long long)w.h_flags[C_ERROR];
            if (st) *st = stats;
            return w.h_flags[C_ERROR] == ERR_OVERFLOW ? hipErrorInvalidValue : hipErrorLaunchTimeOut;
        }
        host_tlast = w.h_flags[C_TLAST];
        host_fresh = true;
        energy = w.h_acc[A_ENERGY];
        if (energy == old_energy) break;               // :1045
    }
    stats.energy = energy;
    stats.accepted = w.h_flags[C_ACCEPTED];
    stats.flow_moves = w.h_flags[C_FLOW_MOVES];
    stats.moves_solved = w.h_flags[C_MOVES_SOLVED];
    stats.moves_run = w.h_flags[C_MOVES_RUN];
    stats.core_sites = w.h_acc[A_CORE_SUM];
    stats.core_max = w.h_flags[C_CORE_MAX];
    stats.outer_iterations = w.h_acc[A_OUTER];
    stats.relax_intervals = w.h_acc[A_RELAX];
    stats.push_phases = w.h_acc[A_PUSH];
    stats.barriers = w.h_acc[A_BARRIERS];
    stats.solve_ms = (double)w.h_acc[A_TICKS] * 1e-5;  // 100 MHz ticks
    stats.barrier_ms = (double)w.h_acc[A_T_BAR] * 1e-5;
    stats.relax_ms = (double)w.h_acc[A_T_RELAX] * 1e-5;
    stats.push_ms = (double)w.h_acc[A_T_PUSH] * 1e-5;
    stats.tail_ms = (double)w.h_acc[A_T_TAIL] * 1e-5;
    stats.tail_rounds = w.h_acc[A_TAIL_ROUNDS];
    stats.max_barrier_wait_ms = (double)w.h_acc[A_MAX_WAIT] * 1e-5;
    if (st) *st = stats;
    return hipSuccess;
}

} // namespace mh
