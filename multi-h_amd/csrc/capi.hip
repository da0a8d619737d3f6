// capi.hip — implementation of the C ABI declared in include/multih_hip.h.
// Owns device buffers, the HIP stream and the per-kernel event timers; every
// computation is a launch of a gfx950 kernel from this directory.  There is no
// CPU fallback: without a usable HIP device mh_create fails (MH_ERR_NO_DEVICE).

#include "../../include/multih_hip.h"
#include "mh_kernels.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

using namespace mh;

static thread_local std::string g_err;

static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(MH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;        // elements
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int launches = 0;
    double total_ms = 0.0;
};

struct mh_engine {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // MultiH ctor members, M/MultiH.cpp:10-21
    double thr_F = 3.0, thr_H = 2.5, locality = 0.002, lambda = 0.5;
    int min_inliers = 0;

    int n = 0;
    bool have_aff = false, have_epi = false, have_graph = false;
    DevBuf<double> x1, y1, x2, y2, a11, a12, a21, a22;
    Epipolar epi{};

    // symmetric weighted graph
    std::vector<int> g_rowptr, g_col, g_w, g_rev;   // host copy of the symmetric graph (built here by the fallback path, else fetched on demand)
    bool g_host_valid = false;
    int g_nnz = 0;
    DevBuf<int> gb_deg, gb_start, gb_cursor, gb_raw, gb_mult, gb_uniq, gb_info, gb_hits_rp, gb_hits_col;   // graph.hip scratch
    int order_n = -1;                        // d_order holds the solver's site order for this many sites
    DevBuf<int> d_rowptr, d_col, d_w, d_rev;

    // model set
    int m = 0;
    bool have_samples = false;
    DevBuf<double> H, H_one;
    DevBuf<int> samples, counts;
    DevBuf<double> R;
    long long ldr = 0;
    DevBuf<int> C;                            // mh_cost_matrix
    long long ldc = 0;
    DevBuf<unsigned char> mask;
    DevBuf<double> moments, min_eig;
    DevBuf<double> cp_pts, cp_H, cp_out;     // mh_compat_trial_stats staging
    DevBuf<int> cp_begin, cp_tri;
    DevBuf<unsigned char> cp_ok;
    // greedy selection (select.hip): two candidate lists, control words, exchange buffers
    DevBuf<int> sel_orig[2], sel_counts, sel_rec, sel_scores, sel_gathered;
    DevBuf<double> sel_cand_H[2], sel_out_H;
    DevBuf<SelRecord> sel_records;             // [0] this rank's offer, [1 .. world] the gathered offers
    DevBuf<long long> sel_counter;
    DevBuf<unsigned long long> sel_keys;
    // transport of the sharded propose stage (mh_set_transport): stream-ordered (RCCL) or host-synchronised (test hook)
    int t_rank = 0, t_world = 1;
    mh_allgather_stream_fn t_stream_fn = nullptr;
    mh_allgather_dev_fn t_host_fn = nullptr;
    void* t_ctx = nullptr;
    // pipelined propose (mh_prefetch_dlt4): the spare batch and the second stream it is prepared on
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_side_pre = nullptr;
    int tune_score32_resident = 12;          // key 24: the FP32 pre-test score as a resident grid with n point slices (12: 1.98 ms against 2.14 hardware-dispatched at 50k x 100k, tools/score32_probe.py); 0 = hardware dispatch, -1 = ~37 500 items
    int occ_score32 = -1, occ_cost32 = -1;   // workgroups per compute unit of the resident score / cost kernels on THIS engine's device (-1: not asked yet)
    int tune_cost32_batched = 0;             // key 28: 1 = experiment: k_cost32 evaluates the near pairs of several models together (score32.hip, cost32_wg_batched; slower: 4.45 vs 4.10 ms)
    int tune_cost32_slice_major = 0;         // key 27: the resident cost-matrix kernel takes its items slice-major (experiment)
    int tune_sweep_slices = 0;               // key 26: > 0 = the resident sweep takes its items slice-major with this many point slices (experiment)
    int tune_dlt_variant = 0;                // key 25: 0 = by context (below), 1 = the LDS-staged proposer everywhere, 2 = the register-resident one everywhere (same bits)
    int tune_cost32_resident = 8;            // key 23: the int32 cost matrix as a resident grid with n point slices (8: 4.12 ms against 4.25 hardware-dispatched at 50k x 100k, tools/cost32_probe.py); 0 = hardware dispatch, -1 = ~37 500 items
    int tune_stream_shift = 0;               // key 22 (experiment): dummy streams created in front of the second / third stream (shifts their hardware queue / pipe)
    std::vector<hipStream_t> dummy_streams;
    int tune_dlt_first = 1;                  // key 20: the sweep waits until the second stream has reached the pending DLT's dispatch (1) or not (0)
    // up to two prefetched batches wait in a FIFO (r04: with the batch after next prepared too, the DLT a sweep has to wait
    // for was dispatched a whole sweep earlier — nothing is handed from stream to stream between two sweeps)
    static constexpr int PF_DEPTH = 2;
    DevBuf<double> pf_H[PF_DEPTH];
    DevBuf<int> pf_samples[PF_DEPTH];
    int pf_m[PF_DEPTH] = { 0, 0 };
    hipEvent_t pf_ev[PF_DEPTH] = { nullptr, nullptr };   // recorded behind the slot's DLT on the second stream
    int pf_head = 0, pf_count = 0;                       // oldest queued slot, number of queued batches
    // best model of a scored batch (mh_select_best).  The (all-gather +) arg-max of batch i runs on a third stream behind an
    // event of sweep i, so sweep i+1 starts at once: the batch's counts buffer goes to the exchange and the next sweep
    // writes the other one (r04)
    hipStream_t xchg_stream = nullptr;
    hipEvent_t ev_sweep = nullptr, ev_x[3] = { nullptr, nullptr, nullptr };
    // two counts buffers wait in a FIFO beside the current one: a buffer handed to exchange k comes back for sweep k + 3, so an
    // exchange has TWO sweeps to finish in before anything waits for it (with one spare buffer it had one)
    DevBuf<int> counts_alt[2];
    int counts_alt_wait[2] = { -1, -1 };          // which ev_x the buffer's last exchange records (-1: none)
    long long xchg_calls = 0;                  // exchanges enqueued on xchg_stream so far (parity selects ev_x)
    long long models_seq = 0, best_models_seq = -1;   // model-set generation; the one the last mh_select_best result belongs to
    bool counts_zeroed_alt[2] = { false, false };   // the same for the two waiting buffers
    bool counts_zeroed = false;                // the current counts buffer was cleared behind the exchange that last read it (the next sweep skips its memset)
    bool counts_fresh = false;                 // the current counts buffer holds the scores of the current model set (a scoring call wrote it)
    bool xchg_pending = false;                 // something enqueued on xchg_stream since the last host wait for it
    DevBuf<unsigned long long> best_key;
    int* h_best = nullptr;                     // mapped pinned: count, global index, sequence number
    int* h_best_dev = nullptr;
    int best_seq = 0;
    int* h_sel = nullptr;                      // mapped pinned mirror of the control words
    int* h_sel_dev = nullptr;
    long long copies_h2d = 0, copies_d2h = 0;  // explicit host<->device copies issued by mh_select_greedy (mh_get_copy_stats)

    // epipolar front half
    int fm = 0;
    DevBuf<double> fund, fund_one;
    DevBuf<int> fund_samples, fund_counts, fund_inl;
    DevBuf<unsigned char> fund_mask, ref_keep, ref_in;
    DevBuf<double> ref_out;

    // reference-style initialisation
    DevBuf<double> loc_H, loc_feat, ms_data, ms_mean;
    DevBuf<int> ms_votes, ms_out, ms_list, ms_pcnt, ms_heads, ms_tickets;
    DevBuf<double> ms_partial, ms_partial2;
    DevBuf<unsigned long long> ms_ticks;     // MULTIH_MS_STATS: phase ticks of the persistent kernel
    DevBuf<int> ms_ctl, ms_pcnt2;            // the persistent tail of a mean-shift batch (meanshift.hip, k_ms_persist)
    int ms_persist_per_cu = -1, ms_persist_per_cu6 = -1;   // workgroups of k_ms_persist<10> / <6> a compute unit holds (-1: not queried; a failed query is not kept)
    int tune_ms_persist = 12;                // key 29: the tail runs persistently once at most this many climbs are left (0 = never)
    long long ms_persist_launches = 0, ms_persist_fallbacks = 0, ms_rounds = 0;

    // labeling
    int cost_L = 0;
    DevBuf<int> cost, labels_in, labels_pts, label_counts;
    DevBuf<int> ew_label, ew_cur, ew_cap, ew_sent, ew_excess, ew_sink, ew_height, ew_decided, ew_flags, ew_core, ew_trace, ew_saved, d_order, d_wsum;
    int comp_moves = 0;                      // > 0: component diagnostic of the first n moves' cores (mh_set_tuning key 21)
    DevBuf<int> ew_comp, ew_comp_out;
    int trace_moves = 0;                     // > 0: k_solve logs 8 ints per move (mh_set_tuning key 8)
    int detail_move = -1;                    // move whose relabels are logged one by one (key 9)
    DevBuf<unsigned char> ew_took;
    int cu_count = 256;
    DevBuf<int> sweep_ctl;                   // work counter + exit counter of the resident sweep (cleared by the launch itself)
    int sweep_wg_per_cu = -1;                // workgroups of the materialising sweep a compute unit holds (-1 = not queried yet)
    int tune_sweep_headroom = 0;             // key 19: workgroup slots the resident sweep leaves free beyond its own occupancy (-1 = hardware dispatch)
    int solve_grid_max = 0;                  // workgroups of the solver launch that can be resident at once (0 = not queried yet)
    double longest_barrier_wait_ms = 0.0;    // longest wait at a grid barrier any completed expansion of this engine has seen
    int last_expand_retries = 0;             // restarts of the last expansion after a barrier timeout (shared GPU)
    long long expand_retries_total = 0;      // ... of all expansions of this engine (mh_get_expand_stats word 22)
    int last_solve_grid = 0;                 // workgroups of the solver launch in the attempt that completed
    int inject_select_failure = 0;           // test hook (key 18): the n-th scoring round of the coming greedy selections fails on this rank
    int inject_barrier_timeouts = 0;         // test hook: the next n expansions' first attempts count as timed out
    DevBuf<long long> ew_acc;
    int* h_flags = nullptr;
    MeanShiftResultBlock* h_ms = nullptr;      // mapped pinned result block of the mean-shift climbs
    MeanShiftResultBlock* h_ms_dev = nullptr;
    int* h_ms_list = nullptr;                  // pinned staging for the first MS_LIST_PREFIX (row, votes) pairs of every climb of a batch
    long long* h_acc = nullptr;
    int* h_flags_dev = nullptr;
    long long* h_acc_dev = nullptr;
    DevBuf<double> sel_pts[4];               // the active points of a greedy-selection round, packed (select.hip)
    DevBuf<int> sel_pack_count;
    DevBuf<int> knn_tmp, knn_part_i;
    DevBuf<float> knn_part_d;

    bool profiling = false;
    KernelTimer timers[MH_K_COUNT_];
    int tune_residual_variant = 0;
    int tune_ld = 0;                         // measurement builds only: row pitch of R in doubles (0 = residual_ld)
    int tune_score_variant = 0;
    int residual_mode = MH_RESIDUAL_FORWARD;
    int tune_ms_batch = 6;                   // mean-shift climb iterations per host round trip
    int tune_reduce = 2;                     // dominance-reduction rounds per launch (0 = off); 2 measured best (loop 0.262 s at 4, 0.250 s at 2)
    int tune_reduce_launches = 1;            // reduction launches per move in front of the solver (1 = the compacting one alone, 2; loop 0.250 vs 0.254 s)
    int tune_recycle = 1;                    // from the second cycle on a label's max-flow starts from the flow its last expansion left (0 = off, A/B)
    int tune_expand[4] = { 128, 512, 1, 256 };
    int tune_cascade_iters = 2;              // key 17: passes of the dominance cascade inside the solver launch (0 = to the fixed point; 2 measured best: 15.6 -> 14.5 ms per LabelingStep at 50k sites)
    int tune_push_mult = 6;                  // push cycles per phase = this x (depth of the last relabel + 3)  // solver: relax rounds per barrier interval, push cycles per phase, push phases per relabel, workgroups
    ExpandStats last_expand{};

    double bbox[4] = { NAN, NAN, NAN, NAN };   // xmin xmax ymin ymax of the source points
    // FP32 pre-test of the score kernels (score32.hip): usable when every coordinate is finite and below 2^20
    bool coords32_ok = false;
    double absmax_x = NAN, absmax_y = NAN, absmax_dst = NAN;
    int tune_score32_tiling = 0;               // key 16: points per lane / models per workgroup of the pre-test kernel (schedule only)
    int tune_score32 = 1;                      // mh_set_tuning key 15: 0 = always the FP64 sweep (A/B; counts are equal by construction)
    DevBuf<float> H32;
    DevBuf<unsigned long long> fb_pairs;
    long long score_pairs = 0;                 // pairs scored through the pre-test since the last reset (mh_get_score_stats)
    Points pts() const { return Points{ x1.p, y1.p, x2.p, y2.p, n, bbox[0], bbox[1], bbox[2], bbox[3] }; }
};

namespace {

// No exception may cross the C ABI (include/multih_hip.h): host-side allocation failures and
// anything else thrown by the standard library become a status code with the text in mh_last_error.
template <typename Fn>
int guarded(Fn&& fn)
{
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        return fail(MH_ERR_INVALID, "out of host memory");
    } catch (const std::exception& ex) {
        return fail(MH_ERR_INVALID, std::string("internal error: ") + ex.what());
    } catch (...) {
        return fail(MH_ERR_INVALID, "internal error");
    }
}

struct ScopedTimer {
    mh_engine* e;
    int k;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(mh_engine* e_, int k_, hipStream_t on = nullptr) : e(e_), k(k_), st(on ? on : e_->stream)
    {
        if (!e->profiling) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, st);
    }
    ~ScopedTimer()
    {
        if (!a) return;
        (void)hipEventRecord(b, st);
        e->timers[k].pending.emplace_back(a, b);
    }
};

void resolve_timers(mh_engine* e)
{
    for (int k = 0; k < MH_K_COUNT_; ++k) {
        KernelTimer& t = e->timers[k];
        for (auto& pr : t.pending) {
            float ms = 0.f;
            if (hipEventSynchronize(pr.second) == hipSuccess &&
                hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                t.total_ms += ms;
                t.launches += 1;
            }
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        t.pending.clear();
    }
}

// Every entry point runs on the engine's device, whatever the calling thread's current HIP
// device is (a host that drives several engines, or torch, may have switched it).
int enter(mh_engine* e)
{
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    HIPCHK(hipSetDevice(e->device));
    return MH_OK;
}

int require_points(mh_engine* e)
{
    int rc = enter(e);
    if (rc) return rc;
    if (e->n <= 0) return fail(MH_ERR_NOT_SET, "correspondences are not set");
    return MH_OK;
}

int require_models(mh_engine* e)
{
    int rc = require_points(e);
    if (rc) return rc;
    if (e->m <= 0) return fail(MH_ERR_NOT_SET, "model set is empty");
    return MH_OK;
}

// The scoring entry points on a rank whose shard of a batch is EMPTY (a transport is installed, more ranks than
// hypotheses): nothing to score, but the call counts — mh_select_best decides "is this a new exchange" from what has been
// scored since the last one, and that has to come out the same on every rank (r04 advisor finding).  *empty = the call is
// done (outputs untouched).
int require_models_or_empty_shard(mh_engine* e, bool* empty)
{
    *empty = false;
    int rc = require_points(e);
    if (rc) return rc;
    if (e->m <= 0 && (e->t_stream_fn || e->t_host_fn)) { *empty = true; e->counts_fresh = true; return MH_OK; }
    if (e->m <= 0) return fail(MH_ERR_NOT_SET, "model set is empty");
    return MH_OK;
}

// (a re-allocated counts buffer is not the one an exchange cleared)
hipError_t reserve_counts(mh_engine* e, size_t n)
{
    if (n <= e->counts.cap) return hipSuccess;
    e->counts_zeroed = false;
    return e->counts.reserve(n);
}

// Host wait for everything the engine has enqueued: the main stream, the DLT prefetch on the second stream and the
// (all-gather +) arg-max on the third.
int quiesce(mh_engine* e)
{
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->side_stream) HIPCHK(hipStreamSynchronize(e->side_stream));
    if (e->xchg_stream) HIPCHK(hipStreamSynchronize(e->xchg_stream));
    e->xchg_pending = false;
    return MH_OK;
}

// The main stream waits (on the device) for the last exchange enqueued on the third stream: called by whatever is about to
// touch the buffers that exchange reads or writes (gathered scores, best key, the counts buffer it was given).
int join_xchg(mh_engine* e)
{
    if (e->xchg_pending && e->xchg_calls > 0)
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_x[(e->xchg_calls - 1) % 3], 0));
    return MH_OK;
}

// Runs fn(begin, end) over [0, n) on up to 16 host threads (one below 20 000 items); fn must only write what its range owns.
template <typename Fn>
static void host_parallel_for(int n, Fn fn)
{
    unsigned t = std::thread::hardware_concurrency();
    if (t == 0) t = 1;
    if (t > 16) t = 16;
    if (n < 20000) t = 1;
    if (t == 1) { fn(0, n); return; }
    std::vector<std::thread> pool;
    const int chunk = (n + (int)t - 1) / (int)t;
    for (unsigned k = 0; k < t; ++k) {
        const int b = (int)k * chunk, en = std::min(n, b + chunk);
        if (b < en) pool.emplace_back(fn, b, en);
    }
    for (auto& th : pool) th.join();
}

// Symmetric weighted CSR with reverse-arc index from a directed hit list.
// setNeighbors semantics (GCoptimization.cpp:1656-1681, M/MultiH.cpp:532-540):
// every directed hit i->j (j != i) appends j to i's list and i to j's list, so the
// pair weight is mult(i,j) = #[i->j] + #[j->i]   (SURVEY A-2).
// Rows are built by several host threads (entries land in a row in any order and are sorted there, so the
// result does not depend on the number of threads).  Fallback of the device construction (graph.hip) for rows too
// long for its per-row LDS sort; 50k points / 0.8 M hits: 38 ms on one thread, 17.6 ms on 16.
int build_sym_graph(mh_engine* e, const int* rowptr, const int* col, int n)
{
    for (int i = 0; i < n; ++i)
        if (rowptr[i + 1] < rowptr[i]) return fail(MH_ERR_INVALID, "rowptr must be non-decreasing");
    std::unique_ptr<std::atomic<int>[]> cnt(new std::atomic<int>[(size_t)n + 1]);
    for (int i = 0; i <= n; ++i) cnt[i].store(0, std::memory_order_relaxed);
    std::atomic<int> bad(0);
    host_parallel_for(n, [&](int b, int en) {
        for (int i = b; i < en; ++i)
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                const int j = col[k];
                if (j < 0 || j >= n) { bad.store(1, std::memory_order_relaxed); continue; }
                if (j == i) continue;
                cnt[i].fetch_add(1, std::memory_order_relaxed);
                cnt[j].fetch_add(1, std::memory_order_relaxed);
            }
    });
    if (bad.load()) return fail(MH_ERR_INVALID, "neighbour index out of range");
    std::vector<long long> start(n + 1, 0);
    for (int i = 0; i < n; ++i) start[i + 1] = start[i] + cnt[i].load(std::memory_order_relaxed);
    if (start[n] > 0x7fffffffll) return fail(MH_ERR_OVERFLOW, "too many neighbour entries");
    std::vector<int> raw((size_t)start[n]);
    for (int i = 0; i < n; ++i) cnt[i].store(0, std::memory_order_relaxed);       // now: entries placed in row i
    host_parallel_for(n, [&](int b, int en) {
        for (int i = b; i < en; ++i)
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                const int j = col[k];
                if (j == i) continue;
                raw[(size_t)(start[i] + cnt[i].fetch_add(1, std::memory_order_relaxed))] = j;
                raw[(size_t)(start[j] + cnt[j].fetch_add(1, std::memory_order_relaxed))] = i;
            }
    });
    // sort each row and fold duplicates in place: (column, multiplicity) pairs at the head of the row
    std::vector<int> mult((size_t)start[n]);
    e->g_rowptr.assign(n + 1, 0);
    host_parallel_for(n, [&](int b, int en) {
        for (int i = b; i < en; ++i) {
            int* rb = raw.data() + start[i];
            int* re = raw.data() + start[i + 1];
            int* mb = mult.data() + start[i];
            std::sort(rb, re);
            int u = 0;
            for (int* p = rb; p < re;) {
                int* q = p;
                while (q < re && *q == *p) ++q;
                rb[u] = *p;
                mb[u] = (int)(q - p);
                ++u;
                p = q;
            }
            e->g_rowptr[i + 1] = u;
        }
    });
    for (int i = 0; i < n; ++i) e->g_rowptr[i + 1] += e->g_rowptr[i];
    const int nnz = e->g_rowptr[n];
    e->g_col.resize(nnz);
    e->g_w.resize(nnz);
    e->g_rev.assign(nnz, -1);
    host_parallel_for(n, [&](int b, int en) {
        for (int i = b; i < en; ++i) {
            const int len = e->g_rowptr[i + 1] - e->g_rowptr[i];
            std::copy(raw.data() + start[i], raw.data() + start[i] + len, e->g_col.data() + e->g_rowptr[i]);
            std::copy(mult.data() + start[i], mult.data() + start[i] + len, e->g_w.data() + e->g_rowptr[i]);
        }
    });
    host_parallel_for(n, [&](int b, int en) {
        for (int i = b; i < en; ++i)
            for (int k = e->g_rowptr[i]; k < e->g_rowptr[i + 1]; ++k) {
                const int j = e->g_col[k];
                const int* rb = e->g_col.data() + e->g_rowptr[j];
                const int* re = e->g_col.data() + e->g_rowptr[j + 1];
                e->g_rev[k] = (int)(std::lower_bound(rb, re, i) - e->g_col.data());
            }
    });
    return MH_OK;
}

// the solver's site order (expand.hip, k_reduce<true>): Fisher-Yates with the engine's counter RNG, fixed seed (results never depend on it)
static int upload_order(mh_engine* e)
{
    const int n = e->n;
    if (e->order_n == n) return MH_OK;
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    unsigned long long z = 0x6d682d6f72646572ull;
    for (int i = n - 1; i > 0; --i) {
        z += 0x9E3779B97F4A7C15ull;
        unsigned long long x = z;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        const int j = (int)(((x >> 32) * (unsigned long long)(i + 1)) >> 32);
        std::swap(order[i], order[j]);
    }
    HIPCHK(e->d_order.reserve(n));
    HIPCHK(hipMemcpyAsync(e->d_order.p, order.data(), sizeof(int) * n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));          // `order` dies with this scope
    e->order_n = n;
    return MH_OK;
}

// per-site weight totals of the graph that is on the device (d_rowptr, d_w)
static int upload_wsum(mh_engine* e)
{
    HIPCHK(e->d_wsum.reserve(e->n));
    HIPCHK(launch_row_weight_sums(e->n, e->d_rowptr.p, e->d_w.p, e->d_wsum.p, e->stream));
    return MH_OK;
}

// host copy (fallback path) -> device
int upload_graph(mh_engine* e)
{
    const int n = e->n, nnz = (int)e->g_col.size();
    HIPCHK(e->d_rowptr.reserve(n + 1));
    HIPCHK(e->d_col.reserve(nnz));
    HIPCHK(e->d_w.reserve(nnz));
    HIPCHK(e->d_rev.reserve(nnz));
    HIPCHK(hipMemcpyAsync(e->d_rowptr.p, e->g_rowptr.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, e->stream));
    if (nnz) {
        HIPCHK(hipMemcpyAsync(e->d_col.p, e->g_col.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->d_w.p, e->g_w.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->d_rev.p, e->g_rev.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    int rc = upload_order(e);
    if (rc) return rc;
    rc = upload_wsum(e);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    e->g_nnz = nnz;
    e->g_host_valid = true;
    e->have_graph = true;
    return MH_OK;
}

// device -> pageable host memory on the engine's stream, complete on return
static hipError_t fetch_ints(mh_engine* e, int* dst, const int* src_dev, size_t count)
{
    hipError_t he = hipMemcpyAsync(dst, src_dev, sizeof(int) * count, hipMemcpyDeviceToHost, e->stream);
    return he != hipSuccess ? he : hipStreamSynchronize(e->stream);
}

// The symmetric graph from directed hits that are on the device (graph.hip): a CSR (rowptr_dev) or a dense
// n x stride table with -1 for "no hit".  Rows of more than SYM_MAX_ROW raw entries take the host path.
// e->gb_info (8 ints) must have been cleared by the caller (its word 2 collects index errors of earlier passes too).
static int device_sym_graph(mh_engine* e, const int* rowptr_dev, int stride, const int* col_dev)
{
    const int n = e->n;
    HIPCHK(e->gb_deg.reserve(n));
    HIPCHK(e->gb_start.reserve((size_t)n + 1));
    HIPCHK(launch_sym_count(n, rowptr_dev, stride, col_dev, e->gb_deg.p, e->gb_start.p, e->gb_info.p, e->stream));
    int info[8] = {}, total = 0;
    HIPCHK(hipMemcpyAsync(info, e->gb_info.p, sizeof(int) * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(&total, e->gb_start.p + n, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (info[2] == 2) return fail(MH_ERR_INVALID, "rowptr must be non-decreasing");
    if (info[2]) return fail(MH_ERR_INVALID, "neighbour index out of range (non-finite coordinates?)");
    if (info[0]) return fail(MH_ERR_OVERFLOW, "too many neighbour entries");
    if (info[1] > SYM_MAX_ROW) {
        // a very dense neighbourhood: build on the host (any row length)
        std::vector<int> rp(n + 1), col;
        if (rowptr_dev) {
            HIPCHK(fetch_ints(e, rp.data(), rowptr_dev, (size_t)n + 1));
            col.resize((size_t)rp[n]);
            if (rp[n] > 0) HIPCHK(fetch_ints(e, col.data(), col_dev, col.size()));
        } else {
            std::vector<int> dense((size_t)n * stride);
            HIPCHK(fetch_ints(e, dense.data(), col_dev, dense.size()));
            for (int i = 0; i < n; ++i) {
                rp[i] = (int)col.size();
                for (int j = 0; j < stride; ++j) if (dense[(size_t)i * stride + j] >= 0) col.push_back(dense[(size_t)i * stride + j]);
            }
            rp[n] = (int)col.size();
        }
        int rc = build_sym_graph(e, rp.data(), col.data(), n);
        if (rc) return rc;
        return upload_graph(e);
    }
    HIPCHK(e->gb_cursor.reserve(n));
    HIPCHK(e->gb_uniq.reserve(n));
    HIPCHK(e->gb_raw.reserve((size_t)std::max(total, 1)));
    HIPCHK(e->gb_mult.reserve((size_t)std::max(total, 1)));
    HIPCHK(e->d_rowptr.reserve((size_t)n + 1));
    HIPCHK(launch_sym_build(n, rowptr_dev, stride, col_dev, e->gb_start.p, e->gb_cursor.p, e->gb_raw.p, e->gb_mult.p,
                            e->gb_uniq.p, e->d_rowptr.p, e->gb_info.p, e->stream));
    int nnz = 0;
    HIPCHK(hipMemcpyAsync(&nnz, e->d_rowptr.p + n, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(e->d_col.reserve((size_t)std::max(nnz, 1)));
    HIPCHK(e->d_w.reserve((size_t)std::max(nnz, 1)));
    HIPCHK(e->d_rev.reserve((size_t)std::max(nnz, 1)));
    HIPCHK(launch_sym_finish(n, e->gb_start.p, e->gb_raw.p, e->gb_mult.p, e->d_rowptr.p, e->d_col.p, e->d_w.p, e->d_rev.p, e->stream));
    int rc = upload_order(e);
    if (rc) return rc;
    rc = upload_wsum(e);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    e->g_nnz = nnz;
    e->g_host_valid = false;
    e->have_graph = true;
    return MH_OK;
}

int ensure_expand_work(mh_engine* e)
{
    const int n = e->n, nnz = e->g_nnz;
    HIPCHK(e->ew_label.reserve(n));
    HIPCHK(e->ew_cur.reserve(n));
    HIPCHK(e->ew_cap.reserve(nnz));
    HIPCHK(e->ew_sent.reserve(nnz));
    HIPCHK(e->ew_excess.reserve(n));
    HIPCHK(e->ew_sink.reserve(n));
    HIPCHK(e->ew_height.reserve(n));
    HIPCHK(e->ew_decided.reserve(n));
    HIPCHK(e->ew_flags.reserve(EXPAND_FLAG_WORDS));
    HIPCHK(e->ew_acc.reserve(EXPAND_ACC_WORDS));
    HIPCHK(e->ew_took.reserve((size_t)n + 2));
    HIPCHK(e->ew_core.reserve((size_t)EXPAND_CORE_SHARDS * n));
    if (!e->h_flags) {
        HIPCHK(hipHostMalloc((void**)&e->h_flags, sizeof(int) * EXPAND_HOST_WORDS, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_flags_dev, e->h_flags, 0));
    }
    if (!e->h_acc) {
        HIPCHK(hipHostMalloc((void**)&e->h_acc, sizeof(long long) * 16, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_acc_dev, e->h_acc, 0));
    }
    return MH_OK;
}

int do_data_cost(mh_engine* e)
{
    const int L = e->m + 1;
    HIPCHK(e->cost.reserve((size_t)e->n * L));
    {
        ScopedTimer t(e, MH_K_DATACOST);
        HIPCHK(launch_data_cost(e->pts(), e->H.p, e->m, e->lambda, e->thr_H * e->thr_H, e->cost.p, e->stream));
    }
    e->cost_L = L;
    return MH_OK;
}

// How many workgroups of the solver launch the device holds at once: the occupancy query's answer for k_solve at its
// default dynamic LDS, times the CUs — and never more than one per CU (the kernel is written for that).
int solve_grid_limit(mh_engine* e)
{
    if (e->solve_grid_max > 0) return MH_OK;
    int per_cu = 0;
    HIPCHK(solver_blocks_per_cu(&per_cu));
    if (per_cu < 1) return fail(MH_ERR_HIP, "the alpha-expansion solver kernel does not fit a compute unit");
    e->solve_grid_max = e->cu_count * 1;
    return MH_OK;
}

// init_dev: device pointer to initial labels (GCO numbering) or null.
int do_expand(mh_engine* e, const int* init_dev, long long* energy, int* cycles)
{
    if (!e->have_graph) return fail(MH_ERR_NOT_SET, "neighbour graph is not set");
    if (e->cost_L != e->m + 1) return fail(MH_ERR_NOT_SET, "data cost is stale; call mh_data_cost first");
    int rc = ensure_expand_work(e);
    if (rc) return rc;
    Graph g{ e->d_rowptr.p, e->d_col.p, e->d_w.p, e->d_rev.p, e->n, e->g_nnz, e->d_order.p, e->d_wsum.p };
    // The solver launch synchronises through a grid barrier, so it must be resident as a whole: one 512-thread workgroup
    // per CU at most (what the occupancy query admits for this kernel is checked once per engine, solve_grid_limit).
    // One process per GPU — the deployment — always is.  Engines of several processes that share a GPU can keep each
    // other's workgroups off the chip; a launch whose barrier then gives up (3 s) is not an error any more: the
    // expansion is restarted from its initial labeling with half the workgroups (results never depend on that number),
    // up to four times — a shared GPU degrades instead of failing.  `mh_set_tuning` key 5 sets the starting number.
    rc = solve_grid_limit(e);
    if (rc) return rc;
    int solve_grid = std::max(1, std::min(e->tune_expand[3], e->solve_grid_max));
    ExpandWork w{ e->ew_label.p, e->ew_cur.p, e->ew_cap.p, e->ew_sent.p, e->ew_excess.p, e->ew_sink.p,
                  e->ew_height.p, e->ew_decided.p, e->ew_took.p, e->ew_core.p, e->ew_flags.p, e->ew_acc.p,
                  e->h_flags, e->h_acc, e->h_flags_dev, e->h_acc_dev,
                  e->tune_expand[0], e->tune_expand[1], e->tune_expand[2], solve_grid, e->tune_push_mult, e->tune_reduce, e->tune_reduce_launches,
                  e->tune_cascade_iters, nullptr, 0, -1, nullptr, nullptr };
    // flow recycling (expand.hip, k_solve): L x (nnz + n) ints, cleared per expansion; left out (every move starts from
    // the zero flow) beyond 8 GiB.  Flows are not kept from one call to the next: measured in the alternation, the
    // re-estimated models move the problems far enough for a kept flow to cost more rounds than the zero flow.
    const size_t recycle_words = (size_t)e->cost_L * ((size_t)g.nnz + (size_t)g.n);
    if (e->tune_recycle && recycle_words <= ((size_t)2 << 30)) {
        HIPCHK(e->ew_saved.reserve(recycle_words));
        HIPCHK(hipMemsetAsync(e->ew_saved.p, 0, sizeof(int) * recycle_words, e->stream));
        w.saved_flow = e->ew_saved.p;
        w.saved_sink = e->ew_saved.p + (size_t)e->cost_L * g.nnz;
    }
    if (e->trace_moves > 0) {
        HIPCHK(e->ew_trace.reserve(8 * (size_t)e->trace_moves + 4 * 2048));
        HIPCHK(hipMemsetAsync(e->ew_trace.p, 0, sizeof(int) * (8 * (size_t)e->trace_moves + 4 * 2048), e->stream));
        w.detail_move = e->detail_move;
        w.trace = e->ew_trace.p;
        w.trace_moves = e->trace_moves;
    }
    if (e->comp_moves > 0) {
        HIPCHK(e->ew_comp.reserve(2 * (size_t)g.n));
        HIPCHK(e->ew_comp_out.reserve(16 * (size_t)e->comp_moves));
        HIPCHK(hipMemsetAsync(e->ew_comp_out.p, 0, sizeof(int) * 16 * (size_t)e->comp_moves, e->stream));
        w.comp_out = e->ew_comp_out.p; w.comp_scratch = e->ew_comp.p; w.comp_moves = e->comp_moves;
    }
    const int potts = (int)std::round(100.0 * e->lambda);     // M/MultiH.h:41, MultiH.cpp:510
    ExpandStats st{};
    {
        ScopedTimer t(e, MH_K_EXPAND);
        hipError_t he = hipSuccess;
        e->last_expand_retries = 0;
        for (int attempt = 0; attempt < 5; ++attempt) {
            w.solve_grid = solve_grid;
            // How long a barrier may wait before the launch gives up and the expansion restarts: a healthy barrier takes
            // about 8 us, so max(20 ms, 50 x the longest steady-state wait this engine has seen) tells "a workgroup is not
            // resident" from "slow" within tens of milliseconds.  The FIRST barrier of a launch is the one that waits for
            // every workgroup to be dispatched — on a GPU shared with another engine's 7 ms sweeps that is a matter of the
            // other work's length, not of this launch's size: it gets 250 ms (ten times the steady limit if that is more).
            // A timed-out attempt is repeated ONCE with the same grid before the grid is halved; only the last attempt (or
            // a launch already down to one workgroup) waits the full 3 s before the call fails.
            const bool last_attempt = attempt == 4 || solve_grid == 1;
            const double steady_ms = std::max(20.0, 50.0 * e->longest_barrier_wait_ms);
            w.barrier_timeout_ticks = last_attempt ? 300000000ll : (long long)(steady_ms * 1e5);
            w.barrier_first_timeout_ticks = last_attempt ? 300000000ll : (long long)(std::max(250.0, 10.0 * steady_ms) * 1e5);
            if (w.saved_flow) HIPCHK(hipMemsetAsync(e->ew_saved.p, 0, sizeof(int) * recycle_words, e->stream));
            HIPCHK(launch_init_labeling(e->cost.p, e->cost_L, e->n, init_dev, w.label, w.cur_cost, e->stream));
            he = run_expansion(g, e->cost.p, e->cost_L, potts, w, 1000, &st, e->stream);
            bool timed_out = he == hipErrorLaunchTimeOut && st.energy == -2;
            if (he == hipSuccess && e->inject_barrier_timeouts > 0) {           // test hook (mh_set_tuning key 14)
                --e->inject_barrier_timeouts;
                timed_out = true;
                he = hipErrorLaunchTimeOut;
                st.energy = -2;
            }
            e->last_solve_grid = solve_grid;
            if (!timed_out || solve_grid == 1 || attempt == 4) break;
            if (attempt >= 1) solve_grid = std::max(1, solve_grid / 2);      // (the first retry keeps the grid)
            ++e->last_expand_retries;
            ++e->expand_retries_total;
        }
        if (he == hipErrorOutOfMemory)
            return fail(MH_ERR_INVALID, "alpha-expansion: more sites than the solver's per-row state holds (about 1.3 million at 256 workgroups)");
        if (he == hipErrorInvalidValue && st.energy == -1)
            return fail(MH_ERR_OVERFLOW, "int32 energy term overflow in alpha-expansion");
        if (he == hipErrorLaunchTimeOut && st.energy == -2)
            return fail(MH_ERR_HIP, "alpha-expansion: the solver's grid barrier timed out even with the launch cut down to a few workgroups "
                                    "(its workgroups were not all resident; is the GPU shared with other persistent launches?)");
        if (he == hipErrorLaunchTimeOut && st.energy == -3)
            return fail(MH_ERR_HIP, "alpha-expansion: push-relabel did not converge within its iteration bound");
        HIPCHK(he);
    }
    e->last_expand = st;
    if (st.max_barrier_wait_ms > e->longest_barrier_wait_ms) e->longest_barrier_wait_ms = std::min(st.max_barrier_wait_ms, 50.0);
    if (st.energy > 0x7fffffffll || st.energy < -0x7fffffffll)
        return fail(MH_ERR_OVERFLOW, "total energy exceeds the reference's int32 EnergyType");
    if (energy) *energy = st.energy;
    if (cycles) *cycles = st.cycles;
    return MH_OK;
}

// cyclic Jacobi, 3x3 symmetric (host copy of the device solver's recurrence)
void host_jacobi3(double* a, double* v, double* d)
{
    const int n = 3;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag = diag + a[i * n + i] * a[i * n + i];
            for (int j = i + 1; j < n; ++j) off = off + a[i * n + j] * a[i * n + j];
        }
        if (off <= 1e-30 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { const double x = a[k * n + p], y = a[k * n + q]; a[k * n + p] = c * x - s * y; a[k * n + q] = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = a[p * n + k], y = a[q * n + k]; a[p * n + k] = c * x - s * y; a[q * n + k] = s * x + c * y; }
                for (int k = 0; k < n; ++k) { const double x = v[k * n + p], y = v[k * n + q]; v[k * n + p] = c * x - s * y; v[k * n + q] = s * x + c * y; }
            }
    }
    for (int i = 0; i < n; ++i) d[i] = a[i * n + i];
}

// Inlier counts of `m` models (device array Hs) over the points `p`: the FP32 pre-test kernel where its preconditions
// hold (forward residual, bounded coordinates), the FP64 sweep otherwise.  Same counts either way.
int score_models(mh_engine* e, const Points& p, const double* Hs, int m, double thr2, const unsigned char* dmask, int* counts_dev)
{
    const bool fwd = e->residual_mode != MH_RESIDUAL_SYMMETRIC;
    if (fwd && e->tune_score32 && e->tune_score_variant == 0 && e->coords32_ok && m > 0 && thr2 >= 0x1p-40 && thr2 <= 0x1p40) {
        HIPCHK(e->H32.reserve((size_t)m * 16));
        HIPCHK(e->fb_pairs.reserve(1));
        if (e->score_pairs == 0) HIPCHK(hipMemsetAsync(e->fb_pairs.p, 0, sizeof(unsigned long long), e->stream));
        HIPCHK(launch_model32(Hs, m, e->absmax_x, e->absmax_y, e->absmax_dst, e->H32.p, e->stream));
        int* ctl = nullptr;
        if (e->tune_score32_resident != 0) {
            if (!e->sweep_ctl.p) {
                HIPCHK(e->sweep_ctl.reserve(2));
                HIPCHK(hipMemsetAsync(e->sweep_ctl.p, 0, sizeof(int) * 2, e->stream));
            }
            ctl = e->sweep_ctl.p;
        }
        HIPCHK(launch_score32(p, Hs, e->H32.p, m, thr2, e->absmax_dst, dmask, counts_dev, e->fb_pairs.p, e->tune_score32_tiling, e->stream,
                              ctl, e->cu_count, e->tune_score32_resident, &e->occ_score32));
        e->score_pairs += (long long)m * p.n;
        return MH_OK;
    }
    HIPCHK(launch_score(p, Hs, m, thr2, dmask, counts_dev, fwd ? e->tune_score_variant : -1, e->stream));
    return MH_OK;
}

__global__ void k_shift_labels(int n, const int* in, int delta, int* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + delta;
}

__global__ void k_split_soa(int n, const double* src, const double* dst, const double* aff,
                            double* x1, double* y1, double* x2, double* y2, double* a11,
                            double* a12, double* a21, double* a22)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    x1[i] = src[2 * i]; y1[i] = src[2 * i + 1];
    x2[i] = dst[2 * i]; y2[i] = dst[2 * i + 1];
    if (aff) { a11[i] = aff[4 * i]; a12[i] = aff[4 * i + 1]; a21[i] = aff[4 * i + 2]; a22[i] = aff[4 * i + 3]; }
}

} // namespace

extern "C" {

int mh_abi_version(void) { return MH_ABI_VERSION; }

const char* mh_last_error(void) { return g_err.c_str(); }

int mh_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int mh_create(mh_engine** out, int device)
{
    return guarded([&]() -> int {
    if (!out) return fail(MH_ERR_INVALID, "out is null");
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0)
        return fail(MH_ERR_NO_DEVICE, "no HIP device visible; the Multi-H engine has no CPU fallback");
    if (device < 0 || device >= c) return fail(MH_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MH_ERR_NO_DEVICE, std::string("device arch is ") + prop.gcnArchName +
                                           "; this library carries gfx950 code objects only");
    mh_engine* e = new (std::nothrow) mh_engine();
    if (!e) return fail(MH_ERR_INVALID, "out of host memory");
    e->device = device;
    e->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t he = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete e; return fail(MH_ERR_HIP, hipGetErrorString(he)); }
    e->stream = e->own_stream;
    *out = e;
    return MH_OK;
    });
}

void mh_destroy(mh_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)quiesce(e);
    resolve_timers(e);
    e->x1.release(); e->y1.release(); e->x2.release(); e->y2.release();
    e->a11.release(); e->a12.release(); e->a21.release(); e->a22.release();
    e->d_rowptr.release(); e->d_col.release(); e->d_w.release(); e->d_rev.release();
    e->H.release(); e->samples.release(); e->counts.release(); e->R.release(); e->C.release(); e->mask.release();
    e->moments.release(); e->min_eig.release();
    e->cp_pts.release(); e->cp_H.release(); e->cp_out.release(); e->cp_begin.release(); e->cp_tri.release(); e->cp_ok.release();
    e->fund.release(); e->fund_one.release(); e->fund_samples.release(); e->fund_counts.release();
    e->fund_inl.release(); e->fund_mask.release(); e->ref_keep.release(); e->ref_in.release(); e->ref_out.release();
    e->loc_H.release(); e->loc_feat.release(); e->ms_data.release(); e->ms_mean.release();
    e->ms_votes.release(); e->ms_out.release(); e->ms_list.release(); e->ms_pcnt.release(); e->ms_heads.release(); e->ms_tickets.release(); e->ms_partial.release();
    e->cost.release(); e->labels_in.release(); e->labels_pts.release(); e->label_counts.release();
    e->ew_label.release(); e->ew_cur.release(); e->ew_cap.release(); e->ew_excess.release();
    e->ew_sink.release(); e->ew_height.release(); e->ew_decided.release(); e->ew_flags.release(); e->ew_acc.release();
    e->ew_comp.release(); e->ew_comp_out.release();
    e->ew_took.release(); e->ew_core.release(); e->ew_sent.release(); e->ew_trace.release(); e->ew_saved.release(); e->d_order.release(); e->d_wsum.release();
    e->knn_tmp.release(); e->knn_part_i.release(); e->knn_part_d.release();
    for (int c = 0; c < 4; ++c) e->sel_pts[c].release();
    e->sel_pack_count.release();
    e->gb_deg.release(); e->gb_start.release(); e->gb_cursor.release(); e->gb_raw.release(); e->gb_mult.release();
    e->gb_uniq.release(); e->gb_info.release(); e->gb_hits_rp.release(); e->gb_hits_col.release();
    if (e->h_flags) (void)hipHostFree(e->h_flags);
    if (e->h_ms) (void)hipHostFree(e->h_ms);
    if (e->h_ms_list) (void)hipHostFree(e->h_ms_list);
    if (e->h_acc) (void)hipHostFree(e->h_acc);
    if (e->h_sel) (void)hipHostFree(e->h_sel);
    for (int b = 0; b < 2; ++b) { e->sel_orig[b].release(); e->sel_cand_H[b].release(); }
    e->sel_counts.release(); e->sel_rec.release(); e->sel_scores.release(); e->sel_gathered.release(); e->sel_out_H.release();
    e->sel_records.release(); e->sel_counter.release(); e->sel_keys.release();
    for (int q = 0; q < mh_engine::PF_DEPTH; ++q) { e->pf_H[q].release(); e->pf_samples[q].release(); if (e->pf_ev[q]) (void)hipEventDestroy(e->pf_ev[q]); }
    e->best_key.release(); e->H32.release(); e->fb_pairs.release();
    if (e->h_best) (void)hipHostFree(e->h_best);
    if (e->ev_main) (void)hipEventDestroy(e->ev_main);
    if (e->ev_side_pre) (void)hipEventDestroy(e->ev_side_pre);
    if (e->side_stream) { (void)hipStreamSynchronize(e->side_stream); (void)hipStreamDestroy(e->side_stream); }
    if (e->xchg_stream) { (void)hipStreamSynchronize(e->xchg_stream); (void)hipStreamDestroy(e->xchg_stream); }
    if (e->ev_sweep) (void)hipEventDestroy(e->ev_sweep);
    for (int b = 0; b < 3; ++b) if (e->ev_x[b]) (void)hipEventDestroy(e->ev_x[b]);
    e->counts_alt[0].release(); e->counts_alt[1].release(); e->sweep_ctl.release();
    for (hipStream_t d : e->dummy_streams) (void)hipStreamDestroy(d);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int mh_set_params(mh_engine* e, double thr_F, double thr_H, double locality, double lambda, int min_inliers)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    if (!(thr_H > 0.0) || !(lambda > 0.0)) return fail(MH_ERR_INVALID, "thr_hom and lambda must be positive");
    e->thr_F = thr_F; e->thr_H = thr_H; e->locality = locality; e->lambda = lambda;
    e->min_inliers = min_inliers;
    e->cost_L = 0;
    return MH_OK;
    });
}

int mh_set_stream(mh_engine* e, void* hip_stream, int external)
{
    return guarded([&]() -> int {
    int rc0 = enter(e);
    if (rc0) return rc0;
    int rcq = quiesce(e);
    if (rcq) return rcq;
    resolve_timers(e);
    e->stream = external ? (hipStream_t)hip_stream : e->own_stream;
    return MH_OK;
    });
}

int mh_synchronize(mh_engine* e)
{
    return guarded([&]() -> int {
    int rc0 = enter(e);
    if (rc0) return rc0;
    rc0 = quiesce(e);
    if (rc0) return rc0;
    resolve_timers(e);
    return MH_OK;
    });
}

int mh_set_correspondences(mh_engine* e, const double* src_xy, const double* dst_xy,
                           const double* affines, int n)
{
    return guarded([&]() -> int {
    int rc0 = enter(e);
    if (rc0) return rc0;
    if (!src_xy || !dst_xy || n <= 0) return fail(MH_ERR_INVALID, "src/dst must be non-null and n > 0");
    // A DLT prefetch in flight on the second stream reads the point arrays this call overwrites, and the batch it
    // prepares belongs to the OLD point set: wait for it and drop it (r03 advisor finding).
    rc0 = quiesce(e);
    if (rc0) return rc0;
    e->pf_count = 0;
    e->pf_head = 0;
    // +1 element of slack: the 16-B vector loads of the residual sweep never cross the end,
    // but keep the allocation even-sized for them.
    const size_t cap = (size_t)n + 2;
    HIPCHK(e->x1.reserve(cap)); HIPCHK(e->y1.reserve(cap));
    HIPCHK(e->x2.reserve(cap)); HIPCHK(e->y2.reserve(cap));
    HIPCHK(e->a11.reserve(cap)); HIPCHK(e->a12.reserve(cap));
    HIPCHK(e->a21.reserve(cap)); HIPCHK(e->a22.reserve(cap));
    struct Staging { DevBuf<double> b; ~Staging() { b.release(); } } st_s, st_d, st_a;   // freed on every return path
    DevBuf<double>&s = st_s.b, &d = st_d.b, &a = st_a.b;
    HIPCHK(s.reserve((size_t)n * 2));
    HIPCHK(d.reserve((size_t)n * 2));
    HIPCHK(hipMemcpyAsync(s.p, src_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d.p, dst_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, e->stream));
    if (affines) {
        HIPCHK(a.reserve((size_t)n * 4));
        HIPCHK(hipMemcpyAsync(a.p, affines, sizeof(double) * 4 * n, hipMemcpyHostToDevice, e->stream));
    }
    hipLaunchKernelGGL(k_split_soa, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, s.p, d.p,
                       affines ? a.p : nullptr, e->x1.p, e->y1.p, e->x2.p, e->y2.p, e->a11.p,
                       e->a12.p, e->a21.p, e->a22.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    if (n != e->n) {
        // everything sized by the previous point set is stale: the residual matrix and its pitch, the sampled batch,
        // the fundamental-matrix hypotheses
        e->m = 0; e->ldr = 0; e->have_samples = false; e->fm = 0;
        e->counts_fresh = false; ++e->models_seq;
    }
    {
        double xmin = src_xy[0], xmax = src_xy[0], ymin = src_xy[1], ymax = src_xy[1], dmax = 0.0;
        bool finite = true, dfinite = true;
        for (int i = 0; i < n; ++i) {
            const double x = src_xy[2 * i], y = src_xy[2 * i + 1];
            finite = finite && std::isfinite(x) && std::isfinite(y);
            xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax;
            ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
            const double a = std::fabs(dst_xy[2 * i]), b = std::fabs(dst_xy[2 * i + 1]);
            dfinite = dfinite && std::isfinite(a) && std::isfinite(b);
            dmax = a > dmax ? a : dmax; dmax = b > dmax ? b : dmax;
        }
        if (!finite) xmin = xmax = ymin = ymax = NAN;
        e->bbox[0] = xmin; e->bbox[1] = xmax; e->bbox[2] = ymin; e->bbox[3] = ymax;
        e->absmax_x = std::max(std::fabs(xmin), std::fabs(xmax));
        e->absmax_y = std::max(std::fabs(ymin), std::fabs(ymax));
        e->absmax_dst = dmax;
        e->coords32_ok = finite && dfinite && e->absmax_x < 0x1p20 && e->absmax_y < 0x1p20 && dmax < 0x1p20;
    }
    e->n = n;
    e->have_aff = affines != nullptr;
    e->have_graph = false;
    e->g_rowptr.clear(); e->g_col.clear(); e->g_w.clear(); e->g_rev.clear();
    e->g_host_valid = false; e->g_nnz = 0;
    e->cost_L = 0;
    return MH_OK;
    });
}

int mh_set_epipolar(mh_engine* e, const double F[9], const double e2[2])
{
    return guarded([&]() -> int {
    if (!e || !F || !e2) return fail(MH_ERR_INVALID, "null argument");
    for (int i = 0; i < 9; ++i) e->epi.F[i] = F[i];
    e->epi.ex = e2[0];
    e->epi.ey = e2[1];
    e->have_epi = true;
    return MH_OK;
    });
}

int mh_set_neighbors_csr(mh_engine* e, const int* rowptr, const int* col, int n)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (!rowptr || n != e->n) return fail(MH_ERR_INVALID, "rowptr null or n != number of correspondences");
    if (rowptr[n] > 0 && !col) return fail(MH_ERR_INVALID, "col is null");
    if (rowptr[0] < 0) return fail(MH_ERR_INVALID, "rowptr must be non-decreasing");
    for (int i = 0; i < n; ++i)
        if (rowptr[i + 1] < rowptr[i]) return fail(MH_ERR_INVALID, "rowptr must be non-decreasing");
    HIPCHK(e->gb_info.reserve(8));
    HIPCHK(hipMemsetAsync(e->gb_info.p, 0, sizeof(int) * 8, e->stream));
    HIPCHK(e->gb_hits_rp.reserve((size_t)n + 1));
    HIPCHK(e->gb_hits_col.reserve((size_t)std::max(rowptr[n], 1)));
    HIPCHK(hipMemcpyAsync(e->gb_hits_rp.p, rowptr, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, e->stream));
    if (rowptr[n] > 0)
        HIPCHK(hipMemcpyAsync(e->gb_hits_col.p, col, sizeof(int) * (size_t)rowptr[n], hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));          // the caller's arrays are free again
    return device_sym_graph(e, e->gb_hits_rp.p, 0, e->gb_hits_col.p);
    });
}

// k nearest hits per query, optionally only those within `radius` (<= 0: no cut).
static int build_knn_graph(mh_engine* e, int k, double radius)
{
    int rc = require_points(e);
    if (rc) return rc;
    if (k < 1 || k > 32 || k >= e->n) return fail(MH_ERR_INVALID, "k must be in [1, 32] and < n");
    const int n = e->n;
    HIPCHK(e->knn_tmp.reserve((size_t)n * k));
    HIPCHK(e->gb_info.reserve(8));
    HIPCHK(hipMemsetAsync(e->gb_info.p, 0, sizeof(int) * 8, e->stream));
    // enough slices of the candidate range to give every SIMD a few waves (one thread per query and slice)
    const int blocks = (n + 255) / 256;
    int splits = (4 * e->cu_count + blocks - 1) / blocks;
    splits = std::max(1, std::min(splits, std::min(KNN_MAX_SPLITS, (n + 1023) / 1024)));
    const int kk = k <= 8 ? 8 : k <= 16 ? 16 : 32;
    if (splits > 1) {
        HIPCHK(e->knn_part_d.reserve((size_t)splits * n * kk));
        HIPCHK(e->knn_part_i.reserve((size_t)splits * n * kk));
    }
    HIPCHK(launch_knn(e->pts(), k, e->knn_tmp.p, splits, e->knn_part_d.p, e->knn_part_i.p, e->stream));
    // the reference's radius (M/MultiH.cpp:252-253) in the kernels' float32 arithmetic; without a radius the pass only
    // validates the indices (non-finite coordinates leave garbage in the k-NN table)
    const float r2 = radius > 0.0 ? (float)radius * (float)radius : INFINITY;
    HIPCHK(launch_hits_filter(e->pts(), k, r2, e->knn_tmp.p, e->gb_info.p + 2, e->stream));
    return device_sym_graph(e, nullptr, k, e->knn_tmp.p);
}

int mh_build_neighbors_knn(mh_engine* e, int k)
{
    return guarded([&]() -> int { return build_knn_graph(e, k, 0.0); });
}

int mh_build_neighbors_knn_radius(mh_engine* e, int k, double radius)
{
    return guarded([&]() -> int { return build_knn_graph(e, k, radius); });
}

int mh_build_neighbors_radius(mh_engine* e, double radius, long long max_hits, long long* hits_out)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (!(radius > 0.0)) return fail(MH_ERR_INVALID, "radius must be positive");
    const int n = e->n;
    const float r2 = (float)radius * (float)radius;
    HIPCHK(e->knn_tmp.reserve((size_t)n + 1));
    HIPCHK(launch_radius_count(e->pts(), r2, e->knn_tmp.p, e->stream));
    std::vector<int> cnt(n), rowptr(n + 1, 0);
    HIPCHK(hipMemcpyAsync(cnt.data(), e->knn_tmp.p, sizeof(int) * n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    long long total = 0;
    for (int i = 0; i < n; ++i) total += cnt[i];
    if (hits_out) *hits_out = total;
    const long long limit = max_hits > 0 ? max_hits : 0x7fffffffll;
    if (total > limit || total > 0x7fffffffll) {
        char msg[160];
        snprintf(msg, sizeof msg, "radius search yields %lld hits (limit %lld): use a smaller radius or k-NN", total, limit);
        return fail(MH_ERR_OVERFLOW, msg);
    }
    for (int i = 0; i < n; ++i) rowptr[i + 1] = rowptr[i] + cnt[i];
    struct Scratch { DevBuf<int> b; ~Scratch() { b.release(); } } s_rp, s_col;
    DevBuf<int>&d_rp = s_rp.b, &d_col = s_col.b;
    HIPCHK(d_rp.reserve((size_t)n + 1));
    HIPCHK(d_col.reserve((size_t)std::max<long long>(total, 1)));
    HIPCHK(hipMemcpyAsync(d_rp.p, rowptr.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, e->stream));
    HIPCHK(launch_radius_fill(e->pts(), r2, d_rp.p, d_col.p, e->stream));
    HIPCHK(e->gb_info.reserve(8));
    HIPCHK(hipMemsetAsync(e->gb_info.p, 0, sizeof(int) * 8, e->stream));
    return device_sym_graph(e, d_rp.p, 0, d_col.p);      // (synchronises before the scratch lists are released)
    });
}

int mh_get_sym_graph(mh_engine* e, int* rowptr, int* col, int* w, int* nnz)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    if (!e->have_graph) return fail(MH_ERR_NOT_SET, "neighbour graph is not set");
    if (!e->g_host_valid) {                            // built on the device: fetch once
        HIPCHK(hipSetDevice(e->device));
        e->g_rowptr.resize((size_t)e->n + 1); e->g_col.resize(e->g_nnz); e->g_w.resize(e->g_nnz); e->g_rev.resize(e->g_nnz);
        HIPCHK(fetch_ints(e, e->g_rowptr.data(), e->d_rowptr.p, (size_t)e->n + 1));
        if (e->g_nnz) {
            HIPCHK(fetch_ints(e, e->g_col.data(), e->d_col.p, (size_t)e->g_nnz));
            HIPCHK(fetch_ints(e, e->g_w.data(), e->d_w.p, (size_t)e->g_nnz));
            HIPCHK(fetch_ints(e, e->g_rev.data(), e->d_rev.p, (size_t)e->g_nnz));
        }
        e->g_host_valid = true;
    }
    if (nnz) *nnz = e->g_nnz;
    if (rowptr) std::copy(e->g_rowptr.begin(), e->g_rowptr.end(), rowptr);
    if (col) std::copy(e->g_col.begin(), e->g_col.end(), col);
    if (w) std::copy(e->g_w.begin(), e->g_w.end(), w);
    return MH_OK;
    });
}

int mh_propose_fund8(mh_engine* e, unsigned long long seed, long long first, int m)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (m <= 0) return fail(MH_ERR_INVALID, "m must be positive");
    if (e->n < 8) return fail(MH_ERR_INVALID, "need at least 8 correspondences");
    HIPCHK(e->fund.reserve((size_t)m * 9));
    HIPCHK(e->fund_samples.reserve((size_t)m * 8));
    HIPCHK(e->fund_counts.reserve(m));
    HIPCHK(launch_fund8(e->pts(), seed, first, m, e->fund_samples.p, e->fund.p, e->stream));
    e->fm = m;
    return MH_OK;
    });
}

int mh_get_fund_hypotheses(mh_engine* e, double* F, int* idx)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (e->fm <= 0) return fail(MH_ERR_NOT_SET, "no fundamental-matrix hypotheses; call mh_propose_fund8");
    if (F) HIPCHK(hipMemcpyAsync(F, e->fund.p, sizeof(double) * 9 * e->fm, hipMemcpyDeviceToHost, e->stream));
    if (idx) HIPCHK(hipMemcpyAsync(idx, e->fund_samples.p, sizeof(int) * 8 * e->fm, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_score_sampson(mh_engine* e, double thr2, int* counts)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (e->fm <= 0) return fail(MH_ERR_NOT_SET, "no fundamental-matrix hypotheses; call mh_propose_fund8");
    HIPCHK(launch_sampson_score(e->pts(), e->fund.p, e->fm, thr2, e->fund_counts.p, e->stream));
    if (counts) {
        HIPCHK(hipMemcpyAsync(counts, e->fund_counts.p, sizeof(int) * e->fm, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return MH_OK;
    });
}

int mh_refit_fundamental(mh_engine* e, const double F_in[9], double thr2, int iterations, double F_out[9],
                         unsigned char* inlier_mask, int* inliers)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (!F_in || !F_out || iterations < 1) return fail(MH_ERR_INVALID, "null F or iterations < 1");
    HIPCHK(e->fund_one.reserve(18));
    HIPCHK(e->fund_inl.reserve(1));
    HIPCHK(e->fund_mask.reserve((size_t)e->n + 2));
    HIPCHK(hipMemcpyAsync(e->fund_one.p, F_in, sizeof(double) * 9, hipMemcpyHostToDevice, e->stream));
    for (int it = 0; it < iterations; ++it) {
        double* in = e->fund_one.p + 9 * (it & 1);
        double* out = e->fund_one.p + 9 * ((it + 1) & 1);
        HIPCHK(launch_fund_refit(e->pts(), in, thr2, out, e->fund_mask.p, e->fund_inl.p, e->stream));
    }
    HIPCHK(hipMemcpyAsync(F_out, e->fund_one.p + 9 * (iterations & 1), sizeof(double) * 9, hipMemcpyDeviceToHost, e->stream));
    if (inlier_mask) HIPCHK(hipMemcpyAsync(inlier_mask, e->fund_mask.p, e->n, hipMemcpyDeviceToHost, e->stream));
    if (inliers) HIPCHK(hipMemcpyAsync(inliers, e->fund_inl.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_epipoles(mh_engine*, const double F[9], double e1[2], double e2[2])
{
    return guarded([&]() -> int {
    if (!F || !e1 || !e2) return fail(MH_ERR_INVALID, "null argument");
    for (int which = 0; which < 2; ++which) {
        double A[9], V[9], D[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0.0;
                for (int k = 0; k < 3; ++k)
                    a = a + (which == 0 ? F[3 * k + i] * F[3 * k + j]      // F^T F  (:795)
                                        : F[3 * i + k] * F[3 * j + k]);    // F F^T  (:789)
                A[3 * i + j] = a;
            }
        host_jacobi3(A, V, D);
        int jm = 0;
        for (int j = 1; j < 3; ++j) if (D[j] < D[jm]) jm = j;
        double* out = which == 0 ? e1 : e2;
        out[0] = V[0 * 3 + jm] / V[2 * 3 + jm];
        out[1] = V[1 * 3 + jm] / V[2 * 3 + jm];
    }
    return MH_OK;
    });
}

int mh_estimate_fundamental(mh_engine* e, unsigned long long seed, int hypotheses, double thr, double F[9],
                            double e2[2], unsigned char* inlier_mask, int* inliers)
{
    return guarded([&]() -> int {
    if (!F || !e2) return fail(MH_ERR_INVALID, "null output");
    int rc = mh_propose_fund8(e, seed, 0, hypotheses);
    if (rc) return rc;
    std::vector<int> counts(hypotheses);
    rc = mh_score_sampson(e, thr * thr, counts.data());
    if (rc) return rc;
    const int best = (int)(std::max_element(counts.begin(), counts.end()) - counts.begin());
    double F0[9];
    HIPCHK(hipMemcpyAsync(F0, e->fund.p + 9 * (size_t)best, sizeof(double) * 9, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // The mask/count reported are the inliers of the LAST refit's input; a final pass on the
    // result makes them the inliers of the returned F.
    rc = mh_refit_fundamental(e, F0, thr * thr, 2, F, nullptr, nullptr);
    if (rc) return rc;
    double Fdummy[9];
    rc = mh_refit_fundamental(e, F, thr * thr, 1, Fdummy, inlier_mask, inliers);
    if (rc) return rc;
    double e1[2];
    return mh_epipoles(e, F, e1, e2);                  // M/MultiH.cpp:786-799
    });
}

int mh_refine_correspondences(mh_engine* e, const double F[9], const double e1[2], const double e2[2],
                              const unsigned char* in_mask, unsigned char* keep, double* refined)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (!F || !e1 || !e2 || !keep || !refined) return fail(MH_ERR_INVALID, "null argument");
    if (!e->have_aff) return fail(MH_ERR_NOT_SET, "affinities are not set");
    HIPCHK(e->ref_keep.reserve((size_t)e->n + 2));
    HIPCHK(e->ref_out.reserve((size_t)e->n * 8));
    const unsigned char* dmask = nullptr;
    if (in_mask) {
        HIPCHK(e->ref_in.reserve((size_t)e->n + 2));
        HIPCHK(hipMemcpyAsync(e->ref_in.p, in_mask, e->n, hipMemcpyHostToDevice, e->stream));
        dmask = e->ref_in.p;
    }
    HIPCHK(hipMemsetAsync(e->ref_out.p, 0, sizeof(double) * 8 * (size_t)e->n, e->stream));
    Affines a{ e->a11.p, e->a12.p, e->a21.p, e->a22.p };
    HIPCHK(launch_refine_points(e->pts(), a, F, e1, e2, dmask, e->ref_keep.p, e->ref_out.p, e->stream));
    HIPCHK(hipMemcpyAsync(keep, e->ref_keep.p, e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(refined, e->ref_out.p, sizeof(double) * 8 * (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_local_homographies(mh_engine* e, double locality, double* H_out, double* feat_out)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (!e->have_aff) return fail(MH_ERR_NOT_SET, "affinities are not set");
    if (!e->have_epi) return fail(MH_ERR_NOT_SET, "fundamental matrix / epipole are not set");
    HIPCHK(e->loc_H.reserve((size_t)e->n * 9));
    HIPCHK(e->loc_feat.reserve((size_t)e->n * 10));
    Affines a{ e->a11.p, e->a12.p, e->a21.p, e->a22.p };
    HIPCHK(launch_haf_point(e->pts(), a, e->epi, locality, e->loc_H.p, e->loc_feat.p, e->stream));
    if (H_out) HIPCHK(hipMemcpyAsync(H_out, e->loc_H.p, sizeof(double) * 9 * e->n, hipMemcpyDeviceToHost, e->stream));
    if (feat_out) HIPCHK(hipMemcpyAsync(feat_out, e->loc_feat.p, sizeof(double) * 10 * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_mean_shift(mh_engine* e, const double* data, int n, int d, double band_width,
                  unsigned long long seed, double* modes, int max_modes, int* assign, int* n_modes)
{
    return guarded([&]() -> int {
    int rc = enter(e);
    if (rc) return rc;
    if (!data || n <= 0 || d <= 0 || d > 16 || !assign || !n_modes)
        return fail(MH_ERR_INVALID, "bad argument (1 <= d <= 16)");
    constexpr int B = MS_BATCH;
    HIPCHK(e->ms_data.reserve((size_t)n * d));
    HIPCHK(e->ms_mean.reserve((size_t)B * 16));
    HIPCHK(e->ms_votes.reserve((size_t)B * n));
    HIPCHK(e->ms_out.reserve((size_t)B * 4));
    HIPCHK(e->ms_list.reserve((size_t)B * 2 * n));
    HIPCHK(e->ms_partial.reserve((size_t)B * 64 * 16));
    HIPCHK(e->ms_pcnt.reserve((size_t)B * 64));
    HIPCHK(hipMemcpyAsync(e->ms_data.p, data, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(e->ms_votes.p, 0, sizeof(int) * (size_t)B * n, e->stream));
    MeanShiftWork w{ e->ms_data.p, n, d, e->ms_mean.p, e->ms_votes.p, e->ms_out.p, e->ms_list.p,
                     e->ms_partial.p, e->ms_pcnt.p };
    const double band_sq = band_width * band_width;                 // MeanShiftClustering.h:31
    const double stop_thresh = 1e-3 * band_width;                   // :48
    constexpr int MS_LIST_PREFIX = 2048;      // pairs per climb that can travel in the batch's one copy; longer lists fetch their rest
    if (!e->h_ms) {                                                 // B result blocks, then the B seed rows
        HIPCHK(hipHostMalloc((void**)&e->h_ms, sizeof(MeanShiftResultBlock) * B + sizeof(int) * B, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_ms_dev, e->h_ms, 0));
    }
    if (!e->h_ms_list) HIPCHK(hipHostMalloc((void**)&e->h_ms_list, sizeof(int) * 2 * MS_LIST_PREFIX * B, hipHostMallocDefault));
    HIPCHK(e->ms_heads.reserve((size_t)B * 2 * MS_LIST_PREFIX));
    HIPCHK(e->ms_tickets.reserve((size_t)B));
    HIPCHK(e->ms_ctl.reserve((size_t)3 * B));
    HIPCHK(e->ms_partial2.reserve((size_t)B * 2 * 64 * 16));
    HIPCHK(e->ms_pcnt2.reserve((size_t)B * 2 * 64));
    HIPCHK(hipMemsetAsync(e->ms_tickets.p, 0, sizeof(int) * (size_t)B, e->stream));
    int* const starts = reinterpret_cast<int*>(e->h_ms + B);
    const int* const starts_dev = reinterpret_cast<const int*>(e->h_ms_dev + B);

    // `init` of the reference (:125-130) is the ascending list of unvisited rows, rebuilt after every
    // climb; a Fenwick tree over the unvisited flags answers "the k-th unvisited row" in O(log n).
    std::vector<int> fen(n + 1, 0), visited(n, 0), list;
    for (int i = 1; i <= n; ++i) { fen[i] += 1; const int j = i + (i & -i); if (j <= n) fen[j] += fen[i]; }
    int top = 1;
    while (top * 2 <= n) top *= 2;
    auto kth_unvisited = [&](int k) {                               // 0-based k
        int pos = 0, rem = k + 1;
        for (int step = top; step > 0; step >>= 1)
            if (pos + step <= n && fen[pos + step] < rem) { pos += step; rem -= fen[pos]; }
        return pos;                                                 // 0-based row index
    };
    auto mark_visited = [&](int row) {
        if (visited[row]) return;
        visited[row] = 1;
        for (int i = row + 1; i <= n; i += i & -i) fen[i] -= 1;
    };
    int unvisited = n;
    // MULTIH_MS_STATS=1: where the call's time goes (a line on stderr at the end) — diagnostic
    const bool ms_stats = std::getenv("MULTIH_MS_STATS") != nullptr;
    double st_persist_us = 0, st_launch_us = 0, st_tail_us = 0;
    long long st_persist_iters = 0, st_persist_rounds = 0, st_persist_climbs = 0, st_launch_rounds = 0, st_tail_climbs = 0, st_batches = 0, st_G = 0;
    std::vector<std::pair<int, int>> st_climbs;              // (iterations, rows touched) of every climb
    if (ms_stats) { HIPCHK(e->ms_ticks.reserve(4)); HIPCHK(hipMemsetAsync(e->ms_ticks.p, 0, sizeof(unsigned long long) * 4, e->stream)); }
    std::vector<double> cent;                                       // modes, d values each
    int n_cent = 0;
    std::vector<std::vector<std::pair<int, int>>> votes;            // per mode: sorted (row, votes)
    unsigned long long counter = 0;
    while (unvisited > 0) {
        // the batch: MS_BATCH seeds drawn from the rows unvisited now (:55-56 for each draw); a small tail draws fewer
        const int climbs = std::min(B, unvisited);
        ++st_batches;
        for (int b = 0; b < climbs; ++b) {
            unsigned long long z = seed + counter++;                // splitmix64
            z += 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z = z ^ (z >> 31);
            const double rnd = (double)(z >> 11) * (1.0 / 9007199254740992.0);
            starts[b] = kth_unvisited((int)std::round(rnd * (double)(unvisited - 1)));
        }
        // a round works on the climbs that have not ended yet (the batch drains: most climbs end within a round or two,
        // a few take dozens); the result block of a climb that has ended keeps what its last round published
        MeanShiftActive active{};
        int n_active = climbs;
        for (int b = 0; b < climbs; ++b) active.climb[b] = (unsigned char)b;
        // r05: once few climbs are left they run to their end in ONE launch (k_ms_persist) instead of a launch per iteration
        // and a host round trip every few; a climb whose workgroups do not all become resident (a shared GPU) comes back
        // untouched, and the call goes on with launched rounds.
        bool persist_ok = e->tune_ms_persist > 0;
        int iters_seen[B];
        for (int b = 0; b < climbs; ++b) iters_seen[b] = 0;
        for (int round = 0; round < 20000 && n_active > 0; ++round) {     // rounds of device-side iterations
            int G = 0;                                         // > 0: this round runs persistently, G workgroups per climb
            if (persist_ok && round > 0 && n_active <= e->tune_ms_persist && ms_persist_supported(n, d)) {
                int& per_cu = d == 10 ? e->ms_persist_per_cu : e->ms_persist_per_cu6;
                if (per_cu < 0) { const int q = ms_persist_occupancy(d); if (q > 0) per_cu = q; }
                const int room = std::max(0, per_cu) * e->cu_count * 7 / 8;      // workgroups that are resident for sure
                const int groups = std::min(64, (n + 255) / 256);
                if (groups * n_active <= room) G = groups;
            }
            ++e->ms_rounds;
            const auto t_round = std::chrono::steady_clock::now();
            const int active_in = n_active;
            if (G > 0) {
                HIPCHK(launch_ms_persist(w, active, n_active, band_sq, stop_thresh, 1 << 20, e->ms_ctl.p, e->ms_partial2.p, e->ms_pcnt2.p,
                                         e->h_ms_dev, e->ms_heads.p, MS_LIST_PREFIX, e->stream, ms_stats ? e->ms_ticks.p : nullptr));
                ++e->ms_persist_launches;
                st_G += G;
            } else {
                HIPCHK(launch_ms_climb(w, active, n_active, round == 0 ? starts_dev : nullptr, band_sq, stop_thresh, e->tune_ms_batch,
                                       e->h_ms_dev, e->ms_heads.p, MS_LIST_PREFIX, e->ms_tickets.p, e->stream));
            }
            HIPCHK(hipStreamSynchronize(e->stream));
            if (ms_stats) {
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_round).count();
                long long its = 0;
                for (int a = 0; a < n_active; ++a) its = std::max<long long>(its, e->h_ms[active.climb[a]].out[0] - iters_seen[active.climb[a]]);
                if (G > 0) { st_persist_us += us; st_persist_iters += its; ++st_persist_rounds; st_persist_climbs += active_in; }
                else { st_launch_us += us; ++st_launch_rounds; if (round > 0) { st_tail_us += us; st_tail_climbs += active_in; } }
            }
            int still = 0;
            for (int a = 0; a < n_active; ++a) {
                const int b = active.climb[a];
                if (!e->h_ms[b].out[1] && !e->h_ms[b].out[3]) {
                    if (G > 0 && e->h_ms[b].out[0] == iters_seen[b]) { persist_ok = false; ++e->ms_persist_fallbacks; }   // its gate closed
                    active.climb[still++] = (unsigned char)b;
                }
                iters_seen[b] = e->h_ms[b].out[0];
            }
            n_active = still;
        }
        if (n_active > 0) {
            // a climb neither converged nor died within the cap: compact and clear the votes (they would leak into the
            // next call's membership lists) and give up loudly
            HIPCHK(launch_ms_collect(w, climbs, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            return fail(MH_ERR_INVALID, "mean shift: a climb did not converge within 20000 rounds of iterations");
        }
        // the heads of all lists in one copy: staged as [position][climb], so the first `longest` pairs of every climb are
        // one contiguous range
        int longest = 0;
        for (int b = 0; b < climbs; ++b) longest = std::max(longest, std::min(e->h_ms[b].out[2], MS_LIST_PREFIX));
        if (longest > 0) {
            HIPCHK(hipMemcpyAsync(e->h_ms_list, e->ms_heads.p, sizeof(int) * 2 * (size_t)B * longest, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
        // apply the climbs in draw order; one whose seed an earlier climb of the batch has visited never started in
        // the reference's terms and is dropped
        if (ms_stats)
            for (int b = 0; b < climbs; ++b) st_climbs.emplace_back(e->h_ms[b].out[0], e->h_ms[b].out[2]);
        for (int b = 0; b < climbs; ++b) {
            const int st = starts[b];
            if (visited[st]) continue;
            const int* out = e->h_ms[b].out;
            const double* mean = e->h_ms[b].mean;
            const int len = out[2];
            list.resize(2 * (size_t)len);
            const int head = std::min(len, MS_LIST_PREFIX);
            for (int k = 0; k < head; ++k) {
                const int* pr = e->h_ms_list + ((size_t)k * B + b) * 2;
                list[2 * k] = pr[0];
                list[2 * k + 1] = pr[1];
            }
            if (len > head) {
                HIPCHK(hipMemcpyAsync(list.data() + 2 * (size_t)head, e->ms_list.p + (size_t)b * 2 * n + 2 * (size_t)head,
                                      sizeof(int) * 2 * (size_t)(len - head), hipMemcpyDeviceToHost, e->stream));
                HIPCHK(hipStreamSynchronize(e->stream));
            }
            std::vector<std::pair<int, int>> mine(len);
            for (int k = 0; k < len; ++k) {
                mine[k] = { list[2 * k], list[2 * k + 1] };
                if (!visited[list[2 * k]]) { mark_visited(list[2 * k]); --unvisited; }
            }
            std::sort(mine.begin(), mine.end());
            if (!out[1]) {
                if (!visited[st]) { mark_visited(st); --unvisited; }    // climb that captured no row
                continue;
            }
            int merge_with = -1;
            // :101-109, first centroid with sqrt(sum) < bandWidth/2.  The running sum of squares only grows, so a
            // centroid is rejected as soon as it exceeds the squared limit by a safe margin; the deciding comparison
            // is the reference's own.
            const double half = band_width / 2, reject = half * half * (1.0 + 1e-9);
            for (int cn = 0; cn < n_cent && merge_with < 0; ++cn) {
                const double* c = cent.data() + (size_t)cn * d;
                double sq = 0.0;
                int j = 0;
                for (; j < d && sq <= reject; ++j) { const double x = mean[j] - c[j]; sq = sq + x * x; }
                if (j == d && std::sqrt(sq) < half) merge_with = cn;
            }
            if (merge_with > -1) {
                double* c = cent.data() + (size_t)merge_with * d;
                for (int j = 0; j < d; ++j) c[j] = 0.5 * (c[j] + mean[j]);
                std::vector<std::pair<int, int>> merged;
                const auto& a = votes[merge_with];
                size_t i = 0, k = 0;
                while (i < a.size() || k < mine.size()) {
                    if (k >= mine.size() || (i < a.size() && a[i].first < mine[k].first)) merged.push_back(a[i++]);
                    else if (i >= a.size() || mine[k].first < a[i].first) merged.push_back(mine[k++]);
                    else { merged.push_back({ a[i].first, a[i].second + mine[k].second }); ++i; ++k; }
                }
                votes[merge_with].swap(merged);
            } else {
                cent.insert(cent.end(), mean, mean + d);
                ++n_cent;
                votes.push_back(mine);
            }
        }
    }
    std::vector<int> best_votes(n, 0);
    for (int i = 0; i < n; ++i) assign[i] = -1;
    for (size_t r = 0; r < votes.size(); ++r)                       // :133-146, first maximum wins
        for (const auto& pr : votes[r])
            if (best_votes[pr.first] < pr.second) { best_votes[pr.first] = pr.second; assign[pr.first] = (int)r; }
    *n_modes = n_cent;
    if (modes) std::copy(cent.begin(), cent.begin() + (size_t)std::min(n_cent, max_modes) * d, modes);
    if (ms_stats) {
        unsigned long long tk[4] = { 0, 0, 0, 0 };
        (void)hipMemcpy(tk, e->ms_ticks.p, sizeof(tk), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mh_mean_shift] persistent kernel, first climb's first workgroup: gate %.1f ms, sweep + tree + partial stores %.1f ms, barrier %.1f ms, "
                        "new mean %.1f ms; mean G %.1f\n", tk[0] * 1e-5, tk[1] * 1e-5, tk[2] * 1e-5, tk[3] * 1e-5, st_persist_rounds ? (double)st_G / st_persist_rounds : 0.0);
    }
    if (ms_stats && !st_climbs.empty()) {
        // how long the climbs are and how many rows they touch: is the tail made of dense or of sparse climbs?
        const int edges[6] = { 2, 6, 12, 30, 100, 1 << 30 };
        int lo = 0;
        for (int q = 0; q < 6; ++q) {
            long long cnt = 0, its = 0;
            std::vector<int> touched;
            for (const auto& c : st_climbs) if (c.first > lo && c.first <= edges[q]) { ++cnt; its += c.first; touched.push_back(c.second); }
            std::sort(touched.begin(), touched.end());
            if (cnt) fprintf(stderr, "[mh_mean_shift]   climbs of %d..%d iterations: %lld (%lld iterations in sum); rows touched: median %d, 90 %% %d, max %d\n",
                             lo + 1, edges[q] > 100000 ? 99999 : edges[q], cnt, its, touched[touched.size() / 2], touched[touched.size() * 9 / 10], touched.back());
            lo = edges[q];
        }
    }
    if (ms_stats)
        fprintf(stderr, "[mh_mean_shift] n %d: %lld batches; launched rounds %lld (%.1f ms, of which rounds after the first %.1f ms on %lld climb-rounds); "
                        "persistent rounds %lld (%.1f ms, %lld climbs, longest climbs %lld iterations in sum = %.1f us per iteration)\n",
                n, st_batches, st_launch_rounds, st_launch_us * 1e-3, st_tail_us * 1e-3, st_tail_climbs, st_persist_rounds, st_persist_us * 1e-3,
                st_persist_climbs, st_persist_iters, st_persist_iters ? st_persist_us / (double)st_persist_iters : 0.0);
    return MH_OK;
    });
}

int mh_propose_dlt4(mh_engine* e, unsigned long long seed, long long first, int m)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (m <= 0) return fail(MH_ERR_INVALID, "m must be positive");
    if (e->n < 4) return fail(MH_ERR_INVALID, "need at least 4 correspondences");
    HIPCHK(e->H.reserve((size_t)m * 9));
    HIPCHK(e->samples.reserve((size_t)m * 4));
    HIPCHK(reserve_counts(e, (size_t)m + 1));
    {
        ScopedTimer t(e, MH_K_DLT4);
        HIPCHK(launch_dlt4(e->pts(), seed, first, m, e->samples.p, e->H.p, e->stream, e->tune_dlt_variant == 1 ? 1 : 0));   // alone on the device: the register form (0.26 against 0.40 ms per 100k)
    }
    e->m = m;
    e->have_samples = true;
    e->cost_L = 0;
    e->counts_fresh = false; ++e->models_seq;
    return MH_OK;
    });
}

int mh_set_models(mh_engine* e, const double* H, int m)
{
    return guarded([&]() -> int {
    if (!e || m < 0 || (m > 0 && !H)) return fail(MH_ERR_INVALID, "null argument or m < 0");
    HIPCHK(hipSetDevice(e->device));
    e->counts_fresh = false; ++e->models_seq;
    if (m == 0) { e->m = 0; e->have_samples = false; e->cost_L = 0; return MH_OK; }     // an empty model set
    HIPCHK(e->H.reserve((size_t)m * 9));
    HIPCHK(reserve_counts(e, (size_t)m + 1));
    HIPCHK(hipMemcpyAsync(e->H.p, H, sizeof(double) * 9 * m, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->m = m;
    e->have_samples = false;
    e->cost_L = 0;
    return MH_OK;
    });
}

int mh_get_models(mh_engine* e, double* H)
{
    return guarded([&]() -> int {
    if (!e || !H) return fail(MH_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->device));
    if (e->m <= 0) return fail(MH_ERR_NOT_SET, "model set is empty");
    HIPCHK(hipMemcpyAsync(H, e->H.p, sizeof(double) * 9 * e->m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_get_model_count(mh_engine* e, int* m)
{
    return guarded([&]() -> int {
    if (!e || !m) return fail(MH_ERR_INVALID, "null argument");
    *m = e->m;
    return MH_OK;
    });
}

int mh_get_samples(mh_engine* e, int* idx)
{
    return guarded([&]() -> int {
    if (!e || !idx) return fail(MH_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->device));
    if (!e->have_samples) return fail(MH_ERR_NOT_SET, "no sampled batch; call mh_propose_dlt4");
    HIPCHK(hipMemcpyAsync(idx, e->samples.p, sizeof(int) * 4 * e->m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_set_residual_mode(mh_engine* e, int mode)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    if (mode != MH_RESIDUAL_FORWARD && mode != MH_RESIDUAL_SYMMETRIC) return fail(MH_ERR_INVALID, "unknown residual mode");
    e->residual_mode = mode;
    return MH_OK;
    });
}

int mh_score(mh_engine* e, double thr2, const unsigned char* point_mask, int* counts)
{
    return guarded([&]() -> int {
    bool empty_shard = false;
    int rc = require_models_or_empty_shard(e, &empty_shard);
    if (rc || empty_shard) return rc;
    HIPCHK(reserve_counts(e, (size_t)e->m + 1));
    const unsigned char* dmask = nullptr;
    if (point_mask) {
        HIPCHK(e->mask.reserve((size_t)e->n + 2));
        HIPCHK(hipMemcpyAsync(e->mask.p, point_mask, e->n, hipMemcpyHostToDevice, e->stream));
        dmask = e->mask.p;
    }
    {
        ScopedTimer t(e, MH_K_SCORE);
        rc = score_models(e, e->pts(), e->H.p, e->m, thr2, dmask, e->counts.p);
        if (rc) return rc;
    }
    e->counts_zeroed = false;
    e->counts_fresh = true;
    if (counts) {
        HIPCHK(hipMemcpyAsync(counts, e->counts.p, sizeof(int) * e->m, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return MH_OK;
    });
}

int mh_residual_matrix(mh_engine* e, double thr2, double* R_host, int* counts)
{
    return guarded([&]() -> int {
    bool empty_shard = false;
    int rc = require_models_or_empty_shard(e, &empty_shard);
    if (rc || empty_shard) return rc;
    e->ldr = residual_ld(e->n);
    size_t r_elems = (size_t)e->m * (size_t)e->ldr;
#ifdef MH_TUNING
    if (e->tune_ld > 0) { e->ldr = std::max<long long>(e->tune_ld, e->ldr); r_elems = (size_t)e->m * (size_t)e->ldr; }
    // the tile-major measurement variants write whole 16-model x 1024-point blocks
    r_elems = std::max(r_elems, (size_t)((e->m + 15) / 16 * 16) * (size_t)((e->n + 1023) / 1024 * 1024));
#endif
    HIPCHK(e->R.reserve(r_elems));
    HIPCHK(reserve_counts(e, (size_t)e->m + 1));
    // The sweep runs as a RESIDENT grid — as many workgroups as the chip holds at the kernel's five waves per SIMD, handing
    // themselves the (model block, point slice) items through a counter (residual.hip, k_residual_resident): 7.27-7.29 ms
    // against 7.48-7.70 for one hardware-dispatched workgroup per item at 50k x 100k, and it leaves 72 registers per SIMD
    // free on every compute unit, which is what a workgroup of the DLT solve needs.  Key 19: -1 = hardware dispatch,
    // h >= 0 = leave h more workgroup slots free.
    int resident = 0;
    if (e->tune_sweep_headroom >= 0 && e->residual_mode != MH_RESIDUAL_SYMMETRIC && e->tune_residual_variant == 0) {
        if (e->sweep_wg_per_cu < 0) e->sweep_wg_per_cu = residual_workgroups_per_cu();
        // With a stream-ordered transport over several ranks, RCCL's own kernel (one or two workgroups for an exchange of this
        // size) has to find room while the NEXT sweep is resident: 32 slots are left free for it unless the caller chose.
        const int headroom = e->tune_sweep_headroom > 0 ? e->tune_sweep_headroom : (e->t_stream_fn && e->t_world > 1 ? 32 : 0);
        resident = e->sweep_wg_per_cu * e->cu_count - headroom;
        if (resident < e->cu_count) resident = 0;
        if (resident > 0 && !e->sweep_ctl.p) {
            HIPCHK(e->sweep_ctl.reserve(2));
            HIPCHK(hipMemsetAsync(e->sweep_ctl.p, 0, sizeof(int) * 2, e->stream));
        }
    }
    // ... and it starts behind the DLT's dispatch, not beside it: a sweep that reaches the chip first fills every
    // workgroup slot and keeps them (its queue is dispatched ahead of the other stream's whatever the priorities), and
    // the DLT then runs after the sweep instead of beside it — the next sweep waits for it (profiles/r04_timeline_*.txt).
    // (only when the pending batch is the very NEXT one: with two batches queued the DLT the next sweep waits for was dispatched
    // a sweep ago, and the one just enqueued has a whole sweep of slack — it runs in this sweep's tail)
    if (e->pf_count == 1 && e->tune_dlt_first && e->ev_side_pre) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_side_pre, 0));
    {
        ScopedTimer t(e, MH_K_RESIDUAL);
        HIPCHK(launch_residual(e->pts(), e->H.p, e->m, thr2, e->R.p, e->ldr, e->counts.p,
                               e->residual_mode == MH_RESIDUAL_SYMMETRIC ? -1 : e->tune_residual_variant,
                               e->stream, e->counts_zeroed, resident, e->sweep_ctl.p, e->tune_sweep_slices, e->tune_sweep_slices > 0 ? 1 : 0));
    }
    e->counts_zeroed = false;
    e->counts_fresh = true;
    if (R_host)
        HIPCHK(hipMemcpy2DAsync(R_host, sizeof(double) * e->n, e->R.p, sizeof(double) * e->ldr,
                                sizeof(double) * e->n, e->m, hipMemcpyDeviceToHost, e->stream));
    if (counts)
        HIPCHK(hipMemcpyAsync(counts, e->counts.p, sizeof(int) * e->m, hipMemcpyDeviceToHost, e->stream));
    if (R_host || counts) HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_cost_matrix(mh_engine* e, int* C_host, int* counts)
{
    return guarded([&]() -> int {
    bool empty_shard = false;
    int rc = require_models_or_empty_shard(e, &empty_shard);
    if (rc || empty_shard) return rc;
    e->ldc = cost_ld(e->n);
    HIPCHK(e->C.reserve((size_t)e->m * (size_t)e->ldc));
    HIPCHK(reserve_counts(e, (size_t)e->m + 1));
    {
        ScopedTimer t(e, MH_K_COSTMATRIX);
        const double thr2 = e->thr_H * e->thr_H;
        if (e->tune_score32 && e->coords32_ok && thr2 >= 0x1p-40 && thr2 <= 0x1p40) {      // the FP32 pre-test (score32.hip); same matrix
            HIPCHK(e->H32.reserve((size_t)e->m * 16));
            HIPCHK(launch_model32(e->H.p, e->m, e->absmax_x, e->absmax_y, e->absmax_dst, e->H32.p, e->stream));
            int* ctl = nullptr;
            if (e->tune_cost32_resident != 0) {
                if (!e->sweep_ctl.p) {
                    HIPCHK(e->sweep_ctl.reserve(2));
                    HIPCHK(hipMemsetAsync(e->sweep_ctl.p, 0, sizeof(int) * 2, e->stream));
                }
                ctl = e->sweep_ctl.p;
            }
            HIPCHK(launch_cost32(e->pts(), e->H.p, e->H32.p, e->m, e->lambda, thr2, e->absmax_dst, e->C.p, e->ldc, e->counts.p, e->stream,
                                 ctl, e->cu_count, e->tune_cost32_resident > 0 ? e->tune_cost32_resident : 0, e->tune_cost32_slice_major, e->tune_cost32_batched, &e->occ_cost32));
        } else
            HIPCHK(launch_cost_matrix(e->pts(), e->H.p, e->m, e->lambda, thr2, e->C.p, e->ldc, e->counts.p, e->stream));
    }
    e->counts_zeroed = false;
    e->counts_fresh = true;
    if (C_host)
        HIPCHK(hipMemcpy2DAsync(C_host, sizeof(int) * e->n, e->C.p, sizeof(int) * e->ldc, sizeof(int) * e->n, e->m,
                                hipMemcpyDeviceToHost, e->stream));
    if (counts) HIPCHK(hipMemcpyAsync(counts, e->counts.p, sizeof(int) * e->m, hipMemcpyDeviceToHost, e->stream));
    if (C_host || counts) HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_get_residual_rows(mh_engine* e, int first, int count, double* rows_host)
{
    return guarded([&]() -> int {
    int rc = require_models(e);
    if (rc) return rc;
    if (!e->R.p || e->ldr <= 0) return fail(MH_ERR_NOT_SET, "residual matrix has not been computed");
    if (first < 0 || count <= 0 || first + count > e->m || !rows_host)
        return fail(MH_ERR_INVALID, "row range out of bounds or null output");
    HIPCHK(hipMemcpy2DAsync(rows_host, sizeof(double) * e->n, e->R.p + (size_t)first * e->ldr,
                            sizeof(double) * e->ldr, sizeof(double) * e->n, count,
                            hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

// all-gather on the engine's stream through whichever transport is set; the host-synchronised hook sees an idle stream
static int exchange(mh_engine* e, const void* send_dev, void* recv_dev, size_t bytes_per_rank, hipStream_t on)
{
    if (e->t_stream_fn) {
        if (e->t_stream_fn(e->t_ctx, send_dev, recv_dev, (unsigned long long)bytes_per_rank, (void*)on) != 0)
            return fail(MH_ERR_INVALID, "all-gather failed (stream-ordered transport)");
        return MH_OK;
    }
    if (!e->t_host_fn) return fail(MH_ERR_NOT_SET, "no transport set (mh_set_transport)");
    HIPCHK(hipStreamSynchronize(on));
    if (e->t_host_fn(e->t_ctx, send_dev, recv_dev, (unsigned long long)bytes_per_rank) != 0)
        return fail(MH_ERR_INVALID, "all-gather failed (host-synchronised transport)");
    return MH_OK;
}

int mh_set_transport(mh_engine* e, int rank, int world, mh_allgather_stream_fn stream_fn, mh_allgather_dev_fn host_fn, void* ctx)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    if (world < 1 || rank < 0 || rank >= world) return fail(MH_ERR_INVALID, "bad rank / world");
    if (stream_fn && host_fn) return fail(MH_ERR_INVALID, "give ONE transport: stream-ordered or host-synchronised");
    if (world > 1 && !stream_fn && !host_fn) return fail(MH_ERR_INVALID, "world > 1 needs a transport");
    if (e->xchg_pending) {                                 // an exchange in flight still uses the old transport
        HIPCHK(hipSetDevice(e->device));
        int rcq = quiesce(e);
        if (rcq) return rcq;
    }
    e->t_rank = rank; e->t_world = world; e->t_stream_fn = stream_fn; e->t_host_fn = host_fn; e->t_ctx = ctx;
    return MH_OK;
    });
}

int mh_select_greedy(mh_engine* e, double thr2, int need, int max_models, unsigned char* point_mask,
                     double* H_out, long long* counters_out, int* counts_out, int* selected_out, long long total_m)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (!H_out || !selected_out || max_models <= 0 || need < 1) return fail(MH_ERR_INVALID, "bad argument");
    const int n = e->n;
    int M = e->m;                                     // M may be 0 on a rank without hypotheses (more ranks than hypotheses)
    // The transport is used whenever one is set — also with world == 1, where a one-rank communicator runs the whole
    // protocol (how the RCCL path is tested on a box with one GPU).
    const bool sharded = e->t_stream_fn || e->t_host_fn;
    const int world = sharded ? e->t_world : 1, rank = sharded ? e->t_rank : 0;
    if (total_m <= 0) total_m = M;
    if (total_m > 0xfffffffell) return fail(MH_ERR_INVALID, "more than 2^32 - 2 hypotheses in a batch");
    // contiguous shards of the whole batch, the first `rem` one hypothesis longer
    const int base = (int)(total_m / world), rem = (int)(total_m % world);
    const int longest = base + (rem ? 1 : 0);
    const int mine = base + (rank < rem ? 1 : 0);
    const unsigned int my_off = (unsigned int)((long long)rank * base + std::min(rank, rem));
    // What can be wrong on THIS rank only — the state of its engine — must not keep it out of the collectives: the other
    // ranks would wait in the all-gather for ever (r03 advisor finding).  Such a failure is remembered (code + text), the
    // rank goes through one round with an empty candidate list and its error word set, every rank reads that word
    // after the exchange and all of them leave together; this rank then reports its own failure.
    int local_rc = MH_OK;
    std::string local_msg;
    auto local_failure = [&](int code, const std::string& msg) { if (local_rc == MH_OK) { local_rc = code; local_msg = msg; } };
    // r05: the selection follows the engine's residual mode — scores (score_models) and claims (k_sel_claim) both on the
    // symmetric transfer error when that is set; the ranks of a sharded batch must agree (their records carry the mode)
    const int symmetric = e->residual_mode == MH_RESIDUAL_SYMMETRIC ? 1 : 0;
    if (!sharded && M <= 0) local_failure(MH_ERR_NOT_SET, "model set is empty");
    else if (M != mine) local_failure(MH_ERR_INVALID, "the resident model set is not this rank's shard of total_m hypotheses");
    if (local_rc != MH_OK && !sharded) return fail(local_rc, local_msg);
    if (local_rc != MH_OK) M = 0;
    rc = join_xchg(e);                                 // an mh_select_best exchange still in flight shares the gather buffer
    if (rc) return rc;
    const size_t cap = (size_t)std::max(M, 1);
    for (int b = 0; b < 2; ++b) { HIPCHK(e->sel_orig[b].reserve(cap)); HIPCHK(e->sel_cand_H[b].reserve(cap * 9)); }
    HIPCHK(e->sel_counts.reserve(cap));
    HIPCHK(e->sel_rec.reserve(8));
    HIPCHK(e->sel_keys.reserve(2));
    HIPCHK(e->sel_out_H.reserve((size_t)max_models * 9));
    HIPCHK(e->sel_counter.reserve(max_models));
    HIPCHK(e->sel_records.reserve((size_t)world + 1));
    HIPCHK(e->mask.reserve((size_t)n + 2));
    if (sharded) {
        HIPCHK(e->sel_scores.reserve((size_t)std::max(longest, 1)));
        HIPCHK(e->sel_gathered.reserve((size_t)world * std::max(longest, 1)));
    }
    if (!e->h_sel) {
        HIPCHK(hipHostMalloc((void**)&e->h_sel, sizeof(int) * 8, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_sel_dev, e->h_sel, 0));
    }
    hipStream_t s = e->stream;
    if (point_mask) { HIPCHK(hipMemcpyAsync(e->mask.p, point_mask, n, hipMemcpyHostToDevice, s)); ++e->copies_h2d; }
    else HIPCHK(hipMemsetAsync(e->mask.p, 1, n, s));
    HIPCHK(hipMemsetAsync(e->sel_rec.p, 0, sizeof(int) * 8, s));
    HIPCHK(hipMemsetAsync(e->sel_keys.p, 0, sizeof(unsigned long long) * 2, s));
    HIPCHK(hipMemsetAsync(e->sel_records.p, 0, sizeof(SelRecord) * ((size_t)world + 1), s));
    if (sharded && longest > 0) HIPCHK(hipMemsetAsync(e->sel_scores.p, 0xff, sizeof(int) * (size_t)longest, s));   // -1: padding
    unsigned long long* key_local = e->sel_keys.p;
    unsigned long long* key_check = e->sel_keys.p + 1;     // the first round's winner as the gathered score vector gives it
    SelRecord* my_record = e->sel_records.p;
    SelRecord* records = sharded ? e->sel_records.p + 1 : e->sel_records.p;

    // The support set only shrinks (the inliers of every selected model leave it), and the score kernel pays per point
    // it sweeps: every round scores the PACKED active points.  Their number is known on the host without a copy — the
    // caller's mask at the start, minus each selected model's count afterwards.
    int active = n;
    if (point_mask) { active = 0; for (int i = 0; i < n; ++i) active += point_mask[i] != 0 ? 1 : 0; }
    for (int c = 0; c < 4; ++c) HIPCHK(e->sel_pts[c].reserve((size_t)n + 2));
    HIPCHK(e->sel_pack_count.reserve(1));

    int Mc = M, cur = 0, selected = 0, packed_as = -1;
    bool first = true;
    for (int round = 0; round < max_models; ++round) {
        const double* Hs = first ? e->H.p : e->sel_cand_H[cur].p;
        const int* orig = first ? nullptr : e->sel_orig[cur].p;
        // the rank-local part of a round: score the candidates.  A failure here does not return before the collectives.
        auto score_round = [&]() -> int {
            if (e->inject_select_failure > 0 && --e->inject_select_failure == 0)
                return fail(MH_ERR_HIP, "greedy selection: injected rank-local failure (test hook, mh_set_tuning key 18)");
            if (Mc <= 0) return MH_OK;
            ScopedTimer t(e, MH_K_SCORE);
            if (active == n) return score_models(e, e->pts(), Hs, Mc, thr2, nullptr, e->sel_counts.p);     // every point is in the support set: no mask to read
            if (active > 0) {
                HIPCHK(launch_sel_pack_points(e->pts(), e->mask.p, e->sel_pts[0].p, e->sel_pts[1].p, e->sel_pts[2].p, e->sel_pts[3].p,
                                              e->sel_pack_count.p, s));
                packed_as = active;
                Points packed = e->pts();                     // (same bounding box: a superset's is valid)
                packed.x1 = e->sel_pts[0].p; packed.y1 = e->sel_pts[1].p; packed.x2 = e->sel_pts[2].p; packed.y2 = e->sel_pts[3].p;
                packed.n = active;
                return score_models(e, packed, Hs, Mc, thr2, nullptr, e->sel_counts.p);
            }
            HIPCHK(hipMemsetAsync(e->sel_counts.p, 0, sizeof(int) * (size_t)Mc, s));
            return MH_OK;
        };
        if (local_rc == MH_OK) {
            const int src = score_round();
            if (src != MH_OK) {
                if (!sharded) return src;
                local_failure(src, g_err);
                Mc = 0;                                        // offer nothing; the error word tells the others
            }
        }
        const int local_err = local_rc != MH_OK ? 1 : 0;
        const bool gather_scores = sharded && first && longest > 0;      // north_star's exchange, once per batch
        HIPCHK(launch_sel_argmax(e->sel_counts.p, orig, Mc, my_off, key_local, gather_scores ? e->sel_scores.p : nullptr, s));
        HIPCHK(launch_sel_record(e->sel_counts.p, orig, Hs, Mc, my_off, key_local, local_err, symmetric, my_record, s));
        if (sharded) {
            if (gather_scores) {
                rc = exchange(e, e->sel_scores.p, e->sel_gathered.p, sizeof(int) * (size_t)longest, s);
                if (rc) return rc;
                HIPCHK(launch_sel_argmax_gathered(e->sel_gathered.p, world, longest, base, rem, key_check, s));
            }
            rc = exchange(e, my_record, records, sizeof(SelRecord), s);     // 88 bytes per rank
            if (rc) return rc;
        }
        HIPCHK(launch_sel_compact(e->sel_counts.p, orig, Hs, Mc, need, records, world, my_off, e->sel_orig[cur ^ 1].p,
                                  e->sel_cand_H[cur ^ 1].p, e->sel_rec.p, s));
        HIPCHK(launch_sel_claim(e->pts(), records, world, gather_scores && !local_err ? key_check : nullptr, thr2, need, e->mask.p, e->sel_rec.p,
                                e->sel_out_H.p, e->sel_counter.p, max_models, s, symmetric));
        HIPCHK(launch_sel_publish(e->sel_rec.p, e->sel_keys.p, my_record, need, e->h_sel_dev, s));
        HIPCHK(hipStreamSynchronize(s));                 // five control words through mapped memory: no copy
        if (local_rc != MH_OK) return fail(local_rc, local_msg);     // (the others have read this rank's error word by now)
        if (e->h_sel[4] != 0)                            // every rank sees the same word, so every rank leaves here
            return fail(e->h_sel[4] == 3 ? MH_ERR_INVALID : MH_ERR_HIP,
                        e->h_sel[4] == 2 ? "greedy selection: the gathered score vector and the ranks' records disagree about the winner"
                        : e->h_sel[4] == 3 ? "greedy selection: the ranks are not in the same residual mode (mh_set_residual_mode)"
                                           : "greedy selection: a rank reported an error");
        const int best = e->h_sel[0];
        if (best < need) break;
        if (counts_out) counts_out[selected] = best;
        ++selected;
        active -= best;                                   // the selected model's inliers have left the support set
        Mc = e->h_sel[2];
        cur ^= 1;
        first = false;
    }
    *selected_out = selected;
    if (selected > 0) {
        HIPCHK(hipMemcpyAsync(H_out, e->sel_out_H.p, sizeof(double) * 9 * (size_t)selected, hipMemcpyDeviceToHost, s));
        ++e->copies_d2h;
        if (counters_out) {
            HIPCHK(hipMemcpyAsync(counters_out, e->sel_counter.p, sizeof(long long) * (size_t)selected, hipMemcpyDeviceToHost, s));
            ++e->copies_d2h;
        }
    }
    if (point_mask) { HIPCHK(hipMemcpyAsync(point_mask, e->mask.p, n, hipMemcpyDeviceToHost, s)); ++e->copies_d2h; }
    int packed_n = packed_as;
    if (packed_as >= 0) { HIPCHK(hipMemcpyAsync(&packed_n, e->sel_pack_count.p, sizeof(int), hipMemcpyDeviceToHost, s)); ++e->copies_d2h; }
    HIPCHK(hipStreamSynchronize(s));
    if (packed_n != packed_as)                           // the host's bookkeeping of the support set against the device's own count
        return fail(MH_ERR_HIP, "greedy selection: the packed support set does not have the expected size");
    return MH_OK;
    });
}

// ---- pipelined propose -------------------------------------------------------------------------
static int ensure_side_stream(mh_engine* e)
{
    if (!e->side_stream) {
        for (int k = 0; k < e->tune_stream_shift; ++k) {
            hipStream_t d = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
            e->dummy_streams.push_back(d);
        }
        // highest priority: the short DLT kernel gets its compute units as soon as the sweep on the main stream frees
        // some, so it is done early in the sweep instead of trickling in behind it and delaying the next one
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&e->side_stream, hipStreamNonBlocking, hi));
    }
    for (int q = 0; q < mh_engine::PF_DEPTH; ++q)
        if (!e->pf_ev[q]) HIPCHK(hipEventCreateWithFlags(&e->pf_ev[q], hipEventDisableTiming));
    if (!e->ev_main) HIPCHK(hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming));
    if (!e->ev_side_pre) HIPCHK(hipEventCreateWithFlags(&e->ev_side_pre, hipEventDisableTiming));
    return MH_OK;
}

int mh_prefetch_dlt4(mh_engine* e, unsigned long long seed, long long first, int m)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (m <= 0) return fail(MH_ERR_INVALID, "m must be positive");
    if (e->n < 4) return fail(MH_ERR_INVALID, "need at least 4 correspondences");
    rc = ensure_side_stream(e);
    if (rc) return rc;
    if (e->pf_count >= mh_engine::PF_DEPTH) return fail(MH_ERR_INVALID, "two batches are already prefetched: adopt one first (mh_adopt_prefetched)");
    const int slot = (e->pf_head + e->pf_count) % mh_engine::PF_DEPTH;
    if (e->pf_H[slot].cap < (size_t)m * 9 || e->pf_samples[slot].cap < (size_t)m * 4) {
        // (re)allocation: nothing may still be reading the spare buffers
        HIPCHK(hipStreamSynchronize(e->side_stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(e->pf_H[slot].reserve((size_t)m * 9));
        HIPCHK(e->pf_samples[slot].reserve((size_t)m * 4));
    }
    // The slot's buffers held a batch that was current before an adoption; kernels of the main stream enqueued up to now
    // may still read them.
    HIPCHK(hipEventRecord(e->ev_main, e->stream));
    HIPCHK(hipStreamWaitEvent(e->side_stream, e->ev_main, 0));
    HIPCHK(hipEventRecord(e->ev_side_pre, e->side_stream));      // the second stream has got as far as this batch's dispatch
    {
        ScopedTimer t(e, MH_K_DLT4, e->side_stream);           // (the kernel's span on the second stream, beside whatever the main one runs)
        // Beside a resident sweep the LDS-staged form is the better one although it is 1.5 x slower alone: its 72 registers
        // fit next to the sweep's five waves per SIMD, so it shares the compute units with the sweep's head, while the
        // register form (128) has to displace sweep workgroups and its run time is added to the step: 0.964 against 0.994 ms
        // per step at the 12 500-hypothesis shard, 1.891 / 1.928 at 25 000, 7.41 / 7.41 at 100 000 (tools/shard_proxy.py DLTFORM=1).
        HIPCHK(launch_dlt4(e->pts(), seed, first, m, e->pf_samples[slot].p, e->pf_H[slot].p, e->side_stream, e->tune_dlt_variant == 2 ? 0 : 1));
    }
    HIPCHK(hipEventRecord(e->pf_ev[slot], e->side_stream));
    e->pf_m[slot] = m;
    ++e->pf_count;
    return MH_OK;
    });
}

int mh_adopt_prefetched(mh_engine* e)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (e->pf_count <= 0) return fail(MH_ERR_NOT_SET, "no prefetched batch (mh_prefetch_dlt4)");
    const int slot = e->pf_head;
    HIPCHK(hipStreamWaitEvent(e->stream, e->pf_ev[slot], 0));    // main-stream work behind this point sees the new batch
    std::swap(e->H, e->pf_H[slot]);
    std::swap(e->samples, e->pf_samples[slot]);
    HIPCHK(reserve_counts(e, (size_t)e->pf_m[slot] + 1));
    e->m = e->pf_m[slot];
    e->pf_head = (e->pf_head + 1) % mh_engine::PF_DEPTH;
    --e->pf_count;
    e->have_samples = true;
    e->cost_L = 0;
    e->counts_fresh = false; ++e->models_seq;
    return MH_OK;
    });
}

// ---- best model of the scored batch --------------------------------------------------------------
static int ensure_xchg_stream(mh_engine* e)
{
    if (!e->xchg_stream) {
        // high priority, like the DLT's stream: the two or three short kernels of an exchange (and RCCL's own) get compute
        // units as soon as the sweep on the main stream frees some
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&e->xchg_stream, hipStreamNonBlocking, hi));
    }
    if (!e->ev_sweep) HIPCHK(hipEventCreateWithFlags(&e->ev_sweep, hipEventDisableTiming));
    for (int b = 0; b < 3; ++b)
        if (!e->ev_x[b]) HIPCHK(hipEventCreateWithFlags(&e->ev_x[b], hipEventDisableTiming));
    return MH_OK;
}

// The exchange is OFF the sweep's critical path (r04): with a stream-ordered transport (or none) the
// all-gather, the arg-max and the publication of batch i are enqueued on a third stream behind an event of sweep i, and
// the main stream goes straight on to sweep i+1.  The ranks' send buffer is the batch's own counts buffer (no padding
// kernel: a shard one shorter than the longest carries its -1 in the element behind its counts), which stays with the
// exchange while the next sweep writes the engine's other counts buffer.  The host-synchronised transport (several ranks
// rehearsing on one GPU) keeps the r03 form: everything on the main stream.
int mh_select_best(mh_engine* e, long long total_m, long long* best_index, int* best_count)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    const bool sharded = e->t_stream_fn || e->t_host_fn;       // also with world == 1: a one-rank communicator runs the exchange
    const int world = sharded ? e->t_world : 1, rank = sharded ? e->t_rank : 0;
    // a rank may hold an EMPTY shard (more ranks than hypotheses): it still takes part in the collective
    if (e->m <= 0 && (!sharded || total_m <= 0)) return fail(MH_ERR_NOT_SET, "model set is empty");
    HIPCHK(reserve_counts(e, (size_t)e->m + 1));
    if (total_m <= 0) total_m = e->m;
    const int base = (int)(total_m / world), rem = (int)(total_m % world);
    const int longest = base + (rem ? 1 : 0);
    HIPCHK(e->best_key.reserve(1));
    if (!e->h_best) {
        HIPCHK(hipHostMalloc((void**)&e->h_best, sizeof(int) * 4, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_best_dev, e->h_best, 0));
        e->h_best[0] = e->h_best[1] = e->h_best[2] = e->h_best[3] = 0;
        HIPCHK(hipMemsetAsync(e->best_key.p, 0, sizeof(unsigned long long), e->stream));
    }
    hipStream_t s = e->stream;
    const bool fetch = best_index || best_count;
    auto result = [&]() -> int {
        if (e->h_best[2] != e->best_seq) return fail(MH_ERR_HIP, "best-model result is stale");
        if (e->h_best[3] != 0) return fail(MH_ERR_HIP, "mh_select_best: a rank reported an error");     // (every rank reads the same word)
        if (best_index) *best_index = e->h_best[1];
        if (best_count) *best_count = e->h_best[0];
        return MH_OK;
    };
    // Is this call a NEW exchange?  Decided from state that is the same on every rank (r04 advisor finding: a rank with an
    // empty shard must not run a collective its peers skip): the model-set generation the last exchange belongs to, and
    // whether anything has been scored since — a scoring call on an empty shard is a no-op that still counts
    // (require_models_or_empty_shard), so ranks that make the same calls agree.
    const bool same_generation = e->best_seq != 0 && e->best_models_seq == e->models_seq;
    if (same_generation && !e->counts_fresh) {
        // nothing has been scored since the last call: that call's result is the answer ("a later call with outputs
        // completes it")
        if (!fetch) return MH_OK;
        rc = quiesce(e);
        if (rc) return rc;
        return result();
    }
    // Rank-local failures do not leave before the collective (their peers would wait in it for ever): the rank sends error
    // markers instead of scores — every rank's arg-max launch sees them and every rank's fetch fails.
    int local_rc = MH_OK;
    const char* local_msg = "";
    if (e->m != base + (rank < rem ? 1 : 0)) { local_rc = MH_ERR_INVALID; local_msg = "the resident model set is not this rank's shard of total_m hypotheses"; }
    else if (e->m > 0 && !e->counts_fresh) { local_rc = MH_ERR_NOT_SET; local_msg = "the batch has not been scored (mh_residual_matrix / mh_score / mh_cost_matrix)"; }
    if (local_rc != MH_OK && !sharded) return fail(local_rc, local_msg);
    if (local_rc != MH_OK) {
        HIPCHK(reserve_counts(e, (size_t)std::max(longest, e->m) + 1));
        if (longest > 0) HIPCHK(hipMemsetAsync(e->counts.p, 0xfe, sizeof(int) * (size_t)longest, s));     // 0xfefefefe < -1: the error marker
        e->counts_zeroed = false;
    }
    const int mine = local_rc != MH_OK ? longest : e->m;           // valid entries at the head of this rank's send buffer
    if (e->t_host_fn) {
        // host-synchronised transport: everything on the main stream
        rc = join_xchg(e);
        if (rc) return rc;
        HIPCHK(e->sel_scores.reserve((size_t)std::max(longest, 1)));
        HIPCHK(e->sel_gathered.reserve((size_t)world * std::max(longest, 1)));
        HIPCHK(launch_pad_scores(e->counts.p, mine, longest, e->sel_scores.p, s));
        rc = exchange(e, e->sel_scores.p, e->sel_gathered.p, sizeof(int) * (size_t)longest, s);     // north_star's all-gather
        if (rc) return rc;
        HIPCHK(launch_best_fused(e->sel_gathered.p, world, longest, base, rem, e->h_best_dev, nullptr, 0, s));
        ++e->best_seq;
        e->best_models_seq = e->models_seq;
        e->counts_fresh = false;                                   // (a later call without a scoring call in between is a completion, on every rank)
        if (fetch || local_rc != MH_OK) HIPCHK(hipStreamSynchronize(s));
    } else {
        rc = ensure_xchg_stream(e);
        if (rc) return rc;
        hipStream_t x = e->xchg_stream;
        if (sharded) {
            if (e->sel_gathered.cap < (size_t)world * longest) {
                rc = quiesce(e);                                   // (re)allocation: an earlier exchange may still write the old buffer
                if (rc) return rc;
                HIPCHK(e->sel_gathered.reserve((size_t)world * longest));
            }
            if (longest > mine)                                    // a shard one shorter than the longest: its padding element (every
                HIPCHK(hipMemsetAsync(e->counts.p + mine, 0xff, sizeof(int) * (size_t)(longest - mine), s));   // counts buffer holds m + 1 ints)
        }
        HIPCHK(hipEventRecord(e->ev_sweep, s));                    // the sweep (and whatever else the main stream holds) up to here
        HIPCHK(hipStreamWaitEvent(x, e->ev_sweep, 0));
        // enqueue-only: this batch's counts buffer comes back to the main stream two calls from now — cleared by the same
        // launch that reads it, so that the sweep that then writes it needs no memset of its own on the main stream
        const bool rotate = !fetch && local_rc == MH_OK;
        int* clear = rotate ? e->counts.p : nullptr;
        const int clear_count = rotate ? (int)e->counts.cap : 0;     // (all of it: the next batch it serves may be larger)
        if (sharded) {
            rc = exchange(e, e->counts.p, e->sel_gathered.p, sizeof(int) * (size_t)longest, x);      // north_star's all-gather
            if (rc) return rc;
            HIPCHK(launch_best_fused(e->sel_gathered.p, world, longest, base, rem, e->h_best_dev, clear, clear_count, x));
        } else {
            HIPCHK(launch_best_fused(e->counts.p, 1, e->m, 0, 0, e->h_best_dev, clear, clear_count, x));
        }
        const int par = (int)(e->xchg_calls % 3);
        HIPCHK(hipEventRecord(e->ev_x[par], x));
        ++e->xchg_calls;
        e->xchg_pending = true;
        ++e->best_seq;
        e->best_models_seq = e->models_seq;
        if (!rotate) {
            HIPCHK(hipStreamSynchronize(x));
            e->xchg_pending = false;
            e->counts_fresh = false;                               // this exchange is done; without a new scoring call the next call returns its result
        } else {
            // this batch's counts stay with the exchange; the next sweep writes the buffer that has waited longest — once the
            // exchange that was given THAT one (two calls ago) is through
            DevBuf<int> given = e->counts;
            const int wait = e->counts_alt_wait[0];
            e->counts = e->counts_alt[0];
            e->counts_zeroed = e->counts_zeroed_alt[0];
            e->counts_alt[0] = e->counts_alt[1]; e->counts_zeroed_alt[0] = e->counts_zeroed_alt[1]; e->counts_alt_wait[0] = e->counts_alt_wait[1];
            e->counts_alt[1] = given; e->counts_zeroed_alt[1] = true; e->counts_alt_wait[1] = par;      // (clear once ev_x[par] has passed)
            e->counts_fresh = false;
            HIPCHK(reserve_counts(e, (size_t)e->m + 1));
            if (wait >= 0) HIPCHK(hipStreamWaitEvent(s, e->ev_x[wait], 0));
        }
    }
    if (local_rc != MH_OK) return fail(local_rc, local_msg);       // (the peers have this rank's markers by now)
    if (fetch) return result();
    return MH_OK;
    });
}

int mh_get_score_stats(mh_engine* e, long long* pairs, long long* pairs_fp64, int reset)
{
    return guarded([&]() -> int {
    int rc = enter(e);
    if (rc) return rc;
    unsigned long long fb = 0;
    if (e->fb_pairs.p && e->score_pairs > 0) {
        HIPCHK(hipMemcpyAsync(&fb, e->fb_pairs.p, sizeof(fb), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (pairs) *pairs = e->score_pairs;
    if (pairs_fp64) *pairs_fp64 = (long long)fb;
    if (reset) e->score_pairs = 0;                 // (the device counter is cleared by the next scoring call)
    return MH_OK;
    });
}

int mh_get_copy_stats(mh_engine* e, long long* h2d, long long* d2h, int reset)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    if (h2d) *h2d = e->copies_h2d;
    if (d2h) *d2h = e->copies_d2h;
    if (reset) { e->copies_h2d = 0; e->copies_d2h = 0; }
    return MH_OK;
    });
}

int mh_inliers_of_model(mh_engine* e, int idx, double thr2, int label_value, int* labels)
{
    return guarded([&]() -> int {
    int rc = require_models(e);
    if (rc) return rc;
    if (idx < 0 || idx >= e->m || !labels) return fail(MH_ERR_INVALID, "bad model index or null labels");
    HIPCHK(e->labels_pts.reserve(e->n));
    HIPCHK(hipMemcpyAsync(e->labels_pts.p, labels, sizeof(int) * e->n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(launch_inliers_of_model(e->pts(), e->H.p, idx, thr2, label_value, e->labels_pts.p, e->stream));
    HIPCHK(hipMemcpyAsync(labels, e->labels_pts.p, sizeof(int) * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_inliers_of_homography(mh_engine* e, const double* H, double thr2, int label_value, int* labels)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (!H || !labels) return fail(MH_ERR_INVALID, "null homography or labels");
    HIPCHK(e->H_one.reserve(9));
    HIPCHK(e->labels_pts.reserve(e->n));
    HIPCHK(hipMemcpyAsync(e->H_one.p, H, sizeof(double) * 9, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->labels_pts.p, labels, sizeof(int) * e->n, hipMemcpyHostToDevice, e->stream));
    HIPCHK(launch_inliers_of_model(e->pts(), e->H_one.p, 0, thr2, label_value, e->labels_pts.p, e->stream));
    HIPCHK(hipMemcpyAsync(labels, e->labels_pts.p, sizeof(int) * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_compat_trial_stats(mh_engine* e, const double* pts_xyxy, const int* cluster_begin, int clusters, const int* tri,
                          const double* H, const unsigned char* ok, int trials, double* stats_out)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    if (clusters < 0 || trials < 0) return fail(MH_ERR_INVALID, "negative cluster or trial count");
    if (clusters == 0 || trials == 0) return MH_OK;
    if (!pts_xyxy || !cluster_begin || !tri || !H || !ok || !stats_out) return fail(MH_ERR_INVALID, "null argument");
    if (cluster_begin[0] != 0) return fail(MH_ERR_INVALID, "cluster_begin[0] must be 0");
    for (int c = 0; c < clusters; ++c)
        if (cluster_begin[c + 1] - cluster_begin[c] < 19)
            return fail(MH_ERR_INVALID, "a cluster of fewer than 19 points: the caller handles those itself (the three stale entries of the reference's buffer reach the median ranks)");
    const size_t total = (size_t)cluster_begin[clusters], ct = (size_t)clusters * (size_t)trials;
    if (ct > (size_t)0x7fffffff) return fail(MH_ERR_INVALID, "too many trials");
    for (size_t i = 0; i < ct; ++i) {
        const int nc = cluster_begin[i / trials + 1] - cluster_begin[i / trials];
        for (int j = 0; j < 3; ++j)
            if (tri[3 * i + j] < 0 || tri[3 * i + j] >= nc) return fail(MH_ERR_INVALID, "a trial draws a point outside its cluster");
    }
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(e->cp_pts.reserve(4 * total)); HIPCHK(e->cp_begin.reserve(clusters + 1)); HIPCHK(e->cp_tri.reserve(3 * ct));
    HIPCHK(e->cp_H.reserve(9 * ct)); HIPCHK(e->cp_ok.reserve(ct)); HIPCHK(e->cp_out.reserve(8 * ct));
    HIPCHK(hipMemcpyAsync(e->cp_pts.p, pts_xyxy, sizeof(double) * 4 * total, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->cp_begin.p, cluster_begin, sizeof(int) * (clusters + 1), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->cp_tri.p, tri, sizeof(int) * 3 * ct, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->cp_H.p, H, sizeof(double) * 9 * ct, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->cp_ok.p, ok, ct, hipMemcpyHostToDevice, e->stream));
    HIPCHK(launch_compat_select(e->cp_pts.p, e->cp_begin.p, clusters, e->cp_tri.p, e->cp_H.p, e->cp_ok.p, trials, e->cp_out.p, e->stream));
    HIPCHK(hipMemcpyAsync(stats_out, e->cp_out.p, sizeof(double) * 8 * ct, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_inlier_moments(mh_engine* e, double thr2, double* moments, double* min_eig)
{
    return guarded([&]() -> int {
    int rc = require_models(e);
    if (rc) return rc;
    HIPCHK(e->moments.reserve((size_t)e->m * 6));
    HIPCHK(e->min_eig.reserve(e->m));
    HIPCHK(launch_moments(e->pts(), e->H.p, e->m, thr2, e->moments.p, e->min_eig.p, e->stream));
    if (moments)
        HIPCHK(hipMemcpyAsync(moments, e->moments.p, sizeof(double) * 6 * e->m, hipMemcpyDeviceToHost, e->stream));
    if (min_eig)
        HIPCHK(hipMemcpyAsync(min_eig, e->min_eig.p, sizeof(double) * e->m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_data_cost(mh_engine* e, int* cost)
{
    return guarded([&]() -> int {
    int rc = require_models(e);
    if (rc) return rc;
    rc = do_data_cost(e);
    if (rc) return rc;
    if (cost) {
        HIPCHK(hipMemcpyAsync(cost, e->cost.p, sizeof(int) * (size_t)e->n * e->cost_L, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return MH_OK;
    });
}

int mh_expand(mh_engine* e, const int* init_labels, int* labels_out, int* energy, int* cycles)
{
    return guarded([&]() -> int {
    int rc = require_models(e);
    if (rc) return rc;
    const int* init_dev = nullptr;
    if (init_labels) {
        for (int i = 0; i < e->n; ++i)
            if (init_labels[i] < 0 || init_labels[i] > e->m)
                return fail(MH_ERR_INVALID, "initial label out of range 0..Nh");
        HIPCHK(e->labels_in.reserve(e->n));
        HIPCHK(hipMemcpyAsync(e->labels_in.p, init_labels, sizeof(int) * e->n, hipMemcpyHostToDevice, e->stream));
        init_dev = e->labels_in.p;
    }
    long long en = 0;
    rc = do_expand(e, init_dev, &en, cycles);
    if (rc) return rc;
    if (energy) *energy = (int)en;
    if (labels_out) {
        HIPCHK(hipMemcpyAsync(labels_out, e->ew_label.p, sizeof(int) * e->n, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return MH_OK;
    });
}

int mh_get_expand_stats(mh_engine* e, long long stats[24])
{
    return guarded([&]() -> int {
    if (!e || !stats) return fail(MH_ERR_INVALID, "null argument");
    const ExpandStats& x = e->last_expand;
    stats[0] = x.cycles;
    stats[1] = x.moves;
    stats[2] = x.accepted;
    stats[3] = x.push_phases;
    stats[4] = x.relax_intervals;
    stats[5] = x.host_syncs;
    stats[6] = x.reduce_launches;
    stats[7] = x.flow_moves;
    stats[8] = x.launches;
    stats[9] = x.moves_run;
    stats[10] = x.moves_solved;
    stats[11] = x.core_sites;
    stats[12] = x.core_max;
    stats[13] = x.barriers;
    stats[14] = x.outer_iterations;
    stats[15] = (long long)(x.solve_ms * 1000.0);      // microseconds inside the solver launches
    stats[16] = (long long)(x.barrier_ms * 1000.0);
    stats[17] = (long long)(x.relax_ms * 1000.0);
    stats[18] = (long long)(x.push_ms * 1000.0);
    stats[19] = (long long)(x.tail_ms * 1000.0);
    stats[20] = e->last_expand_retries;
    stats[21] = e->last_solve_grid;
    stats[22] = e->expand_retries_total;
    stats[23] = (long long)(e->last_expand.max_barrier_wait_ms * 1e3);
    return MH_OK;
    });
}

int mh_get_expand_trace(mh_engine* e, int* trace, int moves)
{
    return guarded([&]() -> int {
    int rc = enter(e);
    if (rc) return rc;
    if (!trace || moves <= 0) return fail(MH_ERR_INVALID, "null trace or moves <= 0");
    if (e->trace_moves <= 0 || !e->ew_trace.p) return fail(MH_ERR_NOT_SET, "tracing is off (mh_set_tuning key 8) or no expansion has run");
    // rows [0, trace_moves): the moves; rows behind them: the relabel log of the detail move (key 9), two relabels per row
    // (the buffer was sized by the trace_moves in force at the last expansion: never read past it)
    const int m = std::min(moves, (int)std::min<size_t>((size_t)e->trace_moves + 1024, e->ew_trace.cap / 8));
    HIPCHK(hipMemcpyAsync(trace, e->ew_trace.p, sizeof(int) * 8 * (size_t)m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_get_core_components(mh_engine* e, int* out, int moves)
{
    return guarded([&]() -> int {
    int rc = enter(e);
    if (rc) return rc;
    if (!out || moves <= 0) return fail(MH_ERR_INVALID, "null output or moves <= 0");
    if (e->comp_moves <= 0 || !e->ew_comp_out.p) return fail(MH_ERR_NOT_SET, "the component diagnostic is off (mh_set_tuning key 21) or no expansion has run");
    const int m = std::min(moves, (int)(e->ew_comp_out.cap / 16));
    HIPCHK(hipMemcpyAsync(out, e->ew_comp_out.p, sizeof(int) * 16 * (size_t)m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
    });
}

int mh_reestimate(mh_engine* e, const int* labels, double* H_out)
{
    return guarded([&]() -> int {
    int rc = require_models(e);
    if (rc) return rc;
    if (!labels) return fail(MH_ERR_INVALID, "labels is null");
    if (!e->have_aff) return fail(MH_ERR_NOT_SET, "affinities are not set");
    if (!e->have_epi) return fail(MH_ERR_NOT_SET, "fundamental matrix / epipole are not set");
    HIPCHK(e->labels_pts.reserve(e->n));
    HIPCHK(e->label_counts.reserve(e->m));
    HIPCHK(hipMemcpyAsync(e->labels_pts.p, labels, sizeof(int) * e->n, hipMemcpyHostToDevice, e->stream));
    Affines a{ e->a11.p, e->a12.p, e->a21.p, e->a22.p };
    {
        ScopedTimer t(e, MH_K_REESTIMATE);
        HIPCHK(launch_reestimate(e->pts(), a, e->labels_pts.p, e->m, e->epi, e->H.p, e->label_counts.p, e->stream));
    }
    e->cost_L = 0;
    if (H_out) {
        HIPCHK(hipMemcpyAsync(H_out, e->H.p, sizeof(double) * 9 * e->m, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return MH_OK;
    });
}

int mh_labeling_step(mh_engine* e, int warm, int* labeling, double* energy, int* cycles)
{
    return guarded([&]() -> int {
    int rc = require_models(e);
    if (rc) return rc;
    if (!labeling) return fail(MH_ERR_INVALID, "labeling is null");
    if (!e->have_aff || !e->have_epi) return fail(MH_ERR_NOT_SET, "affinities / epipolar geometry are not set");
    rc = do_data_cost(e);
    if (rc) return rc;
    const int* init_dev = nullptr;
    const dim3 grid((e->n + 255) / 256), blk(256);
    if (warm) {                                                   // M/MultiH.cpp:525-529
        for (int i = 0; i < e->n; ++i)
            if (labeling[i] < -1 || labeling[i] >= e->m)
                return fail(MH_ERR_INVALID, "warm-start label out of range -1..Nh-1");
        HIPCHK(e->labels_in.reserve(e->n));
        HIPCHK(hipMemcpyAsync(e->labels_in.p, labeling, sizeof(int) * e->n, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(k_shift_labels, grid, blk, 0, e->stream, e->n, e->labels_in.p, 1, e->labels_in.p);
        init_dev = e->labels_in.p;
    }
    long long en = 0;
    rc = do_expand(e, init_dev, &en, cycles);
    if (rc) return rc;
    HIPCHK(e->labels_pts.reserve(e->n));
    HIPCHK(e->label_counts.reserve(e->m));
    hipLaunchKernelGGL(k_shift_labels, grid, blk, 0, e->stream, e->n, e->ew_label.p, -1, e->labels_pts.p); // :547-568
    Affines a{ e->a11.p, e->a12.p, e->a21.p, e->a22.p };
    {
        ScopedTimer t(e, MH_K_REESTIMATE);
        HIPCHK(launch_reestimate(e->pts(), a, e->labels_pts.p, e->m, e->epi, e->H.p, e->label_counts.p, e->stream));
    }
    e->cost_L = 0;                                               // models changed
    HIPCHK(hipMemcpyAsync(labeling, e->labels_pts.p, sizeof(int) * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (energy) *energy = (double)en;
    return MH_OK;
    });
}

int mh_device_buffer(mh_engine* e, int which, void** ptr_dev, unsigned long long* bytes)
{
    return guarded([&]() -> int {
    if (!e || !ptr_dev || !bytes) return fail(MH_ERR_INVALID, "null argument");
    switch (which) {
    case MH_BUF_COUNTS: *ptr_dev = e->counts.p; *bytes = sizeof(int) * (size_t)e->m; break;
    case MH_BUF_MODELS: *ptr_dev = e->H.p; *bytes = sizeof(double) * 9 * (size_t)e->m; break;
    case MH_BUF_RESIDUALS: *ptr_dev = (e->ldr > 0 && e->m > 0) ? e->R.p : nullptr; *bytes = sizeof(double) * (size_t)e->m * (size_t)e->ldr; break;
    case MH_BUF_LABELS: *ptr_dev = e->ew_label.p; *bytes = sizeof(int) * (size_t)e->n; break;
    case MH_BUF_COST: *ptr_dev = e->cost.p; *bytes = sizeof(int) * (size_t)e->n * e->cost_L; break;
    case MH_BUF_GATHERED_SCORES: *ptr_dev = e->sel_gathered.p; *bytes = sizeof(int) * e->sel_gathered.cap; break;
    default: return fail(MH_ERR_INVALID, "unknown buffer id");
    }
    if (!*ptr_dev) return fail(MH_ERR_NOT_SET, "buffer has not been produced yet");
    return MH_OK;
    });
}

int mh_profile_enable(mh_engine* e, int on)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    e->profiling = on != 0;
    return MH_OK;
    });
}

int mh_profile_reset(mh_engine* e)
{
    return guarded([&]() -> int {
    int rc0 = enter(e);
    if (rc0) return rc0;
    HIPCHK(hipStreamSynchronize(e->stream));
    resolve_timers(e);
    for (int k = 0; k < MH_K_COUNT_; ++k) { e->timers[k].launches = 0; e->timers[k].total_ms = 0.0; }
    return MH_OK;
    });
}

int mh_profile_get(mh_engine* e, int kernel, int* launches, double* total_ms)
{
    return guarded([&]() -> int {
    if (!e || kernel < 0 || kernel >= MH_K_COUNT_) return fail(MH_ERR_INVALID, "bad kernel id");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    resolve_timers(e);
    if (launches) *launches = e->timers[kernel].launches;
    if (total_ms) *total_ms = e->timers[kernel].total_ms;
    return MH_OK;
    });
}

int mh_set_tuning(mh_engine* e, int key, int value)
{
    return guarded([&]() -> int {
    if (!e) return fail(MH_ERR_INVALID, "null engine");
#ifdef MH_TUNING
    if (key == 0) { e->tune_residual_variant = value; return MH_OK; }
    if (key == 1) { e->tune_score_variant = value; return MH_OK; }
    if (key == 13 && value >= 0) { e->tune_ld = value; return MH_OK; }
#else
    if (key == 0 || key == 1) {
        if (value == 0) return MH_OK;
        return fail(MH_ERR_INVALID, "residual / score kernel variants exist only in a library built with -DMH_TUNING");
    }
#endif
    if (key >= 2 && key <= 5 && value >= 1) { e->tune_expand[key - 2] = value; return MH_OK; }
    if (key == 6 && value >= 0) { e->tune_reduce = value; return MH_OK; }
    if (key == 7 && value >= 1 && value <= 64) { e->tune_ms_batch = value; return MH_OK; }
    if (key == 29 && value >= 0 && value <= 64) { e->tune_ms_persist = value; return MH_OK; }       // mean shift: persistent tail below this many climbs (0 = off)
    if (key == 8 && value >= 0 && value <= (1 << 20)) { e->trace_moves = value; return MH_OK; }
    if (key == 9 && value >= -1) { e->detail_move = value; return MH_OK; }
    if (key == 10 && value >= 1 && value <= 64) { e->tune_push_mult = value; return MH_OK; }
    if (key == 11 && (value == 0 || value == 1)) { e->tune_recycle = value; return MH_OK; }
    if (key == 12 && (value == 1 || value == 2)) { e->tune_reduce_launches = value; return MH_OK; }
    if (key == 14 && value >= 0 && value <= 8) { e->inject_barrier_timeouts = value; return MH_OK; }
    if (key == 15 && (value == 0 || value == 1)) { e->tune_score32 = value; return MH_OK; }
#ifdef MH_TUNING
    if (key == 16 && value >= 0 && value <= 16) { e->tune_score32_tiling = value; return MH_OK; }
#else
    if (key == 16 && value == 0) return MH_OK;
#endif
    if (key == 17 && value >= 0 && value <= 1000) { e->tune_cascade_iters = value; return MH_OK; }
    if (key == 18 && value >= 0 && value <= 1000) { e->inject_select_failure = value; return MH_OK; }
    if (key == 19 && value >= -1 && value <= 1024) { e->tune_sweep_headroom = value; return MH_OK; }
    if (key == 20 && (value == 0 || value == 1)) { e->tune_dlt_first = value; return MH_OK; }
    if (key == 21 && value >= 0 && value <= (1 << 16)) { e->comp_moves = value; return MH_OK; }
    if (key == 22 && value >= 0 && value <= 16 && !e->side_stream) { e->tune_stream_shift = value; return MH_OK; }
    if (key == 23 && value >= -1 && value <= 64) { e->tune_cost32_resident = value; return MH_OK; }
    if (key == 24 && value >= -1 && value <= 64) { e->tune_score32_resident = value; return MH_OK; }
    if (key == 25 && value >= 0 && value <= 2) { e->tune_dlt_variant = value; return MH_OK; }
#ifdef MH_TUNING
    // measured-and-rejected schedules: measurement libraries only
    if (key == 26 && value >= 0 && value <= 4096) { e->tune_sweep_slices = value; return MH_OK; }
    if (key == 27 && (value == 0 || value == 1)) { e->tune_cost32_slice_major = value; return MH_OK; }
    if (key == 28 && (value == 0 || value == 1)) { e->tune_cost32_batched = value; return MH_OK; }
#else
    if ((key == 26 || key == 27 || key == 28) && value == 0) return MH_OK;
    if (key == 16 || key == 26 || key == 27 || key == 28)
        return fail(MH_ERR_INVALID, "this schedule variant exists only in a library built with -DMH_TUNING");
#endif
    return fail(MH_ERR_INVALID, "unknown tuning key");
    });
}

} // extern "C"
