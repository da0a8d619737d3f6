// capi_engine.hpp — what the translation units of the C ABI share: the engine object, error plumbing and the helpers that
// several entry-point families use.  Internal: nothing here is part of include/multih_hip.h.
//   capi.hip          life cycle, parameters, correspondences, neighbourhood graph, buffers, profiling, tuning keys
//   capi_front.hip    epipolar front half, per-point homographies, mean shift          
//   capi_score.hip    propose, model sets, score / residual matrix / cost matrix, prefetch queue, inlier read-outs
//   capi_select.hip   transport, greedy selection, best model of a batch (the score exchange)
//   capi_label.hip    data cost, alpha-expansion, re-estimation, LabelingStep, post-filter statistics
#pragma once
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

// text of the last error on this thread (mh_last_error)
extern thread_local std::string g_err;

inline int fail(int code, const std::string& msg)
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
    DevBuf<int> sel_orig[2], sel_counts, sel_carried[2], sel_left, sel_rec, sel_scores, sel_gathered;     // (sel_carried / sel_left: r05, the decremental rounds of mh_select_greedy)
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
    int fund_metric = 0;                       // mh_set_fundamental_metric: MH_FUND_SAMPSON / MH_FUND_EPIPOLAR_MAX (changes results: not a tuning key)
    DevBuf<double> fund, fund_one;
    DevBuf<int> fund_samples, fund_counts, fund_inl;
    DevBuf<unsigned char> fund_mask, ref_keep, ref_in, ref_reason;
    int ref_reason_n = 0;                      // rows of the last mh_refine_correspondences (mh_get_refine_reasons)
    DevBuf<double> ref_out;

    // reference-style initialisation
    DevBuf<double> loc_H, loc_feat, ms_data, ms_mean;
    DevBuf<int> ms_votes, ms_out, ms_list, ms_pcnt, ms_heads, ms_tickets;
    DevBuf<double> ms_partial, ms_partial2;
    DevBuf<unsigned long long> ms_ticks;     // MULTIH_MS_STATS: phase ticks of the persistent kernel
    DevBuf<int> ms_ctl, ms_pcnt2;            // the persistent tail of a mean-shift batch (meanshift.hip, k_ms_persist)
    int ms_persist_per_cu = -1, ms_persist_per_cu6 = -1;   // workgroups of k_ms_persist<10> / <6> a compute unit holds (-1: not queried; a failed query is not kept)
    int tune_select_refine = 0;              // key 30: mh_select_greedy refits each round's winner to its inliers before the claim (0 = off)
    DevBuf<double> sel_refit;                // the refit (9 doubles) and its inlier count
    DevBuf<int> sel_refit_ctr;
    DevBuf<double> ms_rs;                    // the mean-shift index of one call (meanshift.hip, k_ms_indexed): rows in cell order,
    DevBuf<int> ms_order, ms_cells, ms_cursor, ms_cellcount;   // position -> row, cell starts, scatter cursors, cell counts
    int tune_ms_indexed = 1;                 // key 32: climbs through the index, a workgroup each, a launch per batch (0: the launched / persistent schedule)
    int tune_ms_dense = 8;                  // key 33: an indexed climb that meets more members than this in an iteration is handed to the per-group kernels
    long long ms_indexed_launches = 0;
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
    // r06: the contexts of the concurrent alpha-moves beyond context 0 (expand.hip): per context cap + sent (2 nnz ints), excess /
    // sink_cap / height / decided (4 n), core (8 n), control words and accumulators; took (n bytes); and the batch's own scratch
    DevBuf<int> ewx_arcs[EXPAND_MAX_CTX - 1], ewx_sites[EXPAND_MAX_CTX - 1], ewx_core[EXPAND_MAX_CTX - 1], ewx_flags[EXPAND_MAX_CTX - 1];
    DevBuf<long long> ewx_acc[EXPAND_MAX_CTX - 1];
    DevBuf<unsigned char> ewx_took[EXPAND_MAX_CTX - 1];
    DevBuf<int> ew_bctl, ew_took_list;
    int* h_batch = nullptr;                  // mapped pinned: k_commit's publication
    int* h_batch_dev = nullptr;
    int tune_expand_ctx = 16;                // key 37: alpha-moves solved together (1 = one after the other, the form until r05)
    int tune_batch_min_labels = 16;          // key 38: label sets of at least this many labels are batched from the first cycle on (smaller ones from the second)
    int tune_batch_spw = 0;                  // key 39: sites per wave in the setup and reduction launches of a batch of moves (16 / 32 / 64; 0 = by the size of the launch); schedule only
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
    int* h_ms_list = nullptr;                  // pinned: the (row, votes) lists of a batch's climbs, packed (grows with the need)
    size_t h_ms_list_pairs = 0;
    long long* h_acc = nullptr;
    int* h_flags_dev = nullptr;
    long long* h_acc_dev = nullptr;
    DevBuf<double> sel_pts[4];               // the active points of a greedy-selection round, packed (select.hip)
    DevBuf<int> sel_pack_count;
    DevBuf<double> sel_gone[4];              // the points the last claim of a greedy-selection round took out of the support set, packed
    int tune_select_decrement = 1;           // key 36: rounds count their candidates on the points that LEFT and subtract (1, default) or count again on what is left (0): same selection
    int tune_knn_grid = 1;                   // key 31: the k-NN table through the grid over the source image (0 = exhaustive pass)
    DevBuf<int> knn_cell, knn_count, knn_start, knn_orig;
    DevBuf<float> knn_P;
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

namespace mhe {

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

// ---- shared helpers (defined in capi.hip unless noted) ----
void resolve_timers(mh_engine* e);
int enter(mh_engine* e);
int require_points(mh_engine* e);
int require_models(mh_engine* e);
int require_models_or_empty_shard(mh_engine* e, bool* empty);
hipError_t reserve_counts(mh_engine* e, size_t n);
int quiesce(mh_engine* e);
int join_xchg(mh_engine* e);
// inlier counts of m models over the points p: FP32 pre-test where its preconditions hold, the FP64 sweep otherwise (capi_score.hip)
int score_models(mh_engine* e, const Points& p, const double* Hs, int m, double thr2, const unsigned char* dmask, int* counts_dev);
// the exchange's stream and events (capi_select.hip); the second stream of the prefetch queue (capi_score.hip)
int ensure_xchg_stream(mh_engine* e);
int ensure_side_stream(mh_engine* e);

} // namespace mhe

using namespace mhe;
