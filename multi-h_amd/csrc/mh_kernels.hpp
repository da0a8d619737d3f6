// mh_kernels.hpp — private launch interface between the C-ABI layer (capi*.hip)
// and the gfx950 kernels.  Nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mh {

// Correspondences resident in HBM as struct-of-arrays (coalesced 16-B loads).
struct Points {
    const double* x1;
    const double* y1;
    const double* x2;
    const double* y2;
    int n;
    // bounding box of the source points (x1, y1), taken when the correspondences were set; NaN when unknown.  The
    // residual sweep uses it to prove, per model, that the projective denominator stays away from zero.
    double xmin, xmax, ymin, ymax;
};

struct Affines {            // a11 a12 a21 a22, SoA
    const double* a11;
    const double* a12;
    const double* a21;
    const double* a22;
};

struct Epipolar {           // passed by value to kernels
    double F[9];
    double ex, ey;
};

// Leading dimension (in doubles) of the residual matrix rows: rows start 128-B aligned.
inline long long residual_ld(int n) { return ((long long)n + 15) & ~15ll; }

// --- residual.hip -----------------------------------------------------------
// variant: 0 default; tuning knob for bench sweeps (see residual.hip).
hipError_t launch_residual(const Points& p, const double* H, int M, double thr2, double* R,
                           long long ldr, int* counts, int variant, hipStream_t s, bool counts_zeroed = false,
                           int resident_grid = 0, int* resident_ctl = nullptr, int slices = 0 /* 0: the launcher's rule */,
                           int slice_major = 0 /* resident grid: consecutive items = the slices of one model block */);
// workgroups of the product's materialising sweep that one compute unit holds at a time (occupancy query)
int residual_workgroups_per_cu();
hipError_t launch_score(const Points& p, const double* H, int M, double thr2,
                        const unsigned char* mask, int* counts, int variant, hipStream_t s);
// --- score32.hip: the same counts through an FP32 pre-test with a rigorous error bound (FP64 only for the pairs it cannot decide)
hipError_t launch_model32(const double* H, int M, double X, double Y, double Cmax, float* H32 /* M x 16 */, hipStream_t s);
// Cmax: the bound on |x2|, |y2| the table was made with; thr2 in [2^-40, 2^40]
hipError_t launch_score32(const Points& p, const double* H, const float* H32, int M, double thr2, double Cmax, const unsigned char* mask,
                          int* counts, unsigned long long* fallback_pairs, int tiling, hipStream_t s, int* resident_ctl = nullptr,
                          int cu_count = 256, int resident_slices = 0, int* occ_cache = nullptr);
// the materialised int32 cost matrix (launch_cost_matrix, datacost.hip) through the same pre-test; H32 made with the same Cmax
hipError_t launch_cost32(const Points& p, const double* H, const float* H32, int M, double lambda, double thr2, double Cmax,
                         int* C, long long ldc, int* counts, hipStream_t s, int* resident_ctl = nullptr, int cu_count = 256,
                         int psplit_override = 0, int slice_major = 0, int batched = 0, int* occ_cache = nullptr);
// occ_cache (both launchers): the caller's per-engine cache of the resident kernel's workgroups per compute unit (-1 = not asked yet)
hipError_t launch_inliers_of_model(const Points& p, const double* H, int idx, double thr2,
                                   int label_value, int* labels, hipStream_t s);
hipError_t launch_moments(const Points& p, const double* H, int M, double thr2, double* moments,
                          double* min_eig, hipStream_t s);

// --- compat.hip: order statistics of the post-filter's trials (HomographyCompatibilityCheck, M/MultiH.cpp:128-196)
hipError_t launch_compat_select(const double* pts /* total x 4 */, const int* begin /* clusters + 1 */, int clusters,
                                const int* tri /* clusters x trials x 3 */, const double* H /* clusters x trials x 9 */,
                                const unsigned char* ok, int trials, double* out /* clusters x trials x 8 */, hipStream_t s);
// the trials' 3-point homographies (GetHomography3PT without refinement, M/MultiH.cpp:154) — H: 9 doubles per (cluster, trial), ok: 1 where finite
hipError_t launch_compat_fit(const double* pts, const int* begin, int clusters, const int* tri, const double F[9], int trials,
                             double* H, unsigned char* ok, hipStream_t s);

// --- dlt4.hip ---------------------------------------------------------------
hipError_t launch_dlt4(const Points& p, unsigned long long seed, long long first, int M,
                       int* idx_out, double* H_out, hipStream_t s, int variant = 0 /* 1: the LDS-staged form */);

hipError_t launch_fund8(const Points& p, unsigned long long seed, long long first, int M,
                        int* idx_out /* M x 8 */, double* F_out, hipStream_t s);

// --- fund.hip ---------------------------------------------------------------
// metric: MH_FUND_SAMPSON (0) or MH_FUND_EPIPOLAR_MAX (1), see fund.hip
hipError_t launch_sampson_score(const Points& p, const double* F, int M, double thr2, int* counts,
                                hipStream_t s, int metric);
// Least-squares 8-point refit of F on the inliers of F_in under `metric` (one workgroup).
hipError_t launch_fund_refit(const Points& p, const double* F_in, double thr2, double* F_out,
                             unsigned char* mask_out, int* count_out, hipStream_t s, int metric);

// --- datacost.hip -----------------------------------------------------------
// the data cost of every model against every point, int32, model-major with pitch ldc (datacost.hip)
inline long long cost_ld(int n) { return ((long long)n + 31) & ~31ll; }
hipError_t launch_cost_matrix(const Points& p, const double* H, int M, double lambda, double thr2, int* C, long long ldc,
                              int* counts, hipStream_t s);
hipError_t launch_data_cost(const Points& p, const double* H, int Nh, double lambda, double thr2,
                            int* cost, hipStream_t s);

// --- reestimate.hip ---------------------------------------------------------
hipError_t launch_reestimate(const Points& p, const Affines& a, const int* labels, int Nh,
                             const Epipolar& ep, double* H, int* counts, hipStream_t s);

hipError_t launch_haf_point(const Points& p, const Affines& a, const Epipolar& ep, double locality,
                            double* H_out /* n x 9, nullable */, double* feat_out /* n x 10, nullable */,
                            hipStream_t s);

// --- refine.hip -------------------------------------------------------------
hipError_t launch_refine_points(const Points& p, const Affines& a, const double F[9], const double e1[2],
                                const double e2[2], const unsigned char* in_mask, unsigned char* keep,
                                double* out /* n x 8 */, unsigned char* reason /* n: MH_REFINE_* */, hipStream_t s);

// --- meanshift.hip ----------------------------------------------------------
constexpr int MS_BATCH = 256;   // climbs per batch: part of the definition of the seed order (meanshift.hip; the oracle draws alike)
struct MeanShiftWork {      // every per-climb array holds MS_BATCH slices
    const double* data;      // n x d row-major
    int n, d;
    double* mean;            // [climb] 16      current mean (in/out)
    int* votes;              // [climb] n       votes of the running climb (members get +1 per iteration)
    int* out;                // [climb] 4       [0] iterations, [1] converged, [2] list length, [3] dead end (no member)
    int* list;               // [climb] 2*n     (index, votes) pairs compacted when the climb has ended
    double* partial;         // [climb] MS_GROUPS x 16 member sums of the running iteration
    int* partial_cnt;        // [climb] MS_GROUPS
};
struct MeanShiftResultBlock { int out[4]; double mean[16]; };
struct MeanShiftActive { unsigned char climb[MS_BATCH]; };       // the climbs a round still works on, passed by value
// The climbs active[0..n_active) side by side without host round trips in between: seed the means with the rows
// starts_dev[climb] (null: continue the running climbs), `iterations` climb iterations (no-ops for a climb that has
// ended), compact the votes of the climbs that have ended into their lists, then publish
// every active climb's control words and mean to result_dev[climb] (mapped pinned).  Workgroups beyond the rows
// (n < 256 * MS_GROUPS) are not launched: their partial sums are +0, which the running sum never notices.
hipError_t launch_ms_climb(const MeanShiftWork& w, const MeanShiftActive& active, int n_active, const int* starts_dev, double band_sq,
                           double stop_thresh, int iterations, MeanShiftResultBlock* result_dev,
                           int* tickets /* MS_BATCH ints, zero */, hipStream_t s);
// r05: the same for the FEW climbs that are still running after the first rounds — to their end (or max_iters) in one
// launch, one workgroup per group of the definition with the thread's rows in registers and a barrier of the climb's own
// per iteration (meanshift.hip, k_ms_persist); a climb whose workgroups were not all resident within 250 ms leaves
// untouched with fell_back (ctl[2 * MS_BATCH + climb]) set.  ctl: 3 x MS_BATCH ints; partial2: MS_BATCH x 2 x 64 x 16
// doubles; partial_cnt2: MS_BATCH x 2 x 64 ints.  Needs n_active x min(64, ceil(n / 256)) workgroups resident at once.
hipError_t launch_ms_persist(const MeanShiftWork& w, const MeanShiftActive& active, int n_active, double band_sq, double stop_thresh,
                             int max_iters, int* ctl, double* partial2, int* partial_cnt2, MeanShiftResultBlock* result_dev,
                             hipStream_t s, unsigned long long* ticks = nullptr);
bool ms_persist_supported(int n, int d);
int ms_persist_occupancy(int d);
// r05: a climb per workgroup, from seed to published result in one launch; the rows reached through a one-coordinate index
// (meanshift.hip, k_ms_indexed).  The index is built once per call; cell width w >= bandWidth^2 (1 + 2^-20) is the caller's
// duty (the three-cell window is exact only then).
struct MeanShiftIndex {
    const double* rs;        // [d][n] the rows in cell order, component-major
    const int* order;        // [n] position in cell order -> row
    const int* cell_start;   // [cells + 1]
    int cells, coord;
    double lo, inv_w;        // cell(x) = clamp(floor((x - lo) * inv_w), 0, cells - 1)
};
bool ms_indexed_supported(int n, int d);
int ms_index_max_cells();
hipError_t launch_ms_index_build(const double* data, int n, int d, const MeanShiftIndex& ix, int* count, int* cursor, int* cell_start,
                                 int* order, double* rs, hipStream_t s);
hipError_t launch_ms_indexed(const MeanShiftWork& w, const MeanShiftActive& active, int n_active, const int* starts_dev,
                             const MeanShiftIndex& ix, double band_sq, double stop_thresh, int max_iters, int dense_limit,
                             int keep, int* running, MeanShiftResultBlock* result_dev, hipStream_t s,
                             unsigned long long* ticks = nullptr);
// the lists of climbs 0 .. climbs-1 packed one behind the other (lengths / offsets: `climbs` ints each, device-visible)
hipError_t launch_ms_pack(const MeanShiftWork& w, int climbs, int longest, const int* offsets_dev, const int* lengths_dev, int* packed,
                          hipStream_t s);
// compacts and clears the votes of all `climbs` climbs, ended or not
hipError_t launch_ms_collect(const MeanShiftWork& w, int climbs, hipStream_t s);

// --- expand.hip -------------------------------------------------------------
struct Graph {              // symmetric weighted CSR in HBM
    const int* rowptr;      // n+1
    const int* col;         // nnz
    const int* w;           // nnz  multiplicity mult(i,j)
    const int* rev;         // nnz  index of the reverse arc
    int n, nnz;
    const int* order;       // n    a fixed pseudo-random permutation of the sites: the alpha-expansion's solver takes its
                            //      core sites in this order so that a workgroup never owns a spatial cluster (expand.hip)
    const int* wsum;        // n    sum of w over the site's row: bounds every n-link total of the site (expand.hip, k_move_setup)
};

struct ExpandWork {         // scratch owned by the engine
    int* label;             // n   current labeling (GCO numbering)
    int* cur_cost;          // n   cost[i][label[i]]
    int* cap;               // nnz capacities of the move's s-t graph
    int* sent;              // nnz cumulative flow per arc, written by the arc's tail only (expand.hip, k_solve)
    int* excess;            // n
    int* sink_cap;          // n
    int* height;            // n
    int* decided;           // n   0 undecided, 1 source side (takes alpha), 2 sink side (keeps its label), 3 not in the graph
    unsigned char* took;    // n   1 where the last solved move moves the site to alpha (applied lazily, see expand.hip)
    int* core;              // 8 x n  compacted list of the sites the dominance reduction left undecided, in 8 shards
    int* flags;             // device control words (EXPAND_FLAG_WORDS ints, see expand.hip)
    long long* acc;         // device 64-bit accumulators (EXPAND_ACC_WORDS)
    int* h_flags;           // pinned, device-mapped host mirror of flags (host address)
    long long* h_acc;       // pinned, device-mapped host mirror of acc (host address)
    int* h_flags_dev;       // device addresses of the two mirrors
    long long* h_acc_dev;
    // schedule knobs of the per-move solver (see expand.hip): relaxation rounds per barrier interval, push cycles per
    // push phase, push phases per global relabel, workgroups of the solver launch
    int relax_rounds, push_cycles, push_phases, solve_grid, push_mult;
    int reduce_rounds;      // dominance-reduction rounds per launch; 0 switches the reduction off (A/B)
    int reduce_launches;    // reduction launches per move in front of the solver: 2, or 1 (the compacting one alone)
    int cascade_iters;      // barrier-separated passes of the dominance cascade inside the solver launch (0 = to its fixed point)
    int* trace;             // optional: 8 ints per move {core sites, workgroups, relabels, relax intervals, push phases,
    int trace_moves;        //   barriers, ticks (100 MHz), ticks inside barriers}; moves beyond trace_moves are not traced
    int detail_move;        // move whose global relabels are logged behind the trace (4 ints each, at most 2048): {active sites,
                            //   largest finite height, relabel intervals so far, ticks so far}; -1 none
    // flow recycling (expand.hip, k_solve): per label the net arc flows (L x nnz) and sink flows (L x n) its last
    // expansion ended with; null = every move starts from the zero flow
    int* saved_flow;
    int* saved_sink;
    // 100 MHz ticks a poll at a grid barrier may last before the launch gives up (a workgroup of it is not resident: a GPU
    // shared with other persistent launches); 0 = 3 s
    long long barrier_timeout_ticks = 0;
    long long barrier_first_timeout_ticks = 0;     // the launch's first barrier (residency of all its workgroups)
    // diagnostic (expand.hip, k_core_components): 16 ints per move for the first comp_moves moves; scratch 2 n ints
    int* comp_out = nullptr;
    int* comp_scratch = nullptr;
    int comp_moves = 0;
    // r06: concurrent alpha-moves (expand.hip, "batches").  n_ctx >= 2: that many consecutive moves are solved together on
    // the SAME labeling, each in a context of its own (the fields above are context 0; ctx[k - 1] holds context k), and
    // committed in order by k_batch_commit, which keeps a move only if it passes a test against what its predecessors in the batch changed.
    int n_ctx = 1;
    struct Ctx { int* cap; int* sent; int* excess; int* sink_cap; int* height; int* decided; unsigned char* took; int* core; int* flags; long long* acc; int* took_list; };
    int* took_list0 = nullptr;    // n   context 0's list of the sites its move takes (the other contexts': Ctx::took_list)
    Ctx ctx[15] = {};     // EXPAND_MAX_CTX - 1
    int* bctl = nullptr;          // 8 + EXPAND_MAX_CTX   batch control words (device)
    int* h_batch = nullptr;       // 8   k_batch_commit's publication, pinned + device-mapped (host address) ...
    int* h_batch_dev = nullptr;   //     ... and its device address
    int batch_min_labels = 0;     // batches only when the label set has at least this many labels
    int batch_spw = 0;            // sites per wave in a batch's setup and reduction launches: 16 / 32 / 64, 0 = by the size of the launch (a move alone: 16)
};
constexpr int EXPAND_MAX_CTX = 16;
static_assert(sizeof(ExpandWork::ctx) / sizeof(ExpandWork::Ctx) == EXPAND_MAX_CTX - 1, "one context is the work area itself");
constexpr int EXPAND_FLAG_WORDS = 896;     // device control block (expand.hip)
constexpr int EXPAND_HOST_WORDS = 32;      // its head, mirrored to the host
constexpr int EXPAND_ACC_WORDS = 64 + 3 * 64 * 16;   // 16 scalars (mirrored to the host) + three striped sums
constexpr int EXPAND_CORE_SHARDS = 8;

struct ExpandStats {
    int cycles;
    long long energy;
    int moves, accepted;                      // moves enqueued; moves that lowered the energy
    long long push_phases, relax_intervals;   // summed over the moves that were solved
    long long host_syncs;
    long long reduce_launches, flow_moves;    // reduction launches; moves that still needed push-relabel
    long long launches;                       // kernel launches of the whole expansion
    long long moves_run;                      // moves not skipped as provably idempotent
    long long moves_solved;                   // of those, moves whose core was not empty (k_solve had work)
    long long core_sites, core_max;           // undecided sites handed to the solver: sum and maximum over moves
    long long barriers, outer_iterations;     // grid barriers / global relabels inside the solver launches
    double solve_ms;                          // time inside the solver launches (device clock of workgroup 0)
    double barrier_ms, relax_ms, push_ms;     // of which: inside grid barriers; global relabels; push phases (the last two include their barriers)
    double tail_ms;                           // of which: relabel/push rounds that began with fewer than 64 rows still holding excess
    long long tail_rounds;
    double max_barrier_wait_ms;               // longest single wait of the leader workgroup at a grid barrier
    // r06, concurrent moves: batches launched; moves committed out of a batch (solved beside others, results kept); moves whose
    // validation against their predecessors' changes failed (re-run alone); moves the HOST did not launch at all because they
    // were provably idempotent; moves run alone (no batch)
    long long batches, batch_committed, batch_invalid, host_skipped, solo_moves;
};

// resident workgroups of the solver launch per CU, as the occupancy query sees k_solve
hipError_t solver_blocks_per_cu(int* blocks);
hipError_t run_expansion(const Graph& g, const int* cost /* n x L */, int L, int potts,
                         ExpandWork& w, int max_cycles, ExpandStats* st, hipStream_t s);
hipError_t launch_init_labeling(const int* cost, int L, int n, const int* init_or_null_dev,
                                int* label, int* cur_cost, hipStream_t s);
hipError_t launch_argmin_labels(const int* cost, int L, int n, int* label, long long* acc,
                                hipStream_t s);

// --- select.hip ------------------------------------------------------------
// packs the points with mask != 0 into cx1.. (any order); *count = their number
hipError_t launch_sel_pack_points(const Points& p, const unsigned char* mask, double* cx1, double* cy1, double* cx2, double* cy2,
                                  int* count, hipStream_t s);
// one rank's offer in a round of the greedy selection: 88 bytes, the unit of the sharded exchange
struct SelRecord { unsigned long long key; double H[9]; int err; int mode; };      // mode: bit 0 the rank's residual mode, bit 1 refitted winners (key 30); the ranks' words must agree
static_assert(sizeof(SelRecord) == 88, "the exchanged record is 88 bytes");
hipError_t launch_sel_argmax(const int* counts, const int* orig, int Mc, unsigned int my_off, unsigned long long* key,
                             int* scores_full, hipStream_t s);
hipError_t launch_sel_argmax_gathered(const int* gathered, int world, int longest, int base, int rem, unsigned long long* key,
                                      hipStream_t s);
hipError_t launch_sel_record(const int* counts, const int* orig, const double* Hs, int Mc, unsigned int my_off,
                             const unsigned long long* key_local, int err, int mode, SelRecord* record, hipStream_t s);
hipError_t launch_sel_compact(const int* counts, const int* orig, const double* Hs, int Mc, int need, const SelRecord* records,
                              int world, unsigned int my_off, int* next_orig, double* next_H, int* rec, int* next_counts, hipStream_t s);
// counts[c] = carried[c] - left[c] (r05: a round of the greedy selection counts its candidates on the points the last claim
// took away and subtracts, instead of counting them again on everything that is left)
hipError_t launch_sel_subtract(const int* carried, const int* left, int Mc, int* counts, hipStream_t s);
hipError_t launch_sel_claim(const Points& p, const SelRecord* records, int world, const unsigned long long* key_check, double thr2,
                            int need, unsigned char* mask, int* rec, double* sel_H, long long* sel_counter, int max_models,
                            hipStream_t s, int symmetric = 0, const double* refit = nullptr,
                            double* cx1 = nullptr, double* cy1 = nullptr, double* cx2 = nullptr, double* cy2 = nullptr /* the points that leave, packed */);
// r05 (mh_set_tuning key 30): the round's winner refitted to its inliers in the support set by the per-label HAF least squares
// (one label); refit = 9 doubles + the refit's inlier count; launch_sel_claim takes it in the winner's place when it is finite and
// explains at least as many points.  labels: n ints, counter / label_count: one int each (scratch).
hipError_t launch_sel_refit(const Points& p, const Affines& a, const Epipolar& ep, const SelRecord* records, int world, double thr2,
                            int need, const unsigned char* mask, int* labels, double* refit, int* counter, int* label_count,
                            hipStream_t s, int symmetric);
hipError_t launch_sel_publish(int* rec, unsigned long long* keys, SelRecord* my_record, int need, int* h_rec_dev, hipStream_t s);
hipError_t launch_best_publish(unsigned long long* key, int* h_best_dev, hipStream_t s);
hipError_t launch_best_fused(int* scores, int world, int longest, int base, int rem, int* h_best_dev, int* clear,
                             int clear_count, hipStream_t s);
hipError_t launch_pad_scores(const int* counts, int m, int longest, int* scores, hipStream_t s);

// --- knn.hip ----------------------------------------------------------------
constexpr int KNN_MAX_SPLITS = 16;      // slices of the candidate range (knn.hip)
hipError_t launch_knn(const Points& p, int k, int* nbr_out /* n x k */, int splits, float* part_d, int* part_i, hipStream_t s);
// r05: the same table through a grid over the source image (cells walked ring by ring until no unexamined point can enter the
// list).  scratch: cell_of n ints, count G*G ints, start G*G + 1 ints (G = knn_grid_cells(n) <= 256), P4 4n floats, orig n ints;
// hipErrorInvalidValue when the points' bounding box is not finite (the caller then takes the exhaustive pass).
int knn_grid_cells(int n);
hipError_t launch_knn_grid(const Points& p, int k, int* nbr_out, int* cell_of, int* count, int* start, float* P4, int* orig, hipStream_t s);
hipError_t launch_radius_count(const Points& p, float r2, int* counts /* n */, hipStream_t s);
hipError_t launch_radius_fill(const Points& p, float r2, const int* rowptr /* n+1 */, int* col /* nnz */, hipStream_t s);

// --- graph.hip ------------------------------------------------------------------
// Symmetric weighted CSR with reverse-arc index from directed hits on the device (a CSR, or rowptr == null: a dense
// n x stride table; column -1 = no hit).  info: 8 device ints the caller cleared —
//   [0] raw total exceeds int32, [1] longest raw row, [2] index error (1 out of range, 2 rowptr decreasing), [4..5] the same two for the folded rows.
constexpr int SYM_MAX_ROW = 1024;         // longest raw row (hits made + received) k_sym_fold sorts in LDS
hipError_t launch_hits_filter(const Points& p, int stride, float r2, int* col, int* err, hipStream_t s);
hipError_t launch_sym_count(int n, const int* rowptr, int stride, const int* col, int* deg, int* start /* n+1 */, int* info, hipStream_t s);
hipError_t launch_sym_build(int n, const int* rowptr, int stride, const int* col, const int* start, int* cursor, int* raw,
                            int* mult, int* uniq, int* out_rowptr /* n+1 */, int* info, hipStream_t s);
// wsum[i] = sum of w over row i
hipError_t launch_row_weight_sums(int n, const int* rowptr, const int* w, int* wsum, hipStream_t s);
hipError_t launch_sym_finish(int n, const int* start, const int* raw, const int* mult, const int* rowptr, int* col, int* w,
                             int* rev, hipStream_t s);

} // namespace mh
