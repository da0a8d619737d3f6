// capi_label.hip: data cost, alpha-expansion, re-estimation, LabelingStep, post-filter statistics — part of the C ABI of include/multih_hip.h (see capi_engine.hpp for the split).
#include "capi_engine.hpp"

namespace {

// Contexts of the concurrent alpha-moves (key 37), bounded by what they cost: a context is 2 nnz + 13 n ints, and the complete
// radius neighbourhood of the reference's rule can hold hundreds of millions of arcs — beyond 8 GiB of contexts fewer moves run
// together (results never depend on the number).
int expand_contexts(const mh_engine* e)
{
    const int want = std::max(1, std::min(e->tune_expand_ctx, EXPAND_MAX_CTX));
    const double per_ctx = 4.0 * (2.0 * (double)e->g_nnz + 13.0 * (double)e->n + 2048.0);
    const int fit = 1 + (int)std::min(64.0, 8.0 * 1024.0 * 1024.0 * 1024.0 / std::max(per_ctx, 1.0));
    return std::max(1, std::min(want, fit));
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
    // r06: the contexts of the concurrent moves (key 37).  Fresh memory is cleared once: a context's capacities and flow counters
    // are read for arcs the current move has not written only together with verdicts that make them irrelevant (expand.hip), but
    // the control words' rings and tickets must start at zero (k_ctl_init) and nothing is gained by leaving the rest to chance.
    const int extra = expand_contexts(e) - 1;
    for (int k = 0; k < extra; ++k) {
        const size_t cap_a = e->ewx_arcs[k].cap, cap_s = e->ewx_sites[k].cap, cap_c = e->ewx_core[k].cap, cap_t = e->ewx_took[k].cap;
        HIPCHK(e->ewx_arcs[k].reserve(2 * (size_t)nnz + 2));
        HIPCHK(e->ewx_sites[k].reserve(5 * (size_t)n + 5));
        HIPCHK(e->ewx_core[k].reserve((size_t)EXPAND_CORE_SHARDS * n));
        HIPCHK(e->ewx_flags[k].reserve(EXPAND_FLAG_WORDS));
        HIPCHK(e->ewx_acc[k].reserve(EXPAND_ACC_WORDS));
        HIPCHK(e->ewx_took[k].reserve((size_t)n + 2));
        if (e->ewx_arcs[k].cap != cap_a) HIPCHK(hipMemsetAsync(e->ewx_arcs[k].p, 0, sizeof(int) * e->ewx_arcs[k].cap, e->stream));
        if (e->ewx_sites[k].cap != cap_s) HIPCHK(hipMemsetAsync(e->ewx_sites[k].p, 0, sizeof(int) * e->ewx_sites[k].cap, e->stream));
        if (e->ewx_core[k].cap != cap_c) HIPCHK(hipMemsetAsync(e->ewx_core[k].p, 0, sizeof(int) * e->ewx_core[k].cap, e->stream));
        if (e->ewx_took[k].cap != cap_t) HIPCHK(hipMemsetAsync(e->ewx_took[k].p, 0, e->ewx_took[k].cap, e->stream));
    }
    if (extra > 0) {
        HIPCHK(e->ew_bctl.reserve(8 + EXPAND_MAX_CTX));
        HIPCHK(e->ew_took_list.reserve((size_t)n + 2));
        if (!e->h_batch) {
            HIPCHK(hipHostMalloc((void**)&e->h_batch, sizeof(int) * 8, hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer((void**)&e->h_batch_dev, e->h_batch, 0));
            for (int i = 0; i < 8; ++i) e->h_batch[i] = 0;
        }
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
    // r06: contexts for concurrent moves (expand.hip, k_commit)
    w.n_ctx = expand_contexts(e);
    for (int k = 0; k + 1 < w.n_ctx; ++k) {
        ExpandWork::Ctx& x = w.ctx[k];
        x.cap = e->ewx_arcs[k].p; x.sent = e->ewx_arcs[k].p + ((size_t)g.nnz + 1);
        x.excess = e->ewx_sites[k].p; x.sink_cap = x.excess + ((size_t)g.n + 1); x.height = x.sink_cap + ((size_t)g.n + 1); x.decided = x.height + ((size_t)g.n + 1);
        x.took = e->ewx_took[k].p; x.core = e->ewx_core[k].p; x.flags = e->ewx_flags[k].p; x.acc = e->ewx_acc[k].p;
        x.took_list = x.decided + ((size_t)g.n + 1);
    }
    if (w.n_ctx > 1) {
        w.bctl = e->ew_bctl.p;
        w.h_batch = e->h_batch; w.h_batch_dev = e->h_batch_dev;
        w.took_list0 = e->ew_took_list.p;
        w.batch_min_labels = e->tune_batch_min_labels;
        w.batch_spw = e->tune_batch_spw;
    }
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

__global__ void k_shift_labels(int n, const int* in, int delta, int* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + delta;
}

} // namespace

extern "C" {

// H / ok given: the caller fitted the trials' homographies (the form until r05); F given instead: the engine fits them itself
static int compat_trial_stats(mh_engine* e, const double* pts_xyxy, const int* cluster_begin, int clusters, const int* tri,
                              const double* H, const unsigned char* ok, const double* F, int trials, double* stats_out, double* H_out,
                              unsigned char* ok_out)
{
    if (!e) return fail(MH_ERR_INVALID, "null engine");
    if (clusters < 0 || trials < 0) return fail(MH_ERR_INVALID, "negative cluster or trial count");
    if (clusters == 0 || trials == 0) return MH_OK;
    if (!pts_xyxy || !cluster_begin || !tri || !stats_out || (!F && (!H || !ok))) return fail(MH_ERR_INVALID, "null argument");
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
    if (F) {
        HIPCHK(launch_compat_fit(e->cp_pts.p, e->cp_begin.p, clusters, e->cp_tri.p, F, trials, e->cp_H.p, e->cp_ok.p, e->stream));
    } else {
        HIPCHK(hipMemcpyAsync(e->cp_H.p, H, sizeof(double) * 9 * ct, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->cp_ok.p, ok, ct, hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(launch_compat_select(e->cp_pts.p, e->cp_begin.p, clusters, e->cp_tri.p, e->cp_H.p, e->cp_ok.p, trials, e->cp_out.p, e->stream));
    HIPCHK(hipMemcpyAsync(stats_out, e->cp_out.p, sizeof(double) * 8 * ct, hipMemcpyDeviceToHost, e->stream));
    if (H_out) HIPCHK(hipMemcpyAsync(H_out, e->cp_H.p, sizeof(double) * 9 * ct, hipMemcpyDeviceToHost, e->stream));
    if (ok_out) HIPCHK(hipMemcpyAsync(ok_out, e->cp_ok.p, ct, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MH_OK;
}

int mh_compat_trial_stats(mh_engine* e, const double* pts_xyxy, const int* cluster_begin, int clusters, const int* tri,
                          const double* H, const unsigned char* ok, int trials, double* stats_out)
{
    return guarded([&]() -> int {
    if (!H || !ok) return fail(MH_ERR_INVALID, "null argument");
    return compat_trial_stats(e, pts_xyxy, cluster_begin, clusters, tri, H, ok, nullptr, trials, stats_out, nullptr, nullptr);
    });
}

int mh_compat_trial_stats_fit(mh_engine* e, const double* pts_xyxy, const int* cluster_begin, int clusters, const int* tri,
                              const double F[9], int trials, double* stats_out, double* H_out, unsigned char* ok_out)
{
    return guarded([&]() -> int {
    if (!F) return fail(MH_ERR_INVALID, "null argument");
    return compat_trial_stats(e, pts_xyxy, cluster_begin, clusters, tri, nullptr, nullptr, F, trials, stats_out, H_out, ok_out);
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

int mh_get_expand_batch_stats(mh_engine* e, long long stats[8])
{
    return guarded([&]() -> int {
    if (!e || !stats) return fail(MH_ERR_INVALID, "null argument");
    const ExpandStats& x = e->last_expand;
    stats[0] = x.batches;
    stats[1] = x.batch_committed;
    stats[2] = x.batch_invalid;
    stats[3] = x.host_skipped;
    stats[4] = x.solo_moves;
    stats[5] = 0;
    stats[6] = expand_contexts(e);
    stats[7] = e->tune_batch_min_labels;
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

} // extern "C"
