// capi.hip — implementation of the C ABI declared in include/multih_hip.h.
// Owns device buffers, the HIP stream and the per-kernel event timers; every
// computation is a launch of a gfx950 kernel from this directory.  There is no
// CPU fallback: without a usable HIP device mh_create fails (MH_ERR_NO_DEVICE).
// r05: one translation unit per entry-point family; this one holds the life cycle, the inputs, the neighbourhood graph,
// buffers, profiling and the tuning keys, and defines the helpers the families share (capi_engine.hpp).

#include "capi_engine.hpp"

thread_local std::string g_err;

namespace mhe {

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

} // namespace mhe

namespace {

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
    for (int k = 0; k < EXPAND_MAX_CTX - 1; ++k) { e->ewx_arcs[k].release(); e->ewx_sites[k].release(); e->ewx_core[k].release(); e->ewx_flags[k].release(); e->ewx_acc[k].release(); e->ewx_took[k].release(); }
    e->ew_bctl.release(); e->ew_took_list.release();
    if (e->h_batch) (void)hipHostFree(e->h_batch);
    e->knn_tmp.release(); e->knn_part_i.release(); e->knn_part_d.release();
    e->knn_cell.release(); e->knn_count.release(); e->knn_start.release(); e->knn_P.release(); e->knn_orig.release();
    for (int c = 0; c < 4; ++c) e->sel_pts[c].release();
    e->sel_pack_count.release();
    for (int c = 0; c < 4; ++c) e->sel_gone[c].release();
    e->sel_carried[0].release(); e->sel_carried[1].release(); e->sel_left.release();
    e->gb_deg.release(); e->gb_start.release(); e->gb_cursor.release(); e->gb_raw.release(); e->gb_mult.release();
    e->gb_uniq.release(); e->gb_info.release(); e->gb_hits_rp.release(); e->gb_hits_col.release();
    if (e->h_flags) (void)hipHostFree(e->h_flags);
    if (e->h_ms) (void)hipHostFree(e->h_ms);
    if (e->h_ms_list) (void)hipHostFree(e->h_ms_list);
    if (e->h_acc) (void)hipHostFree(e->h_acc);
    if (e->h_sel) (void)hipHostFree(e->h_sel);
    for (int b = 0; b < 2; ++b) { e->sel_orig[b].release(); e->sel_cand_H[b].release(); }
    e->sel_counts.release(); e->sel_rec.release(); e->sel_scores.release(); e->sel_gathered.release(); e->sel_out_H.release();
    e->sel_records.release(); e->sel_counter.release(); e->sel_keys.release(); e->sel_refit.release(); e->sel_refit_ctr.release();
    e->ms_partial2.release(); e->ms_ctl.release(); e->ms_pcnt2.release(); e->ms_ticks.release();
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
    // r05: through a grid over the source image when the points' bounding box is finite (knn.hip, k_knn_grid: the same table,
    // 4.5 -> 0.5 ms at 50 000 points and linear in n); key 31 = 0 forces the exhaustive pass (A/B, tests)
    const Points pp = e->pts();
    if (e->tune_knn_grid && std::isfinite(pp.xmin) && std::isfinite(pp.xmax) && std::isfinite(pp.ymin) && std::isfinite(pp.ymax)) {
        const int G = knn_grid_cells(n);
        HIPCHK(e->knn_cell.reserve((size_t)n));
        HIPCHK(e->knn_count.reserve((size_t)G * G));
        HIPCHK(e->knn_start.reserve((size_t)G * G + 1));
        HIPCHK(e->knn_P.reserve((size_t)4 * n));
        HIPCHK(e->knn_orig.reserve((size_t)n));
        const hipError_t hg = launch_knn_grid(pp, k, e->knn_tmp.p, e->knn_cell.p, e->knn_count.p, e->knn_start.p, e->knn_P.p, e->knn_orig.p, e->stream);
        if (hg == hipSuccess) {
            const float r2g = radius > 0.0 ? (float)radius * (float)radius : INFINITY;
            HIPCHK(launch_hits_filter(pp, k, r2g, e->knn_tmp.p, e->gb_info.p + 2, e->stream));
            return device_sym_graph(e, nullptr, k, e->knn_tmp.p);
        }
        if (hg != hipErrorInvalidValue) HIPCHK(hg);      // InvalidValue: no usable grid (all source points coincide, or a cell size beyond float32) — the exhaustive pass below
    }
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
    if (e->xchg_stream) HIPCHK(hipStreamSynchronize(e->xchg_stream));      // MH_K_EXCHANGE's events live there
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
    if (key == 32 && (value == 0 || value == 1)) { e->tune_ms_indexed = value; return MH_OK; }     // mean shift: indexed climbs (1, default) or the launched / persistent schedule (0): same modes
    if (key == 37 && value >= 1 && value <= EXPAND_MAX_CTX) { e->tune_expand_ctx = value; return MH_OK; }     // alpha-moves solved together (1: one after the other) — schedule only
    if (key == 38 && value >= 0 && value <= (1 << 20)) { e->tune_batch_min_labels = value; return MH_OK; }    // ... from the first cycle on for label sets of at least this many labels
    if (key == 39 && (value == 0 || value == 16 || value == 32 || value == 64)) { e->tune_batch_spw = value; return MH_OK; }    // sites per wave in a batch's setup and reduction launches, 0 = by size (schedule only)
    if (key == 36 && (value == 0 || value == 1)) { e->tune_select_decrement = value; return MH_OK; }     // greedy selection: decremental rounds (1, default) — schedule only
    if (key == 33 && value >= 0 && value <= (1 << 20)) { e->tune_ms_dense = value; return MH_OK; }     // ... and the member count beyond which an indexed climb is handed on
    // key 30 CHANGES RESULTS (the one such key the product library accepts): every winner of mh_select_greedy is refitted to its
    // inliers before it claims them.  Sticky per engine; class MultiH sets it on every ProposeModels call (SetProposalRefit).
    if (key == 30 && (value == 0 || value == 1)) { e->tune_select_refine = value; return MH_OK; }
    if (key == 31 && (value == 0 || value == 1)) { e->tune_knn_grid = value; return MH_OK; }        // k-NN through the grid (1, default) or exhaustively (0): same table
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
