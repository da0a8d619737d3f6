// capi_select.hip: transport, greedy selection, the best model of a batch (the score exchange) — part of the C ABI of include/multih_hip.h (see capi_engine.hpp for the split).
#include "capi_engine.hpp"

namespace mhe {

// ---- best model of the scored batch --------------------------------------------------------------
int ensure_xchg_stream(mh_engine* e)
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

} // namespace mhe

namespace {

// all-gather on the engine's stream through whichever transport is set; the host-synchronised hook sees an idle stream
int exchange(mh_engine* e, const void* send_dev, void* recv_dev, size_t bytes_per_rank, hipStream_t on)
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

} // namespace

extern "C" {

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
    // r05, key 30: each round's winner is refitted to its inliers before it claims them (select.hip, launch_sel_refit).  Needs the
    // affinities and the epipolar geometry (the per-label HAF least squares of the loop); a setting, like thr2, that the ranks share.
    // r06 (advisor): affinities / epipolar geometry missing is the state of THIS rank's engine — it goes through local_failure like
    // every other rank-local error (returning here would leave the peers waiting in the all-gather), and the setting itself
    // travels in the records' mode word (bit 1) beside the residual mode (bit 0): ranks that disagree about it would claim with
    // different models, so k_sel_claim raises error 3 on any mismatch of the word.
    const bool refine = e->tune_select_refine != 0;
    const bool refine_usable = refine && e->have_aff && e->have_epi;
    if (refine && !refine_usable)
        local_failure(MH_ERR_NOT_SET, "mh_select_greedy with refitted winners (mh_set_tuning key 30) needs affinities and the epipolar geometry");
    if (refine) {
        HIPCHK(e->labels_pts.reserve((size_t)n));
        HIPCHK(e->sel_refit.reserve(10));
        HIPCHK(e->sel_refit_ctr.reserve(2));
    }
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
    // r05: from the second round on a candidate's count is what it was minus what it counted on the points the last claim
    // took out of the support set — the same integer as counting again on what is left, for a sweep over the few thousand
    // points that left instead of the tens of thousands that stay (key 36; whichever set is smaller is swept).
    const bool decrement = e->tune_select_decrement != 0;
    if (decrement) {
        for (int c = 0; c < 4; ++c) HIPCHK(e->sel_gone[c].reserve((size_t)n + 2));
        for (int b = 0; b < 2; ++b) HIPCHK(e->sel_carried[b].reserve(cap));
        HIPCHK(e->sel_left.reserve(cap));
    }
    int gone = 0;                                         // points the last claim took out

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
            if (decrement && !first && gone <= active) {
                // the candidates' counts of the last round came along with them (k_sel_compact); subtract what left
                if (gone == 0) {
                    HIPCHK(hipMemcpyAsync(e->sel_counts.p, e->sel_carried[cur].p, sizeof(int) * (size_t)Mc, hipMemcpyDeviceToDevice, s));
                    return MH_OK;
                }
                Points left = e->pts();                       // (same bounding box: a superset's is valid)
                left.x1 = e->sel_gone[0].p; left.y1 = e->sel_gone[1].p; left.x2 = e->sel_gone[2].p; left.y2 = e->sel_gone[3].p;
                left.n = gone;
                const int rcs = score_models(e, left, Hs, Mc, thr2, nullptr, e->sel_left.p);
                if (rcs) return rcs;
                HIPCHK(launch_sel_subtract(e->sel_carried[cur].p, e->sel_left.p, Mc, e->sel_counts.p, s));
                return MH_OK;
            }
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
        HIPCHK(launch_sel_record(e->sel_counts.p, orig, Hs, Mc, my_off, key_local, local_err, symmetric | (refine ? 2 : 0), my_record, s));
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
                                  e->sel_cand_H[cur ^ 1].p, e->sel_rec.p, decrement ? e->sel_carried[cur ^ 1].p : nullptr, s));
        const double* refit = nullptr;
        if (refine_usable) {
            // every rank holds all the points and the same records: the refit is computed redundantly, identically
            Affines aff{ e->a11.p, e->a12.p, e->a21.p, e->a22.p };
            HIPCHK(launch_sel_refit(e->pts(), aff, e->epi, records, world, thr2, need, e->mask.p, e->labels_pts.p, e->sel_refit.p,
                                    e->sel_refit_ctr.p, e->sel_refit_ctr.p + 1, s, symmetric));
            refit = e->sel_refit.p;
        }
        HIPCHK(launch_sel_claim(e->pts(), records, world, gather_scores && !local_err ? key_check : nullptr, thr2, need, e->mask.p, e->sel_rec.p,
                                e->sel_out_H.p, e->sel_counter.p, max_models, s, symmetric, refit,
                                decrement ? e->sel_gone[0].p : nullptr, decrement ? e->sel_gone[1].p : nullptr,
                                decrement ? e->sel_gone[2].p : nullptr, decrement ? e->sel_gone[3].p : nullptr));
        HIPCHK(launch_sel_publish(e->sel_rec.p, e->sel_keys.p, my_record, need, e->h_sel_dev, s));
        HIPCHK(hipStreamSynchronize(s));                 // five control words through mapped memory: no copy
        if (local_rc != MH_OK) return fail(local_rc, local_msg);     // (the others have read this rank's error word by now)
        if (e->h_sel[4] != 0)                            // every rank sees the same word, so every rank leaves here
            return fail(e->h_sel[4] == 3 ? MH_ERR_INVALID : MH_ERR_HIP,
                        e->h_sel[4] == 2 ? "greedy selection: the gathered score vector and the ranks' records disagree about the winner"
                        : e->h_sel[4] == 3 ? "greedy selection: the ranks are not in the same residual mode (mh_set_residual_mode) or do not agree on refitted winners (mh_set_tuning key 30)"
                                           : "greedy selection: a rank reported an error");
        const int best = e->h_sel[0];
        if (best < need) break;
        if (counts_out) counts_out[selected] = best;
        ++selected;
        gone = e->h_sel[5];
        active -= gone;                                   // what the claim took out of the support set (the winner's inliers, or its refit's)
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
        {
            ScopedTimer t(e, MH_K_EXCHANGE, x);
            if (sharded) {
                rc = exchange(e, e->counts.p, e->sel_gathered.p, sizeof(int) * (size_t)longest, x);      // north_star's all-gather
                if (rc) return rc;
                HIPCHK(launch_best_fused(e->sel_gathered.p, world, longest, base, rem, e->h_best_dev, clear, clear_count, x));
            } else {
                HIPCHK(launch_best_fused(e->counts.p, 1, e->m, 0, 0, e->h_best_dev, clear, clear_count, x));
            }
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

} // extern "C"
