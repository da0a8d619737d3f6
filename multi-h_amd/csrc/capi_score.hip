// capi_score.hip: propose, model sets, score / residual matrix / cost matrix, the prefetch queue, inlier read-outs — part of the C ABI of include/multih_hip.h (see capi_engine.hpp for the split).
#include "capi_engine.hpp"

namespace mhe {

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

// ---- pipelined propose -------------------------------------------------------------------------
int ensure_side_stream(mh_engine* e)
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

} // namespace mhe

extern "C" {

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
    if (e->tune_sweep_headroom >= 0 && e->residual_mode != MH_RESIDUAL_SYMMETRIC && (e->tune_residual_variant == 0 || (e->tune_residual_variant >= 50 && e->tune_residual_variant <= 52))) {     // (50-52: the product sweep at other work-item sizes, tuning builds)
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

} // extern "C"
