// capi_front.hip: epipolar front half, per-point homographies, mean shift — part of the C ABI of include/multih_hip.h (see capi_engine.hpp for the split).
#include "capi_engine.hpp"

#include <unordered_map>

namespace {

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

} // namespace

extern "C" {

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

int mh_set_fundamental_metric(mh_engine* e, int metric)
{
    return guarded([&]() -> int {
    int rc = enter(e);
    if (rc) return rc;
    if (metric != MH_FUND_SAMPSON && metric != MH_FUND_EPIPOLAR_MAX) return fail(MH_ERR_INVALID, "unknown epipolar error definition");
    e->fund_metric = metric;
    return MH_OK;
    });
}

int mh_score_sampson(mh_engine* e, double thr2, int* counts)
{
    return guarded([&]() -> int {
    int rc = require_points(e);
    if (rc) return rc;
    if (e->fm <= 0) return fail(MH_ERR_NOT_SET, "no fundamental-matrix hypotheses; call mh_propose_fund8");
    HIPCHK(launch_sampson_score(e->pts(), e->fund.p, e->fm, thr2, e->fund_counts.p, e->stream, e->fund_metric));
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
        HIPCHK(launch_fund_refit(e->pts(), in, thr2, out, e->fund_mask.p, e->fund_inl.p, e->stream, e->fund_metric));
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
    HIPCHK(e->ref_reason.reserve((size_t)e->n + 2));
    e->ref_reason_n = 0;
    const unsigned char* dmask = nullptr;
    if (in_mask) {
        HIPCHK(e->ref_in.reserve((size_t)e->n + 2));
        HIPCHK(hipMemcpyAsync(e->ref_in.p, in_mask, e->n, hipMemcpyHostToDevice, e->stream));
        dmask = e->ref_in.p;
    }
    HIPCHK(hipMemsetAsync(e->ref_out.p, 0, sizeof(double) * 8 * (size_t)e->n, e->stream));
    Affines a{ e->a11.p, e->a12.p, e->a21.p, e->a22.p };
    HIPCHK(launch_refine_points(e->pts(), a, F, e1, e2, dmask, e->ref_keep.p, e->ref_out.p, e->ref_reason.p, e->stream));
    HIPCHK(hipMemcpyAsync(keep, e->ref_keep.p, e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(refined, e->ref_out.p, sizeof(double) * 8 * (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->ref_reason_n = e->n;
    return MH_OK;
    });
}

int mh_get_refine_reasons(mh_engine* e, unsigned char* reason, int n)
{
    return guarded([&]() -> int {
    int rc = enter(e);
    if (rc) return rc;
    if (!reason) return fail(MH_ERR_INVALID, "null argument");
    if (e->ref_reason_n <= 0) return fail(MH_ERR_NOT_SET, "no mh_refine_correspondences call to report on");
    if (n != e->ref_reason_n) return fail(MH_ERR_INVALID, "the last mh_refine_correspondences call had another number of rows");
    HIPCHK(hipMemcpyAsync(reason, e->ref_reason.p, (size_t)n, hipMemcpyDeviceToHost, e->stream));
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

// iterations one indexed / persistent launch may run a climb for before it hands it back still running (r06, advisor: 1 << 20 let
// a climb that neither converges nor dies spin for seconds inside one kernel; 120 000 = the launched schedule's own cap of 20 000
// rounds x 6 iterations, after which the host's "did not converge" path fires)
static constexpr int MS_MAX_ITERS_PER_LAUNCH = 120000;

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
    if (!e->h_ms) {                                                 // B result blocks, then the B seed rows, list offsets and list lengths
        HIPCHK(hipHostMalloc((void**)&e->h_ms, sizeof(MeanShiftResultBlock) * B + sizeof(int) * 3 * B, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_ms_dev, e->h_ms, 0));
    }
    // the (row, votes) lists of a batch travel packed, in one copy of exactly their pairs; the pinned buffer grows with the need
    auto list_room = [&](size_t pairs) -> hipError_t {
        if (pairs <= e->h_ms_list_pairs) return hipSuccess;
        size_t want = std::max<size_t>(pairs, std::max<size_t>(1 << 16, 2 * e->h_ms_list_pairs));
        if (e->h_ms_list) { (void)hipHostFree(e->h_ms_list); e->h_ms_list = nullptr; e->h_ms_list_pairs = 0; }
        const hipError_t he = hipHostMalloc((void**)&e->h_ms_list, sizeof(int) * 2 * want, hipHostMallocDefault);
        if (he == hipSuccess) e->h_ms_list_pairs = want;
        return he;
    };
    HIPCHK(e->ms_tickets.reserve((size_t)B));
    HIPCHK(e->ms_ctl.reserve((size_t)3 * B));
    HIPCHK(e->ms_partial2.reserve((size_t)B * 2 * 64 * 16));
    HIPCHK(e->ms_pcnt2.reserve((size_t)B * 2 * 64));
    HIPCHK(hipMemsetAsync(e->ms_tickets.p, 0, sizeof(int) * (size_t)B, e->stream));
    int* const starts = reinterpret_cast<int*>(e->h_ms + B);
    const int* const starts_dev = reinterpret_cast<const int*>(e->h_ms_dev + B);
    int* const list_off = starts + B;
    int* const list_len = starts + 2 * B;

    // r05: the index of this call's rows (k_ms_indexed).  The coordinate with the widest spread of its ordinary values (rows
    // parked at 1e300 and non-finite ones aside) is binned in cells of bandWidth^2 (1 + 2^-20) — wider if that would take
    // more cells than the index holds.  The choice affects speed only: any coordinate, any lo, any w >= that bound is exact.
    // r06 (advisor): only for an ordinary positive bandWidth^2 — the index's exactness rests on cells at least that wide; a huge or
    // non-finite band runs the launched / persistent schedule, which has no such precondition
    const bool indexed = e->tune_ms_indexed != 0 && ms_indexed_supported(n, d) && band_sq > 0.0 && band_sq < 1e299;
    MeanShiftIndex ix{};
    if (indexed) {
        double lo[16], hi[16];
        for (int j = 0; j < d; ++j) { lo[j] = 1e300; hi[j] = -1e300; }
        for (int i = 0; i < n; ++i) {
            const double* row = data + (size_t)i * d;
            for (int j = 0; j < d; ++j) {
                const double x = row[j];
                if (x > -1e299 && x < 1e299) { lo[j] = x < lo[j] ? x : lo[j]; hi[j] = x > hi[j] ? x : hi[j]; }
            }
        }
        int coord = 0;
        double spread = -1.0;
        for (int j = 0; j < d; ++j) { const double sp = hi[j] >= lo[j] ? hi[j] - lo[j] : 0.0; if (sp > spread) { spread = sp; coord = j; } }
        double width = band_sq * (1.0 + 0x1p-20);
        const int max_cells = ms_index_max_cells();
        if (!(spread >= 0.0) || !(spread < 1e299)) spread = 0.0;
        if (spread / width > (double)(max_cells - 2)) width = spread / (double)(max_cells - 2);
        ix.cells = std::max(1, std::min(max_cells, (int)(spread / width) + 2));
        ix.coord = coord;
        ix.lo = hi[coord] >= lo[coord] ? lo[coord] : 0.0;
        ix.inv_w = 1.0 / width;
        HIPCHK(e->ms_rs.reserve((size_t)n * d));
        HIPCHK(e->ms_order.reserve((size_t)n));
        HIPCHK(e->ms_cells.reserve((size_t)ix.cells + 1));
        HIPCHK(e->ms_cursor.reserve((size_t)ix.cells));
        HIPCHK(e->ms_cellcount.reserve((size_t)ix.cells));
        ix.rs = e->ms_rs.p; ix.order = e->ms_order.p; ix.cell_start = e->ms_cells.p;
        HIPCHK(launch_ms_index_build(e->ms_data.p, n, d, ix, e->ms_cellcount.p, e->ms_cursor.p, e->ms_cells.p, e->ms_order.p, e->ms_rs.p, e->stream));
    }

    // `init` of the reference (:125-130) is the ascending list of unvisited rows, rebuilt after every
    // climb; a Fenwick tree over the unvisited flags answers "the k-th unvisited row" in O(log n).
    std::vector<int> fen(n + 1, 0), visited(n, 0);
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
    double st_persist_us = 0, st_launch_us = 0, st_tail_us = 0, st_apply_us = 0, st_merge_us = 0;
    const auto t_call = std::chrono::steady_clock::now();
    long long st_persist_iters = 0, st_persist_rounds = 0, st_persist_climbs = 0, st_launch_rounds = 0, st_tail_climbs = 0, st_batches = 0, st_G = 0;
    std::vector<std::pair<int, int>> st_climbs;              // (iterations, rows touched) of every climb
    if (ms_stats) { HIPCHK(e->ms_ticks.reserve(13)); HIPCHK(hipMemsetAsync(e->ms_ticks.p, 0, sizeof(unsigned long long) * 13, e->stream)); }
    std::vector<double> cent;                                       // modes, d values each
    int n_cent = 0;
    std::vector<std::vector<std::pair<int, int>>> votes;            // per mode: (row, votes), unordered
    std::vector<int> pos(n, -1);                                    // scratch: row -> position in the list being merged into
    // modes binned by their first coordinate (cells of bandWidth / 2; anything beyond +-2^50 cells — rows parked at 1e300 —
    // shares one catch-all cell, where every centroid is a candidate for every other)
    std::unordered_map<long long, std::vector<int>> bins;
    const long long cell_of_special = (1ll << 62);
    const double inv_half = 2.0 / band_width;
    auto cell_of = [&](double x) -> long long {
        const double c = std::floor(x * inv_half);
        return (c > -0x1p50 && c < 0x1p50) ? (long long)c : cell_of_special;      // (false for NaN too)
    };
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
            if (indexed && round == 0) {
                // every climb of the batch from its seed; the dense ones come back running and go on below (persistent
                // when few enough, launched rounds otherwise)
                int keep = 0;                                  // climbs the persistent kernel can take over at once
                if (persist_ok && e->tune_ms_persist > 0 && ms_persist_supported(n, d)) {
                    int& per_cu = d == 10 ? e->ms_persist_per_cu : e->ms_persist_per_cu6;
                    if (per_cu < 0) { const int q = ms_persist_occupancy(d); if (q > 0) per_cu = q; }
                    const int room = std::max(0, per_cu) * e->cu_count * 7 / 8;
                    keep = std::min(e->tune_ms_persist, room / std::min(64, (n + 255) / 256));
                }
                HIPCHK(launch_ms_indexed(w, active, n_active, starts_dev, ix, band_sq, stop_thresh, MS_MAX_ITERS_PER_LAUNCH, e->tune_ms_dense,
                                         keep, e->ms_ctl.p, e->h_ms_dev, e->stream, ms_stats ? e->ms_ticks.p + 5 : nullptr));
                ++e->ms_indexed_launches;
            } else if (G > 0) {
                HIPCHK(launch_ms_persist(w, active, n_active, band_sq, stop_thresh, MS_MAX_ITERS_PER_LAUNCH, e->ms_ctl.p, e->ms_partial2.p, e->ms_pcnt2.p,
                                         e->h_ms_dev, e->stream, ms_stats ? e->ms_ticks.p : nullptr));
                ++e->ms_persist_launches;
                st_G += G;
            } else {
                HIPCHK(launch_ms_climb(w, active, n_active, round == 0 ? starts_dev : nullptr, band_sq, stop_thresh, e->tune_ms_batch,
                                       e->h_ms_dev, e->ms_tickets.p, e->stream));
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
        // all lists in one copy: packed one behind the other on the device (the lengths are in the published results)
        int longest = 0;
        size_t total = 0;
        for (int b = 0; b < climbs; ++b) {
            const int len = std::max(0, e->h_ms[b].out[2]);
            list_off[b] = (int)total;
            list_len[b] = len;
            total += (size_t)len;
            longest = std::max(longest, len);
        }
        if (total > 0) {
            if (total > (size_t)0x3fffffff) return fail(MH_ERR_INVALID, "mean shift: the lists of one batch exceed 2^30 pairs");
            HIPCHK(e->ms_heads.reserve(2 * total));
            HIPCHK(list_room(total));
            HIPCHK(launch_ms_pack(w, climbs, longest, starts_dev + B, starts_dev + 2 * B, e->ms_heads.p, e->stream));
            HIPCHK(hipMemcpyAsync(e->h_ms_list, e->ms_heads.p, sizeof(int) * 2 * total, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
        // apply the climbs in draw order; one whose seed an earlier climb of the batch has visited never started in
        // the reference's terms and is dropped
        if (ms_stats)
            for (int b = 0; b < climbs; ++b) st_climbs.emplace_back(e->h_ms[b].out[0], e->h_ms[b].out[2]);
        const auto t_apply = std::chrono::steady_clock::now();
        for (int b = 0; b < climbs; ++b) {
            const int st = starts[b];
            if (visited[st]) continue;
            const int* out = e->h_ms[b].out;
            const double* mean = e->h_ms[b].mean;
            const int len = out[2];
            const int* list = e->h_ms_list + 2 * (size_t)list_off[b];
            std::vector<std::pair<int, int>> mine(len);
            for (int k = 0; k < len; ++k) {
                mine[k] = { list[2 * k], list[2 * k + 1] };
                if (!visited[list[2 * k]]) { mark_visited(list[2 * k]); --unvisited; }
            }
            const auto t_merge = std::chrono::steady_clock::now();
            struct MergeClock { double* acc; std::chrono::steady_clock::time_point t0; ~MergeClock() { if (acc) *acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } }
                merge_clock{ ms_stats ? &st_merge_us : nullptr, t_merge };
            // (r05: the lists stay in the order the device compacted them — nothing below depends on it; sorting them was a
            // third of the host's share of a call at 50 000 rows)
            if (!out[1]) {
                if (!visited[st]) { mark_visited(st); --unvisited; }    // climb that captured no row
                continue;
            }
            int merge_with = -1;
            // :101-109, first centroid with sqrt(sum) < bandWidth/2.  r05: the centroids are binned by their first coordinate
            // in cells of bandWidth/2 — a centroid that passes differs by less than that in EVERY coordinate, so it lies in
            // the mean's cell or a neighbouring one — and the candidates are tested with the reference's own comparison, the
            // lowest index winning; the scan over all centroids was quadratic in the number of modes (5 500 at 50 000 rows).
            const double half = band_width / 2, reject = half * half * (1.0 + 1e-9);
            const long long cell0 = cell_of(mean[0]);
            for (long long cell = cell0 - 1; cell <= cell0 + 1; ++cell) {
                const auto it = bins.find(cell);
                if (it == bins.end()) continue;
                for (const int cn : it->second) {
                    if (merge_with >= 0 && cn > merge_with) continue;
                    const double* c = cent.data() + (size_t)cn * d;
                    double sq = 0.0;
                    int j = 0;
                    for (; j < d && sq <= reject; ++j) { const double x = mean[j] - c[j]; sq = sq + x * x; }
                    if (j == d && std::sqrt(sq) < half) merge_with = cn;
                }
                if (cell == cell_of_special) break;                       // (the catch-all cell has no neighbours)
            }
            if (merge_with > -1) {
                double* c = cent.data() + (size_t)merge_with * d;
                const long long before = cell_of(c[0]);
                for (int j = 0; j < d; ++j) c[j] = 0.5 * (c[j] + mean[j]);
                const long long after = cell_of(c[0]);
                if (after != before) {
                    auto& vb = bins[before];
                    vb.erase(std::find(vb.begin(), vb.end(), merge_with));
                    bins[after].push_back(merge_with);
                }
                // votes of the two lists added row by row through a row -> position index (no order needed)
                auto& a = votes[merge_with];
                for (size_t i = 0; i < a.size(); ++i) pos[a[i].first] = (int)i;
                for (const auto& pr : mine) {
                    if (pos[pr.first] >= 0) a[pos[pr.first]].second += pr.second;
                    else { pos[pr.first] = (int)a.size(); a.push_back(pr); }
                }
                for (const auto& pr : a) pos[pr.first] = -1;
            } else {
                cent.insert(cent.end(), mean, mean + d);
                bins[cell0].push_back(n_cent);
                ++n_cent;
                votes.push_back(std::move(mine));
            }
        }
        if (ms_stats) st_apply_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_apply).count();
    }
    const auto t_final = std::chrono::steady_clock::now();
    std::vector<int> best_votes(n, 0);
    for (int i = 0; i < n; ++i) assign[i] = -1;
    for (size_t r = 0; r < votes.size(); ++r)                       // :133-146, first maximum wins
        for (const auto& pr : votes[r])
            if (best_votes[pr.first] < pr.second) { best_votes[pr.first] = pr.second; assign[pr.first] = (int)r; }
    *n_modes = n_cent;
    if (modes) std::copy(cent.begin(), cent.begin() + (size_t)std::min(n_cent, max_modes) * d, modes);
    if (ms_stats) {
        unsigned long long tk[5] = { 0, 0, 0, 0, 0 };
        (void)hipMemcpy(tk, e->ms_ticks.p, sizeof(tk), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mh_mean_shift] persistent kernel, first climb's first workgroup: gate %.1f ms, row load %.1f ms, sweep + tree + partial stores %.1f ms, barrier %.1f ms, "
                        "new mean %.1f ms; mean G %.1f\n", tk[4] * 1e-5, tk[0] * 1e-5, tk[1] * 1e-5, tk[2] * 1e-5, tk[3] * 1e-5, st_persist_rounds ? (double)st_G / st_persist_rounds : 0.0);
    }
    if (ms_stats && indexed) {
        unsigned long long tk[8] = {};
        (void)hipMemcpy(tk, e->ms_ticks.p + 5, sizeof(tk), hipMemcpyDeviceToHost);
        for (int o = 0; o < 8; o += 4)
            if (tk[o + 3])
                fprintf(stderr, "[mh_mean_shift] indexed climbs, iterations with %s members: %llu; per iteration: candidates %.2f us, group sums %.2f us, new mean %.2f us\n",
                        o ? "> 64" : "<= 64", tk[o + 3], tk[o] * 0.01 / tk[o + 3], tk[o + 1] * 0.01 / tk[o + 3], tk[o + 2] * 0.01 / tk[o + 3]);
        fprintf(stderr, "[mh_mean_shift] index: coordinate %d, %d cells\n", ix.coord, ix.cells);
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
        fprintf(stderr, "[mh_mean_shift] host: applying the batches' climbs (lists, visited set, vote merging) %.1f ms, of which centroid search and vote merging %.1f ms; final assignment %.1f ms; whole call %.1f ms\n",
                st_apply_us * 1e-3, st_merge_us * 1e-3, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_final).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    if (ms_stats)
        fprintf(stderr, "[mh_mean_shift] n %d: %lld batches; launched rounds %lld (%.1f ms, of which rounds after the first %.1f ms on %lld climb-rounds); "
                        "persistent rounds %lld (%.1f ms, %lld climbs, longest climbs %lld iterations in sum = %.1f us per iteration)\n",
                n, st_batches, st_launch_rounds, st_launch_us * 1e-3, st_tail_us * 1e-3, st_tail_climbs, st_persist_rounds, st_persist_us * 1e-3,
                st_persist_climbs, st_persist_iters, st_persist_iters ? st_persist_us / (double)st_persist_iters : 0.0);
    return MH_OK;
    });
}

} // extern "C"
