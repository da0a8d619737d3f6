// fake_engine.cpp — TEST SCAFFOLDING ONLY (tests/test_host_boundary.py).  A stand-in for libmultih_hip.so with the
// mh_* entry points the host class calls, doing NO real computation: it lets the CPU-only test drive
// MultiH::Process() through every branch that builds cv::Mat objects (refined affinities, models, degenerate
// case) under AddressSanitizer, with cv_shim.h's OpenCV ownership rules.  Never linked into the product.
#include <cmath>
#include <cstring>
#include <vector>

#include "multih_hip.h"

struct mh_engine {
    int n = 0, m = 0;
    std::vector<double> src, dst, aff, H;
    int score_calls = 0;
    int mode = 0;          // 0: two planes survive, 1: only one model is ever proposed (degenerate tail of Process)
};

static int g_mode = 0;
extern "C" void fake_engine_set_mode(int mode) { g_mode = mode; }

extern "C" {
const char* mh_last_error(void) { return "fake engine"; }
int mh_create(mh_engine** out, int) { *out = new mh_engine(); (*out)->mode = g_mode; return MH_OK; }
void mh_destroy(mh_engine* e) { delete e; }
int mh_set_params(mh_engine*, double, double, double, double, int) { return MH_OK; }
int mh_synchronize(mh_engine*) { return MH_OK; }
int mh_set_correspondences(mh_engine* e, const double* s, const double* d, const double* a, int n)
{
    e->n = n;
    e->src.assign(s, s + 2 * n); e->dst.assign(d, d + 2 * n);
    if (a) e->aff.assign(a, a + 4 * n);
    return MH_OK;
}
int mh_set_epipolar(mh_engine*, const double*, const double*) { return MH_OK; }
int mh_set_neighbors_csr(mh_engine*, const int*, const int*, int) { return MH_OK; }
int mh_build_neighbors_knn(mh_engine*, int) { return MH_OK; }
int mh_build_neighbors_radius(mh_engine*, double, long long, long long* hits) { if (hits) *hits = 0; return MH_OK; }
int mh_set_fundamental_metric(mh_engine*, int) { return MH_OK; }
int mh_get_refine_reasons(mh_engine*, unsigned char* r, int n) { for (int i = 0; i < n; ++i) r[i] = (i % 10 < 2) ? 3 : 0; return MH_OK; }
int mh_estimate_fundamental(mh_engine* e, unsigned long long, int, double, double F[9], double e2[2], unsigned char* mask, int* inl)
{
    const double f[9] = { 0, -1, 2000, 1, 0, -1000, -2000, 1000, 0 };      // [e2]_x, e2 = (1000, 2000, 1)
    for (int i = 0; i < 9; ++i) F[i] = f[i] / 2236.0689;
    e2[0] = 1000; e2[1] = 2000;
    if (mask) std::memset(mask, 1, e->n);
    if (inl) *inl = e->n;
    return MH_OK;
}
int mh_epipoles(mh_engine*, const double*, double e1[2], double e2[2]) { e1[0] = 1000; e1[1] = 2000; e2[0] = 1000; e2[1] = 2000; return MH_OK; }
// drops two of every ten points; the refined affinity is the input + 1000 (so a test can
// tell refined values from the caller's)
int mh_refine_correspondences(mh_engine* e, const double*, const double*, const double*, const unsigned char*, unsigned char* keep, double* refined)
{
    for (int i = 0; i < e->n; ++i) {
        keep[i] = (i % 10) >= 2;                         // drops pairs, so index parity (= the fake plane) survives
        refined[8 * i + 0] = e->src[2 * i]; refined[8 * i + 1] = e->src[2 * i + 1];
        refined[8 * i + 2] = e->dst[2 * i]; refined[8 * i + 3] = e->dst[2 * i + 1];
        for (int q = 0; q < 4; ++q) refined[8 * i + 4 + q] = e->aff[4 * i + q] + 1000.0;
    }
    return MH_OK;
}
int mh_local_homographies(mh_engine* e, double, double* H, double* feat)
{
    if (H) for (int i = 0; i < 9 * e->n; ++i) H[i] = (i % 9) % 4 == 0 ? 1.0 : 0.0;
    if (feat) for (int i = 0; i < 10 * e->n; ++i) feat[i] = (double)((i / 10) % 3);
    return MH_OK;
}
int mh_mean_shift(mh_engine*, const double*, int n, int, double, unsigned long long, double*, int, int* assign, int* n_modes)
{
    for (int i = 0; i < n; ++i) assign[i] = i % 3;
    *n_modes = 3;
    return MH_OK;
}
int mh_propose_dlt4(mh_engine* e, unsigned long long, long long, int m)
{
    e->m = m;
    e->H.assign(9 * (size_t)m, 0.0);
    // Two homologies I + e2 v^T around the fake epipole e2 = (1000, 2000, 1): both are compatible with F = [e2]_x
    // (H^T F is skew-symmetric), so the host's merging step and compatibility check accept them on consistent data.
    for (int j = 0; j < m; ++j) {
        double* h = &e->H[9 * (size_t)j];
        const double v = 0.3 * (j % 7);
        h[0] = h[4] = 1.0; h[2] = 1000.0 * v; h[5] = 2000.0 * v; h[8] = 1.0 + v;
    }
    e->score_calls = 0;
    return MH_OK;
}
int mh_set_models(mh_engine* e, const double* H, int m) { e->m = m; e->H.assign(H, H + 9 * (size_t)m); return MH_OK; }
int mh_set_tuning(mh_engine*, int, int) { return MH_OK; }
int mh_get_expand_stats(mh_engine*, long long stats[24]) { for (int i = 0; i < 24; ++i) stats[i] = 0; return MH_OK; }
int mh_get_expand_trace(mh_engine*, int*, int) { return MH_ERR_NOT_SET; }
int mh_get_models(mh_engine* e, double* H) { std::memcpy(H, e->H.data(), sizeof(double) * 9 * (size_t)e->m); return MH_OK; }
int mh_score(mh_engine* e, double, const unsigned char*, int* counts)
{
    // greedy rounds: model `round` is the best with 30 inliers; after two rounds (or one, mode 1) nothing is left
    const int round = e->score_calls++;
    const int rounds = e->mode == 1 ? 1 : 2;
    for (int j = 0; j < e->m; ++j) counts[j] = 0;
    if (round < rounds && round < e->m) counts[round] = 30;
    return MH_OK;
}
int mh_inliers_of_model(mh_engine* e, int idx, double, int label, int* labels)
{
    for (int i = 0; i < e->n; ++i) if ((i % 2) == (idx % 2)) labels[i] = label;
    return MH_OK;
}
int mh_inliers_of_homography(mh_engine* e, const double*, double, int label, int* labels)
{
    for (int i = 0; i < e->n; ++i) if (i % 2) labels[i] = label;
    return MH_OK;
}
int mh_compat_trial_stats(mh_engine*, const double*, const int*, int clusters, const int*, const double*, const unsigned char*, int trials,
                          double* out)
{
    for (long long i = 0; i < 8ll * clusters * trials; ++i) out[i] = 0.0;      // every cluster looks compatible
    return MH_OK;
}
int mh_compat_trial_stats_fit(mh_engine*, const double*, const int*, int clusters, const int*, const double*, int trials, double* out, double*,
                              unsigned char*)
{
    for (long long i = 0; i < 8ll * clusters * trials; ++i) out[i] = 0.0;
    return MH_OK;
}
int mh_inlier_moments(mh_engine* e, double, double* mom, double* mineig)
{
    for (int j = 0; j < e->m; ++j) { for (int q = 0; q < 6; ++q) mom[6 * j + q] = 20.0; mineig[j] = 1.0; }
    return MH_OK;
}
int mh_labeling_step(mh_engine* e, int, int* labeling, double* energy, int* cycles)
{
    for (int i = 0; i < e->n; ++i) labeling[i] = e->m > 0 ? i % e->m : -1;
    *energy = 1234.0;
    if (cycles) *cycles = 1;
    return MH_OK;
}
}
extern "C" int mh_build_neighbors_knn_radius(mh_engine*, int, double) { return MH_OK; }
extern "C" int mh_set_transport(mh_engine*, int, int, mh_allgather_stream_fn, mh_allgather_dev_fn, void*) { return MH_OK; }
extern "C" int mh_select_greedy(mh_engine* e, double, int, int max_models, unsigned char* mask, double* H_out, long long* counters,
                                int* counts, int* selected, long long)
{
    // model `j` is the j-th pick with 30 inliers (every other point); two picks, or one in mode 1
    int k = e->mode == 1 ? 1 : 2;
    if (k > max_models) k = max_models;
    if (k > e->m) k = e->m;
    for (int j = 0; j < k; ++j) {
        std::memcpy(H_out + 9 * j, &e->H[9 * (size_t)j], sizeof(double) * 9);
        if (counters) counters[j] = j;
        if (counts) counts[j] = 30;
        if (mask) for (int i = 0; i < e->n; ++i) if ((i % 2) == (j % 2)) mask[i] = 0;
    }
    *selected = k;
    return MH_OK;
}
