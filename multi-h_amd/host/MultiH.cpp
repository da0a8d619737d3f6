// MultiH.cpp — the reference's orchestrator (M/MultiH.cpp) re-created over the
// gfx950 engine's C ABI.  Every heavy step is one mh_* call; this file holds the
// control flow of Process / ClusterMergingAndLabeling / MergingStep /
// LabelingStep with the reference's stop rules and conventions.
#include "MultiH.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <mutex>

#include "merge_step.h"
#include "multih_hip.h"

namespace {

// A matrix that OWNS its storage, filled from a raw array: allocate, then copy.  (cv::Mat(rows, cols, type, ptr) would
// only be a header over the caller's memory.)
cv::Mat OwnedMat(int rows, int cols, const double* v)
{
    cv::Mat m(rows, cols, CV_64F);
    double* p = reinterpret_cast<double*>(m.data);
    for (int i = 0; i < rows * cols; ++i) p[i] = v[i];
    return m;
}

cv::Mat MatFrom9(const double* h) { return OwnedMat(3, 3, h); }

bool Check(int rc, const char* what)
{
    if (rc == MH_OK) return true;
    std::cerr << "[Multi-H] engine error in " << what << ": " << mh_last_error() << "\n";
    return false;
}

} // namespace

MultiH::MultiH(double _thr_fund_mat, double _thr_hom, double _locality, double _lambda,
               int _minimum_inlier_number)
    : minimum_inlier_number(_minimum_inlier_number),
      threshold_fundamental_matrix(_thr_fund_mat),
      threshold_homography(_thr_hom),
      sqr_threshold_homography(_thr_hom * _thr_hom),
      locality_lambda(_locality),
      energy_lambda(_lambda),
      affine_threshold(DEFAULT_AFFINE_THRESHOLD),
      straightness_threshold(DEFAULT_LINENESS_THRESHOLD)
{
    for (double& f : fundamental_matrix) f = 0.0;
    epipole_2[0] = epipole_2[1] = 0.0;
}

MultiH::~MultiH() { Release(); }

// Engines are REUSED from one MultiH object to the next (r04).  The reference's harness makes a new MultiH per image pair
// (M/main.cpp:262); creating and destroying an engine — streams, a few dozen device buffers, their release with the device
// synchronisations that implies — costs 2.5 ms, a third of a whole Process() on a scene of 500 correspondences.  An engine
// carries no result from one call to the next: every Process() sets parameters, transport and correspondences anew, and
// mh_set_correspondences drops everything derived from the previous point set.  Engines that were given tuning knobs, or
// whose stream does not drain cleanly, are destroyed instead.  MULTIH_ENGINE_POOL=0 switches the reuse off.
namespace {
struct EnginePool {
    std::mutex mu;
    std::vector<std::pair<int, mh_engine*>> idle;      // (device, engine), at most two
};
EnginePool& engine_pool()
{
    static EnginePool* p = new EnginePool();           // never destroyed: the HIP runtime may be gone by the time statics are
    return *p;
}
bool engine_pool_enabled()
{
    const char* v = std::getenv("MULTIH_ENGINE_POOL");
    return !(v && v[0] == '0');
}
} // namespace

extern "C" __attribute__((visibility("default")))
void mhh_release_engine_pool()
{
    EnginePool& p = engine_pool();
    std::lock_guard<std::mutex> lock(p.mu);
    for (auto& de : p.idle) mh_destroy(de.second);
    p.idle.clear();
}

void MultiH::Release()
{
    if (!engine) return;
    if (engine_tuning.empty() && engine_pool_enabled() && mh_synchronize(engine) == MH_OK &&
        mh_set_transport(engine, 0, 1, nullptr, nullptr, nullptr) == MH_OK) {
        EnginePool& p = engine_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.idle.size() < 2) {
            p.idle.emplace_back(device, engine);
            engine = nullptr;
            return;
        }
    }
    mh_destroy(engine);
    engine = nullptr;
}

void MultiH::SetEpipolarGeometry(const double F[9], const double e2[2])
{
    for (int i = 0; i < 9; ++i) fundamental_matrix[i] = F[i];
    epipole_2[0] = e2[0];
    epipole_2[1] = e2[1];
    have_epipolar = true;
}

void MultiH::SetNeighbours(const std::vector<std::vector<int>>& hits) { neighbours = hits; }

void MultiH::SetInitialHomographies(const std::vector<cv::Mat>& Hs)
{
    initial_homographies.clear();
    for (const cv::Mat& h : Hs) initial_homographies.push_back(h.clone());
}

void MultiH::SetProposal(uint64_t seed, int hypotheses, int max_models)
{
    proposal_seed = seed;
    proposal_hypotheses = hypotheses;
    proposal_max_models = max_models;
}

bool MultiH::Process(std::vector<cv::Point2d> _srcPoints, std::vector<cv::Point2d> _dstPoints,
                     std::vector<cv::Mat> _affines)
{
    printf("[Multi-H] Processing has been started.\n");
    src_points_original = _srcPoints;
    dst_points_original = _dstPoints;
    affinities_original = _affines;
    return Process();
}

bool MultiH::EnsureEngine()
{
    if (!engine && engine_tuning.empty() && engine_pool_enabled()) {
        EnginePool& p = engine_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        for (size_t i = 0; i < p.idle.size(); ++i)
            if (p.idle[i].first == device) { engine = p.idle[i].second; p.idle.erase(p.idle.begin() + (long)i); break; }
    }
    if (!engine) {
        if (!Check(mh_create(&engine, device), "mh_create")) return false;
        for (const auto& kv : engine_tuning)
            if (!Check(mh_set_tuning(engine, kv.first, kv.second), "mh_set_tuning")) return false;
    }
    if (!Check(mh_set_transport(engine, shard_rank, std::max(shard_world, 1), shard_stream_allgather, shard_allgather, shard_ctx),
               "mh_set_transport"))
        return false;
    if (!Check(mh_set_fundamental_metric(engine, fundamental_metric), "mh_set_fundamental_metric")) return false;   // (sticky, and engines are reused)
    return Check(mh_set_params(engine, threshold_fundamental_matrix, threshold_homography,
                               locality_lambda, energy_lambda, minimum_inlier_number),
                 "mh_set_params");
}

namespace {
// an engine for a helper outside any MultiH object: from the pool when one is idle on `device`
mh_engine* BorrowEngine(int device)
{
    if (engine_pool_enabled()) {
        EnginePool& p = engine_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        for (size_t i = 0; i < p.idle.size(); ++i)
            if (p.idle[i].first == device) { mh_engine* e = p.idle[i].second; p.idle.erase(p.idle.begin() + (long)i); return e; }
    }
    mh_engine* e = nullptr;
    return Check(mh_create(&e, device), "mh_create") ? e : nullptr;
}
void ReturnEngine(int device, mh_engine* e)
{
    if (!e) return;
    if (engine_pool_enabled() && mh_synchronize(e) == MH_OK) {
        EnginePool& p = engine_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.idle.size() < 2) { p.idle.emplace_back(device, e); return; }
    }
    mh_destroy(e);
}
} // namespace

bool multih::FilterCorrespondencesByEpipolarGeometry(std::vector<cv::Point2d>& srcPoints, std::vector<cv::Point2d>& dstPoints,
                                                     std::vector<cv::Mat>& affines, double threshold, uint64_t seed,
                                                     int hypotheses, int metric, int device, std::vector<unsigned char>* mask_out)
{
    const size_t n = srcPoints.size();
    if (n < 8 || dstPoints.size() != n || affines.size() != n) return false;
    mh_engine* e = BorrowEngine(device);
    if (!e) return false;
    std::vector<double> s(2 * n), d(2 * n);
    for (size_t i = 0; i < n; ++i) {
        s[2 * i] = srcPoints[i].x; s[2 * i + 1] = srcPoints[i].y;
        d[2 * i] = dstPoints[i].x; d[2 * i + 1] = dstPoints[i].y;
    }
    std::vector<unsigned char> mask(n, 0);
    double F[9], e2[2];
    int inl = 0;
    const bool ok = Check(mh_set_fundamental_metric(e, metric), "mh_set_fundamental_metric") &&
                    Check(mh_set_correspondences(e, s.data(), d.data(), nullptr, (int)n), "mh_set_correspondences") &&
                    Check(mh_estimate_fundamental(e, seed, hypotheses, threshold, F, e2, mask.data(), &inl), "mh_estimate_fundamental");
    ReturnEngine(device, e);
    if (!ok) return false;
    if (mask_out) *mask_out = mask;
    size_t k = 0;
    for (size_t i = 0; i < n; ++i)
        if (mask[i]) {
            if (k != i) { srcPoints[k] = srcPoints[i]; dstPoints[k] = dstPoints[i]; affines[k] = affines[i]; }
            ++k;
        }
    srcPoints.resize(k); dstPoints.resize(k); affines.resize(k);
    return true;
}

bool MultiH::Process()
{
    if (src_points_original.size() < 8 || dst_points_original.size() != src_points_original.size() ||
        affinities_original.size() != src_points_original.size()) {
        std::cerr << "Error: Features are not set!\n";                       // M/MultiH.cpp:48
        return false;
    }
    const bool timing = std::getenv("MULTIH_TIMING") != nullptr;          // diagnostic: where Process() spends its time
    const auto t_process = std::chrono::system_clock::now();
    auto stage = [&](const char* what) {
        if (timing) printf("[Multi-H] %s done %.1f ms after Process() began\n", what,
                           std::chrono::duration<double, std::milli>(std::chrono::system_clock::now() - t_process).count());
    };
    if (!EnsureEngine()) return false;
    stage("engine");

    // GetFundamentalMatrixAndRefineData (M/MultiH.cpp:52, :770-848; §8(f) row 4): with
    // SetEpipolarGeometry the caller's F / e2 are used and the points are taken as already
    // filtered; otherwise F is estimated on the GPU below.
    src_points = src_points_original;
    dst_points = dst_points_original;
    affinities = affinities_original;
    degenerate_case = false;
    cluster_homographies.clear();
    labeling.clear();

    const int N = static_cast<int>(src_points.size());
    front_stages = FrontStages();
    front_stages.input = front_stages.in_ransac_mask = front_stages.triangulated = front_stages.affine_consistent = N;
    std::vector<double> s(2 * (size_t)N), d(2 * (size_t)N), a(4 * (size_t)N);
    for (int i = 0; i < N; ++i) {
        s[2 * i] = src_points[i].x; s[2 * i + 1] = src_points[i].y;
        d[2 * i] = dst_points[i].x; d[2 * i + 1] = dst_points[i].y;
        const cv::Mat& A = affinities[i];
        a[4 * i] = A.at<double>(0, 0); a[4 * i + 1] = A.at<double>(0, 1);
        a[4 * i + 2] = A.at<double>(1, 0); a[4 * i + 3] = A.at<double>(1, 1);
    }
    if (!Check(mh_set_correspondences(engine, s.data(), d.data(), a.data(), N), "mh_set_correspondences"))
        return false;
    stage("correspondences on the device");

    if (!have_epipolar) {
        // GetFundamentalMatrixAndRefineData (M/MultiH.cpp:770-848) on the GPU: RANSAC over normalised
        // 8-point hypotheses with Sampson scoring, LS refit, epipoles from F^T F / F F^T, then the
        // per-correspondence refinement of :807-838.
        std::vector<unsigned char> mask(N, 0);
        int inl = 0;
        const bool ok = Check(mh_estimate_fundamental(engine, proposal_seed ^ 0xf00dull, fundamental_hypotheses,
                                                      threshold_fundamental_matrix, fundamental_matrix, epipole_2,
                                                      mask.data(), &inl),
                              "mh_estimate_fundamental");
        double nrm = 0.0;
        for (double f : fundamental_matrix) nrm += f * f;
        degenerate_case = !ok || !(std::sqrt(nrm) >= 1e-5) || !std::isfinite(epipole_2[0]) ||
                          !std::isfinite(epipole_2[1]);                             // :779
        if (degenerate_case) {
            printf("[Multi-H] Degenerate case, the fundamental matrix cannot be estimated.\n");
        } else {
            // :807-838 on the GPU: Hartley-Sturm correction, affine consistency filter, optimal affinity
            double e1[2];
            std::vector<unsigned char> keep(N, 0);
            std::vector<double> refined(8 * (size_t)N);
            if (!Check(mh_epipoles(engine, fundamental_matrix, e1, epipole_2), "mh_epipoles") ||
                !Check(mh_refine_correspondences(engine, fundamental_matrix, e1, epipole_2, mask.data(), keep.data(),
                                                 refined.data()),
                       "mh_refine_correspondences"))
                return false;
            std::vector<unsigned char> reason(N, 0);
            if (!Check(mh_get_refine_reasons(engine, reason.data(), N), "mh_get_refine_reasons")) return false;
            front_stages.in_ransac_mask = front_stages.triangulated = front_stages.affine_consistent = 0;
            for (int i = 0; i < N; ++i) {
                if (reason[i] != MH_REFINE_NOT_IN_MASK) ++front_stages.in_ransac_mask;
                if (reason[i] == MH_REFINE_KEPT || reason[i] == MH_REFINE_AFFINE_TEST) ++front_stages.triangulated;
                if (reason[i] == MH_REFINE_KEPT) ++front_stages.affine_consistent;
            }
            std::vector<cv::Point2d> s2, d2;
            std::vector<cv::Mat> a2;
            for (int i = 0; i < N; ++i)
                if (keep[i]) {
                    const double* r = &refined[8 * (size_t)i];
                    s2.push_back(cv::Point2d(r[0], r[1]));
                    d2.push_back(cv::Point2d(r[2], r[3]));
                    a2.push_back(OwnedMat(2, 2, r + 4));
                }
            printf("[Multi-H] %d points kept from the initial %d after filtering.\n", (int)s2.size(), N);   // :840
            if (log_to_console)
                printf("[Multi-H]   stages: %d in the RANSAC mask at %.2f px, %d after OptimalTriangulation, %d after distanceError <= 1\n",
                       front_stages.in_ransac_mask, threshold_fundamental_matrix, front_stages.triangulated, front_stages.affine_consistent);
            if (s2.size() < 8) {
                degenerate_case = true;
                printf("[Multi-H] Degenerate case, not enough points remained.\n");                         // :845
            } else {
                src_points.swap(s2); dst_points.swap(d2); affinities.swap(a2);
                const int K = static_cast<int>(src_points.size());
                std::vector<double> ss(2 * (size_t)K), dd(2 * (size_t)K), aa(4 * (size_t)K);
                for (int i = 0; i < K; ++i) {
                    ss[2 * i] = src_points[i].x; ss[2 * i + 1] = src_points[i].y;
                    dd[2 * i] = dst_points[i].x; dd[2 * i + 1] = dst_points[i].y;
                    const cv::Mat& A = affinities[i];
                    aa[4 * i] = A.at<double>(0, 0); aa[4 * i + 1] = A.at<double>(0, 1);
                    aa[4 * i + 2] = A.at<double>(1, 0); aa[4 * i + 3] = A.at<double>(1, 1);
                }
                if (!Check(mh_set_correspondences(engine, ss.data(), dd.data(), aa.data(), K), "mh_set_correspondences"))
                    return false;
            }
        }
    }

    if (degenerate_case) {
        HandleDegenerateCase();
        return true;
    }
    if (!Check(mh_set_epipolar(engine, fundamental_matrix, epipole_2), "mh_set_epipolar")) return false;

    if (!initial_homographies.empty()) {
        for (const cv::Mat& h : initial_homographies) cluster_homographies.push_back(h.clone());
    } else if (init_mode == INIT_STABLE_SETS) {
        auto t0 = std::chrono::system_clock::now();
        if (!EstablishStablePointSets()) return false;
        std::chrono::duration<double> el = std::chrono::system_clock::now() - t0;
        printf("[Multi-H] Stable cluster estimation time = %f secs\n", el.count());          // :74
    } else if (!ProposeInitialModels()) {
        return false;
    }

    stage("initial models");
    ClusterMergingAndLabeling();
    stage("neighbourhood + alternation");

    if (cluster_homographies.size() > 1 && run_compatibility_check) {        // :78-86
        const int before = static_cast<int>(cluster_homographies.size());
        auto t0 = std::chrono::system_clock::now();
        post_filter_failed = false;
        HomographyCompatibilityCheck();
        if (post_filter_failed) return false;                                // the engine refused: no host fallback
        std::chrono::duration<double> el = std::chrono::system_clock::now() - t0;
        printf("[Multi-H] Compatibility check time = %f secs (%d clusters removed from %d)\n", el.count(),
               before - (int)cluster_homographies.size(), before);
    }

    if (cluster_homographies.size() <= 1) {                                  // :88-94
        labeling.clear();
        labeling.resize(src_points.size(), -1);
        cluster_homographies.resize(0);
        HandleDegenerateCase();
    }
    return true;
}

void MultiH::HomographyCompatibilityCheck()
{
    const int N = static_cast<int>(src_points.size());
    const int nh = static_cast<int>(cluster_homographies.size());
    if (nh == 0 || (int)labeling.size() != N) return;
    std::vector<double> s(2 * (size_t)N), d(2 * (size_t)N), H(9 * (size_t)nh);
    for (int i = 0; i < N; ++i) {
        s[2 * i] = src_points[i].x; s[2 * i + 1] = src_points[i].y;
        d[2 * i] = dst_points[i].x; d[2 * i + 1] = dst_points[i].y;
    }
    for (int i = 0; i < nh; ++i) {
        const double* p = reinterpret_cast<const double*>(cluster_homographies[i].data);
        for (int k = 0; k < 9; ++k) H[9 * (size_t)i + k] = p[k];
    }
    // the trials' 3-point fits (r06) and their order statistics come from the engine (csrc/compat.hip); the replay of the draws
    // and the reference's stale-buffer bookkeeping are host work (merge_step.cpp)
    multih::CompatStatsFn on_engine;
    if (engine)
        on_engine = [this](const double* pts, const int* begin, int clusters, const int* tri, const double* Ht,
                           const unsigned char* ok, int trials, double* out) {
            if (!Ht) return Check(mh_compat_trial_stats_fit(engine, pts, begin, clusters, tri, fundamental_matrix, trials, out, nullptr, nullptr),
                                  "mh_compat_trial_stats_fit");
            return Check(mh_compat_trial_stats(engine, pts, begin, clusters, tri, Ht, ok, trials, out), "mh_compat_trial_stats");
        };
    bool failed = false;
    const int kept = multih::CompatibilityCheck(s.data(), d.data(), N, labeling.data(), H.data(), nh,
                                                fundamental_matrix, sqr_threshold_homography,
                                                minimum_inlier_number, proposal_seed ^ 0xc0117a7ull, nullptr,
                                                engine ? &on_engine : nullptr, &failed, engine != nullptr);
    if (failed) { post_filter_failed = true; return; }
    cluster_homographies.clear();
    for (int i = 0; i < kept; ++i) cluster_homographies.push_back(MatFrom9(&H[9 * (size_t)i]));
}

// DrawClusters, M/MultiH.cpp:314-350: one filled circle per labelled correspondence in each image, colour by plane.
// The reference's fixed palette (:322-332) for the first eleven planes; it writes all eleven entries whatever the
// number of planes (out of bounds below eleven planes), which is not reproduced: the palette is sized first.
void MultiH::DrawClusters(cv::Mat& img1, cv::Mat& img2, int size)
{
    static const double fixed[11][3] = { { 255, 0, 0 }, { 255, 255, 255 }, { 0, 0, 255 }, { 255, 255, 0 }, { 255, 0, 255 },
                                         { 0, 255, 255 }, { 0, 255, 0 }, { 127, 255, 0 }, { 0, 255, 127 }, { 127, 127, 0 },
                                         { 63, 255, 127 } };
    const size_t k = cluster_homographies.size();
    std::vector<cv::Scalar> colors(std::max<size_t>(k + 1, 12));
    colors[0] = cv::Scalar(0, 0, 0);
    uint64_t z = proposal_seed ^ 0xc0105ull;                              // planes beyond the palette: counter RNG, not rand()
    for (size_t i = 1; i < colors.size(); ++i) {
        if (i <= 11) { colors[i] = cv::Scalar(fixed[i - 1][0], fixed[i - 1][1], fixed[i - 1][2]); continue; }
        double c[3];
        for (double& v : c) {
            z += 0x9E3779B97F4A7C15ull;
            uint64_t x = z;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
            x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
            v = 255.0 * (double)((x ^ (x >> 31)) >> 11) * (1.0 / 9007199254740992.0);
        }
        colors[i] = cv::Scalar(c[0], c[1], c[2]);
    }
    for (size_t i = 0; i < labeling.size(); ++i) {
        if (labeling[i] == -1) continue;                                                  // :336
        const cv::Scalar& col = colors[(size_t)labeling[i] + 1];
        if (k == 1 && i < src_points_original.size()) {                                   // degenerate path labels the ORIGINAL points, :339-343
            cv::circle(img1, src_points_original[i], size, col, -1);
            cv::circle(img2, dst_points_original[i], size, col, -1);
        } else if (i < src_points.size()) {
            cv::circle(img1, src_points[i], size, col, -1);
            cv::circle(img2, dst_points[i], size, col, -1);
        }
    }
}

bool MultiH::UploadModels()
{
    const int nh = static_cast<int>(cluster_homographies.size());
    std::vector<double> H(9 * (size_t)nh);
    for (int i = 0; i < nh; ++i) {
        const double* p = reinterpret_cast<const double*>(cluster_homographies[i].data);
        for (int k = 0; k < 9; ++k) H[9 * (size_t)i + k] = p[k];
    }
    return Check(mh_set_models(engine, H.data(), nh), "mh_set_models");
}

bool MultiH::DownloadModels(int count)
{
    std::vector<double> H(9 * (size_t)count);
    if (!Check(mh_get_models(engine, H.data()), "mh_get_models")) return false;
    cluster_homographies.clear();
    for (int i = 0; i < count; ++i) cluster_homographies.push_back(MatFrom9(&H[9 * (size_t)i]));
    return true;
}

// EstablishStablePointSets (M/MultiH.cpp:604-694) preceded by ComputeLocalHomographies (:696-717):
// point-wise HAF homographies and their 10-D features on the GPU (mh_local_homographies), mean
// shift with bandwidth thr_H whose climbs run on the GPU (mh_mean_shift), then per cluster of at
// least three points one least-squares 3-point homography with LM refinement (host, :664-688).
bool MultiH::EstablishStablePointSets()
{
    const int N = static_cast<int>(src_points.size());
    const bool timing = std::getenv("MULTIH_TIMING") != nullptr;          // diagnostic: where the initialisation's time goes
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    std::vector<double> feat(10 * (size_t)N);
    if (!Check(mh_local_homographies(engine, locality_lambda, nullptr, feat.data()), "mh_local_homographies"))
        return false;
    const double ms_local = ms_since(t0);
    // a degenerate per-point solve (NaN features) would poison the L1 ball test; park such rows far away
    for (double& f : feat) if (!std::isfinite(f)) f = 1e300;
    std::vector<int> assign(N);
    int k = 0;
    if (!Check(mh_mean_shift(engine, feat.data(), N, 10, threshold_homography, proposal_seed ^ 0x57ab1eull,
                             nullptr, 0, assign.data(), &k),
               "mh_mean_shift"))
        return false;
    const double ms_shift = ms_since(t0) - ms_local;
    std::vector<std::vector<int>> members(k);
    for (int i = 0; i < N; ++i) if (assign[i] >= 0) members[assign[i]].push_back(i);
    {
        // one 3-point fit with LM per cluster of at least three points (:664-688), on the host's cores (r06), taken in cluster order
        std::vector<double> sxy(2 * (size_t)N), dxy(2 * (size_t)N), Hc;
        std::vector<unsigned char> okc;
        for (int i = 0; i < N; ++i) {
            sxy[2 * (size_t)i] = src_points[i].x; sxy[2 * (size_t)i + 1] = src_points[i].y;
            dxy[2 * (size_t)i] = dst_points[i].x; dxy[2 * (size_t)i + 1] = dst_points[i].y;
        }
        multih::Homography3PTClusters(sxy.data(), dxy.data(), members, fundamental_matrix, Hc, okc);
        for (int c = 0; c < k; ++c)
            if (okc[c]) cluster_homographies.push_back(MatFrom9(&Hc[9 * (size_t)c]));
    }
    if (timing)
        printf("[Multi-H] stable sets: per-point homographies %.2f ms, mean shift %.2f ms (%d modes), 3-point fits %.2f ms\n",
               ms_local, ms_shift, k, ms_since(t0) - ms_local - ms_shift);
    if (log_to_console)
        printf("[Multi-H] Number of stable, local clusters = %d\n", (int)cluster_homographies.size());   // :693
    return true;
}

// north_star "propose": a batch of minimal-sample DLT hypotheses scored against all points,
// then greedy selection — take the best-supported hypothesis, remove its inliers from the
// support mask, re-score, repeat (the sequential-RANSAC scheme of the dead
// M/MultipleHomographies.h:146-175, where points already claimed are skipped via usabilityMask).
bool MultiH::ProposeInitialModels()
{
    std::vector<unsigned char> mask(src_points.size(), 1);
    const bool ok = ProposeModels(proposal_seed, 0, proposal_hypotheses, proposal_max_models, mask);
    if (ok && log_to_console)
        printf("[Multi-H] Proposed %d models from %d DLT hypotheses\n", (int)cluster_homographies.size(),
               proposal_hypotheses);
    return ok;
}

// `mask`: 1 = point still unexplained (in/out).  Appends the selected models to cluster_homographies.
// One hypothesis batch -> mh_propose_dlt4 (this rank's shard of it) -> mh_select_greedy: scoring, arg-max, claiming
// the winner's inliers and pruning the candidates all stay on the device; per round the host reads three control
// words.  Sharded (SetSharding): rank r owns counters [first + off_r, first + off_r + m_r); the ranks all-gather
// their int32 score vectors and the H each offers through the caller's transport on DEVICE buffers, and the selection
// order is the single-GPU one (highest score, lowest counter on ties), so world sizes 1..G give identical model lists.
bool MultiH::ProposeModels(uint64_t seed, long long first, int M, int max_models, std::vector<unsigned char>& mask)
{
    const int need = std::max(minimum_inlier_number, 8);
    if (M <= 0 || max_models <= 0) return true;
    const int W = std::max(shard_world, 1), base = M / W, rem = M % W;
    const int mine = base + (shard_rank < rem ? 1 : 0);
    const long long off = (long long)shard_rank * base + std::min(shard_rank, rem);
    if (mine > 0) {
        if (!Check(mh_propose_dlt4(engine, seed, first + off, mine), "mh_propose_dlt4")) return false;
    } else if (!Check(mh_set_models(engine, nullptr, 0), "mh_set_models")) {
        return false;
    }
    std::vector<double> H(9 * (size_t)max_models);
    int selected = 0;
    if (!Check(mh_set_tuning(engine, 30, proposal_refit ? 1 : 0), "mh_set_tuning(30)")) return false;
    if (!Check(mh_select_greedy(engine, sqr_threshold_homography, need, max_models, mask.data(), H.data(), nullptr, nullptr,
                                &selected, (long long)M),
               "mh_select_greedy"))
        return false;
    for (int i = 0; i < selected; ++i) cluster_homographies.push_back(MatFrom9(&H[9 * (size_t)i]));
    return true;
}

void MultiH::SetSharding(int rank, int world, AllGatherFn fn, void* ctx)
{
    shard_stream_allgather = nullptr;
    if (world <= 1 || !fn || rank < 0 || rank >= world) {
        shard_rank = 0; shard_world = 1; shard_allgather = nullptr; shard_ctx = nullptr;
        return;
    }
    shard_rank = rank; shard_world = world; shard_allgather = fn; shard_ctx = ctx;
}

void MultiH::SetShardingStream(int rank, int world, StreamAllGatherFn fn, void* ctx)
{
    shard_allgather = nullptr;
    if (world < 1 || !fn || rank < 0 || rank >= world) {
        shard_rank = 0; shard_world = 1; shard_stream_allgather = nullptr; shard_ctx = nullptr;
        return;
    }
    shard_rank = rank; shard_world = world; shard_stream_allgather = fn; shard_ctx = ctx;      // world == 1: a one-rank communicator
}

void MultiH::ClusterMergingAndLabeling()
{
    auto start = std::chrono::system_clock::now();
    const int N = static_cast<int>(src_points.size());

    // Neighbourhood (M/MultiH.cpp:233-253).  Given hits are used as they are; otherwise the engine builds them in the
    // same float32 (x1,y1,x2,y2) space (MultiH.h, SetNeighbourK / SetNeighbourRadius).
    bool ok;
    if (!neighbours.empty()) {
        std::vector<int> rowptr(N + 1, 0), col;
        for (int i = 0; i < N; ++i) {
            if (i < (int)neighbours.size()) col.insert(col.end(), neighbours[i].begin(), neighbours[i].end());
            rowptr[i + 1] = static_cast<int>(col.size());
        }
        ok = Check(mh_set_neighbors_csr(engine, rowptr.data(), col.data(), N), "mh_set_neighbors_csr");
    } else if (neighbour_mode == NEIGHBOURS_APPROX) {
        // the reference's own rule as FLANN's default search answers it (MultiH.h, SetNeighbourApprox): built on the host —
        // where the reference builds it — from the float32 (x1, y1, x2, y2) vectors of :233-250, handed over as directed hits
        std::vector<double> pv(4 * (size_t)N);
        for (int i = 0; i < N; ++i) {
            pv[4 * (size_t)i] = (double)static_cast<float>(src_points[i].x);
            pv[4 * (size_t)i + 1] = (double)static_cast<float>(src_points[i].y);
            pv[4 * (size_t)i + 2] = (double)static_cast<float>(dst_points[i].x);
            pv[4 * (size_t)i + 3] = (double)static_cast<float>(dst_points[i].y);
        }
        std::vector<std::vector<int>> hits;
        multih::ApproxNeighbourHits(pv.data(), N, approx_trees, approx_checks, 1.0 / locality_lambda, approx_seed, hits);
        std::vector<int> rowptr(N + 1, 0), col;
        for (int i = 0; i < N; ++i) {
            col.insert(col.end(), hits[i].begin(), hits[i].end());
            rowptr[i + 1] = static_cast<int>(col.size());
        }
        if (log_to_console) printf("[Multi-H] %d approximate neighbourhood hits (%d trees, %d checks)\n", (int)col.size(), approx_trees, approx_checks);
        ok = Check(mh_set_neighbors_csr(engine, rowptr.data(), col.data(), N), "mh_set_neighbors_csr");
    } else if (neighbour_mode == NEIGHBOURS_KNN) {
        // the k nearest hits within the reference's radius 1 / locality_lambda (M/MultiH.cpp:252-253); see MultiH.h
        ok = Check(mh_build_neighbors_knn_radius(engine, std::min(knn, N - 1), 1.0 / locality_lambda), "mh_build_neighbors_knn_radius");
    } else {
        // the complete radius list, answered exactly; it grows like N^2, and beyond `neighbour_max_hits` the engine refuses
        long long hits = 0;
        const int rc = mh_build_neighbors_radius(engine, neighbour_radius, neighbour_max_hits, &hits);
        if (rc == MH_ERR_OVERFLOW) {
            printf("[Multi-H] %lld neighbourhood hits within %.1f px exceed the limit of %lld: using the %d nearest hits instead\n",
                   hits, neighbour_radius, neighbour_max_hits, std::min(knn, N - 1));
            ok = Check(mh_build_neighbors_knn_radius(engine, std::min(knn, N - 1), neighbour_radius), "mh_build_neighbors_knn_radius");
        } else {
            ok = Check(rc, "mh_build_neighbors_radius");
            if (ok && log_to_console) printf("[Multi-H] %lld neighbourhood hits within %.1f px\n", hits, neighbour_radius);
        }
    }
    std::chrono::duration<double> el = std::chrono::system_clock::now() - start;
    printf("[Multi-H] Adjacency-matrix calculation time = %f secs\n", el.count());      // :258
    if (!ok) { cluster_homographies.clear(); return; }

    start = std::chrono::system_clock::now();
    int iteration_number = 0;
    labeling.resize(N, -1);                                                             // :263
    double lastEnergy = INT_MAX;
    int not_changed_number = 0;
    labeling_steps_run = 0;

    const bool timing = std::getenv("MULTIH_TIMING") != nullptr;                        // diagnostic: where the loop's time goes
    double merge_s = 0.0, label_s = 0.0;
    auto seconds_since = [](std::chrono::time_point<std::chrono::system_clock> t0) {
        return std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count();
    };
    while (iteration_number++ < MAX_ITERATION_NUMBER) {                                 // :267
        bool changed = false;
        const auto t_merge = std::chrono::system_clock::now();
        const bool merged = MergingStep(changed);
        merge_s += seconds_since(t_merge);
        if (!merged) break;
        if (iter_hypotheses > 0 && iteration_number > 1) {
            // PEARL re-proposal on the points the current labeling leaves unexplained
            std::vector<unsigned char> mask(N);
            for (int i = 0; i < N; ++i) mask[i] = labeling[i] < 0 ? 1 : 0;
            const size_t before = cluster_homographies.size();
            if (!ProposeModels(proposal_seed, (long long)iteration_number * iter_hypotheses + proposal_hypotheses,
                               iter_hypotheses, iter_max_new, mask))
                break;
            if (cluster_homographies.size() != before) changed = true;
        }
        if (changed) not_changed_number = 0;
        else ++not_changed_number;

        if (cluster_homographies.size() == 1) {                                         // :280-285
            labeling.resize(N, -1);
            ComputeInliersOfHomography(0);
            break;
        } else if (cluster_homographies.size() == 0)
            break;

        double energy;
        const auto t_label = std::chrono::system_clock::now();
        const bool labelled = LabelingStep(energy, changed);
        label_s += seconds_since(t_label);
        if (labelled) ++labeling_steps_run;
        if (timing) {
            long long st[24] = {};
            (void)mh_get_expand_stats(engine, st);
            printf("[Multi-H] iteration %d: %d clusters, changed %d, merging %.1f ms so far, labeling %.1f ms so far (this step %.1f ms: "
                   "%lld cycles, %lld moves solved, core %lld / max %lld, %lld relabels, %lld barriers, solver %.1f ms of which tail rounds %.1f ms)\n",
                   iteration_number, (int)cluster_homographies.size(), changed ? 1 : 0, merge_s * 1e3, label_s * 1e3,
                   seconds_since(t_label) * 1e3, st[0], st[10], st[11], st[12], st[14], st[13], (double)st[15] * 1e-3, (double)st[19] * 1e-3);
            std::vector<int> tr(8 * 64, 0);              // per-move log, when the caller switched it on (mh_set_tuning key 8)
            if (mh_get_expand_trace(engine, tr.data(), 64) == MH_OK)
                for (int mv = 0; mv < 64; ++mv)
                    if (tr[8 * mv + 6] > 200000)          // moves of more than 2 ms
                        printf("[Multi-H]    move %d: core %d, %d workgroups, %d relabels, %d intervals, %d push phases, %d barriers, %.1f ms\n",
                               mv, tr[8 * mv], tr[8 * mv + 1], tr[8 * mv + 2], tr[8 * mv + 3], tr[8 * mv + 4], tr[8 * mv + 5],
                               tr[8 * mv + 6] * 1e-5);
        }
        {
            // a labeling that ran on a cut-down solver grid (barrier timeouts on a shared GPU) is not silent: same
            // labels, but slower — say so whenever it happens
            long long st[24] = {};
            if (labelled && mh_get_expand_stats(engine, st) == MH_OK && st[20] > 0)
                printf("[Multi-H] iteration %d: the alpha-expansion's solver launch was restarted %lld time(s) after a grid-barrier "
                       "timeout and ran on %lld workgroups (is the GPU shared?)\n", iteration_number, st[20], st[21]);
        }
        if (!labelled) break;
        if (log_to_console)
            printf("Iteration %d.   Number of clusters = %d   Energy = %f\n", iteration_number,
                   (int)cluster_homographies.size(), energy);

        if ((!changed && std::abs(lastEnergy - energy) < CONVERGENCE_THRESHOLD) || not_changed_number > 10 ||
            (fixed_iterations > 0 && iteration_number >= fixed_iterations)) {           // :295
            final_energy = energy;
            break;
        }
        lastEnergy = energy;
    }
    printf("[Multi-H] Optimization started... Iteration %d\n", iteration_number);       // :305
    el = std::chrono::system_clock::now() - start;
    loop_seconds = el.count();
    printf("[Multi-H] Alternating optimization time = %f secs\n", el.count());          // :310
    final_iteration_number = iteration_number - 1;                                      // :311
}

// MergingStep, M/MultiH.cpp:352-471: models -> 6-D features -> mean-shift modes -> one
// homography per mode -> inlier scoring + collinearity filter -> `changed` iff the count changed.
// Mode -> homography is GetHomography3PT with its LM refinement (:995-1055, multih::Homography3PT).
// The N x modes scoring and the 3x3 scatter eigen test (:430-463) run on the GPU
// (mh_inlier_moments).
// MergingStep (M/MultiH.cpp:352-471) on a given engine: host candidates (merge_step.cpp MergeCandidates), then the N x
// candidates scoring and the collinearity filter on the GPU (mh_inlier_moments; :430-463).  kept: the surviving
// candidates, 9 doubles each.  Returns an MH_* status.  The correspondences must be resident in the engine; its model
// set is replaced by the candidates.
static int MergingStepOnEngine(mh_engine* engine, const double* H, int nh, const double F[9], double thr_h, double straightness,
                               uint64_t seed, std::vector<double>& kept, uint64_t* draws)
{
    kept.clear();
    std::vector<double> cand;
    const int nc = multih::MergeCandidates(H, nh, F, thr_h, seed, nullptr, nullptr, cand, nullptr, draws);
    if (nc == 0) return MH_OK;
    int rc = mh_set_models(engine, cand.data(), nc);
    if (rc != MH_OK) return rc;
    std::vector<double> mom(6 * (size_t)nc), mineig(nc);
    rc = mh_inlier_moments(engine, thr_h * thr_h, mom.data(), mineig.data());
    if (rc != MH_OK) return rc;
    for (int i = 0; i < nc; ++i) {
        const int inl = static_cast<int>(mom[6 * (size_t)i]);
        if (mineig[i] < straightness || inl < 3) continue;                             // :462
        kept.insert(kept.end(), cand.begin() + 9 * (size_t)i, cand.begin() + 9 * (size_t)(i + 1));
    }
    return MH_OK;
}

// test hook (tests/test_gpu_alternation.py): one MergingStep of the product on `engine`, to be compared bit for bit with
// the oracle's mho_merging_step.  kept: capacity nh x 9; returns the number kept or a negative MH_* status.
extern "C" __attribute__((visibility("default")))
int mhh_merging_step(mh_engine* engine, const double* H, int nh, const double* F, double thr_h, double straightness,
                     unsigned long long seed, double* kept_out, int* changed, unsigned long long* draws)
{
    std::vector<double> kept;
    uint64_t d = 0;
    const int rc = MergingStepOnEngine(engine, H, nh, F, thr_h, straightness, seed, kept, &d);
    if (rc != MH_OK) return rc;
    std::copy(kept.begin(), kept.end(), kept_out);
    if (changed) *changed = (int)(kept.size() / 9) != nh;
    if (draws) *draws = d;
    return (int)(kept.size() / 9);
}

bool MultiH::MergingStep(bool& changed)
{
    const int nh = static_cast<int>(cluster_homographies.size());
    changed = false;
    if (nh == 0) return true;
    std::vector<double> H(9 * (size_t)nh);
    for (int i = 0; i < nh; ++i) {
        const double* p = reinterpret_cast<const double*>(cluster_homographies[i].data);
        for (int k = 0; k < 9; ++k) H[9 * (size_t)i + k] = p[k];
    }
    std::vector<double> kept9;
    const int rc = MergingStepOnEngine(engine, H.data(), nh, fundamental_matrix, threshold_homography, straightness_threshold,
                                       proposal_seed ^ 0x4d53u ^ (merge_rng_counter << 20), kept9, nullptr);
    ++merge_rng_counter;
    if (!Check(rc, "MergingStep (mh_set_models / mh_inlier_moments)")) return false;
    std::vector<cv::Mat> kept;
    for (size_t i = 0; i + 9 <= kept9.size(); i += 9) kept.push_back(MatFrom9(&kept9[i]));
    changed = kept.size() != cluster_homographies.size();                               // :468
    if (changed) cluster_homographies = kept;                                           // :469-470
    return true;
}

bool MultiH::LabelingStep(double& energy, bool changed)
{
    energy = 0;
    if (!UploadModels()) return false;
    int cycles = 0;
    if (!Check(mh_labeling_step(engine, changed ? 0 : 1, labeling.data(), &energy, &cycles), "mh_labeling_step"))
        return false;
    return DownloadModels(static_cast<int>(cluster_homographies.size()));
}

void MultiH::ComputeInliersOfHomography(int idx)
{
    if (!UploadModels()) return;
    Check(mh_inliers_of_model(engine, idx, sqr_threshold_homography, idx, labeling.data()), "mh_inliers_of_model");
}

// HandleDegenerateCase (M/MultiH.cpp:719-741) calls cv::findHomography(RANSAC) on the ORIGINAL
// points and labels its inliers 0.  Here: best-supported of `proposal_hypotheses` DLT hypotheses.
void MultiH::HandleDegenerateCase()
{
    const int N = static_cast<int>(src_points_original.size());
    labeling.assign(N, -1);
    // The engine may still hold the FILTERED, refined correspondences (Process() re-uploads them after
    // GetFundamentalMatrixAndRefineData); the reference fits and labels the originals here (:725-739).
    {
        std::vector<double> s(2 * (size_t)N), d(2 * (size_t)N), a(4 * (size_t)N);
        for (int i = 0; i < N; ++i) {
            s[2 * i] = src_points_original[i].x; s[2 * i + 1] = src_points_original[i].y;
            d[2 * i] = dst_points_original[i].x; d[2 * i + 1] = dst_points_original[i].y;
            const cv::Mat& A = affinities_original[i];
            a[4 * i] = A.at<double>(0, 0); a[4 * i + 1] = A.at<double>(0, 1);
            a[4 * i + 2] = A.at<double>(1, 0); a[4 * i + 3] = A.at<double>(1, 1);
        }
        if (!Check(mh_set_correspondences(engine, s.data(), d.data(), a.data(), N), "mh_set_correspondences")) return;
    }
    const int M = std::max(proposal_hypotheses, 1000);
    if (!Check(mh_propose_dlt4(engine, proposal_seed ^ 0xdeadull, 0, M), "mh_propose_dlt4")) return;
    std::vector<int> counts(M);
    if (!Check(mh_score(engine, sqr_threshold_homography, nullptr, counts.data()), "mh_score")) return;
    const int best = static_cast<int>(std::max_element(counts.begin(), counts.end()) - counts.begin());
    std::vector<double> H(9 * (size_t)M);
    if (!Check(mh_get_models(engine, H.data()), "mh_get_models")) return;
    if (!Check(mh_inliers_of_model(engine, best, sqr_threshold_homography, 0, labeling.data()), "mh_inliers_of_model"))
        return;
    cluster_homographies.push_back(MatFrom9(&H[9 * (size_t)best]));
}

// Sharding for the next mhh_run_process call of this process (one process per GPU).
static int g_shard_rank = 0, g_shard_world = 1;
static MultiH::AllGatherFn g_shard_fn = nullptr;
static MultiH::StreamAllGatherFn g_shard_stream_fn = nullptr;
static void* g_shard_ctx = nullptr;
extern "C" __attribute__((visibility("default")))
void mhh_set_sharding(int rank, int world, MultiH::AllGatherFn fn, void* ctx)
{
    g_shard_rank = rank; g_shard_world = world; g_shard_fn = fn; g_shard_stream_fn = nullptr; g_shard_ctx = ctx;
}

// Stream-ordered transport (RCCL: host/rccl_transport.cpp) for the next mhh_run_process calls; fn = NULL returns to
// whatever mhh_set_sharding set.  world == 1 is allowed: the whole sharded protocol on a one-rank communicator.
extern "C" __attribute__((visibility("default")))
void mhh_set_sharding_stream(int rank, int world, MultiH::StreamAllGatherFn fn, void* ctx)
{
    g_shard_rank = rank; g_shard_world = world; g_shard_stream_fn = fn; g_shard_fn = nullptr; g_shard_ctx = ctx;
}

static int g_device = 0;
extern "C" __attribute__((visibility("default")))
void mhh_set_device(int device) { g_device = device; }

// Neighbourhood of the next mhh_run_process calls: radius > 0 selects the complete radius list, else k > 0 the k nearest
// hits within 1 / locality; both 0: the class default (k = 16).
static int g_knn = 0, g_post_filter = 1;
static double g_radius = 0.0;
extern "C" __attribute__((visibility("default")))
void mhh_set_post_filter(int on) { g_post_filter = on; }
extern "C" __attribute__((visibility("default")))
void mhh_set_neighbourhood(int knn_k, double radius) { g_knn = knn_k; g_radius = radius; }
// MultiH::SetNeighbourApprox for the next mhh_run_process calls (trees <= 0 switches it off again)
static int g_approx_trees = 0, g_approx_checks = 32;
static unsigned long long g_approx_seed = 0;
extern "C" __attribute__((visibility("default")))
void mhh_set_neighbourhood_approx(int trees, int checks, unsigned long long seed) { g_approx_trees = trees; g_approx_checks = checks; g_approx_seed = seed; }
// bound on the hits of the complete radius list (MultiH::SetNeighbourRadius' max_hits); <= 0 restores the default
static long long g_max_hits = 0;
extern "C" __attribute__((visibility("default")))
void mhh_set_neighbour_max_hits(long long max_hits) { g_max_hits = max_hits; }
// caller-supplied directed hit lists (MultiH::SetNeighbours) for the next mhh_run_process calls, CSR; n <= 0 clears them.
// For tools/neighbourhood_sweep.py: which neighbourhood rule reproduces the reference's recorded barrsmith result.
static std::vector<std::vector<int>> g_hits;
extern "C" __attribute__((visibility("default")))
void mhh_set_neighbour_hits(const int* rowptr, const int* col, int n)
{
    g_hits.clear();
    if (n <= 0 || !rowptr || !col) return;
    g_hits.resize(n);
    for (int i = 0; i < n; ++i) g_hits[i].assign(col + rowptr[i], col + rowptr[i + 1]);
}
// MultiH::SetProposalRefit for the next mhh_run_process calls (default 1)
static int g_proposal_refit = 1;
extern "C" __attribute__((visibility("default")))
void mhh_set_proposal_refit(int on) { g_proposal_refit = on; }
// schedule knobs (mh_set_tuning) for the engines of the next mhh_run_process calls; key < 0 clears the list
static std::vector<std::pair<int, int>> g_tuning;
extern "C" __attribute__((visibility("default")))
void mhh_set_engine_tuning(int key, int value) { if (key < 0) g_tuning.clear(); else g_tuning.emplace_back(key, value); }

// LabelingSteps the last mhh_run_process ran (MultiH::GetLabelingStepsRun)
static int g_labeling_steps = 0;
extern "C" __attribute__((visibility("default")))
int mhh_get_labeling_steps() { return g_labeling_steps; }
// MultiH::SetFundamentalMetric for the next mhh_run_process calls (< 0: the class default), and the stage table of the last one
static int g_fund_metric = -1;
static int g_front_stages[4] = { 0, 0, 0, 0 };
extern "C" __attribute__((visibility("default")))
void mhh_set_fundamental_metric(int metric) { g_fund_metric = metric; }
extern "C" __attribute__((visibility("default")))
void mhh_get_front_stages(int out[4]) { for (int i = 0; i < 4; ++i) out[i] = g_front_stages[i]; }
// multih::FilterCorrespondencesByEpipolarGeometry on plain arrays: mask (n flags) out; returns the number kept, -1 on failure
extern "C" __attribute__((visibility("default")))
int mhh_filter_correspondences(const double* src_xy, const double* dst_xy, int n, double threshold, unsigned long long seed,
                               int hypotheses, int metric, int device, unsigned char* mask)
{
    std::vector<cv::Point2d> s(n), d(n);
    std::vector<cv::Mat> a(n);
    for (int i = 0; i < n; ++i) {
        s[i] = cv::Point2d(src_xy[2 * i], src_xy[2 * i + 1]);
        d[i] = cv::Point2d(dst_xy[2 * i], dst_xy[2 * i + 1]);
    }
    std::vector<unsigned char> m;
    if (!multih::FilterCorrespondencesByEpipolarGeometry(s, d, a, threshold, seed, hypotheses, metric, device, &m)) return -1;
    for (int i = 0; i < n; ++i) mask[i] = m[i];
    return (int)s.size();
}

// multih::CompatibilityCheck with the trials' order statistics from an engine (mh_compat_trial_stats), as Process() runs
// it, returning the per-cluster median-of-medians too; -1 if the engine fails.  For the GPU tests.
static int g_compat_fits_on_engine = 1;           // mhh_compatibility_medians_on_engine: 1 = the fits on the device too (as Process() runs it since r06), 0 = on the host
extern "C" __attribute__((visibility("default")))
void mhh_set_compat_fits_on_engine(int on) { g_compat_fits_on_engine = on; }
extern "C" __attribute__((visibility("default")))
int mhh_compatibility_medians_on_engine(mh_engine* engine, const double* src_xy, const double* dst_xy, int n, int* labels, double* H,
                                        int nh, const double* F, double sqr_thr, int min_inliers, unsigned long long seed,
                                        double* medians)
{
    multih::CompatStatsFn fn = [engine, F](const double* pts, const int* begin, int clusters, const int* tri, const double* Ht,
                                           const unsigned char* ok, int trials, double* out) {
        if (!Ht) return mh_compat_trial_stats_fit(engine, pts, begin, clusters, tri, F, trials, out, nullptr, nullptr) == MH_OK;
        return mh_compat_trial_stats(engine, pts, begin, clusters, tri, Ht, ok, trials, out) == MH_OK;
    };
    bool failed = false;
    const int kept = multih::CompatibilityCheck(src_xy, dst_xy, n, labels, H, nh, F, sqr_thr, min_inliers, seed, medians, &fn, &failed,
                                                g_compat_fits_on_engine != 0);
    return failed ? -1 : kept;
}

// ---- C hook for the GPU-side integration test (ctypes; plain arrays in/out) ----------------
extern "C" __attribute__((visibility("default")))
int mhh_run_process(const double* src_xy, const double* dst_xy, const double* aff, int n,
                    const double* F, const double* e2, double thr_F, double thr_H, double locality,
                    double lambda, int min_inliers, unsigned long long seed, int hypotheses,
                    int max_models, int fixed_iterations, const double* init_H, int n_init,
                    int* labels_out, double* H_out, int max_H, int* iterations, double* energy,
                    double* loop_seconds, int iter_hypotheses, int iter_max_new)
{
    std::vector<cv::Point2d> s(n), d(n);
    std::vector<cv::Mat> a(n);
    // r06: the affinities as NON-owning 2 x 2 headers over one copy of the caller's array (cv::Mat(rows, cols, type, data): the
    // same with the real library) — 50 000 matrices that each own a 32-byte allocation cost 5 ms to build and to copy, a third
    // of what this hook reported as "Process()" at configs[4].  The copy outlives the call below.
    std::vector<double> aff_copy(aff, aff + 4 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        s[i] = cv::Point2d(src_xy[2 * i], src_xy[2 * i + 1]);
        d[i] = cv::Point2d(dst_xy[2 * i], dst_xy[2 * i + 1]);
        a[i] = cv::Mat(2, 2, CV_64F, aff_copy.data() + 4 * (size_t)i);
    }
    MultiH mh(thr_F, thr_H, locality, lambda, min_inliers);
    if (F && e2) mh.SetEpipolarGeometry(F, e2);
    mh.SetProposal(seed, hypotheses, max_models);
    mh.SetFixedIterations(fixed_iterations);
    mh.SetIterativeProposal(iter_hypotheses, iter_max_new < 0 ? 4 : iter_max_new);
    if (g_shard_stream_fn) mh.SetShardingStream(g_shard_rank, g_shard_world, g_shard_stream_fn, g_shard_ctx);
    else mh.SetSharding(g_shard_rank, g_shard_world, g_shard_fn, g_shard_ctx);
    mh.SetDevice(g_device);
    mh.SetCompatibilityCheck(g_post_filter != 0);
    mh.SetProposalRefit(g_proposal_refit != 0);
    if (g_fund_metric >= 0) mh.SetFundamentalMetric(g_fund_metric);
    for (const auto& kv : g_tuning) mh.SetEngineTuning(kv.first, kv.second);
    if (g_radius > 0.0 && g_max_hits > 0) { mh.SetNeighbourRadius(g_radius, g_max_hits); if (g_knn > 0) mh.SetFallbackK(g_knn); }
    else if (g_radius > 0.0) mh.SetNeighbourRadius(g_radius);
    else if (g_knn > 0) mh.SetNeighbourK(g_knn);
    if (g_approx_trees > 0) mh.SetNeighbourApprox(g_approx_trees, g_approx_checks, g_approx_seed);
    if (!g_hits.empty() && (int)g_hits.size() == n) mh.SetNeighbours(g_hits);
    if (iter_max_new < 0) mh.SetInitialisation(MultiH::INIT_STABLE_SETS);      // test hook: negative = reference-style init
    if (init_H && n_init > 0) {
        std::vector<cv::Mat> hs;
        for (int i = 0; i < n_init; ++i) hs.push_back(MatFrom9(init_H + 9 * (size_t)i));
        mh.SetInitialHomographies(hs);
    }
    if (!mh.Process(s, d, a)) return -1;
    {
        const MultiH::FrontStages st = mh.GetFrontStages();
        g_front_stages[0] = st.input; g_front_stages[1] = st.in_ransac_mask; g_front_stages[2] = st.triangulated; g_front_stages[3] = st.affine_consistent;
    }
    std::vector<int> lab;
    mh.GetLabels(lab);
    for (size_t i = 0; i < lab.size() && i < (size_t)n; ++i) labels_out[i] = lab[i];
    const int k = mh.GetClusterNumber();
    for (int i = 1; i <= k && i <= max_H; ++i) {
        cv::Mat h = mh.GetHomography(i);
        for (int q = 0; q < 9; ++q) H_out[9 * (size_t)(i - 1) + q] = reinterpret_cast<double*>(h.data)[q];
    }
    if (iterations) *iterations = mh.GetIterationNumber();
    if (energy) *energy = mh.GetEnergy();
    if (loop_seconds) *loop_seconds = mh.GetLastLoopSeconds();
    g_labeling_steps = mh.GetLabelingStepsRun();
    return k;
}
