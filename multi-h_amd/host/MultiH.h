// MultiH.h — host-side mirror of the reference's class `MultiH`
// (M/MultiH.h:20-149, M/ = /root/reference/MultiH/MultiH/) implemented over the
// C ABI of the gfx950 engine (include/multih_hip.h).  Same class name, same
// public signatures, same defaults, same label / index conventions, same error
// behaviour (Process() returns false and writes "Error: Features are not set!"
// to stderr, M/MultiH.cpp:44-50), so `ApplyMultiH` (M/main.cpp:232-311) can
// construct it, call Process and read the getters unchanged.
//
// Scope: Process() follows the reference's Process() (M/MultiH.cpp:42-98) stage by stage with every heavy
// stage on the GPU.  The front half (GetFundamentalMatrixAndRefineData, M/MultiH.cpp:770-848) has two routes: the
// caller hands over F and the epipole (SetEpipolarGeometry — e.g. the reference's own OpenCV code; the points are
// then taken as already refined), or the engine estimates them itself (8-point RANSAC with Sampson scoring and a
// least-squares refit, then the per-correspondence Hartley-Sturm correction, affine-consistency filter and optimal
// affinity of :807-838).  Initial models: SetInitialHomographies, the reference's stable point sets
// (INIT_STABLE_SETS: per-point HAF homographies -> mean shift -> 3-point fits, :604-717), or — the default — random
// minimal samples with the batched 4-point DLT (north_star).  Neighbour hits can be supplied (SetNeighbours) or are
// built on the GPU.
#pragma once

#include <cstdint>
#include <utility>
#include <vector>

#include "cv_shim.h"

#define DEFAULT_THRESHOLD_FUNDAMENTAL_MATRIX 3.0    // M/MultiH.h:7
#define DEFAULT_THRESHOLD_HOMOGRAPHY 2.5            // :8
#define DEFAULT_LOCALITY 0.002                      // :9
#define DEFAULT_LAMBDA 0.5                          // :10
#define DEFAULT_AFFINE_THRESHOLD 1.0                // :12
#define DEFAULT_LINENESS_THRESHOLD 0.005            // :13
#define MAX_ITERATION_NUMBER 500                    // :14
#define CONVERGENCE_THRESHOLD 1e-5                  // :15

struct mh_engine;

class MultiH {
public:
    MultiH(double _thr_fund_mat = DEFAULT_THRESHOLD_FUNDAMENTAL_MATRIX,
           double _thr_hom = DEFAULT_THRESHOLD_HOMOGRAPHY, double _locality = DEFAULT_LOCALITY,
           double _lambda = DEFAULT_LAMBDA, int _minimum_inlier_number = 0);
    ~MultiH();
    void Release();

    // M/MultiH.h:58-59.  Arguments by value, copied into *_original (M/MultiH.cpp:32-40).
    bool Process(std::vector<cv::Point2d> _srcPoints, std::vector<cv::Point2d> _dstPoints,
                 std::vector<cv::Mat> _affines);
    bool Process();

    // M/MultiH.h:61-75
    int GetLabel(int idx) { return labeling[idx]; }
    void GetLabels(std::vector<int>& _labeling) { _labeling = labeling; }
    void GetSourcePoints(std::vector<cv::Point2d>& _src_points) { _src_points = src_points; }
    // The reference returns the SOURCE points here (M/MultiH.h:64, SURVEY A-9); result files
    // written by the harness therefore carry x2,y2 = x1,y1.  Reproduced for drop-in parity.
    void GetDestinationPoints(std::vector<cv::Point2d>& _dst_points) { _dst_points = src_points; }
    void GetAffinities(std::vector<cv::Mat>& _affinities) { _affinities = affinities; }
    int GetPointNumber() { return static_cast<int>(labeling.size()); }
    int GetClusterNumber() { return static_cast<int>(cluster_homographies.size()); }
    int GetIterationNumber() { return final_iteration_number; }
    cv::Mat GetHomography(int idx) { return cluster_homographies[idx - 1]; }   // 1-based, :69
    double GetEnergy() { return final_energy; }
    double GetHomographyThreshold() { return threshold_homography; }
    // M/MultiH.h:71, M/MultiH.cpp:314-350: a filled circle per labelled correspondence, colour by plane.
    void DrawClusters(cv::Mat& img1, cv::Mat& img2, int size);
    // Post-filter of the reference (M/MultiH.cpp:100-222): multih::CompatibilityCheck with the trials' order statistics
    // from the engine (mh_compat_trial_stats).
    void HomographyCompatibilityCheck();

    // ---- extension points: outputs of the reference's OpenCV front half ----
    // fundamental_matrix (row-major) and epipole_2 = (x, y, 1) (M/MultiH.cpp:775-799).
    void SetEpipolarGeometry(const double F[9], const double e2[2]);
    // `neighbours` (M/MultiH.cpp:252-253): per query the hit indices (self allowed).
    void SetNeighbours(const std::vector<std::vector<int>>& hits);
    // Without SetNeighbours the hits are built on the GPU in the reference's float32 (x1,y1,x2,y2) space.  The reference
    // asks FLANN for every correspondence within 1 / locality_lambda pixels (radiusMatch, :252-253), but with FLANN's
    // default search (4 randomised KD-trees, 32 checks) the answer is a few dozen approximate nearest neighbours
    // inside that ball, never the ball itself — which at the harness defaults holds hundreds of points, enough for the
    // Potts term to swamp the data term (on the reference's own barrsmith pair the complete list leaves no plane
    // standing).  Default here: the k = 16 EXACT nearest hits that lie within the reference's radius.
    void SetNeighbourK(int k) { neighbour_mode = NEIGHBOURS_KNN; knn = k; }
    // The complete radius list instead (every hit within `radius`, exact), bounded by `max_hits` because it grows like
    // N^2: beyond the bound the k nearest hits within the radius are used and a line says so.
    void SetNeighbourRadius(double radius, long long max_hits = 1ll << 28) { neighbour_mode = NEIGHBOURS_RADIUS; neighbour_radius = radius; neighbour_max_hits = max_hits; }
    // r05: the reference's neighbourhood as FLANN's DEFAULT search answers it — `trees` randomised KD-trees (4), best-bin-first
    // with `checks` examined points per query (32), the hits among them inside 1 / locality_lambda — re-enacted on the host
    // with the engine's counter RNG (approx_neighbours.cpp; OpenCV / FLANN are outside /root/reference: restated from the
    // published algorithm, parity unpinned).  About 29 one-way hits per correspondence, 70 % of the exact 31 nearest.
    void SetNeighbourApprox(int trees = 4, int checks = 32, uint64_t seed = 0x464c414e4eull)
    { neighbour_mode = NEIGHBOURS_APPROX; approx_trees = trees; approx_checks = checks; approx_seed = seed; }
    // k of that fallback (default 16) without leaving the radius mode
    void SetFallbackK(int k) { knn = k; }
    // Initial cluster_homographies (what EstablishStablePointSets hands to the loop).
    void SetInitialHomographies(const std::vector<cv::Mat>& Hs);
    // How the initial models are made when SetInitialHomographies was not called:
    //   INIT_DLT          random minimal samples -> batched 4-point DLT -> greedy selection (north_star)
    //   INIT_STABLE_SETS  the reference's own route: per-point HAF homographies
    //                     (ComputeLocalHomographies, M/MultiH.cpp:696-717) -> 10-D mean shift ->
    //                     one 3-point LSQ homography per cluster of >= 3 points
    //                     (EstablishStablePointSets, :604-694), all heavy steps on the GPU.
    enum { INIT_DLT = 0, INIT_STABLE_SETS = 1 };
    void SetInitialisation(int mode) { init_mode = mode; }
    // Propose step used when no initial models are given: `hypotheses` random 4-tuples ->
    // DLT -> greedy selection of at most `max_models` models with >= max(min inliers, 8).
    void SetProposal(uint64_t seed, int hypotheses, int max_models);
    // A CAP on the loop's iterations (north_star configs[4]: "20 propose-expand iterations"): with n > 0 the loop also stops
    // after its n-th LabelingStep.  The reference's own stop rule (:295: no change and the same energy, or more than ten
    // unchanged iterations) stays in force — a loop that has converged after three iterations runs three, not n.  0 = the
    // reference's rule alone.  How many LabelingSteps the last Process() ran: GetLabelingStepsRun() (GetIterationNumber()
    // is the reference's iteration_number - 1, :311, i.e. one less when the loop ended behind a LabelingStep).
    void SetFixedIterations(int n) { fixed_iterations = n; }
    int GetLabelingStepsRun() const { return labeling_steps_run; }
    // PEARL-style re-proposal (north_star "propose-expand iterations"; the reference proposes only
    // once, before the loop): every iteration samples `hypotheses` fresh DLT hypotheses, scores them
    // on the points currently labelled outlier and appends at most `max_new` models that gather
    // >= max(min inliers, 8) of them.  0 hypotheses (default) keeps the reference's behaviour.
    void SetIterativeProposal(int hypotheses, int max_new) { iter_hypotheses = hypotheses; iter_max_new = max_new; }
    // r05: every winner of the greedy selection is refitted to the correspondences it explains — the loop's per-label HAF
    // least squares with one label (M/MultiH.cpp:913-989) — before it claims them, and the refit takes its place when it
    // explains at least as many (mh_set_tuning key 30).  A hypothesis fitted to four matches explains 60-70 % of its plane;
    // what it leaves behind used to feed models that sit between two planes.  Default on.
    void SetProposalRefit(bool on) { proposal_refit = on; }
    // Multi-GPU propose stage (BASELINE configs[3] and [4]): one process per GPU, every rank holds all
    // correspondences and owns a contiguous shard of each hypothesis batch (the hypotheses are a pure function of
    // (seed, counter), so the union over ranks is the single-GPU batch).  In the first greedy round the ranks all-gather
    // their int32 inlier scores — north_star's exchange —, in every round one 88-byte record each (best score and its
    // position in the batch, that hypothesis' H), and run the same first-maximum selection on the device.  Labeling and re-estimation run replicated and deterministic, so the ranks stay
    // identical without a broadcast.  `fn` is the transport: it receives DEVICE pointers (the engine's resident score
    // buffer on the send side), so RCCL (`ncclAllGather`, or torch.distributed's all_gather_into_tensor as in
    // multi-h_amd/sharding.py) works on them in place: send `bytes_per_rank` bytes, receive world * bytes_per_rank in
    // rank order, return 0 once the result is complete in `recv_dev`.  The engine's stream is idle during the call.
    typedef int (*AllGatherFn)(void* ctx, const void* send_dev, void* recv_dev, unsigned long long bytes_per_rank);
    void SetSharding(int rank, int world, AllGatherFn fn, void* ctx);
    // The same with a STREAM-ORDERED transport: `fn` enqueues the all-gather on `hip_stream` (the engine's) and returns at
    // once — RCCL's ncclAllGather; include/multih_rccl.h + host/rccl_transport.cpp provide it (mhr_allgather) together with the communicator
    // bootstrap.  No host synchronisation and no Python in the exchange.  world == 1 with a transport runs the whole
    // sharded protocol on a one-rank communicator.
    typedef int (*StreamAllGatherFn)(void* ctx, const void* send_dev, void* recv_dev, unsigned long long bytes_per_rank,
                                     void* hip_stream);
    void SetShardingStream(int rank, int world, StreamAllGatherFn fn, void* ctx);
    // Number of 8-point hypotheses of the GPU F estimation used when SetEpipolarGeometry was not called.
    void SetFundamentalHypotheses(int n) { fundamental_hypotheses = n; }
    // r06: what the F estimation compares with threshold_fundamental_matrix — FUND_EPIPOLAR_MAX: the squared distance of a
    // point to the epipolar line of its partner, the larger of the two images, which is what the reference's
    // cv::findFundamentalMat(CV_FM_RANSAC, thr) thresholds (M/MultiH.cpp:775; OpenCV 3.1.0, restated); FUND_SAMPSON: the
    // Sampson distance (at most half of it at equal F — the build's definition until r05: at 2.6 px it let through what the
    // reference's call rejects from 1.84 px on).  include/multih_hip.h, mh_set_fundamental_metric.
    enum { FUND_SAMPSON = 0, FUND_EPIPOLAR_MAX = 1 };
    void SetFundamentalMetric(int m) { fundamental_metric = m; }
    // r06: how many correspondences each stage of GetFundamentalMatrixAndRefineData (M/MultiH.cpp:770-848) let through in
    // the last Process(): the input, the RANSAC mask (:809), OptimalTriangulation (:815-817), distanceError <= 1 (:826).
    // All equal to the input when the caller supplied F (SetEpipolarGeometry): the points are then taken as they are.
    struct FrontStages { int input = 0, in_ransac_mask = 0, triangulated = 0, affine_consistent = 0; };
    FrontStages GetFrontStages() const { return front_stages; }
    // The post-filter of Process() (HomographyCompatibilityCheck, M/MultiH.cpp:78-86) can be switched off to look at
    // what the merge <-> label loop itself produced (parity tests against the oracle of that loop).
    void SetCompatibilityCheck(bool on) { run_compatibility_check = on; }
    void SetDevice(int d) { device = d; }
    // schedule knob of the engine (mh_set_tuning; results never depend on it), applied when the engine is created
    void SetEngineTuning(int key, int value) { engine_tuning.emplace_back(key, value); }
    void SetVerbose(bool v) { log_to_console = v; }
    double GetLastLoopSeconds() const { return loop_seconds; }

protected:
    std::vector<cv::Point2d> src_points_original, dst_points_original;   // M/MultiH.h:78
    std::vector<cv::Point2d> src_points, dst_points;                     // :79
    std::vector<cv::Mat> affinities_original, affinities;                // :80
    double fundamental_matrix[9];
    double epipole_2[2];
    bool have_epipolar = false;
    bool log_to_console = false;
    bool degenerate_case = false;
    double final_energy = 0.0;
    std::vector<std::vector<int>> neighbours;                            // :86 (trainIdx only)
    std::vector<cv::Mat> cluster_homographies;                           // :87
    std::vector<int> labeling;                                           // :88
    int minimum_inlier_number;
    double threshold_fundamental_matrix;
    double threshold_homography, sqr_threshold_homography;
    double locality_lambda, energy_lambda;
    double affine_threshold, straightness_threshold;
    int final_iteration_number = 0;

    // engine state
    mh_engine* engine = nullptr;
    int device = 0;
    std::vector<std::pair<int, int>> engine_tuning;
    enum { NEIGHBOURS_KNN = 0, NEIGHBOURS_RADIUS = 1, NEIGHBOURS_APPROX = 2 };
    int approx_trees = 4, approx_checks = 32;
    uint64_t approx_seed = 0;
    int neighbour_mode = NEIGHBOURS_KNN;
    int knn = 16;
    double neighbour_radius = 0.0;
    long long neighbour_max_hits = 1ll << 28;
    uint64_t proposal_seed = 1234;
    int proposal_hypotheses = 10000;
    int proposal_max_models = 32;
    bool proposal_refit = true;
    int fixed_iterations = 0;
    int iter_hypotheses = 0, iter_max_new = 4;
    int fundamental_hypotheses = 4000;
    int fundamental_metric = FUND_EPIPOLAR_MAX;
    FrontStages front_stages;
    int init_mode = INIT_DLT;
    bool run_compatibility_check = true;
    bool post_filter_failed = false;             // the engine's part of HomographyCompatibilityCheck returned an error
    uint64_t merge_rng_counter = 0;
    double loop_seconds = 0.0;
    int labeling_steps_run = 0;
    std::vector<cv::Mat> initial_homographies;
    int shard_rank = 0, shard_world = 1;
    AllGatherFn shard_allgather = nullptr;
    StreamAllGatherFn shard_stream_allgather = nullptr;
    void* shard_ctx = nullptr;

    bool EnsureEngine();
    bool UploadModels();
    bool DownloadModels(int count);
    bool ProposeInitialModels();          // north_star propose: DLT batch + greedy selection
    bool EstablishStablePointSets();      // M/MultiH.cpp:604-694 (with ComputeLocalHomographies :696-717)
    bool ProposeModels(uint64_t seed, long long first, int hypotheses, int max_models,
                       std::vector<unsigned char>& mask);
    void ClusterMergingAndLabeling();     // M/MultiH.cpp:224-312
    bool MergingStep(bool& changed);      // :352-471
    bool LabelingStep(double& energy, bool changed);   // :513-602
    void ComputeInliersOfHomography(int idx);          // :743-768
    void HandleDegenerateCase();                       // :719-741 (single best DLT model)
};

namespace multih {
// r06: the load-time filter of the reference's CALLER — LoadPointsFromFile, M/main.cpp:399-409:
//     findFundamentalMat(srcPoints, dstPoints, CV_FM_RANSAC, 2.0, 0.99, mask);   then every row with mask[i] == 0 is erased
// — through the engine's own estimator (mh_estimate_fundamental: `hypotheses` normalised 8-point fits from counter-RNG
// 8-tuples, the best-supported one refitted twice to its inliers; cv::findFundamentalMat itself — 7-point samples, at most
// 1 000 of them, its own RNG, no refit — is outside /root/reference and not reproduced).  `metric` as
// MultiH::SetFundamentalMetric.  Erases the rejected rows from all three vectors (order kept, as the reference's backward
// erase loop keeps it) and returns false — leaving them untouched — when there are fewer than 8 rows, the sizes differ or the
// engine fails.  mask_out (nullable): one flag per INPUT row.
bool FilterCorrespondencesByEpipolarGeometry(std::vector<cv::Point2d>& srcPoints, std::vector<cv::Point2d>& dstPoints,
                                             std::vector<cv::Mat>& affines, double threshold = 2.0,
                                             uint64_t seed = 1234, int hypotheses = 4000,
                                             int metric = MultiH::FUND_EPIPOLAR_MAX, int device = 0,
                                             std::vector<unsigned char>* mask_out = nullptr);
}
