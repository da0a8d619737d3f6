/* multih_hip.h — C ABI of the MI355X-native Multi-H hot-path engine.
 *
 * This is the drop-in boundary: plain pointers and sizes, no
 * C++/torch types, status codes instead of exceptions.  The reference has no
 * FFI — its boundary is the C++ class `MultiH` (M/MultiH.h:20-149, "M/" =
 * /root/reference/MultiH/MultiH/).  `multi-h_amd/host/MultiH.{h,cpp}` re-creates
 * that class on top of these entry points; INTEGRATION.md shows the binding a
 * reference maintainer would add.  Each entry point cites the reference code
 * whose arithmetic it replaces.
 *
 * Conventions
 *   - all pointers are HOST pointers unless a name ends in `_dev`;
 *   - homographies are 9 doubles, row-major (cv::Mat 3x3 CV_64F `.data`);
 *   - labels at this boundary: -1 = outlier, 0..Nh-1 = plane (M/MultiH.cpp:547-568);
 *   - every function returns MH_OK (0) or a negative MH_ERR_*; mh_last_error()
 *     gives the message of the calling thread's last failure;
 *   - an engine is used from one thread at a time (as the reference class is);
 *   - the engine fails loudly (MH_ERR_NO_DEVICE) when no gfx950 device is
 *     usable: there is NO CPU fallback anywhere behind this header.
 */
#ifndef MULTIH_HIP_H
#define MULTIH_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MH_ABI_VERSION 2

/* exported even when the library is built with -fvisibility=hidden */
#if defined(__GNUC__) || defined(__clang__)
#define MH_API __attribute__((visibility("default")))
#else
#define MH_API
#endif

enum {
    MH_OK = 0,
    MH_ERR_NO_DEVICE = -1,   /* no HIP device / wrong arch */
    MH_ERR_INVALID = -2,     /* bad argument or call order */
    MH_ERR_HIP = -3,         /* a HIP runtime call failed */
    MH_ERR_NOT_SET = -4,     /* required input (points, F, neighbours, models) missing */
    MH_ERR_OVERFLOW = -5     /* an int32 energy of the reference's type would overflow */
};

typedef struct mh_engine mh_engine;

/* ---- library ---------------------------------------------------------- */
MH_API int mh_abi_version(void);
MH_API const char* mh_last_error(void);
MH_API int mh_device_count(void);              /* number of visible HIP devices (0 if none) */

/* ---- lifetime / parameters -------------------------------------------- */
/* MultiH::MultiH, M/MultiH.h:49-53 + M/MultiH.cpp:10-21.  One HIP stream per engine. */
MH_API int mh_create(mh_engine** out, int device);
MH_API void mh_destroy(mh_engine* e);
/* thr_F, thr_H, locality, lambda, min_inliers: the ctor arguments (defaults 3.0, 2.5, 0.002, 0.5, 0). */
MH_API int mh_set_params(mh_engine* e, double thr_fund_mat, double thr_hom, double locality,
                  double lambda, int min_inliers);
/* external != 0: launch on the caller's hipStream_t `hip_stream` (e.g. torch's current stream;
 * NULL is the HIP default stream).  external == 0: back to the engine's own stream. */
MH_API int mh_set_stream(mh_engine* e, void* hip_stream, int external);
MH_API int mh_synchronize(mh_engine* e);

/* ---- inputs ------------------------------------------------------------ */
/* Correspondences as the reference holds them after filtering: src_points, dst_points
 * (vector<cv::Point2d> -> n x 2 doubles each) and affinities (2x2 CV_64F -> n x 4 doubles,
 * order a11 a12 a21 a22, M/MultiH.cpp:928-931).  affines may be NULL if re-estimation is
 * not used.  Stored in HBM as struct-of-arrays. */
MH_API int mh_set_correspondences(mh_engine* e, const double* src_xy, const double* dst_xy,
                           const double* affines, int n);
/* fundamental_matrix (row-major 9) and epipole_2 (x, y with third coordinate 1),
 * the members GetFundamentalMatrixAndRefineData computes (M/MultiH.cpp:775-799). */
MH_API int mh_set_epipolar(mh_engine* e, const double F[9], const double e2[2]);
/* `neighbours` of ClusterMergingAndLabeling (M/MultiH.cpp:252-253) as a DIRECTED hit list in
 * CSR form: row i lists the trainIdx hits of query i (self hits allowed; skipped as in
 * M/MultiH.cpp:537).  Multiplicity semantics of setNeighbors are reproduced (SURVEY A-2). */
MH_API int mh_set_neighbors_csr(mh_engine* e, const int* rowptr, const int* col, int n);
/* Exact k-NN hit list built on the GPU in float32 (x1,y1,x2,y2) space — the engine's own
 * replacement for the FLANN radius search (deviation documented). */
MH_API int mh_build_neighbors_knn(mh_engine* e, int k);
/* The same k nearest hits, but only those within `radius` of the query (radius <= 0: no cut): the reference's
 * radius (1/locality_lambda) with the list bounded the way FLANN's default search bounds it in practice
 * (32 checks: a few dozen approximate nearest neighbours, never the full ball). */
MH_API int mh_build_neighbors_knn_radius(mh_engine* e, int k, double radius);
/* The reference's neighbourhood rule itself (M/MultiH.cpp:252-253, radiusMatch with maxDistance =
 * 1/locality_lambda), answered exactly instead of by FLANN's randomised KD-trees: query i hits
 * every j (itself included) whose float32 squared distance in (x1,y1,x2,y2) is <= radius^2.
 * The hit count grows like n^2 * radius^4, so `max_hits` (0 = 2^31-1) bounds it: beyond the bound
 * the call fails with MH_ERR_OVERFLOW and leaves the current graph untouched.  hits_out
 * (nullable) receives the number of directed hits found. */
MH_API int mh_build_neighbors_radius(mh_engine* e, double radius, long long max_hits, long long* hits_out);
/* Copy the symmetric weighted graph the engine derived (rowptr n+1, col/w nnz).  Pass NULLs to
 * query nnz only.  The graph is built and kept on the device (csrc/graph.hip); the host copy is fetched by the
 * first call after a build. */
MH_API int mh_get_sym_graph(mh_engine* e, int* rowptr, int* col, int* w, int* nnz);

/* ---- epipolar front half --------------------------- */
/* The reference estimates F inside Process() with cv::findFundamentalMat(RANSAC)
 * (M/MultiH.cpp:775) and takes the epipole from F F^T (:786-793).  These entry points are the
 * engine's own GPU version of that step (normalised 8-point hypotheses from counter-RNG 8-tuples,
 * Sampson-distance scoring, least-squares refit on the inliers); results are inputs for
 * mh_set_epipolar.  They do not touch the homography model set. */
/* What "distance to the epipolar geometry" means in mh_score_sampson / mh_refit_fundamental / mh_estimate_fundamental
 * (r06; part of the result's definition, so an entry point of its own and not an mh_set_tuning key):
 *   MH_FUND_SAMPSON       e^2 / (a^2 + b^2 + a'^2 + b'^2)            — the engine's default since r02
 *   MH_FUND_EPIPOLAR_MAX  max(e^2 / (a^2 + b^2), e^2 / (a'^2 + b'^2)) — the squared point-to-epipolar-line distance, the
 *                         larger of the two images: what cv::findFundamentalMat compares with its threshold (OpenCV 3.1.0
 *                         FMEstimatorCallback::computeError; the call sites are M/main.cpp:400 at 2.0 px and
 *                         M/MultiH.cpp:775 at threshold_fundamental_matrix).  At least twice the Sampson value at equal F (equal when both line normals have the same length).
 * Stays set until changed; mh_set_correspondences does not reset it. */
#define MH_FUND_SAMPSON 0
#define MH_FUND_EPIPOLAR_MAX 1
MH_API int mh_set_fundamental_metric(mh_engine* e, int metric);
MH_API int mh_propose_fund8(mh_engine* e, unsigned long long seed, long long first, int m);
MH_API int mh_get_fund_hypotheses(mh_engine* e, double* F /* m x 9 */, int* idx /* m x 8, nullable */);
MH_API int mh_score_sampson(mh_engine* e, double thr2, int* counts /* m */);
/* `iterations` rounds of {inliers of F (Sampson d < thr2) -> normalised LS 8-point -> rank 2}. */
MH_API int mh_refit_fundamental(mh_engine* e, const double F_in[9], double thr2, int iterations,
                         double F_out[9], unsigned char* inlier_mask /* n, nullable */, int* inliers);
/* propose(hypotheses) -> score -> best -> 2 refits -> epipole e2 = null(F^T), third coordinate 1. */
MH_API int mh_estimate_fundamental(mh_engine* e, unsigned long long seed, int hypotheses, double thr,
                            double F[9], double e2[2], unsigned char* inlier_mask /* n, nullable */,
                            int* inliers);

/* Epipoles of F with third coordinate 1: e1 = eigenvector of F^T F, e2 = eigenvector of F F^T with the
 * smallest eigenvalue (M/MultiH.cpp:786-799).  Host arithmetic only; e may be NULL. */
MH_API int mh_epipoles(mh_engine* e, const double F[9], double e1[2], double e2[2]);
/* Per-correspondence refinement of GetFundamentalMatrixAndRefineData (M/MultiH.cpp:807-838) on the
 * GPU: Hartley-Sturm correction (OptimalTriangulation :1116-1188), affine consistency filter
 * (distanceError > 1 drops the point, :824-827) and the optimal affinity (:1190-1223).
 * in_mask (n, nullable): only points with mask != 0 are processed (the F-RANSAC mask of :809).
 * keep (n): 1 where the point survives; refined (n x 8): x1 y1 x2 y2 a11 a12 a21 a22 of survivors. */
MH_API int mh_refine_correspondences(mh_engine* e, const double F[9], const double e1[2], const double e2[2],
                              const unsigned char* in_mask, unsigned char* keep, double* refined);
/* r06: at which stage of M/MultiH.cpp:807-838 each row of the LAST mh_refine_correspondences call left (n = its row count):
 * the stage table of the harness (rows after the RANSAC mask, after OptimalTriangulation, after distanceError > 1). */
#define MH_REFINE_KEPT 0
#define MH_REFINE_NOT_IN_MASK 1          /* :809  mask[i] == 0 */
#define MH_REFINE_TRIANGULATION 2        /* :815-817  OptimalTriangulation failed (optimum at infinity, :1170-1175) */
#define MH_REFINE_AFFINE_TEST 3          /* :826  distanceError > 1 (a NaN is dropped here too) */
MH_API int mh_get_refine_reasons(mh_engine* e, unsigned char* reason /* n */, int n);

/* ---- reference-style initialisation ------------- */
/* ComputeLocalHomographies (M/MultiH.cpp:696-717, GetHomographyHAF :850-911): one homography per
 * correspondence from its affinity, F and e2; H_out (n x 9, nullable).  feat_out (n x 10, nullable)
 * receives the feature vectors EstablishStablePointSets clusters (:617-644) with `locality` as the
 * coordinate weight. */
MH_API int mh_local_homographies(mh_engine* e, double locality, double* H_out, double* feat_out);
/* MeanShiftClustering<double>::Cluster (MeanShiftClustering.h:23-157) on n rows of d <= 16 doubles:
 * every climb (:62-123) runs on the GPU, the sequential seed/merge/vote logic (:52-56,:100-146) on
 * the host; seeds come from the counter RNG instead of rand(), 256 at a time from the rows unvisited
 * at that moment (their climbs share the launches; a seed visited by an earlier climb of its batch
 * is dropped — HISTORY.md 3.8).  modes: up to max_modes x d;
 * assign: per row the index of its mode (-1 if none); n_modes: number of modes found. */
MH_API int mh_mean_shift(mh_engine* e, const double* data, int n, int d, double band_width,
                  unsigned long long seed, double* modes, int max_modes, int* assign, int* n_modes);

/* ---- propose: hypothesis batch ----------------------------------------- */
/* Sample `m` 4-tuples with the counter RNG (seed, first..first+m-1) and solve the normalised
 * 4-point DLT for each on the GPU (north_star; stands where the reference calls
 * cv::findHomography: M/MultiH.cpp:725, M/MultipleHomographies.h:144,339).  The batch stays
 * resident in HBM as the engine's current model set. */
MH_API int mh_propose_dlt4(mh_engine* e, unsigned long long seed, long long first, int m);
/* Upload an explicit model set (m x 9) as the current model set. */
MH_API int mh_set_models(mh_engine* e, const double* H, int m);
MH_API int mh_get_models(mh_engine* e, double* H /* m x 9 */);
MH_API int mh_get_model_count(mh_engine* e, int* m);
MH_API int mh_get_samples(mh_engine* e, int* idx /* m x 4, valid after mh_propose_dlt4 */);

/* ---- score -------------------------------------------------------------- */
/* Residual definition used by mh_score / mh_residual_matrix / mh_get_residual_rows:
 *   MH_RESIDUAL_FORWARD   d2 = |H p1 - p2|^2 — the reference's formula (M/MultiH.cpp:434-441), default;
 *   MH_RESIDUAL_SYMMETRIC d2 = |H p1 - p2|^2 + |H^-1 p2 - p1|^2 — north_star's "symmetric transfer",
 *                         an extension with no reference counterpart (H^-1 = adjugate, same rounding
 *                         discipline; the definition is checked in exact rationals, tests/test_symmetric_exact.py).
 *                         Used by mh_score, mh_residual_matrix, mh_get_residual_rows and (r05) mh_select_greedy;
 *                         data cost / labeling always use the forward formula. */
enum { MH_RESIDUAL_FORWARD = 0, MH_RESIDUAL_SYMMETRIC = 1 };
MH_API int mh_set_residual_mode(mh_engine* e, int mode);
/* Inlier count of every current model over all points, forward transfer error, strict
 * d2 < thr2 (M/MultiH.cpp:430-443).  point_mask (n bytes, nullable) restricts the count to
 * points with mask != 0.  counts (m ints, nullable) receives a host copy; the counts also
 * stay resident (mh_device_buffer MH_BUF_COUNTS). */
MH_API int mh_score(mh_engine* e, double thr2, const unsigned char* point_mask, int* counts);
/* The N x M residual matrix of north_star, model-major: R[m*n + i] = d2(point i, model m),
 * written to HBM by one kernel that also produces the inlier counts.  R_host (nullable)
 * receives a host copy — leave NULL to keep the matrix on the device only. */
MH_API int mh_residual_matrix(mh_engine* e, double thr2, double* R_host, int* counts);
/* The s = 4 variant of the matrix: the int32 PEARL data cost (dataEnergy, M/MultiH.cpp:473-504, label =
 * model + 1) of every current model against every point, model-major C[m*n + i], written to HBM by one kernel that also
 * produces the inlier counts (strict d2 < thr_hom^2).  Uses the engine's lambda and thr_hom (mh_set_params).  C_host
 * (nullable) receives a host copy. */
MH_API int mh_cost_matrix(mh_engine* e, int* C_host, int* counts);
/* Copy rows [first, first+count) of the resident residual matrix (valid after
 * mh_residual_matrix) to the host, tightly packed count x n doubles. */
MH_API int mh_get_residual_rows(mh_engine* e, int first, int count, double* rows_host);
/* Label points with the single model idx where inlier (ComputeInliersOfHomography,
 * M/MultiH.cpp:743-768): labels[i] = label_value if d2 < thr2, else unchanged. */
MH_API int mh_inliers_of_model(mh_engine* e, int idx, double thr2, int label_value, int* labels /* in/out n */);
/* Same labelling for a homography that is NOT in the resident model set (row-major H[9]); the
 * model set is left untouched.  Used by the sharded propose stage, where the winning hypothesis
 * of a round may live on another rank. */
MH_API int mh_inliers_of_homography(mh_engine* e, const double* H, double thr2, int label_value, int* labels /* in/out n */);
/* Trial statistics of HomographyCompatibilityCheck (M/MultiH.cpp:128-196; host/merge_step.cpp ClusterMedian) for
 * `clusters` clusters of at least 19 points each.  pts_xyxy: the clusters' points back to back in cluster order, 4 doubles
 * each (x1 y1 x2 y2); cluster_begin[c] .. cluster_begin[c+1] (cluster_begin[0] = 0); per (cluster, trial): tri = the
 * positions inside the cluster of the 3 points the trial drew (:140-151), H = the trial's 3-point homography (:154,
 * row-major 9), ok = 0 when that fit failed (all distances then count as NaN -> 1e300).  stats_out: 8 doubles per
 * (cluster, trial): the order statistics of the N-3 squared transfer errors (:158-173) at ranks k-3 .. k+1, k = (N-3)/2,
 * then their three largest values ascending — all a trial's "median" (:175-176) can depend on, bit-identical to sorting
 * on the host.  Independent of the resident correspondences. */
MH_API int mh_compat_trial_stats(mh_engine* e, const double* pts_xyxy, const int* cluster_begin, int clusters, const int* tri,
                                 const double* H, const unsigned char* ok, int trials, double* stats_out);
/* r06: the same with the trials' 3-point homographies fitted ON THE DEVICE (GetHomography3PT without refinement, M/MultiH.cpp:154,
 * :995-1050) from F — until r05 the host fitted the 501 x clusters of them.  Bit-identical to the host's fits
 * (host/merge_step.cpp, Homography3PTLinear).  H_out (clusters x trials x 9) / ok_out (clusters x trials): nullable, the fits. */
MH_API int mh_compat_trial_stats_fit(mh_engine* e, const double* pts_xyxy, const int* cluster_begin, int clusters, const int* tri,
                              const double F[9], int trials, double* stats_out, double* H_out, unsigned char* ok_out);
/* ---- multi-GPU transport (one process per GPU, hypotheses sharded, correspondences replicated) ------
 * The reference is a single process; the one exchange north_star adds is an all-gather of per-model inlier scores.
 * The engine calls the transport with DEVICE pointers: send `bytes_per_rank` bytes, receive world * bytes_per_rank in
 * rank order.  Two kinds (give exactly one; both NULL with world = 1 clears the transport):
 *   stream_fn  enqueues the collective on `hip_stream` (the engine's stream) and returns at once — ncclAllGather of
 *              RCCL; multi-h_amd/host/rccl_transport.cpp is that function.  No host synchronisation, no Python.
 *   host_fn    is called with the engine's stream idle and returns once recv_dev is complete (the torch.distributed
 *              hook of multi-h_amd/sharding.py, used to rehearse several ranks on one GPU over gloo).
 * Rank r owns the hypotheses [r * base + min(r, rem), ...) of a batch of total_m (base = total_m / world, rem = total_m
 * % world, the first rem shards one longer) as its resident model set. */
typedef int (*mh_allgather_stream_fn)(void* ctx, const void* send_dev, void* recv_dev, unsigned long long bytes_per_rank,
                                      void* hip_stream);
typedef int (*mh_allgather_dev_fn)(void* ctx, const void* send_dev, void* recv_dev, unsigned long long bytes_per_rank);
MH_API int mh_set_transport(mh_engine* e, int rank, int world, mh_allgather_stream_fn stream_fn, mh_allgather_dev_fn host_fn,
                            void* ctx);
/* Greedy model selection over the resident hypothesis batch, on the device (csrc/select.hip): up to `max_models`
 * rounds of {score the candidates over the points still in the support mask, take the best — highest count, lowest
 * position in the WHOLE batch on ties —, stop if it has fewer than `need` inliers, take its inliers out of the mask}.
 * This is the sequential-RANSAC scheme of the dead M/MultipleHomographies.h:146-175 behind MultiH::ProposeModels.
 * point_mask (n bytes, nullable = all ones): in = points that may support a model, out = points no selected model
 * explains.  H_out: max_models x 9; counters_out / counts_out (nullable): position of each selected hypothesis in the
 * whole batch and its inlier count when selected.  Per round the host reads five control words from mapped memory; no
 * host<->device copy is issued inside the loop (mh_get_copy_stats).  Scores AND claims on the engine's residual mode
 * (r05: the symmetric transfer error too; until r04 the mode was refused).
 * Sharded batch (a transport is set; total_m = size of the whole batch, 0 = the resident set is the whole batch): in the
 * first round the ranks all-gather their int32 score vectors (north_star's exchange); in every round they all-gather
 * one 88-byte record each — {best score and its position, that hypothesis' H, an error word} — and pick the same
 * winner.  Outputs do not depend on the number of ranks.  A rank-local failure (the state of that rank's engine: wrong
 * shard size, a failing scoring launch) does not keep the rank out of the round's collectives: it offers
 * nothing, sets the record's error word, and after the exchange every rank leaves the loop — the failing one with its
 * own error, the others with MH_ERR_HIP "a rank reported an error".  Arguments (thr2, need, max_models, total_m) must be
 * the same on every rank. */
MH_API int mh_select_greedy(mh_engine* e, double thr2, int need, int max_models, unsigned char* point_mask,
                            double* H_out, long long* counters_out, int* counts_out, int* selected_out, long long total_m);
/* Pipelined propose: mh_prefetch_dlt4 prepares the batch (seed, first .. first+m-1) in one of the engine's spare model buffers
 * on a second stream, concurrently with whatever the main stream is doing (csrc/dlt4.hip); mh_adopt_prefetched makes the OLDEST
 * prepared batch the current model set — the main stream waits for the side stream's event, the host does not wait at all.
 * Up to TWO batches may be prepared ahead (a third mh_prefetch_dlt4 answers MH_ERR_INVALID until one is adopted).  With the
 * batch after next prepared too — adopt i, prefetch i+2, sweep i — the DLT a sweep waits for was dispatched a whole sweep
 * earlier and nothing is handed from stream to stream between two sweeps; with one batch ahead the sweep is held until the
 * second stream has reached the DLT's dispatch (a sweep that reaches the chip first leaves the DLT no room until it ends).
 * Same hypotheses, bit for bit, as mh_propose_dlt4.  mh_set_correspondences drops whatever is queued. */
MH_API int mh_prefetch_dlt4(mh_engine* e, unsigned long long seed, long long first, int m);
MH_API int mh_adopt_prefetched(mh_engine* e);
/* Best-supported model of the scored batch (highest resident inlier count, lowest position in the whole batch on ties) by
 * the engine's own arg-max kernel.  With a transport the ranks first all-gather their int32 score vectors (BASELINE
 * configs[3]; the gathered vector stays resident: MH_BUF_GATHERED_SCORES) and every rank finds the same winner; a rank
 * whose shard is empty (more ranks than hypotheses) takes part with nothing to offer.
 * best_index / best_count both NULL: ENQUEUE ONLY.  With a stream-ordered transport (or none) the all-gather, the arg-max
 * and the publication then run on a stream of their own behind an event of the sweep, OFF the main stream's critical
 * path: the next sweep starts at once and writes the engine's other counts buffer (after such a call MH_BUF_COUNTS is
 * undefined until the next scoring call).  A later call with outputs — before anything else has been scored — or
 * mh_synchronize completes it; the ranks' send buffer is the batch's own counts buffer (no padding kernel).  The
 * host-synchronised transport runs the same steps on the main stream.
 * Whether a call is a NEW exchange is decided from state that is the same on every rank (r05): the model-set generation
 * the last exchange belongs to and whether a scoring call has been made since — on an empty shard the scoring entry points
 * are no-ops that succeed and count.  The ranks make the same calls; a rank-local failure (a model set that is not the
 * rank's shard, an unscored batch) travels through the collective as an error marker: that rank returns its own error,
 * every other rank's next fetch fails with MH_ERR_HIP "a rank reported an error", none is left waiting. */
MH_API int mh_select_best(mh_engine* e, long long total_m, long long* best_index, int* best_count);
/* mh_score and mh_select_greedy decide most (point, model) pairs in FP32 with a rigorous error bound and only the pairs
 * within that bound of the threshold in FP64 (csrc/score32.hip; the counts are the FP64 formula's, bit for bit).  pairs:
 * pairs scored that way since the last reset; pairs_fp64: how many of them needed the FP64 formula. */
MH_API int mh_get_score_stats(mh_engine* e, long long* pairs, long long* pairs_fp64, int reset);
/* Number of explicit host<->device copies mh_select_greedy has issued since the last reset. */
MH_API int mh_get_copy_stats(mh_engine* e, long long* h2d, long long* d2h, int reset);
/* Per-model inlier moments {n, Sx, Sy, Sxx, Sxy, Syy} and smallest eigenvalue of the 3x3
 * scatter — the collinearity test of MergingStep (M/MultiH.cpp:446-463). */
MH_API int mh_inlier_moments(mh_engine* e, double thr2, double* moments /* m x 6 */, double* min_eig /* m */);

/* ---- label --------------------------------------------------------------- */
/* dataEnergy (M/MultiH.cpp:473-504) for every (site, label): cost[i*(Nh+1)+l], int32, label 0 =
 * outlier, l>=1 = current model l-1.  cost (nullable) receives a host copy. */
MH_API int mh_data_cost(mh_engine* e, int* cost);
/* alpha-expansion over the current data cost and neighbour graph with the Potts term
 * round(100*lambda) (M/MultiH.cpp:506-511; GCoptimization.cpp:975-1058,1212-1289).
 * labels: in = initial labeling in GCO numbering 0..Nh (NULL = all zero), out = result in GCO
 * numbering.  energy: final int32 energy; cycles: executed cycles. */
MH_API int mh_expand(mh_engine* e, const int* init_labels, int* labels_out, int* energy, int* cycles);
/* Counters of the last alpha-expansion: {0 cycles, 1 moves, 2 accepted moves, 3 push phases, 4 relaxation
 * intervals, 5 host synchronisations, 6 dominance-reduction launches, 7 moves that still needed
 * push-relabel after the reduction, 8 kernel launches, 9 moves actually run (the others were skipped on the
 * device as provably idempotent), 10 moves whose undecided core was not empty, 11 sum and 12 maximum of the
 * core sizes, 13 grid barriers, 14 global relabels, 15 microseconds spent inside the solver launches, of which
 * 16 inside grid barriers, 17 in global relabels and 18 in push phases (both including their barriers), 19 in
 * relabel/push rounds that began with fewer than 64 solver rows still holding excess (the tail of a move), 20 restarts of
 * the expansion after a grid-barrier timeout (a GPU shared with other persistent launches: each restart halves the
 * solver's workgroups; results never depend on their number), 21 workgroups of the solver launch in the attempt that
 * completed, 22 restarts of all expansions of this engine so far, 23 the longest single wait at a grid barrier in microseconds (the give-up time of
 * the next expansion's first attempts is max(20 ms, 50 x the longest such wait the engine has seen); r05: the FIRST barrier of a
 * launch — the one that waits for every workgroup to be dispatched — gets max(250 ms, ten times that), a timed-out attempt is
 * repeated once with the same grid before the grid is halved, word 22 = restarts of all expansions of this engine so far; only
 * the last attempt waits 3 s)}. */
MH_API int mh_get_expand_stats(mh_engine* e, long long stats[24]);
/* r06: the concurrent alpha-moves of the last expansion (csrc/expand.hip, k_commit): [0] batches launched, [1] moves kept out of a
 * batch (solved beside others on the same labeling and validated), [2] batches that ended at a move whose test failed (it heads the
 * next batch), [3] moves the host never launched because they were provably idempotent, [4] moves run alone, [5] reserved (0),
 * [6] moves per batch (mh_set_tuning key 37), [7] label count from which the first cycle is batched too (key 38). */
MH_API int mh_get_expand_batch_stats(mh_engine* e, long long stats[8]);
/* Per-move log of the last alpha-expansion's solver launches (diagnostic; enabled with mh_set_tuning key 8 = number of
 * moves to log): 8 ints per move {undecided core sites, workgroups, global relabels, relabel intervals, push phases, grid
 * barriers, 100 MHz ticks inside the launch, of which inside barriers}; all zero for moves that were skipped or had an empty core. */
MH_API int mh_get_expand_trace(mh_engine* e, int* trace /* moves x 8 */, int moves);
/* Diagnostic (mh_set_tuning key 21 = number of moves): connected components of each move's undecided core, 16 ints per move
 * {core sites, components, largest, second largest, sites in components of <= 64 / 256 / 1024 / 2048 / 8192 sites, components of
 * those sizes, propagation rounds, 0}; zeros for skipped moves.  Never changes a result. */
MH_API int mh_get_core_components(mh_engine* e, int* out /* moves x 16 */, int moves);
/* GetHomographyHAFNonminimal for every label (M/MultiH.cpp:913-989 + the 1/lambda rescale of
 * Homography_RefineHAFCallback.h:33-34).  labels: -1..Nh-1 per point.  Updates the current
 * model set in place; H_out (nullable) receives a host copy. */
MH_API int mh_reestimate(mh_engine* e, const int* labels, double* H_out);
/* LabelingStep (M/MultiH.cpp:513-602): data cost -> expansion (warm start iff warm != 0, from
 * `labeling` + 1) -> labels - 1 -> re-estimation.  labeling: in/out n ints (-1..Nh-1). */
MH_API int mh_labeling_step(mh_engine* e, int warm, int* labeling, double* energy, int* cycles);

/* ---- device-side access (bench / multi-GPU plumbing) --------------------- */
enum { MH_BUF_COUNTS = 0, MH_BUF_MODELS = 1, MH_BUF_RESIDUALS = 2, MH_BUF_LABELS = 3, MH_BUF_COST = 4, MH_BUF_GATHERED_SCORES = 5 };
/* Device pointer and size in bytes of a resident buffer (valid until the next call that
 * re-allocates it).  Used to wrap the per-model scores in a tensor for the RCCL all-gather. */
MH_API int mh_device_buffer(mh_engine* e, int which, void** ptr_dev, unsigned long long* bytes);

/* Per-kernel HIP-event timing on the engine's stream.  When enabled every launch of the
 * instrumented kernels is bracketed by events; stats are resolved at the next synchronize. */
enum { MH_K_DLT4 = 0, MH_K_RESIDUAL = 1, MH_K_SCORE = 2, MH_K_DATACOST = 3, MH_K_EXPAND = 4,
       MH_K_REESTIMATE = 5, MH_K_COSTMATRIX = 6,
       /* r06: what mh_select_best enqueues on the exchange stream — the all-gather of the scores (when a transport is set)
        * and the arg-max launch behind it — from the moment the batch's sweep has ended to the end of that launch: a rank's
        * wait for its peers plus the wire time, per exchange */
       MH_K_EXCHANGE = 7, MH_K_COUNT_ = 8 };
MH_API int mh_profile_enable(mh_engine* e, int on);
MH_API int mh_profile_reset(mh_engine* e);
MH_API int mh_profile_get(mh_engine* e, int kernel, int* launches, double* total_ms);
/* mh_set_tuning — every key in one table (r06).  Class:
 *   S  product switch, SCHEDULE ONLY: accepted by the product library, never changes a result (defaults are the measured optima)
 *   R  product switch that CHANGES RESULTS — key 30 alone
 *   D  diagnostic: what is logged / read back, never a result
 *   T  test hook: injects a failure
 *   X  experiment of a measurement build: accepted only by a library compiled with -DMH_TUNING (python multi-h_amd/build.py
 *      --tuning); the product library takes the value 0 and answers MH_ERR_INVALID to anything else
 *
 *  key class default  what
 *   0   X     0       residual kernel variant (other tilings, nt stores, compiler division, store-only calibration, fused multiply-adds: one not bit-exact)
 *   1   X     0       score kernel variant
 *   2   S     128     alpha-expansion solver: frontier (relax) rounds per barrier interval
 *   3   S     512     ... most push cycles per phase
 *   4   S     1       ... push phases per global relabel
 *   5   S     256     ... workgroups of the solver launch
 *   6   S     2       dominance-reduction rounds per launch (0 = off)
 *   7   S     6       mean shift: climb iterations per host round trip (launched schedule)
 *   8   D     0       moves logged by mh_get_expand_trace (0 = off)
 *   9   D     -1      the move whose relabels are logged one by one
 *  10   S     6       push cycles per phase as a multiple of the last relabel's depth
 *  11   S     1       flow recycling between the cycles of an expansion (0: every move from the zero flow)
 *  12   S     1       dominance-reduction launches per move (1 or 2)
 *  13   X     0       row pitch of R in doubles (0 = the product's)
 *  14   T     0       the first attempts of the next n expansions count as barrier time-outs (restart with fewer workgroups)
 *  15   S     1       FP32 pre-test of the score kernels (0: the FP64 sweep for every pair; counts equal by construction)
 *  16   X     0       tiling of that pre-test kernel
 *  17   S     2       passes of the dominance cascade inside the solver launch (0 = to its fixed point)
 *  18   T     0       the n-th scoring round of the coming greedy selections fails on this rank (collective exit of mh_select_greedy)
 *  19   S     0       residual sweep as a resident grid: workgroup slots left free beyond its own occupancy (-1: one hardware-dispatched workgroup per item)
 *  20   S     1       the sweep is held until the second stream has reached a pending DLT prefetch's dispatch
 *  21   D     0       moves whose core components are diagnosed (mh_get_core_components)
 *  22   S     0       dummy streams created in front of the engine's second stream, before its first use (placement experiment kept in the product: no effect on results)
 *  23   S     8       int32 cost matrix as a resident grid: point slices (0: one workgroup per item; -1: about 37 500 items)
 *  24   S     12      the same for the FP32 pre-test score
 *  25   S     0       DLT proposer's form: 0 by context (registers + DPP for mh_propose_dlt4, LDS-staged for mh_prefetch_dlt4), 1 LDS everywhere, 2 registers everywhere — same bits
 *  26   X     0       resident residual sweep takes its items slice-major with this many point slices
 *  27   X     0       the same order for the resident cost-matrix kernel
 *  28   X     0       cost-matrix kernel evaluates the near pairs of several models together (same matrix; slower)
 *  29   S     12      mean shift: at most this many running climbs of a batch finish in one persistent launch (0 = a launch per iteration)
 *  30   R     0       mh_select_greedy refits every round's winner to the correspondences of the support set it explains (the loop's
 *                     per-label HAF least squares, M/MultiH.cpp:913-989, one label; needs affinities and the epipolar geometry) before it
 *                     claims them; the refit takes the hypothesis' place when it is finite and explains at least as many.  Sticky per engine;
 *                     class MultiH sets it on every call (SetProposalRefit, default on).  The ranks of a sharded batch must agree on it
 *                     (their records carry it; r06).
 *  31   S     1       k-NN table through a grid over the source image (0: the exhaustive pass) — the same table
 *  32   S     1       mean shift: a workgroup per climb through a one-coordinate index, one launch per batch (0: keys 7 / 29's schedule)
 *  33   S     8       ... members per iteration beyond which an indexed climb counts as dense and is handed to the persistent kernel
 *  36   S     1       greedy selection: rounds after the first count on the points the last claim took out and subtract (0: count again on what is left)
 *  37   S     16      alpha-expansion: this many consecutive moves are solved together on the same labeling and committed in order, each
 *                     validated against what its predecessors changed (1: one move after the other, the form until r05) — same labels,
 *                     energies and cycle counts
 *  38   S     16      ... from the first cycle on for label sets of at least this many labels (smaller sets from the second cycle)
 *  39   S     0       ... sites per wavefront in the setup and reduction launches of a batch of moves: 16 / 32 / 64, 0 = by the size of the
 *                     launch (moves x sites); a move alone takes 16, a batch's energy-difference launch 64
 * (34 and 35 are not assigned.) */
MH_API int mh_set_tuning(mh_engine* e, int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* MULTIH_HIP_H */
