// merge_step.h — host-side pieces of MergingStep (M/MultiH.cpp:352-471): the 6-D
// feature map, the mean-shift mode seeker and the 3-point homography from a mode.
// Nh (number of models) is tens to hundreds, so these run on the host; the
// N x Nh scoring they feed is done by the GPU engine.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace multih {

// Images of (0,0), (1,0), (0,1) under H (M/MultiH.cpp:364-387): 6 doubles per model.
void HomographyFeatures(const double* H /* nh x 9 */, int nh, double* feat /* nh x 6 */);

struct MeanShiftResult {
    int dim = 0;
    std::vector<double> modes;                 // k x dim
    std::vector<std::vector<int>> members;     // per mode: row indices voted to it
};

// MeanShiftClustering<double>::Cluster (MeanShiftClustering.h:23-157): flat kernel, membership
// test  sum_j sqrt(d_j^2) < bandWidth^2  (an L1 ball, quirk A-8), stop when the mean moves less
// than 1e-3*bandWidth, merge modes closer than bandWidth/2, assign rows by votes.  The
// reference draws seeds with the unseeded C rand() (:54-56); here draw number c uses
// u = (splitmix64(seed + c) >> 11) * 2^-53, so runs are reproducible.  `draws` returns the
// number of draws consumed.  A shift that captures no row (the reference would spin forever
// on a NaN mean) ends that seed's climb.
void MeanShiftCluster(const double* data, int num_pts, int num_dim, double band_width,
                      uint64_t seed, MeanShiftResult& out, uint64_t* draws = nullptr);

// GetHomography3PT (M/MultiH.cpp:995-1055) WITHOUT the LM refinement (:1052-1053): Hartley
// normalisation of both point sets (Homography_Refine3PTCallback.h:146-272), normalised F and
// epipole (:1009-1017), the 2n x 3 least-squares system (:1019-1038), H rows (:1040-1050),
// de-normalisation (:1054).  n >= 3 points; pts are x,y pairs.  Returns false when the
// normal equations are singular.
bool Homography3PTLinear(const double* pts1, const double* pts2, int n, const double F[9],
                         double H[9]);

// GetHomography3PT including RefineHomography3PT (Homography_Refine3PTCallback.h:7-143): the third
// row of H is refined on the NORMALISED data by Levenberg-Marquardt — a restatement of the
// reference's copy of OpenCV's LMSolverImpl::run (M/Utilities.hpp:750-869: lambda/lc schedule,
// 1000 iterations, eps = FLT_EPSILON, DECOMP_EIG solves done with a Jacobi eigen-decomposition) with
// the callback's residual x2 - xi and its (approximate) Jacobian ex*s*[x1 y1 1], ey*s*[x1 y1 1].
// `iterations` (optional) returns the LM iteration count.
bool Homography3PT(const double* pts1, const double* pts2, int n, const double F[9], double H[9],
                   bool do_numerical_refinement, int* iterations = nullptr);

// The host half of MergingStep (M/MultiH.cpp:352-428): 6-D features of the nh models (:364-390), their mean-shift
// modes at band width thr_h (:394-397), and one LM-refined 3-point homography per mode from the images of (0,0), (1,0),
// (0,1) (:408-427).  feat (nh x 6) and modes (k x 6) are outputs for tests (nullable / may be ignored); cand receives
// 9 doubles per mode whose fit succeeded, cand_mode (nullable) the mode each candidate came from.  Returns the number
// of candidates.  The N x candidates scoring and the collinearity filter (:430-463) are the engine's.
int MergeCandidates(const double* H /* nh x 9 */, int nh, const double F[9], double thr_h, uint64_t seed,
                    std::vector<double>* feat, std::vector<double>* modes, std::vector<double>& cand,
                    std::vector<int>* cand_mode = nullptr, uint64_t* draws = nullptr);

// HomographyCompatibilityCheck (M/MultiH.cpp:100-222): per cluster 501 trials of {3 random cluster
// points -> GetHomography3PT without refinement -> median squared transfer error of the remaining
// points}; a cluster whose median-of-medians exceeds thr^2*81/16, or that has fewer than
// min_inliers points, is removed and the labels are compacted.  The reference's quirks are kept:
// the even-length "median" averages elements size/2 and size/2+1 (:176,:193), and the distance
// buffer keeps N entries, the last three of them stale from the previous trial (:136,:175).
// rand() (:142) is replaced by the engine's splitmix64 counter RNG.
// labels: in/out (-1..nh-1); H: nh x 9 in, compacted in place; returns the new model count.
// stats_fn (optional): where the trials' order statistics are computed for the clusters of at least 19 points — the
// engine's mh_compat_trial_stats (same arguments; returns false on failure, reported through *failed with the labels
// and H untouched).  Without it they are computed here on the host's cores; the 3-point fits, the replay of the draws
// and the threading of the reference's three stale buffer entries through the trials stay on the host either way.
using CompatStatsFn = std::function<bool(const double* pts_xyxy, const int* cluster_begin, int clusters, const int* tri,
                                         const double* H, const unsigned char* ok, int trials, double* stats_out)>;
int CompatibilityCheck(const double* src_xy, const double* dst_xy, int n, int* labels, double* H, int nh,
                       const double F[9], double sqr_thr, int min_inliers, uint64_t seed,
                       double* medians = nullptr /* nh, optional: per-cluster median-of-medians */,
                       const CompatStatsFn* stats_fn = nullptr, bool* failed = nullptr,
                       // r06: stats_fn fits the trials' 3-point homographies itself (the engine's mh_compat_trial_stats_fit): it is
                       // called with H = ok = nullptr and the host fits none
                       bool stats_fn_fits = false);

// r06: one 3-point least-squares homography with LM refinement per cluster of at least three members (EstablishStablePointSets,
// M/MultiH.cpp:664-688), the clusters spread over the host's cores (each fit is independent; the results are taken in cluster
// order, so nothing depends on the number of threads).  H: 9 doubles per cluster; ok: 1 where the cluster had >= 3 members and its
// fit is finite.
void Homography3PTClusters(const double* src_xy, const double* dst_xy, const std::vector<std::vector<int>>& members, const double F[9],
                           std::vector<double>& H, std::vector<unsigned char>& ok);

// The reference's neighbourhood as FLANN's default search answers it (approx_neighbours.cpp): `trees` randomised KD-trees,
// best-bin-first with `checks` examined points per query, the hits among them within `radius`; pv = n x 4 float32-rounded
// (x1, y1, x2, y2) as doubles.
void ApproxNeighbourHits(const double* pv, int n, int trees, int checks, double radius, uint64_t seed,
                         std::vector<std::vector<int>>& hits);

} // namespace multih
