// merge_step.cpp — see merge_step.h.  Plain C++17, no OpenCV, no HIP.
#include "merge_step.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>

namespace multih {

void HomographyFeatures(const double* H, int nh, double* feat)
{
    for (int i = 0; i < nh; ++i) {
        const double* h = H + 9 * (size_t)i;
        double* f = feat + 6 * (size_t)i;
        const double s1 = h[8];                      // image of [0,0,1]  (M/MultiH.cpp:368-370)
        f[0] = h[2] / s1;
        f[1] = h[5] / s1;
        const double s2 = h[6] + h[8];               // image of [1,0,1]  (:373-375)
        f[2] = (h[0] + h[2]) / s2;
        f[3] = (h[3] + h[5]) / s2;
        const double s3 = h[7] + h[8];               // image of [0,1,1]  (:378-380)
        f[4] = (h[1] + h[2]) / s3;
        f[5] = (h[4] + h[5]) / s3;
    }
}

static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static double l2(const std::vector<double>& a, const std::vector<double>& b)
{
    double s = 0.0;
    for (size_t j = 0; j < a.size(); ++j) { const double d = a[j] - b[j]; s = s + d * d; }
    return std::sqrt(s);
}

void MeanShiftCluster(const double* data, int num_pts, int num_dim, double band_width,
                      uint64_t seed, MeanShiftResult& out, uint64_t* draws)
{
    out.dim = num_dim;
    out.modes.clear();
    out.members.clear();
    uint64_t counter = 0;
    const double band_sq = band_width * band_width;             // MeanShiftClustering.h:31
    const double stop_thresh = 1e-3 * band_width;               // :48
    // `init` of the reference (:125-130) is the ascending list of the unvisited rows, rebuilt after every climb; a Fenwick tree over
    // the unvisited flags answers "the k-th unvisited row" without the list (r06; the engine's mh_mean_shift does the same)
    std::vector<int> visited(num_pts, 0), fen(num_pts + 1, 0);
    for (int i = 1; i <= num_pts; ++i) { fen[i] += 1; const int j = i + (i & -i); if (j <= num_pts) fen[j] += fen[i]; }
    int top = 1;
    while (top * 2 <= num_pts) top *= 2;
    int unvisited = num_pts;
    auto kth_unvisited = [&](int k) {                               // 0-based k -> 0-based row
        int pos = 0, rem = k + 1;
        for (int step = top; step > 0; step >>= 1)
            if (pos + step <= num_pts && fen[pos + step] < rem) { pos += step; rem -= fen[pos]; }
        return pos;
    };
    auto mark_visited = [&](int row) {
        if (visited[row]) return;
        visited[row] = 1;
        --unvisited;
        for (int i = row + 1; i <= num_pts; i += i & -i) fen[i] -= 1;
    };
    std::vector<std::vector<double>> cent;
    // per mode the rows that voted for it and how often (r06: sparse — a mode's climbs touch a handful of the rows); `slot` is the
    // scratch that finds a row's entry when a later climb is merged into the mode
    std::vector<std::vector<std::pair<int, int>>> votes;
    std::vector<int> slot(num_pts, -1);

    // r06: an index on ONE coordinate.  A climb iteration tests every row against the window sum_j sqrt(r_j^2) < band_sq (:78-85); a
    // member's every |r_j| is bounded by that sum, so only rows whose coordinate c lies within band_sq (1 + 2^-20) of the mean's can be
    // members (the slack covers the rounding of r*r and of its root: the computed term is >= |r_c| (1 - 2^-51)).  The rows are sorted on
    // the coordinate with the widest spread once; an iteration takes the rows of its window, puts them back into ASCENDING ROW ORDER —
    // the order in which the full scan adds its members up — and applies the reference's own test to them: the same members, the same
    // sums in the same order, the same bits, without touching the other rows.  (The first MergingStep of the reference's route at
    // configs[4] shifts 1 564 models: 31 -> 3 ms.)  Non-finite coordinates and windows holding most of the rows fall back to the full scan.
    int cdim = 0;
    std::vector<int> perm;                                      // rows in ascending order of coordinate cdim
    std::vector<double> key;                                    // ... and their coordinate
    bool indexed = num_pts >= 64 && band_sq > 0.0 && band_sq < 1e299;
    if (indexed) {
        double best = -1.0;
        for (int j = 0; j < num_dim && indexed; ++j) {
            double lo = data[j], hi = data[j];
            for (int i = 0; i < num_pts; ++i) {
                const double x = data[(size_t)i * num_dim + j];
                if (!(std::fabs(x) < 1e299)) { indexed = false; break; }
                lo = std::min(lo, x); hi = std::max(hi, x);
            }
            if (hi - lo > best) { best = hi - lo; cdim = j; }
        }
    }
    if (indexed) {
        perm.resize(num_pts);
        for (int i = 0; i < num_pts; ++i) perm[i] = i;
        std::sort(perm.begin(), perm.end(), [&](int a, int b) {
            const double xa = data[(size_t)a * num_dim + cdim], xb = data[(size_t)b * num_dim + cdim];
            return xa < xb || (xa == xb && a < b);
        });
        key.resize(num_pts);
        for (int i = 0; i < num_pts; ++i) key[i] = data[(size_t)perm[i] * num_dim + cdim];
    }
    const double reach = band_sq * (1.0 + 0x1p-20);
    std::vector<int> cand, touched;                             // rows of the window in row order; rows this climb has voted for
    std::vector<int> my_votes(num_pts, 0);

    while (unvisited > 0) {
        const double rnd = (double)(splitmix64(seed + counter++) >> 11) * (1.0 / 9007199254740992.0);
        const int temp = (int)std::round(rnd * (double)(unvisited - 1));     // :55
        const int st = kth_unvisited(temp);
        std::vector<double> mean(data + (size_t)st * num_dim, data + (size_t)(st + 1) * num_dim);
        for (int i : touched) my_votes[i] = 0;
        touched.clear();
        for (int guard = 0; guard < 100000; ++guard) {
            const std::vector<double> old = mean;
            std::vector<double> acc(num_dim, 0.0);
            int in = 0;
            auto visit = [&](int i) {
                double dist = 0.0;
                for (int j = 0; j < num_dim; ++j) {                 // :78-83 (L1 norm via sqrt of square)
                    const double r = old[j] - data[(size_t)i * num_dim + j];
                    dist += std::sqrt(r * r);
                }
                if (dist < band_sq) {                                // :85
                    if (my_votes[i]++ == 0) touched.push_back(i);
                    ++in;
                    for (int j = 0; j < num_dim; ++j) acc[j] = acc[j] + data[(size_t)i * num_dim + j];
                    mark_visited(i);
                }
            };
            bool scanned = false;
            if (indexed && std::fabs(old[cdim]) < 1e299) {
                const int lo = (int)(std::lower_bound(key.begin(), key.end(), old[cdim] - reach) - key.begin());
                const int hi = (int)(std::upper_bound(key.begin(), key.end(), old[cdim] + reach) - key.begin());
                if (2 * (hi - lo) < num_pts) {
                    cand.assign(perm.begin() + lo, perm.begin() + hi);
                    std::sort(cand.begin(), cand.end());
                    for (int i : cand) visit(i);
                    scanned = true;
                }
            }
            if (!scanned) for (int i = 0; i < num_pts; ++i) visit(i);
            if (in == 0) { mean = old; mark_visited(st); break; }   // reference: NaN mean, endless loop
            const double inv = 1.0 / (double)in;                    // cv::Mat / scalar scales by 1/s (:96)
            for (int j = 0; j < num_dim; ++j) mean[j] = acc[j] * inv;
            if (l2(mean, old) < stop_thresh) {                      // :98
                int merge_with = -1;
                for (size_t cn = 0; cn < cent.size(); ++cn)
                    if (l2(mean, cent[cn]) < band_width / 2) { merge_with = (int)cn; break; }   // :101-109
                if (merge_with > -1) {
                    for (int j = 0; j < num_dim; ++j) cent[merge_with][j] = 0.5 * (cent[merge_with][j] + mean[j]);
                    std::vector<std::pair<int, int>>& v = votes[merge_with];
                    for (size_t q = 0; q < v.size(); ++q) slot[v[q].first] = (int)q;
                    for (int i : touched) {
                        if (slot[i] >= 0) v[slot[i]].second += my_votes[i];
                        else v.emplace_back(i, my_votes[i]);
                    }
                    for (const auto& e : v) slot[e.first] = -1;
                } else {
                    cent.push_back(mean);
                    votes.emplace_back();
                    for (int i : touched) votes.back().emplace_back(i, my_votes[i]);
                }
                break;
            }
        }
    }

    // :133-146: a row belongs to the mode that voted for it most often, the FIRST such mode on ties (strict <, modes in order)
    std::vector<int> best_votes(num_pts, 0), best_idx(num_pts, -1);
    for (size_t r = 0; r < votes.size(); ++r)
        for (const auto& e : votes[r])
            if (best_votes[e.first] < e.second) { best_votes[e.first] = e.second; best_idx[e.first] = (int)r; }
    out.members.assign(cent.size(), {});
    for (int i = 0; i < num_pts; ++i) if (best_idx[i] >= 0) out.members[best_idx[i]].push_back(i);
    for (auto& c : cent) out.modes.insert(out.modes.end(), c.begin(), c.end());
    if (draws) *draws = counter;
}

// --- small dense helpers -----------------------------------------------------
static void mat3_mul(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s = s + a[3 * i + k] * b[3 * k + j];
            c[3 * i + j] = s;
        }
}

static void jacobi3(double* a, double* v, double* d)
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

// NormalizePoints, Homography_Refine3PTCallback.h:161-196 (row-matrix branch)
static void normalize_points(const double* pts, int n, std::vector<double>& out, double T[9])
{
    double cx = 0.0, cy = 0.0;
    for (int i = 0; i < n; ++i) { cx = cx + pts[2 * i]; cy = cy + pts[2 * i + 1]; }
    const double invn = 1.0 / (double)n;
    cx = invn * cx; cy = invn * cy;
    double avg = 0.0;
    out.resize(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        out[2 * i] = pts[2 * i] - cx;
        out[2 * i + 1] = pts[2 * i + 1] - cy;
        avg = avg + std::sqrt(out[2 * i] * out[2 * i] + out[2 * i + 1] * out[2 * i + 1]);
    }
    avg = avg / n;
    const double ratio = std::sqrt(2.0) / avg;
    for (int i = 0; i < 2 * n; ++i) out[i] = out[i] * ratio;
    const double t[9] = { ratio, 0, -cx * ratio, 0, ratio, -cy * ratio, 0, 0, 1 };
    std::memcpy(T, t, sizeof(t));
}

// T.inv() of the similarity T = [r 0 tx; 0 r ty; 0 0 1] (:1009, :1054), computed FROM T as the reference computes it from
// T — not from the centroid behind it, which would round differently.
static void similarity_inverse(const double T[9], double Ti[9])
{
    const double ir = 1.0 / T[0];
    const double t[9] = { ir, 0, -T[2] * ir, 0, ir, -T[5] * ir, 0, 0, 1 };
    std::memcpy(Ti, t, sizeof(t));
}

// x = pinv(A) b (x nullable) and / or pinv(A) itself (P nullable) for a symmetric 3 x 3 A through its eigen-decomposition,
// eigenvalues within 2 eps sum|w| of zero dropped — what cv::solve / cv::invert do with DECOMP_EIG (M/Utilities.hpp:806,:830)
// and the stand-in for A.inv(DECOMP_SVD) * b on the normal equations (:1038).
static void sym_eig_solve3(const double A[9], const double* b, double* x, double* P)
{
    double a[9], v[9], w[3];
    std::memcpy(a, A, sizeof(a));
    jacobi3(a, v, w);
    double cut = 0.0;
    for (int k = 0; k < 3; ++k) cut = cut + std::fabs(w[k]);
    cut = cut * (2.0 * 2.220446049250313e-16);
    if (x) for (int i = 0; i < 3; ++i) x[i] = 0.0;
    if (P) for (int i = 0; i < 9; ++i) P[i] = 0.0;
    for (int k = 0; k < 3; ++k) {
        if (std::fabs(w[k]) <= cut) continue;
        if (x) {
            double proj = 0.0;
            for (int i = 0; i < 3; ++i) proj = proj + v[3 * i + k] * b[i];
            proj = proj / w[k];
            for (int i = 0; i < 3; ++i) x[i] = x[i] + proj * v[3 * i + k];
        }
        if (P)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) P[3 * i + j] = P[3 * i + j] + v[3 * i + k] * v[3 * j + k] / w[k];
    }
}

namespace {

// Everything GetHomography3PT computes before the optional refinement (M/MultiH.cpp:1003-1050): normalised points,
// normalised F and epipole, and the third row of the normalised H from the 2n x 3 least-squares system.
struct Normalised3PT {
    std::vector<double> p1, p2;
    double T1[9], T2[9], T2i[9], Fn[9], e0, e1, h3[3];
};

void solve_normalised_3pt(const double* pts1, const double* pts2, int n, const double F[9], Normalised3PT& q)
{
    double T1i[9];
    normalize_points(pts1, n, q.p1, q.T1);
    normalize_points(pts2, n, q.p2, q.T2);
    similarity_inverse(q.T1, T1i);
    similarity_inverse(q.T2, q.T2i);
    // Fn = T2^-T * F * T1^-1   (:1009)
    double T2it[9], tmp[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2it[3 * i + j] = q.T2i[3 * j + i];
    mat3_mul(T2it, F, tmp);
    mat3_mul(tmp, T1i, q.Fn);
    // epipole of the normalised F: eigenvector of Fn*Fn^T with the smallest eigenvalue (:1013-1017)
    double FFt[9], Fnt[9], v[9], d[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fnt[3 * i + j] = q.Fn[3 * j + i];
    mat3_mul(q.Fn, Fnt, FFt);
    jacobi3(FFt, v, d);
    int jm = 0;
    for (int j = 1; j < 3; ++j) if (d[j] < d[jm]) jm = j;
    q.e0 = v[0 * 3 + jm] / v[2 * 3 + jm];
    q.e1 = v[1 * 3 + jm] / v[2 * 3 + jm];
    // the 2n x 3 system (:1019-1037), solved through its normal equations (the reference: SVD pseudo-inverse, :1038);
    // A^T A and A^T b are summed row by row in the order of the rows
    std::vector<double> A(6 * (size_t)n), rhs(2 * (size_t)n);
    const double* Fn = q.Fn;
    for (int i = 0; i < n; ++i) {
        const double x1 = q.p1[2 * i], y1 = q.p1[2 * i + 1], x2 = q.p2[2 * i], y2 = q.p2[2 * i + 1];
        double* r = &A[6 * (size_t)i];
        r[0] = q.e0 * x1 - x2 * x1; r[1] = q.e0 * y1 - x2 * y1; r[2] = q.e0 - x2;
        r[3] = q.e1 * x1 - y2 * x1; r[4] = q.e1 * y1 - y2 * y1; r[5] = q.e1 - y2;
        rhs[2 * i] = -(x1 * Fn[3] + y1 * Fn[4] + Fn[5]);
        rhs[2 * i + 1] = (x1 * Fn[0] + y1 * Fn[1] + Fn[2]);
    }
    double AtA[9], Atb[3];
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int i = 0; i < 2 * n; ++i) s = s + A[3 * (size_t)i + a] * A[3 * (size_t)i + c];
            AtA[3 * a + c] = s;
        }
        double s = 0.0;
        for (int i = 0; i < 2 * n; ++i) s = s + A[3 * (size_t)i + a] * rhs[i];
        Atb[a] = s;
    }
    sym_eig_solve3(AtA, Atb, q.h3, nullptr);
}

// rows of the normalised H from its third row (:1040-1050) and the de-normalisation H = T2^-1 Hn T1 (:1054)
bool assemble_3pt(const Normalised3PT& q, const double h3[3], double H[9])
{
    double Hn[9], tmp[9];
    Hn[6] = h3[0]; Hn[7] = h3[1]; Hn[8] = h3[2];
    Hn[3] = q.e1 * h3[0] - q.Fn[0]; Hn[4] = q.e1 * h3[1] - q.Fn[1]; Hn[5] = q.e1 * h3[2] - q.Fn[2];     // :1045-1047
    Hn[0] = q.e0 * h3[0] + q.Fn[3]; Hn[1] = q.e0 * h3[1] + q.Fn[4]; Hn[2] = q.e0 * h3[2] + q.Fn[5];     // :1048-1050
    mat3_mul(q.T2i, Hn, tmp);
    mat3_mul(tmp, q.T1, H);
    for (int i = 0; i < 9; ++i) if (!std::isfinite(H[i])) return false;
    return true;
}

} // namespace

bool Homography3PTLinear(const double* pts1, const double* pts2, int n, const double F[9], double H[9])
{
    Normalised3PT q;
    solve_normalised_3pt(pts1, pts2, n, F, q);
    return assemble_3pt(q, q.h3, H);
}

// ---- LM refinement of the 3-point homography ------------------------------------------------
namespace {

struct Lm3 {
    const double* p1; const double* p2; int n; const double* Fn; double e0, e1;
    // Homography_Refine3PTCallback::compute (Homography_Refine3PTCallback.h:105-140)
    void compute(const double h[3], std::vector<double>& err, std::vector<double>* J) const
    {
        err.resize(2 * (size_t)n);
        if (J) J->resize(6 * (size_t)n);
        const double h21 = e1 * h[0] - Fn[0], h22 = e1 * h[1] - Fn[1], h23 = e1 * h[2] - Fn[2];
        const double h11 = e0 * h[0] + Fn[3], h12 = e0 * h[1] + Fn[4], h13 = e0 * h[2] + Fn[5];
        for (int i = 0; i < n; ++i) {
            const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
            double s = h[0] * x1 + h[1] * y1 + h[2];
            s = std::fabs(s) > 2.220446049250313e-16 ? 1. / s : 0;
            const double xi = (h11 * x1 + h12 * y1 + h13) * s;
            const double yi = (h21 * x1 + h22 * y1 + h23) * s;
            err[2 * i] = x2 - xi;
            err[2 * i + 1] = y2 - yi;
            if (J) {
                double* j = &(*J)[6 * (size_t)i];
                j[0] = e0 * s * x1; j[1] = e0 * s * y1; j[2] = e0 * s;
                j[3] = e1 * s * x1; j[4] = e1 * s * y1; j[5] = e1 * s;
            }
        }
    }
};

void atA_atb(const std::vector<double>& J, const std::vector<double>& r, int rows, double A[9], double v[3])
{
    for (int i = 0; i < 9; ++i) A[i] = 0.0;
    for (int i = 0; i < 3; ++i) v[i] = 0.0;
    for (int q = 0; q < rows; ++q)
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) A[3 * a + b] += J[3 * (size_t)q + a] * J[3 * (size_t)q + b];
            v[a] += J[3 * (size_t)q + a] * r[q];
        }
}

double sumsq(const std::vector<double>& r) { double s = 0; for (double x : r) s += x * x; return s; }
double maxabs(const double* x, size_t n) { double m = 0; for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(x[i])); return m; }

// LMSolverImpl::run, M/Utilities.hpp:762-869, for 3 parameters.
int lm_run3(const Lm3& cb, double x[3], int max_iters)
{
    const double epsx = 1.1920928955078125e-07, epsf = 1.1920928955078125e-07;   // FLT_EPSILON (:758)
    const double DEPS = 2.220446049250313e-16;
    std::vector<double> r, rd, J;
    cb.compute(x, r, &J);
    double S = sumsq(r);
    double A[9], v[3], D[3];
    atA_atb(J, r, 2 * cb.n, A, v);
    for (int i = 0; i < 3; ++i) D[i] = A[4 * i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        double Ap[9], P[9], d[3], xd[3];
        std::memcpy(Ap, A, sizeof(Ap));
        for (int i = 0; i < 3; ++i) Ap[4 * i] += lambda * D[i];
        sym_eig_solve3(Ap, v, d, nullptr);                          // solve(Ap, v, d, DECOMP_EIG), :806
        for (int i = 0; i < 3; ++i) xd[i] = x[i] - d[i];
        cb.compute(xd, rd, nullptr);
        const double Sd = sumsq(rd);
        double temp_d[3];
        for (int i = 0; i < 3; ++i)
            temp_d[i] = -(A[3 * i] * d[0] + A[3 * i + 1] * d[1] + A[3 * i + 2] * d[2]) + 2 * v[i];
        const double dS = d[0] * temp_d[0] + d[1] * temp_d[1] + d[2] * temp_d[2];
        const double R = (S - Sd) / (std::fabs(dS) > DEPS ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            const double t = d[0] * v[0] + d[1] * v[1] + d[2] * v[2];
            double nu = (Sd - S) / (std::fabs(t) > DEPS ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                sym_eig_solve3(A, nullptr, nullptr, P);               // invert(A, Ap, DECOMP_EIG), :830
                double maxval = DEPS;
                for (int i = 0; i < 3; ++i) maxval = std::max(maxval, std::fabs(P[4 * i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            for (int i = 0; i < 3; ++i) x[i] = xd[i];
            cb.compute(x, r, &J);
            atA_atb(J, r, 2 * cb.n, A, v);
        }
        ++iter;
        const bool proceed = iter < max_iters && maxabs(d, 3) >= epsx && maxabs(r.data(), r.size()) >= epsf;
        if (!proceed) break;
    }
    return iter;
}

} // namespace

bool Homography3PT(const double* pts1, const double* pts2, int n, const double F[9], double H[9],
                   bool do_numerical_refinement, int* iterations)
{
    if (iterations) *iterations = 0;
    // The refinement runs where the reference runs it (:1052-1054): on norm_pts1 / norm_pts2 with the normalised F and
    // epipole, starting from the third row of the normalised H as the linear solve left it.
    Normalised3PT q;
    solve_normalised_3pt(pts1, pts2, n, F, q);
    double h3[3] = { q.h3[0], q.h3[1], q.h3[2] };
    if (do_numerical_refinement) {
        Lm3 cb{ q.p1.data(), q.p2.data(), n, q.Fn, q.e0, q.e1 };
        const int it = lm_run3(cb, h3, 1000);
        if (iterations) *iterations = it;
    }
    return assemble_3pt(q, h3, H);                                                       // 3PTCallback.h:46-51, :1054
}

// ---- HomographyCompatibilityCheck ----------------------------------------------------------
// What the reference does per cluster (M/MultiH.cpp:128-196), and what its trial loop reduces to:
//   * each trial erases 3 randomly indexed points from the cluster's vectors (:140-151) and appends
//     them again at the end (:180-189) — only the ORDER of the points carries over between trials;
//   * the distance buffer has N entries of which the trial rewrites the first N-3 (:158-173), then
//     sorts all N (:175): the last three hold the three largest values of the previous trial's
//     sorted buffer.  So trial t sees  new_t (N-3 values)  U  stale_t (3 values)  with
//     stale_0 = {0,0,0} and stale_{t+1} = top3(new_t U stale_t);
//   * its "median" is element rest/2 of that sorted buffer, or the mean of elements rest/2 and
//     rest/2+1 when rest = N-3 is even (:176).
// Hence: (1) replay the index shuffling alone (sequential, integers only); (2) per trial, on all
// host cores, fit the 3-point homography and take from new_t only what the result can depend on —
// the order statistics of ranks rest/2-3 .. rest/2+1 and the three largest values — with
// std::nth_element instead of a full sort; (3) thread the three stale values through the trials.
// Elements of new_t below rank rest/2-3 are <= everything kept and elements above rank rest/2+1 are
// >= everything kept, so ranks rest/2 and rest/2+1 of the union are ranks 3 and 4 of the eight
// kept values (five order statistics + three stale).
namespace {

struct TrialStats {
    double mid[5];      // order statistics of new_t at ranks k-3 .. k+1 (k = rest/2)
    double top[3];      // three largest values of new_t, ascending
};

// Squared transfer errors of the N-3 points trial `tri` did not draw (:154-173); NaN -> 1e300.
void TrialDistances(const std::vector<double>& sx, const std::vector<double>& dx, const int* tri,
                    const double F[9], double* dist)
{
    const int N = (int)(sx.size() / 2);
    double ms[6], md[6];
    for (int j = 0; j < 3; ++j) {
        ms[2 * j] = sx[2 * tri[j]]; ms[2 * j + 1] = sx[2 * tri[j] + 1];
        md[2 * j] = dx[2 * tri[j]]; md[2 * j + 1] = dx[2 * tri[j] + 1];
    }
    double Hc[9];
    const bool ok = Homography3PTLinear(ms, md, 3, F, Hc);                    // do_numerical_refinement = false, :154
    int w = 0;
    for (int i = 0; i < N; ++i) {
        if (i == tri[0] || i == tri[1] || i == tri[2]) continue;
        double d2 = std::nan("");
        if (ok) {
            const double ox = sx[2 * i], oy = sx[2 * i + 1];
            const double ss = Hc[6] * ox + Hc[7] * oy + Hc[8];
            const double x1 = (Hc[0] * ox + Hc[1] * oy + Hc[2]) / ss;
            const double y1 = (Hc[3] * ox + Hc[4] * oy + Hc[5]) / ss;
            const double ddx = dx[2 * i] - x1, ddy = dx[2 * i + 1] - y1;
            d2 = ddx * ddx + ddy * ddy;
        }
        dist[w++] = std::isnan(d2) ? 1e300 : d2;
    }
}

void TrialSelect(std::vector<double>& dist, TrialStats& out)
{
    const int rest = (int)dist.size();
    const int k = rest / 2;
    const int lo = k - 3, hi = k + 1;                  // caller guarantees lo >= 0 and hi + 3 < rest
    std::nth_element(dist.begin(), dist.begin() + lo, dist.end());
    std::partial_sort(dist.begin() + lo + 1, dist.begin() + hi + 1, dist.end());
    for (int j = 0; j < 5; ++j) out.mid[j] = dist[lo + j];
    // everything from hi+1 on is >= dist[hi]; its three largest are the three largest overall
    std::partial_sort(dist.begin() + hi + 1, dist.begin() + hi + 4, dist.end(), std::greater<double>());
    out.top[0] = dist[hi + 3]; out.top[1] = dist[hi + 2]; out.top[2] = dist[hi + 1];
}

// Which three points every trial of one cluster draws (:140-151, :180-189): positions in the cluster's original order.
// The reference erases from and appends to a vector; only "the idx-th point still in the list" and "append" are needed,
// so the list is a Fenwick tree over N + 3 * trials slots (erase = find the idx-th live slot, O(log N)) — the vector's
// 1 503 erases cost 5 ms on a 37 000-point scene.
std::vector<int> ReplayDraws(int N, int trials, uint64_t seed, uint64_t& counter)
{
    const int cap = N + 3 * trials;
    int top = 1;
    while (top * 2 <= cap) top *= 2;
    std::vector<int> fen(cap + 1, 0), val(cap, 0);
    auto add = [&](int slot, int d) { for (int i = slot + 1; i <= cap; i += i & -i) fen[i] += d; };
    for (int i = 0; i < N; ++i) { val[i] = i; }
    for (int i = 1; i <= cap; ++i) {                                  // linear build: N live slots in front
        fen[i] += i <= N ? 1 : 0;
        const int j = i + (i & -i);
        if (j <= cap) fen[j] += fen[i];
    }
    auto kth = [&](int k) {                                           // slot of the k-th live entry, k from 0
        int pos = 0, left = k + 1;
        for (int step = top; step > 0; step >>= 1)
            if (pos + step <= cap && fen[pos + step] < left) { pos += step; left -= fen[pos]; }
        return pos;                                                   // 0-based slot
    };
    std::vector<int> tri(3 * (size_t)trials);
    int live = N, next = N;
    for (int t = 0; t < trials; ++t) {
        for (int j = 0; j < 3; ++j) {
            const double u = (double)(splitmix64(seed + counter++) >> 11) * (1.0 / 9007199254740992.0);
            const int idx = (int)((live - 1) * u);                    // :142
            const int slot = kth(idx);
            tri[3 * (size_t)t + j] = val[slot];
            add(slot, -1);                                            // :149-150
            --live;
        }
        for (int j = 2; j >= 0; --j) {                                // :180-189: sample j returns to slot N-j-1
            val[next] = tri[3 * (size_t)t + j];
            add(next, +1);
            ++next; ++live;
        }
    }
    return tri;
}

// The 3-point homography of one trial (:154, do_numerical_refinement = false).
bool TrialHomography(const std::vector<double>& sx, const std::vector<double>& dx, const int* tri, const double F[9], double Hc[9])
{
    double ms[6], md[6];
    for (int j = 0; j < 3; ++j) {
        ms[2 * j] = sx[2 * tri[j]]; ms[2 * j + 1] = sx[2 * tri[j] + 1];
        md[2 * j] = dx[2 * tri[j]]; md[2 * j + 1] = dx[2 * tri[j] + 1];
    }
    return Homography3PTLinear(ms, md, 3, F, Hc);
}

// `work(i)` for i in [0, count) on the host's cores.
template <typename Fn>
void ParallelFor(int count, size_t cost, Fn work)
{
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    if (nthreads > 32) nthreads = 32;
    // a thread costs some tens of microseconds to start and join: one per ~150 000 units of work (a 3-point fit is ~400),
    // so the post-filter of a two-plane scene (1 002 fits) runs on three threads, not on thirty-two
    const size_t useful = cost / 150000;
    if (useful < nthreads) nthreads = useful < 1 ? 1 : (unsigned)useful;
    std::atomic<int> next(0);
    auto loop = [&]() { for (;;) { const int i = next.fetch_add(1); if (i >= count) break; work(i); } };
    if (nthreads == 1) { loop(); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(loop);
    for (auto& th : pool) th.join();
}

// The literal buffer semantics, for clusters so small that the three stale entries reach the median ranks (rest < 16).
double TinyClusterMedian(const std::vector<double>& sx, const std::vector<double>& dx, const std::vector<int>& tri,
                         const double F[9], int trials)
{
    const int N = (int)(sx.size() / 2), rest = N - 3;
    std::vector<double> distances(trials), dist(N, 0.0);
    for (int t = 0; t < trials; ++t) {
        TrialDistances(sx, dx, &tri[3 * (size_t)t], F, dist.data());          // first N-3 entries
        std::sort(dist.begin(), dist.end());                                  // all N entries, 3 of them stale (:175)
        distances[t] = (rest % 2) ? dist[rest / 2] : 0.5 * (dist[rest / 2] + dist[rest / 2 + 1]);   // :176
    }
    std::sort(distances.begin(), distances.end());
    return trials % 2 ? distances[trials / 2] : 0.5 * (distances[trials / 2] + distances[trials / 2 + 1]);   // :193
}

// (3) of the header: thread the three stale entries through the trials, then the median of the trials' medians (:193).
double MedianFromStats(const TrialStats* st, int rest, int trials)
{
    std::vector<double> distances(trials);
    double stale[3] = { 0.0, 0.0, 0.0 };
    for (int t = 0; t < trials; ++t) {
        double u[8] = { st[t].mid[0], st[t].mid[1], st[t].mid[2], st[t].mid[3], st[t].mid[4],
                        stale[0], stale[1], stale[2] };
        std::sort(u, u + 8);
        distances[t] = (rest % 2) ? u[3] : 0.5 * (u[3] + u[4]);               // ranks rest/2 and rest/2+1 of the union
        double v[6] = { st[t].top[0], st[t].top[1], st[t].top[2], stale[0], stale[1], stale[2] };
        std::sort(v, v + 6);
        stale[0] = v[3]; stale[1] = v[4]; stale[2] = v[5];
    }
    std::sort(distances.begin(), distances.end());
    return trials % 2 ? distances[trials / 2] : 0.5 * (distances[trials / 2] + distances[trials / 2 + 1]);   // :193
}

} // namespace

static_assert(sizeof(TrialStats) == 8 * sizeof(double), "the statistics travel as 8 doubles per trial");

int CompatibilityCheck(const double* src_xy, const double* dst_xy, int n, int* labels, double* H, int nh,
                       const double F[9], double sqr_thr, int min_inliers, uint64_t seed, double* medians,
                       const CompatStatsFn* stats_fn, bool* failed, bool stats_fn_fits)
{
    const int trials = 501;                                               // MAX(501, MIN(501, ...)), :128
    if (failed) *failed = false;
    std::vector<std::vector<double>> src(nh), dst(nh);
    {
        std::vector<int> count(nh, 0);
        for (int i = 0; i < n; ++i) if (labels[i] > -1 && labels[i] < nh) ++count[labels[i]];
        for (int c = 0; c < nh; ++c) { src[c].reserve(2 * (size_t)count[c]); dst[c].reserve(2 * (size_t)count[c]); }
    }
    for (int i = 0; i < n; ++i) {
        const int l = labels[i];
        if (l > -1 && l < nh) {
            src[l].push_back(src_xy[2 * i]); src[l].push_back(src_xy[2 * i + 1]);
            dst[l].push_back(dst_xy[2 * i]); dst[l].push_back(dst_xy[2 * i + 1]);
        }
    }
    std::vector<char> remove(nh, 0), has_median(nh, 0);
    std::vector<double> median(nh, std::nan(""));
    // (1) the draws of every cluster that gets a median, in cluster order: one RNG counter runs through them, and every such
    // cluster takes exactly 3 * trials draws, so a cluster's first counter is known up front and the replays run side by side
    std::vector<std::vector<int>> tri(nh);
    std::vector<int> big;                                                 // clusters whose trials go through the order statistics
    std::vector<int> drawn;                                               // clusters that get a median, in order
    std::vector<uint64_t> first_counter;
    size_t replay_cost = 0;
    for (int c = 0; c < nh; ++c) {
        const int N = (int)(src[c].size() / 2);
        if (N >= std::max(min_inliers, 4)) {
            first_counter.push_back(3 * (uint64_t)trials * drawn.size());
            drawn.push_back(c);
            has_median[c] = 1;
            if (N - 3 >= 16) big.push_back(c);
            replay_cost += 8 * (size_t)N + 180 * (size_t)trials + (N - 3 < 16 ? 600 * (size_t)trials : 0);
        } else if (N < min_inliers) {
            remove[c] = 1;                                                // :199-200
        }
    }
    ParallelFor((int)drawn.size(), replay_cost, [&](int q) {
        const int c = drawn[q];
        const int N = (int)(src[c].size() / 2);
        uint64_t counter = first_counter[q];
        tri[c] = ReplayDraws(N, trials, seed, counter);
        if (N - 3 < 16) median[c] = TinyClusterMedian(src[c], dst[c], tri[c], F, trials);
    });
    // (2) per trial, independent of each other: the 3-point fit, then the order statistics of its distances
    if (!big.empty()) {
        const int nb = (int)big.size();
        std::vector<TrialStats> st((size_t)nb * trials);
        if (stats_fn && *stats_fn) {
            std::vector<int> begin(nb + 1, 0);
            for (int b = 0; b < nb; ++b) begin[b + 1] = begin[b] + (int)(src[big[b]].size() / 2);
            std::vector<double> pts(4 * (size_t)begin[nb]), Ht(stats_fn_fits ? 0 : 9 * (size_t)nb * trials);
            std::vector<int> tr(3 * (size_t)nb * trials);
            std::vector<unsigned char> ok(stats_fn_fits ? 0 : (size_t)nb * trials);
            for (int b = 0; b < nb; ++b) {
                const std::vector<double>&sx = src[big[b]], &dx = dst[big[b]];
                double* q = &pts[4 * (size_t)begin[b]];
                for (size_t i = 0; i < sx.size() / 2; ++i) { q[4 * i] = sx[2 * i]; q[4 * i + 1] = sx[2 * i + 1]; q[4 * i + 2] = dx[2 * i]; q[4 * i + 3] = dx[2 * i + 1]; }
                std::copy(tri[big[b]].begin(), tri[big[b]].end(), tr.begin() + 3 * (size_t)b * trials);
            }
            if (!stats_fn_fits)
                ParallelFor(nb * trials, (size_t)nb * trials * 400, [&](int i) {
                    const int b = i / trials, t = i % trials;
                    ok[i] = TrialHomography(src[big[b]], dst[big[b]], &tri[big[b]][3 * (size_t)t], F, &Ht[9 * (size_t)i]) ? 1 : 0;
                    if (!ok[i]) for (int k = 0; k < 9; ++k) Ht[9 * (size_t)i + k] = 0.0;
                });
            if (!(*stats_fn)(pts.data(), begin.data(), nb, tr.data(), stats_fn_fits ? nullptr : Ht.data(), stats_fn_fits ? nullptr : ok.data(), trials,
                             reinterpret_cast<double*>(st.data()))) {
                if (failed) *failed = true;
                return nh;
            }
        } else {
            size_t cost = 0;
            for (int c : big) cost += src[c].size() / 2 * (size_t)trials;
            ParallelFor(nb * trials, cost, [&](int i) {
                const int b = i / trials, t = i % trials;
                static thread_local std::vector<double> buf;
                buf.resize(src[big[b]].size() / 2 - 3);
                TrialDistances(src[big[b]], dst[big[b]], &tri[big[b]][3 * (size_t)t], F, buf.data());
                TrialSelect(buf, st[i]);
            });
        }
        ParallelFor(nb, (size_t)nb * trials * 150, [&](int b) {
            median[big[b]] = MedianFromStats(&st[(size_t)b * trials], (int)(src[big[b]].size() / 2) - 3, trials);
        });
    }
    for (int c = 0; c < nh; ++c) {
        if (medians) medians[c] = median[c];
        if (has_median[c]) remove[c] = median[c] > sqr_thr * 81.0 / 16.0;  // :195
    }
    int kept = nh;
    for (int c = nh - 1; c >= 0; --c) {                                       // :208-221
        if (!remove[c]) continue;
        for (int j = 0; j < n; ++j) {
            if (labels[j] == c) labels[j] = -1;
            else if (labels[j] > c) --labels[j];
        }
        for (int q = c; q + 1 < kept; ++q) std::memcpy(H + 9 * (size_t)q, H + 9 * (size_t)(q + 1), 9 * sizeof(double));
        --kept;
    }
    return kept;
}

void Homography3PTClusters(const double* src_xy, const double* dst_xy, const std::vector<std::vector<int>>& members, const double F[9],
                           std::vector<double>& H, std::vector<unsigned char>& ok)
{
    const int k = (int)members.size();
    H.assign(9 * (size_t)k, 0.0);
    ok.assign((size_t)k, 0);
    size_t cost = 0;
    for (const auto& m : members) if (m.size() >= 3) cost += 600 + 40 * m.size();
    ParallelFor(k, cost, [&](int c) {
        const int ni = (int)members[c].size();
        if (ni < 3) return;                                                                 // :667
        std::vector<double> p1(2 * (size_t)ni), p2(2 * (size_t)ni);
        for (int j = 0; j < ni; ++j) {
            const int idx = members[c][j];
            p1[2 * j] = src_xy[2 * (size_t)idx]; p1[2 * j + 1] = src_xy[2 * (size_t)idx + 1];
            p2[2 * j] = dst_xy[2 * (size_t)idx]; p2[2 * j + 1] = dst_xy[2 * (size_t)idx + 1];
        }
        ok[c] = Homography3PT(p1.data(), p2.data(), ni, F, &H[9 * (size_t)c], true) ? 1 : 0;  // :685
    });
}

int MergeCandidates(const double* H, int nh, const double F[9], double thr_h, uint64_t seed, std::vector<double>* feat_out,
                    std::vector<double>* modes_out, std::vector<double>& cand, std::vector<int>* cand_mode, uint64_t* draws)
{
    std::vector<double> feat(6 * (size_t)nh);
    HomographyFeatures(H, nh, feat.data());
    MeanShiftResult ms;
    MeanShiftCluster(feat.data(), nh, 6, thr_h, seed, ms, draws);
    const int k = static_cast<int>(ms.members.size());
    cand.clear();
    if (cand_mode) cand_mode->clear();
    const double pts1[6] = { 0, 0, 1, 0, 0, 1 };                                        // :408
    for (int i = 0; i < k; ++i) {
        double Hc[9];
        if (Homography3PT(pts1, &ms.modes[6 * (size_t)i], 3, F, Hc, true)) {            // :427
            cand.insert(cand.end(), Hc, Hc + 9);
            if (cand_mode) cand_mode->push_back(i);
        }
    }
    if (feat_out) *feat_out = std::move(feat);
    if (modes_out) *modes_out = std::move(ms.modes);
    return static_cast<int>(cand.size() / 9);
}

} // namespace multih

// ---- C hooks for the CPU-side tests (no GPU needed) ----------------------------
extern "C" {

// MergeCandidates for tests: feat nh x 6, modes up to nh x 6 (*n_modes of them), cand up to nh x 9, cand_mode up to nh.
__attribute__((visibility("default")))
int mhh_merge_candidates(const double* H, int nh, const double* F, double thr_h, unsigned long long seed, double* feat,
                         double* modes, int* n_modes, double* cand, int* cand_mode, unsigned long long* draws)
{
    std::vector<double> f, m, c;
    std::vector<int> cm;
    uint64_t d = 0;
    const int nc = multih::MergeCandidates(H, nh, F, thr_h, seed, &f, &m, c, &cm, &d);
    std::copy(f.begin(), f.end(), feat);
    std::copy(m.begin(), m.end(), modes);
    *n_modes = (int)(m.size() / 6);
    std::copy(c.begin(), c.end(), cand);
    std::copy(cm.begin(), cm.end(), cand_mode);
    if (draws) *draws = d;
    return nc;
}

__attribute__((visibility("default")))
void mhh_homography_features(const double* H, int nh, double* feat) { multih::HomographyFeatures(H, nh, feat); }

// modes_out: up to max_modes*dim doubles; assign_out: per row the mode index (-1 none).
__attribute__((visibility("default")))
int mhh_mean_shift(const double* data, int num_pts, int num_dim, double band_width,
                   unsigned long long seed, double* modes_out, int max_modes, int* assign_out,
                   unsigned long long* draws)
{
    multih::MeanShiftResult r;
    uint64_t d = 0;
    multih::MeanShiftCluster(data, num_pts, num_dim, band_width, seed, r, &d);
    const int k = (int)r.members.size();
    for (int i = 0; i < num_pts; ++i) assign_out[i] = -1;
    for (int c = 0; c < k; ++c)
        for (int i : r.members[c]) assign_out[i] = c;
    for (int c = 0; c < k && c < max_modes; ++c)
        for (int j = 0; j < num_dim; ++j) modes_out[c * num_dim + j] = r.modes[(size_t)c * num_dim + j];
    if (draws) *draws = d;
    return k;
}

__attribute__((visibility("default")))
int mhh_homography_3pt(const double* pts1, const double* pts2, int n, const double* F, double* H)
{
    return multih::Homography3PTLinear(pts1, pts2, n, F, H) ? 1 : 0;
}

__attribute__((visibility("default")))
int mhh_homography_3pt_refined(const double* pts1, const double* pts2, int n, const double* F, double* H,
                               int* iterations)
{
    return multih::Homography3PT(pts1, pts2, n, F, H, true, iterations) ? 1 : 0;
}

__attribute__((visibility("default")))
int mhh_compatibility_check(const double* src_xy, const double* dst_xy, int n, int* labels, double* H, int nh,
                            const double* F, double sqr_thr, int min_inliers, unsigned long long seed)
{
    return multih::CompatibilityCheck(src_xy, dst_xy, n, labels, H, nh, F, sqr_thr, min_inliers, seed);
}

// same, also returning the per-cluster median-of-medians (NaN where the cluster was not tested)
__attribute__((visibility("default")))
int mhh_compatibility_medians(const double* src_xy, const double* dst_xy, int n, int* labels, double* H, int nh,
                              const double* F, double sqr_thr, int min_inliers, unsigned long long seed,
                              double* medians)
{
    return multih::CompatibilityCheck(src_xy, dst_xy, n, labels, H, nh, F, sqr_thr, min_inliers, seed, medians);
}

}
