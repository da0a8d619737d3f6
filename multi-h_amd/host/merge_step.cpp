// merge_step.cpp — see merge_step.h.  Plain C++17, no OpenCV, no HIP.
#include "merge_step.h"

#include <algorithm>
#include <cmath>
#include <cstring>

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
    std::vector<int> init(num_pts), visited(num_pts, 0);
    for (int i = 0; i < num_pts; ++i) init[i] = i;
    std::vector<std::vector<double>> cent;
    std::vector<std::vector<int>> votes;

    while (!init.empty()) {
        const double rnd = (double)(splitmix64(seed + counter++) >> 11) * (1.0 / 9007199254740992.0);
        const int temp = (int)std::round(rnd * (double)(init.size() - 1));   // :55
        const int st = init[temp];
        std::vector<double> mean(data + (size_t)st * num_dim, data + (size_t)(st + 1) * num_dim);
        std::vector<int> my_votes(num_pts, 0);
        for (int guard = 0; guard < 100000; ++guard) {
            const std::vector<double> old = mean;
            std::vector<double> acc(num_dim, 0.0);
            int in = 0;
            for (int i = 0; i < num_pts; ++i) {
                double dist = 0.0;
                for (int j = 0; j < num_dim; ++j) {                 // :78-83 (L1 norm via sqrt of square)
                    const double r = old[j] - data[(size_t)i * num_dim + j];
                    dist += std::sqrt(r * r);
                }
                if (dist < band_sq) {                                // :85
                    ++my_votes[i];
                    ++in;
                    for (int j = 0; j < num_dim; ++j) acc[j] = acc[j] + data[(size_t)i * num_dim + j];
                    visited[i] = 1;
                }
            }
            if (in == 0) { mean = old; visited[st] = 1; break; }    // reference: NaN mean, endless loop
            const double inv = 1.0 / (double)in;                    // cv::Mat / scalar scales by 1/s (:96)
            for (int j = 0; j < num_dim; ++j) mean[j] = acc[j] * inv;
            if (l2(mean, old) < stop_thresh) {                      // :98
                int merge_with = -1;
                for (size_t cn = 0; cn < cent.size(); ++cn)
                    if (l2(mean, cent[cn]) < band_width / 2) { merge_with = (int)cn; break; }   // :101-109
                if (merge_with > -1) {
                    for (int j = 0; j < num_dim; ++j) cent[merge_with][j] = 0.5 * (cent[merge_with][j] + mean[j]);
                    for (int i = 0; i < num_pts; ++i) votes[merge_with][i] += my_votes[i];
                } else {
                    cent.push_back(mean);
                    votes.push_back(my_votes);
                }
                break;
            }
        }
        init.clear();                                               // :125-130
        for (int i = 0; i < num_pts; ++i) if (!visited[i]) init.push_back(i);
    }

    std::vector<int> best_votes(num_pts, 0), best_idx(num_pts, -1);  // :133-146
    for (size_t r = 0; r < votes.size(); ++r)
        for (int i = 0; i < num_pts; ++i)
            if (best_votes[i] < votes[r][i]) { best_votes[i] = votes[r][i]; best_idx[i] = (int)r; }
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
static void normalize_points(const double* pts, int n, std::vector<double>& out, double T[9], double Tinv[9])
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
    const double ti[9] = { 1.0 / ratio, 0, cx, 0, 1.0 / ratio, cy, 0, 0, 1 };
    std::memcpy(T, t, sizeof(t));
    std::memcpy(Tinv, ti, sizeof(ti));
}

bool Homography3PTLinear(const double* pts1, const double* pts2, int n, const double F[9], double H[9])
{
    std::vector<double> p1, p2;
    double T1[9], T1i[9], T2[9], T2i[9];
    normalize_points(pts1, n, p1, T1, T1i);
    normalize_points(pts2, n, p2, T2, T2i);
    // Fn = T2^-T * F * T1^-1   (M/MultiH.cpp:1009)
    double T2it[9], tmp[9], Fn[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2it[3 * i + j] = T2i[3 * j + i];
    mat3_mul(T2it, F, tmp);
    mat3_mul(tmp, T1i, Fn);
    // epipole of the normalised F: eigenvector of Fn*Fn^T with the smallest eigenvalue (:1013-1017)
    double FFt[9], Fnt[9], v[9], d[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fnt[3 * i + j] = Fn[3 * j + i];
    mat3_mul(Fn, Fnt, FFt);
    jacobi3(FFt, v, d);
    int jm = 0;
    for (int j = 1; j < 3; ++j) if (d[j] < d[jm]) jm = j;
    const double e0 = v[0 * 3 + jm] / v[2 * 3 + jm], e1 = v[1 * 3 + jm] / v[2 * 3 + jm];
    // normal equations of the 2n x 3 system (:1019-1038; the reference solves with an SVD pseudo-inverse)
    double AtA[9] = { 0 }, Atb[3] = { 0 };
    for (int i = 0; i < n; ++i) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        const double r0[3] = { e0 * x1 - x2 * x1, e0 * y1 - x2 * y1, e0 - x2 };
        const double r1[3] = { e1 * x1 - y2 * x1, e1 * y1 - y2 * y1, e1 - y2 };
        const double b0 = -(x1 * Fn[3] + y1 * Fn[4] + Fn[5]);
        const double b1 = (x1 * Fn[0] + y1 * Fn[1] + Fn[2]);
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) AtA[3 * a + b] += r0[a] * r0[b] + r1[a] * r1[b];
            Atb[a] += r0[a] * b0 + r1[a] * b1;
        }
    }
    // 3x3 symmetric solve by cofactors
    const double* m = AtA;
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double inv[9] = {
        c00 / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
        c01 / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
        c02 / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det };
    double r[3];
    for (int a = 0; a < 3; ++a) r[a] = inv[3 * a] * Atb[0] + inv[3 * a + 1] * Atb[1] + inv[3 * a + 2] * Atb[2];
    double Hn[9];
    Hn[6] = r[0]; Hn[7] = r[1]; Hn[8] = r[2];
    Hn[3] = e1 * Hn[6] - Fn[0]; Hn[4] = e1 * Hn[7] - Fn[1]; Hn[5] = e1 * Hn[8] - Fn[2];     // :1045-1047
    Hn[0] = e0 * Hn[6] + Fn[3]; Hn[1] = e0 * Hn[7] + Fn[4]; Hn[2] = e0 * Hn[8] + Fn[5];     // :1048-1050
    mat3_mul(T2i, Hn, tmp);                                                                 // :1054
    mat3_mul(tmp, T1, H);
    for (int i = 0; i < 9; ++i) if (!std::isfinite(H[i])) return false;
    return true;
}

} // namespace multih

// ---- C hooks for the CPU-side tests (no GPU needed) ----------------------------
extern "C" {

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

}
