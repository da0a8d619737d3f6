// Sanitizer driver for the host-side MergingStep pieces (multi-h_amd/host/merge_step.cpp): built by
// tests/test_host_sanitizers.py with -fsanitize=address,undefined and with -fsanitize=thread (the
// compatibility check spreads its trials over threads).  GPU ASan is not available on this pool, so
// the host code — the part with std::vector index arithmetic — is what gets the sanitizer runs.
// Synthetic data only; prints a checksum so the two builds can be compared.
#include "merge_step.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

static uint64_t rng_state = 88172645463325252ull;
static double urand()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}
static double nrand() { return std::sqrt(-2.0 * std::log(urand() + 1e-300)) * std::cos(6.283185307179586 * urand()); }

static void apply(const double* H, double x, double y, double* u, double* v)
{
    const double s = H[6] * x + H[7] * y + H[8];
    *u = (H[0] * x + H[1] * y + H[2]) / s;
    *v = (H[3] * x + H[4] * y + H[5]) / s;
}

int main(int argc, char** argv)
{
    const bool only_threads = argc > 1;
    // two planes related by H = A + e2 * v^T share the epipolar geometry F = [e2]x A
    const double A[9] = { 1.02, 0.01, 3.0, -0.015, 0.99, -2.0, 0.0, 0.0, 1.0 };
    const double e2[3] = { 900.0, 450.0, 1.0 };
    const double ex[9] = { 0, -e2[2], e2[1], e2[2], 0, -e2[0], -e2[1], e2[0], 0 };
    double F[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) F[3 * i + j] = ex[3 * i] * A[j] + ex[3 * i + 1] * A[3 + j] + ex[3 * i + 2] * A[6 + j];
    const double vs[3][3] = { { 1e-5, -2e-5, 0.01 }, { -3e-5, 1e-5, -0.02 }, { 2e-5, 2e-5, 0.03 } };
    double H[3][9];
    for (int p = 0; p < 3; ++p)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) H[p][3 * i + j] = A[3 * i + j] + e2[i] * vs[p][j];

    const int per[3] = { 900, 401, 30 };
    std::vector<double> src, dst;
    std::vector<int> labels;
    for (int p = 0; p < 3; ++p)
        for (int i = 0; i < per[p]; ++i) {
            const double x = 1000.0 * urand(), y = 1000.0 * urand();
            double u, v;
            apply(H[p], x, y, &u, &v);
            src.push_back(x); src.push_back(y);
            dst.push_back(u + 0.3 * nrand()); dst.push_back(v + 0.3 * nrand());
            labels.push_back(p);
        }
    for (int i = 0; i < 120; ++i) {                         // a cluster of scrambled matches + unlabelled points
        src.push_back(1000.0 * urand()); src.push_back(1000.0 * urand());
        dst.push_back(1000.0 * urand()); dst.push_back(1000.0 * urand());
        labels.push_back(i < 100 ? 3 : -1);
    }
    const int n = (int)labels.size();
    double sum = 0.0;

    // compatibility check (threads inside): clusters 0,1 survive, 3 (scrambled) and 2 (< min inliers) go
    std::vector<double> Hs(4 * 9);
    for (int p = 0; p < 3; ++p) for (int q = 0; q < 9; ++q) Hs[9 * p + q] = H[p][q];
    for (int q = 0; q < 9; ++q) Hs[27 + q] = H[0][q];
    std::vector<double> med(4);
    std::vector<int> lab = labels;
    const int kept = multih::CompatibilityCheck(src.data(), dst.data(), n, lab.data(), Hs.data(), 4, F, 2.2 * 2.2, 50, 7, med.data());
    std::printf("compat kept=%d medians %.6g %.6g %.6g %.6g\n", kept, med[0], med[1], med[2], med[3]);
    if (kept != 2) { std::printf("unexpected cluster count\n"); return 2; }
    sum += med[0] + med[1] + med[3];
    if (only_threads) { std::printf("checksum %.12g\n", sum); return 0; }

    // feature map + mean shift over perturbed copies of the three models
    const int nh = 60;
    std::vector<double> Hm(9 * nh), feat(6 * nh);
    for (int i = 0; i < nh; ++i)
        for (int q = 0; q < 9; ++q) Hm[9 * i + q] = H[i % 3][q] * (1.0 + 1e-4 * nrand());
    multih::HomographyFeatures(Hm.data(), nh, feat.data());
    multih::MeanShiftResult ms;
    uint64_t draws = 0;
    multih::MeanShiftCluster(feat.data(), nh, 6, 2.2, 99, ms, &draws);
    std::printf("mean shift: %d modes, %llu draws\n", (int)ms.members.size(), (unsigned long long)draws);
    for (double m : ms.modes) sum += m;

    // 3-point homography, linear and LM-refined, minimal and over-determined
    const double canon[6] = { 0, 0, 1, 0, 0, 1 };
    for (size_t k = 0; k < ms.members.size(); ++k) {
        double Hl[9], Hr[9];
        int it = 0;
        const bool a = multih::Homography3PTLinear(canon, &ms.modes[6 * k], 3, F, Hl);
        const bool b = multih::Homography3PT(canon, &ms.modes[6 * k], 3, F, Hr, true, &it);
        if (a) sum += Hl[0] / Hl[8];
        if (b) sum += Hr[4] / Hr[8] + it;
    }
    {
        double Hl[9], Hr[9];
        int it = 0;
        multih::Homography3PTLinear(src.data(), dst.data(), 40, F, Hl);
        multih::Homography3PT(src.data(), dst.data(), 40, F, Hr, true, &it);
        sum += Hl[1] / Hl[8] + Hr[1] / Hr[8];
        // degenerate input: three identical points
        const double same[6] = { 5, 5, 5, 5, 5, 5 };
        double Hd[9];
        const bool ok = multih::Homography3PTLinear(same, same, 3, F, Hd);
        std::printf("degenerate 3PT accepted=%d\n", (int)ok);
    }
    // the FLANN-like neighbourhood (host/approx_neighbours.cpp): forests over the float32 point vectors, with exact
    // duplicates and a coordinate shared by many points (the halving rule of the tree build)
    {
        const int n = (int)labels.size();
        std::vector<double> pv(4 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            pv[4 * (size_t)i] = (double)(float)src[2 * i]; pv[4 * (size_t)i + 1] = (double)(float)src[2 * i + 1];
            pv[4 * (size_t)i + 2] = (double)(float)dst[2 * i]; pv[4 * (size_t)i + 3] = (double)(float)dst[2 * i + 1];
        }
        for (int i = 10; i < 40; ++i) for (int d = 0; d < 4; ++d) pv[4 * (size_t)i + d] = pv[40 + d];
        for (int i = 100; i < 300; ++i) pv[4 * (size_t)i] = 77.0;
        std::vector<std::vector<int>> hits;
        multih::ApproxNeighbourHits(pv.data(), n, 4, 32, 200.0, 99, hits);
        multih::ApproxNeighbourHits(pv.data(), n, 1, 3, 1e9, 5, hits);
        multih::ApproxNeighbourHits(pv.data(), 1, 4, 32, 200.0, 99, hits);
        multih::ApproxNeighbourHits(pv.data(), n, 4, 32, 200.0, 99, hits);
        for (int i = 0; i < n; ++i) sum += (double)hits[i].size();
    }
    std::printf("checksum %.12g\n", sum);
    return 0;
}
