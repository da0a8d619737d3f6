// reestimate.hip — per-label HAF non-minimal homography re-estimation, gfx950.
//
// GetHomographyHAFNonminimal (M/MultiH.cpp:913-989) for every label in one
// launch, plus the in-place 1/lambda rescale RefineHomographyHAF applies
// (Homography_RefineHAFCallback.h:33-34).  The LM loop that follows in the
// reference does not change the output (SURVEY A-3) and is not reproduced.
//
// Per point with label l: six design rows (4 columns) from the affinity
// (a11 a12 a21 a22), the correspondence, F and the epipole e2 (:938-966); the
// reference forms the 6n x 4 matrix and multiplies A^T A with OpenCV gemm.
// Here one workgroup per label accumulates the 10 unique A^T A entries
// directly: thread t of 256 adds its sites t, t+256, ... in increasing order,
// then a binary tree v[t] += v[t+s], s = 128..1 (the engine's deterministic
// FP64 summation order, identical to oracle/mh_oracle.cpp TreeAcc).  Thread 0
// solves the 4x4 symmetric eigen-problem (cyclic Jacobi, stands for cv::eigen
// :973), takes the eigenvector of the smallest eigenvalue (:975-982), builds
// rows 1-2 of H (:984-989) and rescales.  A label with no points keeps its H
// (:592-593).  Nh is small (10..100): the kernel is latency-bound by design.

#include "mh_kernels.hpp"
#include "mh_device.hpp"

namespace mh {

__global__ void __launch_bounds__(256)
k_haf_reestimate(const double* __restrict__ x1, const double* __restrict__ y1,
                 const double* __restrict__ x2, const double* __restrict__ y2,
                 const double* __restrict__ a11p, const double* __restrict__ a12p,
                 const double* __restrict__ a21p, const double* __restrict__ a22p, int N,
                 const int* __restrict__ labels, Epipolar ep, double* __restrict__ H,
                 int* __restrict__ counts)
{
    const int l = blockIdx.x;
    const int t = threadIdx.x;
    const double* F = ep.F;
    const double ex = ep.ex, ey = ep.ey;

    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.0;
    int cnt = 0;
    // A lane's members are added up in index order (the oracle's order).  The loop is latency bound, and most points
    // carry another label: the labels of 32 of the lane's points are fetched together (one trip to memory), the
    // correspondences and affinities only of those that match, four at a time (r05; before: all eight doubles of EVERY
    // point, four points per trip — 85 of the launch's 110 us at 50 000 points, whatever the label's size).  The additions
    // keep their order.
    auto add_point = [&](const double (&in)[8]) {
        const double a11 = in[0], a12 = in[1], a21 = in[2], a22 = in[3];
        const double px = in[4], py = in[5], qx = in[6], qy = in[7];
        double r[6][4];
        r[0][0] = a11 * px + qx - ex; r[0][1] = a11 * py;           r[0][2] = a11; r[0][3] = -F[3];
        r[1][0] = a12 * px;           r[1][1] = a12 * py + qx - ex; r[1][2] = a12; r[1][3] = -F[4];
        r[2][0] = a21 * px + qy - ey; r[2][1] = a21 * py;           r[2][2] = a21; r[2][3] = F[0];
        r[3][0] = a22 * px;           r[3][1] = a22 * py + qy - ey; r[3][2] = a22; r[3][3] = F[1];
        r[4][0] = ex * px - qx * px;  r[4][1] = ex * py - qx * py;  r[4][2] = ex - qx;
        r[4][3] = px * F[3] + py * F[4] + F[5];
        r[5][0] = ey * px - qy * px;  r[5][1] = ey * py - qy * py;  r[5][2] = ey - qy;
        r[5][3] = -(px * F[0] + py * F[1] + F[2]);
        int k = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = i; jj < 4; ++jj) {
                double s = r[0][i] * r[0][jj];
#pragma unroll
                for (int q = 1; q < 6; ++q) s = s + r[q][i] * r[q][jj];
                acc[k] = acc[k] + s;
                ++k;
            }
        ++cnt;
    };
    constexpr int BATCH = 32, UNROLL = 4;
    for (int n0 = t; n0 < N; n0 += 256 * BATCH) {
        int lb[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {                    // 32 independent loads in flight
            const int n = n0 + 256 * j;
            lb[j] = labels[n < N ? n : 0];
        }
        unsigned match = 0;                                  // bit j: point n0 + 256 j carries label l
#pragma unroll
        for (int j = 0; j < BATCH; ++j) match |= (n0 + 256 * j < N && lb[j] == l) ? 1u << j : 0u;
        while (match) {                                      // ascending j = ascending point index
            int idx[UNROLL];
            double in[UNROLL][8];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                idx[u] = -1;
                if (match) {
                    const int j = __builtin_ctz(match);
                    match &= match - 1u;
                    idx[u] = n0 + 256 * j;
                }
                const int m = idx[u] >= 0 ? idx[u] : 0;
                in[u][0] = a11p[m]; in[u][1] = a12p[m]; in[u][2] = a21p[m]; in[u][3] = a22p[m];
                in[u][4] = x1[m]; in[u][5] = y1[m]; in[u][6] = x2[m]; in[u][7] = y2[m];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (idx[u] >= 0) add_point(in[u]);
        }
    }

    __shared__ double sv[256][10];
    __shared__ int sc[256];
#pragma unroll
    for (int k = 0; k < 10; ++k) sv[t][k] = acc[k];
    sc[t] = cnt;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int k = 0; k < 10; ++k) sv[t][k] = sv[t][k] + sv[t + s][k];
            sc[t] += sc[t + s];
        }
        __syncthreads();
    }
    if (t != 0) return;
    if (counts) counts[l] = sc[0];
    if (sc[0] == 0) return;

    double a[16], v[16], d[4];
    int k = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j) { a[i * 4 + j] = sv[0][k]; a[j * 4 + i] = sv[0][k]; ++k; }
    jacobi_sym_dev(4, a, v, d);
    int jm = 0;
    for (int j = 1; j < 4; ++j) if (d[j] < d[jm]) jm = j;
    const double h6 = v[0 * 4 + jm], h7 = v[1 * 4 + jm], h8 = v[2 * 4 + jm], lam = v[3 * 4 + jm];
    double h[9];
    h[6] = h6; h[7] = h7; h[8] = h8;
    h[3] = ey * h6 - lam * F[0];
    h[4] = ey * h7 - lam * F[1];
    h[5] = ey * h8 - lam * F[2];
    h[0] = ex * h6 + lam * F[3];
    h[1] = ex * h7 + lam * F[4];
    h[2] = ex * h8 + lam * F[5];
    const double lam2 = (h[0] - ex * h[6]) / F[3];      // RefineHAFCallback.h:33
    const double inv = 1.0 / lam2;                      // :34  H = H * (1.0 / lambda)
    double* out = H + 9 * (size_t)l;
    for (int q = 0; q < 9; ++q) out[q] = h[q] * inv;
}

// ---------------------------------------------------------------------------
// Per-point homographies (ComputeLocalHomographies, M/MultiH.cpp:696-717 -> GetHomographyHAF
// :850-911): the six HAF rows of ONE correspondence, A^T A (entry = sum over the six rows in
// order), eigenvector of the smallest eigenvalue, H rows from e2/F/lambda, H / h33 (:910).
// Also emits the 10-D feature vector EstablishStablePointSets clusters (:617-644): images of
// (0,0),(1,0),(0,1) under H as x1 x2 x3 y1 y2 y3, then x1,y1,x2,y2 of the point times locality.
// One thread per point (the 4x4 Jacobi is a few hundred flops).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_haf_point(const double* __restrict__ x1, const double* __restrict__ y1,
            const double* __restrict__ x2, const double* __restrict__ y2,
            const double* __restrict__ a11p, const double* __restrict__ a12p,
            const double* __restrict__ a21p, const double* __restrict__ a22p, int N, Epipolar ep,
            double locality, double* __restrict__ H_out, double* __restrict__ feat_out)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double* F = ep.F;
    const double ex = ep.ex, ey = ep.ey;
    const double a11 = a11p[n], a12 = a12p[n], a21 = a21p[n], a22 = a22p[n];
    const double px = x1[n], py = y1[n], qx = x2[n], qy = y2[n];
    double r[6][4];
    r[0][0] = a11 * px + qx - ex; r[0][1] = a11 * py;           r[0][2] = a11; r[0][3] = -F[3];
    r[1][0] = a12 * px;           r[1][1] = a12 * py + qx - ex; r[1][2] = a12; r[1][3] = -F[4];
    r[2][0] = a21 * px + qy - ey; r[2][1] = a21 * py;           r[2][2] = a21; r[2][3] = F[0];
    r[3][0] = a22 * px;           r[3][1] = a22 * py + qy - ey; r[3][2] = a22; r[3][3] = F[1];
    r[4][0] = ex * px - qx * px;  r[4][1] = ex * py - qx * py;  r[4][2] = ex - qx;
    r[4][3] = px * F[3] + py * F[4] + F[5];
    r[5][0] = ey * px - qy * px;  r[5][1] = ey * py - qy * py;  r[5][2] = ey - qy;
    r[5][3] = -(px * F[0] + py * F[1] + F[2]);
    double a[16], v[16], d[4];
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j) {
            double s = r[0][i] * r[0][j];
            for (int q = 1; q < 6; ++q) s = s + r[q][i] * r[q][j];
            a[i * 4 + j] = s;
            a[j * 4 + i] = s;
        }
    jacobi_sym_dev(4, a, v, d);
    int jm = 0;
    for (int j = 1; j < 4; ++j) if (d[j] < d[jm]) jm = j;
    const double h6 = v[0 * 4 + jm], h7 = v[1 * 4 + jm], h8 = v[2 * 4 + jm], lam = v[3 * 4 + jm];
    double h[9];
    h[6] = h6; h[7] = h7; h[8] = h8;
    h[3] = ey * h6 - lam * F[0];
    h[4] = ey * h7 - lam * F[1];
    h[5] = ey * h8 - lam * F[2];
    h[0] = ex * h6 + lam * F[3];
    h[1] = ex * h7 + lam * F[4];
    h[2] = ex * h8 + lam * F[5];
    const double inv = 1.0 / h[8];                      // H = H / h33, cv::Mat / scalar scales by 1/s
    for (int q = 0; q < 9; ++q) h[q] = h[q] * inv;
    if (H_out) for (int q = 0; q < 9; ++q) H_out[9 * (size_t)n + q] = h[q];
    if (feat_out) {
        double* f = feat_out + 10 * (size_t)n;
        const double s1 = h[8], s2 = h[6] + h[8], s3 = h[7] + h[8];
        f[0] = h[2] / s1; f[1] = (h[0] + h[2]) / s2; f[2] = (h[1] + h[2]) / s3;
        f[3] = h[5] / s1; f[4] = (h[3] + h[5]) / s2; f[5] = (h[4] + h[5]) / s3;
        f[6] = px * locality; f[7] = py * locality; f[8] = qx * locality; f[9] = qy * locality;
    }
}

hipError_t launch_haf_point(const Points& p, const Affines& a, const Epipolar& ep, double locality,
                            double* H_out, double* feat_out, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_haf_point, dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2,
                       a.a11, a.a12, a.a21, a.a22, p.n, ep, locality, H_out, feat_out);
    return hipGetLastError();
}

hipError_t launch_reestimate(const Points& p, const Affines& a, const int* labels, int Nh,
                             const Epipolar& ep, double* H, int* counts, hipStream_t s)
{
    if (Nh <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_haf_reestimate, dim3(Nh), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, a.a11,
                       a.a12, a.a21, a.a22, p.n, labels, ep, H, counts);
    return hipGetLastError();
}

} // namespace mh
