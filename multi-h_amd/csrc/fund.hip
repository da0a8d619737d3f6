// fund.hip — epipolar front half on the GPU: Sampson scoring of a batch of
// fundamental-matrix hypotheses and the least-squares 8-point refit on the inliers.  Stands where the
// reference calls cv::findFundamentalMat(RANSAC) (M/MultiH.cpp:775; M/main.cpp:400) — OpenCV code that
// is not under /root/reference, so the arithmetic here is the engine's own definition (restated in
// oracle/mh_oracle.cpp, "parity unpinned").
//
// Sampson distance of p1=(x,y), p2=(u,v) under F (row-major f0..f8):
//     a = (f0 x + f1 y) + f2      b = (f3 x + f4 y) + f5      c = (f6 x + f7 y) + f8      (F p1)
//     a' = (f0 u + f3 v) + f6     b' = (f1 u + f4 v) + f7                                  (F^T p2)
//     e = (u a + v b) + c         d = (e*e) / (((a*a + b*b) + a'*a') + b'*b')
// inlier <=> d < thr^2 (strict).  Compiled with -ffp-contract=off like everything else.
//
// r06: a second error definition, selected per engine (mh_set_fundamental_metric): the one cv::findFundamentalMat
// thresholds (OpenCV 3.1.0 calib3d, FMEstimatorCallback::computeError — published source outside /root/reference, restated;
// parity unpinned): the squared distance of each point to the epipolar line of the other, the LARGER of the two,
//     d = max((e*e) / (a*a + b*b), (e*e) / (a'*a' + b'*b'))
// with e, a, b, a', b' as above (e is the same bilinear form seen from either image).  At equal F it is at least twice the
// Sampson distance (e^2 / min(A, B) against e^2 / (A + B)), so a threshold in pixels means what the reference's caller means by it (M/main.cpp:400: 2.0 px,
// M/MultiH.cpp:775: threshold_fundamental_matrix).

#include "mh_kernels.hpp"
#include "mh_device.hpp"

namespace mh {

template <int METRIC>
__device__ __forceinline__ double epipolar_d(const double* f, double x, double y, double u, double v)
{
    const double a = f[0] * x + f[1] * y + f[2];
    const double b = f[3] * x + f[4] * y + f[5];
    const double c = f[6] * x + f[7] * y + f[8];
    const double a2 = f[0] * u + f[3] * v + f[6];
    const double b2 = f[1] * u + f[4] * v + f[7];
    const double e = u * a + v * b + c;
    if (METRIC == 0) return (e * e) / (a * a + b * b + a2 * a2 + b2 * b2);
    const double d2 = (e * e) / (a * a + b * b);          // p2 to the line F p1
    const double d1 = (e * e) / (a2 * a2 + b2 * b2);      // p1 to the line F^T p2
    return d1 > d2 ? d1 : d2;
}

// Same decomposition as k_residual: a workgroup owns MC hypotheses (coefficients in LDS) and sweeps
// all points held PPL per lane in registers; counts via ballot + s_bcnt1, finished in the workgroup.
template <int PPL, int MC, int METRIC>
__global__ void __launch_bounds__(256)
k_sampson(const double* __restrict__ x1, const double* __restrict__ y1,
          const double* __restrict__ x2, const double* __restrict__ y2, int N,
          const double* __restrict__ F, int M, double thr2, int* __restrict__ counts)
{
    constexpr int TILE = 256 * PPL;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * MC;
    __shared__ double s_f[MC * 9];
    for (int i = threadIdx.x; i < MC * 9; i += 256) {
        const size_t g = (size_t)m0 * 9 + i;
        s_f[i] = (g < (size_t)M * 9) ? F[g] : 0.0;
    }
    __syncthreads();
    int cnt = 0;
    for (int base = 0; base < N; base += TILE) {
        double px[PPL], py[PPL], qx[PPL], qy[PPL];
        bool ok[PPL];
#pragma unroll
        for (int c = 0; c < PPL; ++c) {
            const int n = base + c * 256 + threadIdx.x;
            ok[c] = n < N;
            px[c] = ok[c] ? x1[n] : 0.0; py[c] = ok[c] ? y1[n] : 0.0;
            qx[c] = ok[c] ? x2[n] : 0.0; qy[c] = ok[c] ? y2[n] : 0.0;
        }
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            if (m0 + mi >= M) break;
            const double* f = s_f + 9 * mi;
            int c_m = 0;
#pragma unroll
            for (int c = 0; c < PPL; ++c) {
                const double d = epipolar_d<METRIC>(f, px[c], py[c], qx[c], qy[c]);
                c_m += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok[c] && d < thr2));
            }
            cnt += (lane == mi) ? c_m : 0;
        }
    }
    __shared__ int s_cnt[4][MC];
    if (lane < MC) s_cnt[wave][lane] = cnt;
    __syncthreads();
    if (threadIdx.x < MC && m0 + (int)threadIdx.x < M) {
        const int t = threadIdx.x;
        counts[m0 + t] = s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
    }
}

hipError_t launch_sampson_score(const Points& p, const double* F, int M, double thr2, int* counts,
                                hipStream_t s, int metric)
{
    if (M <= 0 || p.n <= 0) return hipSuccess;
    if (metric == 0)
        hipLaunchKernelGGL((k_sampson<4, 16, 0>), dim3((M + 15) / 16), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2,
                           p.n, F, M, thr2, counts);
    else
        hipLaunchKernelGGL((k_sampson<4, 16, 1>), dim3((M + 15) / 16), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2,
                           p.n, F, M, thr2, counts);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Refit: normalised least-squares 8-point on the inliers of F_in.  One workgroup, three strided
// passes over the points, every FP64 sum in the engine's deterministic order (thread t of 256 adds
// its points t, t+256, ... then a binary tree), so the oracle reproduces it bit for bit:
//   pass 1  inlier mask, count, centroid sums of both images
//   pass 2  mean distance to the centroid in both images  -> scales s = sqrt(2)/mean
//   pass 3  the 45 unique entries of A^T A of the normalised design rows
// thread 0: 9x9 cyclic Jacobi, eigenvector of the smallest eigenvalue, rank-2 projection,
// de-normalisation, unit Frobenius norm, F[8] >= 0.
// ---------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void tree_reduce(double (*sv)[K], int t)
{
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int k = 0; k < K; ++k) sv[t][k] = sv[t][k] + sv[t + s][k];
        }
        __syncthreads();
    }
}

template <int METRIC>
__global__ void __launch_bounds__(256)
k_fund_refit(const double* __restrict__ x1, const double* __restrict__ y1,
             const double* __restrict__ x2, const double* __restrict__ y2, int N,
             const double* __restrict__ F_in, double thr2, double* __restrict__ F_out,
             unsigned char* __restrict__ mask_out, int* __restrict__ count_out)
{
    const int t = threadIdx.x;
    __shared__ double sv[256][45];
    __shared__ int sc[256];
    __shared__ double s_par[8];          // cx1 cy1 cx2 cy2 s1 s2
    double f[9];
    for (int i = 0; i < 9; ++i) f[i] = F_in[i];

    // pass 1
    {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int cnt = 0;
        for (int n = t; n < N; n += 256) {
            const double x = x1[n], y = y1[n], u = x2[n], v = y2[n];
            const bool in = epipolar_d<METRIC>(f, x, y, u, v) < thr2;
            if (mask_out) mask_out[n] = in ? 1 : 0;
            if (in) { a0 = a0 + x; a1 = a1 + y; a2 = a2 + u; a3 = a3 + v; ++cnt; }
        }
        sv[t][0] = a0; sv[t][1] = a1; sv[t][2] = a2; sv[t][3] = a3;
        sc[t] = cnt;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (t < s) {
                for (int k = 0; k < 4; ++k) sv[t][k] = sv[t][k] + sv[t + s][k];
                sc[t] += sc[t + s];
            }
            __syncthreads();
        }
        if (t == 0) {
            const int c = sc[0];
            if (count_out) *count_out = c;
            const double inv = 1.0 / (double)c;
            s_par[0] = sv[0][0] * inv; s_par[1] = sv[0][1] * inv; s_par[2] = sv[0][2] * inv; s_par[3] = sv[0][3] * inv;
        }
        __syncthreads();
    }
    const int count = sc[0];
    if (count < 8) {                     // not enough support: hand the input back unchanged
        if (t < 9) F_out[t] = f[t];
        return;
    }
    const double cx1 = s_par[0], cy1 = s_par[1], cx2 = s_par[2], cy2 = s_par[3];
    __syncthreads();
    // pass 2
    {
        double d1 = 0.0, d2 = 0.0;
        for (int n = t; n < N; n += 256) {
            const double x = x1[n], y = y1[n], u = x2[n], v = y2[n];
            if (epipolar_d<METRIC>(f, x, y, u, v) < thr2) {
                const double ax = x - cx1, ay = y - cy1, bx = u - cx2, by = v - cy2;
                d1 = d1 + sqrt(ax * ax + ay * ay);
                d2 = d2 + sqrt(bx * bx + by * by);
            }
        }
        sv[t][0] = d1; sv[t][1] = d2;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (t < s) { sv[t][0] = sv[t][0] + sv[t + s][0]; sv[t][1] = sv[t][1] + sv[t + s][1]; }
            __syncthreads();
        }
        if (t == 0) {
            s_par[4] = sqrt(2.0) / (sv[0][0] / (double)count);
            s_par[5] = sqrt(2.0) / (sv[0][1] / (double)count);
        }
        __syncthreads();
    }
    const double s1 = s_par[4], s2 = s_par[5];
    __syncthreads();
    // pass 3
    {
        double acc[45];
#pragma unroll
        for (int k = 0; k < 45; ++k) acc[k] = 0.0;
        for (int n = t; n < N; n += 256) {
            const double x0 = x1[n], y0 = y1[n], u0 = x2[n], v0 = y2[n];
            if (epipolar_d<METRIC>(f, x0, y0, u0, v0) < thr2) {
                const double x = (x0 - cx1) * s1, y = (y0 - cy1) * s1, u = (u0 - cx2) * s2, v = (v0 - cy2) * s2;
                const double r[9] = { u * x, u * y, u, v * x, v * y, v, x, y, 1.0 };
                int k = 0;
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int j = i; j < 9; ++j) { acc[k] = acc[k] + r[i] * r[j]; ++k; }
            }
        }
#pragma unroll
        for (int k = 0; k < 45; ++k) sv[t][k] = acc[k];
        tree_reduce<45>(sv, t);
    }
    if (t != 0) return;
    double A[81], V[81], D[9];
    {
        int k = 0;
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) { A[i * 9 + j] = sv[0][k]; A[j * 9 + i] = sv[0][k]; ++k; }
    }
    jacobi_sym_dev(9, A, V, D);
    int jm = 0;
    for (int j = 1; j < 9; ++j) if (D[j] < D[jm]) jm = j;
    double g[9];
    for (int j = 0; j < 9; ++j) g[j] = V[j * 9 + jm];
    // rank 2, de-normalise — same steps as k_fund8
    double Mm[9], V3[9], D3[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a = a + g[3 * k + i] * g[3 * k + j];
            Mm[3 * i + j] = a;
        }
    jacobi_sym_dev(3, Mm, V3, D3);
    int j3 = 0;
    for (int j = 1; j < 3; ++j) if (D3[j] < D3[j3]) j3 = j;
    const double v0 = V3[0 * 3 + j3], v1 = V3[1 * 3 + j3], v2 = V3[2 * 3 + j3];
    double Fn[9];
    for (int r = 0; r < 3; ++r) {
        const double w = (g[3 * r] * v0 + g[3 * r + 1] * v1) + g[3 * r + 2] * v2;
        Fn[3 * r] = g[3 * r] - w * v0;
        Fn[3 * r + 1] = g[3 * r + 1] - w * v1;
        Fn[3 * r + 2] = g[3 * r + 2] - w * v2;
    }
    double B[9];
    for (int r = 0; r < 3; ++r) {
        const double a = Fn[3 * r] * s1, b = Fn[3 * r + 1] * s1;
        B[3 * r] = a; B[3 * r + 1] = b;
        B[3 * r + 2] = (Fn[3 * r + 2] - a * cx1) - b * cy1;
    }
    double Fh[9];
    const double tx = s2 * cx2, ty = s2 * cy2;
    for (int j = 0; j < 3; ++j) {
        Fh[j] = s2 * B[j];
        Fh[3 + j] = s2 * B[3 + j];
        Fh[6 + j] = (B[6 + j] - tx * B[j]) - ty * B[3 + j];
    }
    double fro = 0.0;
    for (int j = 0; j < 9; ++j) fro = fro + Fh[j] * Fh[j];
    double scl = 1.0 / sqrt(fro);
    if (Fh[8] < 0.0) scl = -scl;
    for (int j = 0; j < 9; ++j) F_out[j] = Fh[j] * scl;
}

hipError_t launch_fund_refit(const Points& p, const double* F_in, double thr2, double* F_out,
                             unsigned char* mask_out, int* count_out, hipStream_t s, int metric)
{
    if (metric == 0)
        hipLaunchKernelGGL(k_fund_refit<0>, dim3(1), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, F_in, thr2,
                           F_out, mask_out, count_out);
    else
        hipLaunchKernelGGL(k_fund_refit<1>, dim3(1), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, F_in, thr2,
                           F_out, mask_out, count_out);
    return hipGetLastError();
}

} // namespace mh
