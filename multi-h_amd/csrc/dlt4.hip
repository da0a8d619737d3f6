// dlt4.hip — propose step: counter-RNG 4-tuples + batched normalised 4-point
// DLT, gfx950.  Stands where the reference calls cv::findHomography on minimal
// samples (M/MultiH.cpp:725, M/MultipleHomographies.h:118-144,328-339); there
// is no in-tree reference arithmetic, the definition is DESIGN.md 3.3 (HISTORY.md 3.3).
//
// Per hypothesis: Hartley-normalise the 4 correspondences (centroid, mean
// distance sqrt 2 — the recipe of NormalizePoints,
// Homography_Refine3PTCallback.h:236-271), build the 8x9 DLT matrix A and find
// its null vector with a one-sided (Hestenes) Jacobi SVD: column pairs of
// W = [A; I9] are rotated until all columns of the A-part are mutually
// orthogonal; the I-part column under the vanishing A-column is the null
// vector.  De-normalise, scale to unit Frobenius norm, h33 >= 0.
//
// Mapping: a wavefront solves 16 hypotheses at once, 4 lanes each.  The
// round-robin schedule gives 9 rounds of 4 DISJOINT column pairs per sweep;
// lane (slot s = lane>>4) rotates pair s of the round for hypothesis lane&15.
// W lives in LDS as W[wave][row*9+col][hyp] (17x9 doubles per hypothesis,
// 19.1 KiB per wave) so the 16 hypotheses of a wave sit in consecutive banks.
// Disjoint pairs commute exactly, so the result equals a serial sweep in
// schedule order bit for bit.  All operations round once (-ffp-contract=off).

#include "mh_kernels.hpp"
#include "mh_device.hpp"

namespace mh {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// K distinct indices for hypothesis m: draw c gives r = splitmix64(seed + (m<<8) + c),
// idx = ((r>>32)*N)>>32; duplicates inside the tuple are rejected; at most MAXDRAW (<= 256) draws.
template <int K, int MAXDRAW>
__device__ __forceinline__ void sample_tuple(unsigned long long seed, unsigned long long m,
                                             unsigned int N, int* out)
{
    int got = 0;
    for (unsigned int c = 0; c < MAXDRAW && got < K; ++c) {
        const unsigned long long r = splitmix64(seed + (m << 8) + c);
        const int idx = (int)(((r >> 32) * (unsigned long long)N) >> 32);
        bool dup = false;
        for (int k = 0; k < got; ++k) dup = dup || (out[k] == idx);
        if (!dup) out[got++] = idx;
    }
    for (; got < K; ++got) out[got] = out[0];
}

__device__ __forceinline__ void sample4(unsigned long long seed, unsigned long long m,
                                        unsigned int N, int out[4])
{
    sample_tuple<4, 64>(seed, m, N, out);
}

// circle-method schedule: round r, slot k (1..4): a = (r+k)%9, b = (r+9-k)%9
__device__ __forceinline__ void rr_pair(int r, int slot, int& p, int& q)
{
    const int k = slot + 1;
    const int a = (r + k) % 9, b = (r + 9 - k) % 9;
    p = a < b ? a : b;
    q = a < b ? b : a;
}

constexpr int HPW = 16;                  // hypotheses per wave
constexpr int WROWS = 17, WCOLS = 9;

// A wavefront owns its 16 hypotheses and its slice of LDS, so the rounds only need ordering inside
// the wave: LDS instructions of one wave execute in program order, the fence keeps the compiler from
// moving LDS accesses across the round boundary.  (Workgroup barriers here cost ~25 % of the kernel.)
__device__ __forceinline__ void wave_sync()
{
#ifdef MH_DLT_WAVE_FENCES          // the r01-r03 form: wave-scope release / acquire fences (the backend emits s_waitcnt for them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#endif
}

// One-sided (Hestenes) Jacobi on W = [A (8x9); I9] in LDS, 4 lanes per hypothesis: lane slot s
// rotates pair s of each round-robin round.  Shared by the homography (DLT) and fundamental-matrix
// (8-point) proposers: both reduce to the null vector of an 8x9 matrix.
__device__ __forceinline__ void null9_sweeps(double (*W)[HPW], int hs, int slot, bool live)
{
#define WE(r, c) W[(r) * WCOLS + (c)][hs]
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < 9; ++r) {
            int p, q;
            rr_pair(r, slot, p, q);
            double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double wp = WE(i, p), wq = WE(i, q);
                alpha = alpha + wp * wp;
                beta = beta + wq * wq;
                gamma = gamma + wp * wq;
            }
            // rotate only if |gamma| > 1e-15*sqrt(alpha*beta) and neither column has already
            // vanished (norm < 1e-14; Hartley-normalised data).  NaN never rotates.
            const bool rot = live && (gamma != 0.0) && (gamma * gamma > 1e-30 * (alpha * beta)) &&
                             (alpha >= 1e-28) && (beta >= 1e-28);
            if (rot) {
                ++rotated;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double sg = (zeta >= 0.0) ? 1.0 : -1.0;
                const double t = sg / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t);
                const double s = c * t;
#pragma unroll
                for (int i = 0; i < WROWS; ++i) {
                    const double wp = WE(i, p), wq = WE(i, q);
                    WE(i, p) = c * wp - s * wq;
                    WE(i, q) = s * wp + c * wq;
                }
            }
            wave_sync();
        }
        if (!__any(rotated)) break;      // per wave: the 16 hypotheses of this wave are done
    }
#undef WE
}

// null vector = the I-part column under the A-part column of smallest norm (first minimum wins)
__device__ __forceinline__ void null9_vector(double (*W)[HPW], int hs, double g[9])
{
#define WE(r, c) W[(r) * WCOLS + (c)][hs]
    int jm = 0;
    double best = 0.0;
    for (int j = 0; j < 9; ++j) {
        double a = 0.0;
        for (int i = 0; i < 8; ++i) a = a + WE(i, j) * WE(i, j);
        if (j == 0 || a < best) { best = a; jm = j; }
    }
    for (int j = 0; j < 9; ++j) g[j] = WE(8 + j, jm);
#undef WE
}

// At most 72 VGPRs: that is what the resident residual sweep leaves free on every SIMD (5 waves of 88 registers), so a
// workgroup of this kernel fits beside it on any compute unit (residual.hip, k_residual_resident).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(72)))
k_dlt4(const double* __restrict__ x1, const double* __restrict__ y1,
       const double* __restrict__ x2, const double* __restrict__ y2, int N,
       unsigned long long seed, long long first, int M, int* __restrict__ idx_out,
       double* __restrict__ H_out)
{
    // Issue priority above the sweep's waves (priority 0).  A SIMD issues from its oldest ready wave first; beside five
    // resident sweep waves that always have an FP64 instruction ready, a wave of this kernel hardly ever issued: the DLT
    // of the next batch took the whole sweep and its own run time again after it, whatever the stream's priority and
    // however many workgroup slots stood free (profiles/r04_timeline_*.txt).  Its total VALU work is a few per cent of
    // the sweep's, so the sweep does not notice.
    __builtin_amdgcn_s_setprio(3);
    __shared__ double sW[4][WROWS * WCOLS][HPW];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int hs = lane & (HPW - 1);
    const int slot = lane >> 4;
    const int m = (blockIdx.x * 4 + wave) * HPW + hs;
    const bool live = m < M;
    double (*W)[HPW] = sW[wave];
#define WE(r, c) W[(r) * WCOLS + (c)][hs]

    // ---- sample + normalise (all 4 lanes of a hypothesis redundantly) ----
    int id[4] = { 0, 0, 0, 0 };
    double sx[4], sy[4], dx[4], dy[4];
    if (live) sample4(seed, (unsigned long long)(first + m), (unsigned int)N, id);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sx[k] = x1[id[k]]; sy[k] = y1[id[k]]; dx[k] = x2[id[k]]; dy[k] = y2[id[k]];
    }
    if (live && slot == 0 && idx_out) {
#pragma unroll
        for (int k = 0; k < 4; ++k) idx_out[4 * (size_t)m + k] = id[k];
    }
    double cx1 = ((sx[0] + sx[1]) + sx[2]) + sx[3], cy1 = ((sy[0] + sy[1]) + sy[2]) + sy[3];
    double cx2 = ((dx[0] + dx[1]) + dx[2]) + dx[3], cy2 = ((dy[0] + dy[1]) + dy[2]) + dy[3];
    cx1 = cx1 * 0.25; cy1 = cy1 * 0.25; cx2 = cx2 * 0.25; cy2 = cy2 * 0.25;
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double ax = sx[k] - cx1, ay = sy[k] - cy1, bx = dx[k] - cx2, by = dy[k] - cy2;
        d1 = d1 + sqrt(ax * ax + ay * ay);
        d2 = d2 + sqrt(bx * bx + by * by);
    }
    const double s1 = sqrt(2.0) / (d1 * 0.25), s2 = sqrt(2.0) / (d2 * 0.25);

    // ---- fill W: lane `slot` writes the two rows of correspondence `slot` ----
    {
        double x = 0.0, y = 0.0, u = 0.0, v = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k == slot) {
                x = (sx[k] - cx1) * s1; y = (sy[k] - cy1) * s1;
                u = (dx[k] - cx2) * s2; v = (dy[k] - cy2) * s2;
            }
        const int r0 = 2 * slot, r1 = 2 * slot + 1;
        WE(r0, 0) = -x; WE(r0, 1) = -y; WE(r0, 2) = -1.0; WE(r0, 3) = 0.0; WE(r0, 4) = 0.0;
        WE(r0, 5) = 0.0; WE(r0, 6) = u * x; WE(r0, 7) = u * y; WE(r0, 8) = u;
        WE(r1, 0) = 0.0; WE(r1, 1) = 0.0; WE(r1, 2) = 0.0; WE(r1, 3) = -x; WE(r1, 4) = -y;
        WE(r1, 5) = -1.0; WE(r1, 6) = v * x; WE(r1, 7) = v * y; WE(r1, 8) = v;
        // identity part: slot s writes rows 8+3s .. 8+3s+2 (slot 3: none)
        if (slot < 3)
            for (int i = 3 * slot; i < 3 * slot + 3; ++i)
                for (int j = 0; j < 9; ++j) WE(8 + i, j) = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();

    null9_sweeps(W, hs, slot, live);

    // ---- extract null vector, de-normalise (slot 0 of each live hypothesis) ----
    if (live && slot == 0) {
        double g[9];
        null9_vector(W, hs, g);
        double A1[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double a = g[3 * r], b = g[3 * r + 1], c = g[3 * r + 2];
            A1[3 * r] = a * s1;
            A1[3 * r + 1] = b * s1;
            A1[3 * r + 2] = (c - (a * s1) * cx1) - (b * s1) * cy1;
        }
        const double is2 = 1.0 / s2;
        double Hh[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Hh[j] = A1[j] * is2 + cx2 * A1[6 + j];
            Hh[3 + j] = A1[3 + j] * is2 + cy2 * A1[6 + j];
            Hh[6 + j] = A1[6 + j];
        }
        double fro = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) fro = fro + Hh[j] * Hh[j];
        double sc = 1.0 / sqrt(fro);
        if (Hh[8] < 0.0) sc = -sc;
        double* out = H_out + 9 * (size_t)m;
#pragma unroll
        for (int j = 0; j < 9; ++j) out[j] = Hh[j] * sc;
    }
#undef WE
}

// ---------------------------------------------------------------------------
// k_fund8 — batched normalised 8-point fundamental-matrix hypotheses (front half, SURVEY §8(f) row 4:
// stands where the reference calls cv::findFundamentalMat(RANSAC), M/MultiH.cpp:775, M/main.cpp:400).
// Same machinery as k_dlt4: counter-RNG 8-tuples, Hartley normalisation, the 8x9 design matrix
// (row = [u x, u y, u, v x, v y, v, x, y, 1] for p1=(x,y), p2=(u,v)), null vector by the LDS-staged
// one-sided Jacobi; then rank 2 is enforced (F <- F (I - v3 v3^T), v3 = eigenvector of F^T F with the
// smallest eigenvalue), the normalisation is undone (F = T2^T Fn T1), unit Frobenius norm, F[8] >= 0.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_fund8(const double* __restrict__ x1, const double* __restrict__ y1,
        const double* __restrict__ x2, const double* __restrict__ y2, int N,
        unsigned long long seed, long long first, int M, int* __restrict__ idx_out,
        double* __restrict__ F_out)
{
    __shared__ double sW[4][WROWS * WCOLS][HPW];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int hs = lane & (HPW - 1);
    const int slot = lane >> 4;
    const int m = (blockIdx.x * 4 + wave) * HPW + hs;
    const bool live = m < M;
    double (*W)[HPW] = sW[wave];
#define WE(r, c) W[(r) * WCOLS + (c)][hs]

    int id[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (live) sample_tuple<8, 256>(seed, (unsigned long long)(first + m), (unsigned int)N, id);
    double sx[8], sy[8], dx[8], dy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sx[k] = x1[id[k]]; sy[k] = y1[id[k]]; dx[k] = x2[id[k]]; dy[k] = y2[id[k]]; }
    if (live && slot == 0 && idx_out) {
#pragma unroll
        for (int k = 0; k < 8; ++k) idx_out[8 * (size_t)m + k] = id[k];
    }
    double cx1 = sx[0], cy1 = sy[0], cx2 = dx[0], cy2 = dy[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { cx1 = cx1 + sx[k]; cy1 = cy1 + sy[k]; cx2 = cx2 + dx[k]; cy2 = cy2 + dy[k]; }
    cx1 = cx1 * 0.125; cy1 = cy1 * 0.125; cx2 = cx2 * 0.125; cy2 = cy2 * 0.125;
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double ax = sx[k] - cx1, ay = sy[k] - cy1, bx = dx[k] - cx2, by = dy[k] - cy2;
        d1 = d1 + sqrt(ax * ax + ay * ay);
        d2 = d2 + sqrt(bx * bx + by * by);
    }
    const double s1 = sqrt(2.0) / (d1 * 0.125), s2 = sqrt(2.0) / (d2 * 0.125);

    // lane `slot` writes rows 2*slot and 2*slot+1 (correspondences 2*slot, 2*slot+1)
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((k >> 1) == slot) {
            const double x = (sx[k] - cx1) * s1, y = (sy[k] - cy1) * s1;
            const double u = (dx[k] - cx2) * s2, v = (dy[k] - cy2) * s2;
            WE(k, 0) = u * x; WE(k, 1) = u * y; WE(k, 2) = u;
            WE(k, 3) = v * x; WE(k, 4) = v * y; WE(k, 5) = v;
            WE(k, 6) = x; WE(k, 7) = y; WE(k, 8) = 1.0;
        }
    if (slot < 3)
        for (int i = 3 * slot; i < 3 * slot + 3; ++i)
            for (int j = 0; j < 9; ++j) WE(8 + i, j) = (i == j) ? 1.0 : 0.0;
    __syncthreads();

    null9_sweeps(W, hs, slot, live);

    if (live && slot == 0) {
        double g[9];
        null9_vector(W, hs, g);
        // rank 2: M = Fn^T Fn, v3 = eigenvector of the smallest eigenvalue, Fn <- Fn - (Fn v3) v3^T
        double Mm[9], V[9], D[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0.0;
                for (int k = 0; k < 3; ++k) a = a + g[3 * k + i] * g[3 * k + j];
                Mm[3 * i + j] = a;
            }
        jacobi_sym_dev(3, Mm, V, D);
        int jm = 0;
        for (int j = 1; j < 3; ++j) if (D[j] < D[jm]) jm = j;
        const double v0 = V[0 * 3 + jm], v1 = V[1 * 3 + jm], v2 = V[2 * 3 + jm];
        double Fn[9];
        for (int r = 0; r < 3; ++r) {
            const double w = (g[3 * r] * v0 + g[3 * r + 1] * v1) + g[3 * r + 2] * v2;
            Fn[3 * r] = g[3 * r] - w * v0;
            Fn[3 * r + 1] = g[3 * r + 1] - w * v1;
            Fn[3 * r + 2] = g[3 * r + 2] - w * v2;
        }
        // F = T2^T Fn T1
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
        double sc = 1.0 / sqrt(fro);
        if (Fh[8] < 0.0) sc = -sc;
        double* out = F_out + 9 * (size_t)m;
        for (int j = 0; j < 9; ++j) out[j] = Fh[j] * sc;
    }
#undef WE
}

hipError_t launch_fund8(const Points& p, unsigned long long seed, long long first, int M,
                        int* idx_out, double* F_out, hipStream_t s)
{
    if (M <= 0) return hipSuccess;
    const int per_block = 4 * HPW;
    hipLaunchKernelGGL(k_fund8, dim3((M + per_block - 1) / per_block), dim3(256), 0, s, p.x1, p.y1,
                       p.x2, p.y2, p.n, seed, first, M, idx_out, F_out);
    return hipGetLastError();
}

hipError_t launch_dlt4(const Points& p, unsigned long long seed, long long first, int M,
                       int* idx_out, double* H_out, hipStream_t s)
{
    if (M <= 0) return hipSuccess;
    const int per_block = 4 * HPW;
    hipLaunchKernelGGL(k_dlt4, dim3((M + per_block - 1) / per_block), dim3(256), 0, s, p.x1, p.y1,
                       p.x2, p.y2, p.n, seed, first, M, idx_out, H_out);
    return hipGetLastError();
}

} // namespace mh
