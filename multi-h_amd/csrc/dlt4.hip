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

// The r01-r04 form of the proposer: W staged in LDS (78 KB per workgroup, two workgroups per compute unit).  It is what
// mh_prefetch_dlt4 launches (variant 1 of launch_dlt4): slower alone than the register-resident form below (0.40 against
// 0.26 ms per 100 000 hypotheses, same bits), but the one of the two that fits beside a resident sweep (capi_score.hip, mh_prefetch_dlt4).
// At most 72 VGPRs: that is what the resident residual sweep leaves free on every SIMD (5 waves of 88 registers), so a
// workgroup of this kernel fits beside it on any compute unit (residual.hip, k_residual_resident).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(72)))
k_dlt4_lds(const double* __restrict__ x1, const double* __restrict__ y1,
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
// Register-resident form (r04).  The same rotations in the same order, but W never touches LDS: the four lanes of a
// hypothesis hold the nine columns of W (17 doubles each) in registers and hand them round with DPP row shifts.
//
// Lane layout inside a row of 16 lanes: lane = 4 * slot + j, so the four slots of hypothesis j sit in the four BANKS
// (groups of 4 lanes) of the row, a row shift by 4 lanes moves a column to the neighbouring slot, and DPP's bank mask
// lets one slot keep what it has.  Round r of the circle schedule pairs positions (k, 9 - k), k = 1..4, where position
// j holds column (r + j) % 9 and position 0 sits out; slot k - 1 keeps position k in its "A" registers and position
// 9 - k in its "B" registers, slot 0 also the idle position in "C".  After a round every column moves down one position:
//      A[slot] <- A[slot + 1]   (slot 3: its own B)        one DPP move per dword, written INTO the old B registers
//      B[slot] <- B[slot - 1]   (slot 0: the idle column)  one DPP move per dword, written INTO the old C registers
//      idle    <- A[slot 0]                                nothing to do: the old A registers now play C
// so the three register sets swap roles with period 3 and a sweep is three times three rounds.  68 DPP moves per round
// stand where the LDS form had 84 LDS instructions and their latency; no LDS, any number of workgroups per compute unit.
//
// A slot may hold the higher-numbered column of its pair in "A" (the schedule's p < q is by column number).  Swapping
// the roles of the two columns negates zeta, t and s exactly and leaves c, gamma, alpha * beta unchanged; the updated
// columns are then the same sums with the operands in the other order, i.e. the same bits — except for zeta == 0, where
// the reference order takes t = +1; `swapped` picks the sign there.
// ---------------------------------------------------------------------------
struct Col { double v[WROWS]; };

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp_merge(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

// one step of the circle: a -> (plays idle), b <- a shifted down (plays A), c <- b shifted up (plays B)
__device__ __forceinline__ void shift_columns(Col& a, Col& b, Col& c)
{
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
        c.v[i] = dpp_merge<0x114 /* row_shr:4 */, 0xE>(c.v[i], b.v[i]);   // slots 1..3 take B of the slot below; slot 0 keeps the idle column
        b.v[i] = dpp_merge<0x104 /* row_shl:4 */, 0x7>(b.v[i], a.v[i]);   // slots 0..2 take A of the slot above; slot 3 keeps its own B
    }
}

__device__ __forceinline__ int rotate_pair(Col& P, Col& Q, bool swapped, bool live)
{
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double wp = P.v[i], wq = Q.v[i];
        alpha = alpha + wp * wp;
        beta = beta + wq * wq;
        gamma = gamma + wp * wq;
    }
    const bool rot = live && (gamma != 0.0) && (gamma * gamma > 1e-30 * (alpha * beta)) &&
                     (alpha >= 1e-28) && (beta >= 1e-28);
    if (rot) {
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const bool pos = swapped ? (zeta > 0.0) : (zeta >= 0.0);
        const double sg = pos ? 1.0 : -1.0;
        const double t = sg / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t);
        const double s = c * t;
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const double wp = P.v[i], wq = Q.v[i];
            P.v[i] = c * wp - s * wq;
            Q.v[i] = s * wp + c * wq;
        }
    }
    return rot ? 1 : 0;
}

// bit r set: in round r slot s holds the higher-numbered column of its pair in the position-k registers
__host__ __device__ constexpr unsigned swapped_rounds(int s)
{
    unsigned m = 0;
    for (int r = 0; r < 9; ++r) {
        const int k = s + 1, a = (r + k) % 9, b = (r + 9 - k) % 9;
        if (a > b) m |= 1u << r;
    }
    return m;
}

// column c of the 8 x 9 design matrix over the identity, c a per-lane value
__device__ __forceinline__ void dlt_column(int c, const double* x, const double* y, const double* u, const double* v, Col& out)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double e0[9] = { -x[k], -y[k], -1.0, 0.0, 0.0, 0.0, u[k] * x[k], u[k] * y[k], u[k] };
        const double e1[9] = { 0.0, 0.0, 0.0, -x[k], -y[k], -1.0, v[k] * x[k], v[k] * y[k], v[k] };
        double r0 = e0[0], r1 = e1[0];
#pragma unroll
        for (int j = 1; j < 9; ++j) {
            r0 = (c == j) ? e0[j] : r0;
            r1 = (c == j) ? e1[j] : r1;
        }
        out.v[2 * k] = r0;
        out.v[2 * k + 1] = r1;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) out.v[8 + i] = (i == c) ? 1.0 : 0.0;
}

__device__ __forceinline__ double column_norm(const Col& a)
{
    double n = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) n = n + a.v[i] * a.v[i];
    return n;
}

__global__ void __launch_bounds__(256)
k_dlt4(const double* __restrict__ x1, const double* __restrict__ y1,
       const double* __restrict__ x2, const double* __restrict__ y2, int N,
       unsigned long long seed, long long first, int M, int* __restrict__ idx_out,
       double* __restrict__ H_out)
{
    __builtin_amdgcn_s_setprio(3);           // (as k_dlt4_lds; this form runs alone on the device, where it changes nothing)
    __shared__ double sN[4][9][HPW];         // column norms, then the null vector: 1.1 KB per wave
    __shared__ double sK[4][6][HPW];         // the normalisation, parked while the sweeps need the registers
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot = (lane >> 2) & 3;
    const int hs = (lane >> 4) * 4 + (lane & 3);
    const int m = (blockIdx.x * 4 + wave) * HPW + hs;
    const bool live = m < M;

    Col A, B, C;
    {
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
        if (slot == 0) {
            sK[wave][0][hs] = s1; sK[wave][1][hs] = s2; sK[wave][2][hs] = cx1;
            sK[wave][3][hs] = cy1; sK[wave][4][hs] = cx2; sK[wave][5][hs] = cy2;
        }
        double x[4], y[4], u[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = (sx[k] - cx1) * s1; y[k] = (sy[k] - cy1) * s1;
            u[k] = (dx[k] - cx2) * s2; v[k] = (dy[k] - cy2) * s2;
        }
        dlt_column(slot + 1, x, y, u, v, A);
        dlt_column(8 - slot, x, y, u, v, B);
        dlt_column(0, x, y, u, v, C);
    }

    const unsigned sw = slot == 0 ? swapped_rounds(0) : slot == 1 ? swapped_rounds(1) : slot == 2 ? swapped_rounds(2) : swapped_rounds(3);
    for (int sweep = 0; sweep < 30; ++sweep) {
        unsigned long long rotated = 0;                               // lanes that rotated in this sweep (wave-uniform)
#pragma unroll 1
        for (int r = 0; r < 9; r += 3) {
            rotated |= __ballot(rotate_pair(A, B, (sw >> r) & 1u, live));
            shift_columns(A, B, C);                                   // roles now: A <- B, B <- C, idle <- A
            rotated |= __ballot(rotate_pair(B, C, (sw >> (r + 1)) & 1u, live));
            shift_columns(B, C, A);                                   // A <- C, B <- A, idle <- B
            rotated |= __ballot(rotate_pair(C, A, (sw >> (r + 2)) & 1u, live));
            shift_columns(C, A, B);                                   // back to A, B, idle = C
        }
        if (rotated == 0) break;                                      // per wave: its 16 hypotheses are done
    }

    // after whole sweeps the columns are back where they started: A = column slot + 1, B = column 8 - slot, C = column 0
    sN[wave][slot + 1][hs] = column_norm(A);
    sN[wave][8 - slot][hs] = column_norm(B);
    if (slot == 0) sN[wave][0][hs] = column_norm(C);
    wave_sync();
    int jm = 0;
    {
        double best = 0.0;
        for (int j = 0; j < 9; ++j) {
            const double a = sN[wave][j][hs];
            if (j == 0 || a < best) { best = a; jm = j; }
        }
    }
    wave_sync();
    if (jm == slot + 1) {
#pragma unroll
        for (int j = 0; j < 9; ++j) sN[wave][j][hs] = A.v[8 + j];
    } else if (jm == 8 - slot) {
#pragma unroll
        for (int j = 0; j < 9; ++j) sN[wave][j][hs] = B.v[8 + j];
    } else if (jm == 0 && slot == 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j) sN[wave][j][hs] = C.v[8 + j];
    }
    wave_sync();

    if (live && slot == 0) {
        double g[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) g[j] = sN[wave][j][hs];
        const double s1 = sK[wave][0][hs], s2 = sK[wave][1][hs], cx1 = sK[wave][2][hs];
        const double cy1 = sK[wave][3][hs], cx2 = sK[wave][4][hs], cy2 = sK[wave][5][hs];
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
                       int* idx_out, double* H_out, hipStream_t s, int variant)
{
    if (M <= 0) return hipSuccess;
    const int per_block = 4 * HPW;
    if (variant == 1)
        hipLaunchKernelGGL(k_dlt4_lds, dim3((M + per_block - 1) / per_block), dim3(256), 0, s, p.x1, p.y1,
                           p.x2, p.y2, p.n, seed, first, M, idx_out, H_out);
    else
        hipLaunchKernelGGL(k_dlt4, dim3((M + per_block - 1) / per_block), dim3(256), 0, s, p.x1, p.y1,
                           p.x2, p.y2, p.n, seed, first, M, idx_out, H_out);
    return hipGetLastError();
}

} // namespace mh
