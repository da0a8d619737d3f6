// score32.hip — inlier scoring with an FP32 pre-test, gfx950.  Counts are the reference's, bit for bit.
//
// A score needs, per (point, model) pair, only the DECISION  d2 < thr^2  of the reference's FP64 formula
// (M/MultiH.cpp:434-443), never d2 itself.  This kernel evaluates the forward transfer error in FP32 (fused
// multiply-adds, hardware reciprocal: about a third of the FP64 sweep's issue cycles per pair) together with a RIGOROUS
// bound B on |d2_fp32 - d2_fp64|, decides the pairs for which the bound leaves no doubt
//         d2_fp32 + B <  thr^2   ->  inlier            d2_fp32 - B >= thr^2   ->  not an inlier
// (evaluated as |d2_fp32 - thr^2| > B with the sign of the difference telling which)
// and recomputes the others — pairs within B of the threshold, pairs near a model's horizon, anything that produced a
// NaN or an infinity on the way — with the FP64 formula in the reference's own operation order.  The result is the count
// the FP64 kernel (residual.hip) gives; the tests compare the two and put thresholds exactly ON residual values.
//
// The bound.  u = 2^-24.  Inputs are rounded to FP32 (relative error u each).  With X, Y = max |x|, |y| over all source
// points, per MODEL (k_model32, in FP64, rounded up):
//     E_s = 5u (|h6| X + |h7| Y + |h8|)          bounds |s_fp32 - s|   (two fmas on rounded inputs: (1+u)^4 - 1 < 5u)
//     E_n = 5u max(|h0| X + |h1| Y + |h2|, |h3| X + |h4| Y + |h5|)     the same for both numerators
// per PAIR, from the FP32 values (sigma = |s_fp32|, r = rcp(s_fp32) with |r sigma - 1| <= 3u, m = max(|u|, |v|)):
//     a pair is only decided in FP32 if sigma >= 64 E_s  (then the true |s| >= 63/64 sigma and the quotient's error is
//     first-order); the quotient n/s computed as n_fp32 * r then errs by at most
//         E_q = 1.1 (E_n + m E_s) |r| + 5u m                                     (1.1 covers 64/63, the reciprocal's
//                                                                                 3u and the second-order term)
//     dx = x2 - u errs by   E = E_q + u max(|x2|, |y2|) + 1.01u max(|dx|, |dy|)  (input rounding, the subtraction)
//     d2 = dx^2 + dy^2 errs by   B32 <= 2 E (2 max(|dx|, |dy|) + E) + 2.2u d2
// The FP64 value the reference computes differs from the exact one by the same expressions with 2^-53 for u (a few more
// roundings, no fma): less than 2^-27 B32.  B = 1.01 B32 covers that and the rounding of the bound's own evaluation (a
// dozen FP32 operations, every term non-negative).  Models or points outside the magnitudes for which "relative error
// u per operation" holds (overflow, underflow to subnormals) are not eligible: a model with a coefficient >= 2^100, not
// finite, or with E_s or E_n below 2^-80 gets tau = NaN (every comparison against it is false) and all its pairs go to FP64; the launcher uses this kernel
// only when every coordinate is finite and below 2^20 in magnitude.  A NaN anywhere makes both comparisons false, which
// also sends the pair to FP64.
//
// Most pairs are nowhere near the threshold — a random hypothesis maps a point hundreds of pixels from its match — and
// for them a much cheaper sufficient test decides "not an inlier" before d2, or even a quotient, is formed.  It works on
//         Wx = x2 s - nx = s dx,     Wy = y2 s - ny = s dy          (one fma each; no reciprocal)
// With Cmax = max |x2|, |y2| over all points, the computed Wx^ = fl(x2~ s~ - nx~) satisfies
//         |Wx^ - Wx| <= Cmax (1+u) E_s + E_n + u Cmax |s| + 1.01u |Wx^|         (s~, nx~ as above, x2~ = fl32(x2), the fma's rounding)
// so with the per-model constant A = 1.01 (Cmax E_s + E_n) and W^ = max(|Wx^|, |Wy^|), if
//         sigma >= 64 E_s,     W^ >= 1.12 thr sigma,     W^ >= 25 A,     W^ >= 25.4 u Cmax sigma
// then (|s| <= 65/64 sigma) the true max(|dx|, |dy|) = max(|Wx|, |Wy|) / |s| is at least
//         W^ (1 - 1.01u - 0.04 - 0.04) / (65/64 sigma) >= 0.9057 W^ / sigma >= 1.014 thr
// and the true d2 at least 1.028 thr^2 — a margin of 2.8 % against the 2^-50 by which the reference's own roundings can move
// d2.  (The factor was 2.5 at first; hypotheses fitted to four matches are often nearly right for a whole plane, 3 % of
// the pairs of a DLT batch lie within 5 pixels, and every pair that fails this test costs the full bound below.)  The second and fourth condition are one comparison against k1 sigma with the
// per-launch constant k1 = max(1.12 thr, 25.4 u Cmax) (rounded up); the third against the per-model constant 25.2 A.
// Products cannot overflow: eligible models have |h6| X + |h7| Y + |h8| and both numerators' sums below 2^100 and
// coordinates are below 2^20; they do not underflow into the subnormals either where it matters: W^ >= k1 sigma with
// sigma >= 2^-74 (E_s >= 2^-80) and thr^2 >= 2^-40 (the launcher's precondition) is a normal number.  When all 64 lanes
// of a wave pass this for a pair, the wave moves on (a wave-uniform branch); otherwise the pair takes the full bound.
// (Until late r03 this test was formed on the quotient, with a reciprocal and two multiplies more per pair.)
//
// An FP32 instruction with a scalar-register operand issues in 4 cycles, with vector operands only in 2
// (tools/ubench/valu_cost.hip), so the per-model constants are staged in LDS once per workgroup and broadcast into
// VGPRs per model (LDS instructions do not take VALU issue slots).
//
// Work split as k_residual: a 256-thread workgroup owns MC models (64 by default) and sweeps a slice of the points, a lane
// holds PPL = 4 points.
#include "mh_kernels.hpp"
#include "mh_device.hpp"

#include <cmath>

namespace mh {

constexpr float U32 = 5.9604644775390625e-08f;       // 2^-24

// per model: 9 coefficients in FP32, then 1.1 E_s, 1.1 E_n, tau = 64 E_s (or NaN: not eligible), 25.2 A (the cheap test), 3 pad
__global__ void __launch_bounds__(256)
k_model32(const double* __restrict__ H, int M, double X, double Y, double Cmax, float* __restrict__ out)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const double* h = H + 9 * (size_t)m;
    float* o = out + 16 * (size_t)m;
    bool ok = true;
    for (int q = 0; q < 9; ++q) {
        ok = ok && fabs(h[q]) < 0x1p100;              // (false for NaN / inf)
        o[q] = (float)h[q];
    }
    const double u = 0x1p-24;
    const double as = fabs(h[6]) * X + fabs(h[7]) * Y + fabs(h[8]);
    const double an = fmax(fabs(h[0]) * X + fabs(h[1]) * Y + fabs(h[2]), fabs(h[3]) * X + fabs(h[4]) * Y + fabs(h[5]));
    const double es = 5.0 * u * as, en = 5.0 * u * an;
    ok = ok && es >= 0x1p-80 && en >= 0x1p-80 && as < 0x1p100 && an < 0x1p100;     // (as, an bound every product formed from this model)
    const double up = 1.0 + 0x1p-22;                  // round the bounds UP on their way to FP32
    o[9] = (float)(1.1 * es * up);
    o[10] = (float)(1.1 * en * up);
    o[11] = ok ? (float)(64.0 * es * up) : NAN;        // NaN, not +inf: an overflowed |s| = inf must not pass "|s| >= tau"
    o[12] = (float)(25.2 * 1.01 * (Cmax * es + en) * up);      // 25.2 A (25 = 1 / 0.04, the rest covers the test's own rounding)
    o[13] = o[14] = o[15] = 0.f;
}

// score32_wg: the work of ONE workgroup — model block bx (MC models), point slice by.
template <int PPL, int MC, bool MASK>
__device__ __forceinline__ void
score32_wg(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
           const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
           double thr2, float thr2_f, float c_thr, float k1, int* __restrict__ counts, const unsigned char* __restrict__ mask,
           int psplit, unsigned long long* __restrict__ fallback_pairs, const int bx, const int by)
{
    constexpr int WAVE_PTS = 64 * PPL, TILE = 4 * WAVE_PTS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = bx * MC;
    __shared__ float4 s_m[MC * 4];               // this workgroup's rows of the model table
    for (int i = threadIdx.x; i < MC * 4; i += 256) {
        const size_t g = (size_t)m0 * 4 + i;
        s_m[i] = g < (size_t)M * 4 ? reinterpret_cast<const float4*>(H32)[g] : make_float4(0.f, 0.f, 0.f, NAN);
    }
    __syncthreads();
    // kernel-argument constants that enter FP32 instructions: VGPR copies, made once
    float vthr2, vc_thr, vk1;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vthr2) : "s"(thr2_f));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vc_thr) : "s"(c_thr));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vk1) : "s"(k1));
    int cnt = 0;                                 // lane mi of each wave counts model m0 + mi
    unsigned long long fb = 0;                   // pairs this lane sent to FP64 (diagnostic)
    for (int base = by * TILE; base < N; base += psplit * TILE) {
        const int n0 = base + wave * WAVE_PTS + lane * PPL;
        // only the FP32 copies of the points stay in registers; the rare FP64 decision reloads its point (L2-resident)
        float fx[PPL], fy[PPL], gx[PPL], gy[PPL], cx[PPL];
        unsigned long long okm[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int n = n0 + q;
            bool ok = n < N;
            const double px = ok ? x1[n] : 1.0, py = ok ? y1[n] : 1.0, qx = ok ? x2[n] : 1.0, qy = ok ? y2[n] : 1.0;
            if (MASK && ok) ok = mask[n] != 0;
            okm[q] = __builtin_amdgcn_ballot_w64(ok);
            fx[q] = (float)px; fy[q] = (float)py; gx[q] = (float)qx; gy[q] = (float)qy;
            cx[q] = U32 * fmaxf(fabsf(gx[q]), fabsf(gy[q])) * 1.0000002f;
        }
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            const int m = m0 + mi;
            if (m >= M) break;
            // broadcast LDS reads: every lane gets the model's constants in VGPRs
            const float4 ma = s_m[4 * mi], mb = s_m[4 * mi + 1], mc = s_m[4 * mi + 2], md = s_m[4 * mi + 3];
            const float h0 = ma.x, h1 = ma.y, h2 = ma.z, h3 = ma.w, h4 = mb.x, h5 = mb.y, h6 = mb.z, h7 = mb.w, h8 = mc.x;
            const float es = mc.y, en = mc.z, tau = mc.w, a25 = md.x;
            // Pass one, all PPL pairs: the cheap test.  Nothing but the PPL lane masks survives it, so it needs few registers.
            unsigned long long farq[PPL], all_far = ~0ull;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const float s = __builtin_fmaf(h6, fx[q], __builtin_fmaf(h7, fy[q], h8));
                const float nx = __builtin_fmaf(h0, fx[q], __builtin_fmaf(h1, fy[q], h2));
                const float ny = __builtin_fmaf(h3, fx[q], __builtin_fmaf(h4, fy[q], h5));
                const float wx = __builtin_fmaf(gx[q], s, -nx), wy = __builtin_fmaf(gy[q], s, -ny);      // s dx, s dy
                const float W = fmaxf(fabsf(wx), fabsf(wy));
                farq[q] = __builtin_amdgcn_ballot_w64(fabsf(s) >= tau) &
                          __builtin_amdgcn_ballot_w64(W >= fmaxf(vk1 * fabsf(s), a25));
                all_far &= farq[q];
            }
            int c_m = 0;
            if (all_far != ~0ull) {
                // Pass two, only for the pairs in which some lane is not provably far out: the full bound (the few FP32
                // operations of pass one are simply done again; this is the rare path)
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    if (farq[q] == ~0ull) continue;
                    asm volatile("; score32: full bound");           // (keeps the two passes' arithmetic apart)
                    const float s = __builtin_fmaf(h6, fx[q], __builtin_fmaf(h7, fy[q], h8));
                    const float nx = __builtin_fmaf(h0, fx[q], __builtin_fmaf(h1, fy[q], h2));
                    const float ny = __builtin_fmaf(h3, fx[q], __builtin_fmaf(h4, fy[q], h5));
                    const float r = __builtin_amdgcn_rcpf(s);
                    const float uu = nx * r, vv = ny * r;
                    const float dx = gx[q] - uu, dy = gy[q] - vv;
                    const float w = fmaxf(fabsf(dx), fabsf(dy));
                    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                    // the bound (every term >= 0); B0 = everything but the 2.2u d2 term, which the constant c_thr absorbs: a pair
                    // decided as inlier has d2 < thr^2, and "d2 - B0 >= thr^2 (1 + 3u)" implies "d2 (1 - 2.2u) - B0 >= thr^2"
                    const float mm = fmaxf(fabsf(uu), fabsf(vv));
                    const float eq = __builtin_fmaf(__builtin_fmaf(mm, es, en), fabsf(r), (5.0f * U32) * mm);
                    const float E = __builtin_fmaf(1.01f * U32, w, eq + cx[q]);
                    const float B0 = __builtin_fmaf(2.02f * E, __builtin_fmaf(2.0f, w, E), vc_thr);
                    const float t = d2 - vthr2;
                    // lane masks straight from the compares; the logic on them is scalar
                    const unsigned long long trust = __builtin_amdgcn_ballot_w64(fabsf(s) >= tau);
                    const unsigned long long clear = __builtin_amdgcn_ballot_w64(fabsf(t) > B0);       // (false for NaN)
                    const unsigned long long below = __builtin_amdgcn_ballot_w64(t < 0.0f);
                    const unsigned long long decided = (trust & clear) | farq[q];
                    unsigned long long inl = trust & clear & below;
                    if (decided != ~0ull) {                               // rarer still: some lane's pair is too close to call
                        const bool need = ((decided >> lane) & 1ull) == 0ull;
                        bool in64 = false;
                        if (need) {
                            const double* h = H + 9 * (size_t)m;
                            const int n = n0 + q < N ? n0 + q : N - 1;      // (a padding lane: its result is masked out by okm)
                            const double e2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], x1[n], y1[n], x2[n], y2[n]);
                            in64 = e2 < thr2;
                            ++fb;
                        }
                        inl |= __builtin_amdgcn_ballot_w64(in64);
                    }
                    c_m += __builtin_popcountll(inl & okm[q]);
                }
            }
            const int c_new = __builtin_amdgcn_readlane(cnt, mi) + c_m;
            asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(cnt) : "s"(c_new), "s"(mi) : "m0");
        }
    }
    __shared__ int s_cnt[4][MC];
    if (lane < MC) s_cnt[wave][lane] = cnt;
    __syncthreads();
    if (threadIdx.x < MC && m0 + (int)threadIdx.x < M) {
        const int t = threadIdx.x;
        const int c = s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
        if (psplit == 1) counts[m0 + t] = c;
        else atomicAdd(&counts[m0 + t], c);
    }
    if (fallback_pairs) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) fb += __shfl_xor(fb, o, 64);
        if (lane == 0 && fb) atomicAdd(fallback_pairs, fb);
    }
}

template <int PPL, int MC, bool MASK, int MINW = 1>
__global__ void __launch_bounds__(256, MINW)
k_score32(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
          const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
          double thr2, float thr2_f, float c_thr, float k1, int* __restrict__ counts, const unsigned char* __restrict__ mask,
          int psplit, unsigned long long* __restrict__ fallback_pairs)
{
    score32_wg<PPL, MC, MASK>(x1, y1, x2, y2, N, H, H32, M, thr2, thr2_f, c_thr, k1, counts, mask, psplit, fallback_pairs, blockIdx.x, blockIdx.y);
}

// The same work items walked by a resident grid that hands them out through a counter (as k_residual_resident, residual.hip).
template <int PPL, int MC, bool MASK, int MINW = 1>
__global__ void __launch_bounds__(256, MINW)
k_score32_resident(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
                   const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
                   double thr2, float thr2_f, float c_thr, float k1, int* __restrict__ counts, const unsigned char* __restrict__ mask,
                   int psplit, unsigned long long* __restrict__ fallback_pairs, int gx, int nitems, int* __restrict__ ctl)
{
    __shared__ int s_item;
#pragma unroll 1
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&ctl[0], 1);
        __syncthreads();
        const int item = s_item;
        if (item >= nitems) break;
        const int by = item / gx, bx = item - by * gx;
        score32_wg<PPL, MC, MASK>(x1, y1, x2, y2, N, H, H32, M, thr2, thr2_f, c_thr, k1, counts, mask, psplit, fallback_pairs, bx, by);
        __syncthreads();
    }
    if (threadIdx.x == 0 && atomicAdd(&ctl[1], 1) == (int)gridDim.x - 1) {
        ctl[1] = 0;
        __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------
// k_cost32 — the materialised int32 data-cost matrix (k_cost_matrix of datacost.hip: the s = 4 variant of SURVEY 8(d))
// behind the same pre-test.  dataEnergy (M/MultiH.cpp:473-504) of a pair whose d2 is at least T = thr^2 81/16 is the
// constant 2 round(lam T), and for a random hypothesis that is nearly every pair: the cheap test above with
// k1 = max(1.12 x 9/4 thr, 25.4 u Cmax) (launch_cost32) proves max(|dx|, |dy|) >= 1.014 x 9/4 thr, i.e. d2 >= 1.028 T — a
// 2.8 % margin beyond the truncation threshold — per LANE; the other lanes (both |dx| and |dy| within 1.12 x 9/4 thr, models
// not eligible for FP32, NaN anywhere) evaluate the reference's
// FP64 formula — fwd_d2, the IEEE division d2 / T, C round() — exactly as k_cost_matrix does.  Same matrix, same fused
// inlier counts, bit for bit.  Measured at 50k x 100k DLT hypotheses: 7.7 -> 4.2 ms (the store stream alone would take 3.6 ms:
// 3.3 % of the pairs of such a batch are near — hypotheses fitted to four matches are often nearly right for a whole
// plane — and 42 % of the wave-model iterations contain one, each costing a pass through the IEEE formula).
// ---------------------------------------------------------------------------
// Lanes of ONE wave hand data to each other through LDS below (no workgroup barrier: the other waves are not involved).
// The LDS operations of a wave complete in program order, so all that is needed is that the COMPILER keeps the writes in
// front of the cross-lane reads: a compiler-level memory barrier and a wave barrier (a scheduling fence; no instruction).
// (r03 advisor finding.  A wave-scope release / acquire fence pair was tried first: the backend emits s_waitcnt for it and
// the kernel went from 4.3 to 4.9 ms.)
__device__ __forceinline__ void wave_lds_handover()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// cost32_wg: the work of ONE workgroup — model block bx (MC models), point slice by.
template <int MC, int WAVES>
__device__ __forceinline__ void
cost32_wg(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
          const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
          double lam, double T, double thr2, float k1, int* __restrict__ C, long long ldc, int* __restrict__ counts, int psplit,
          const int bx, const int by)
{
    constexpr int PPL = 4, WAVE_PTS = 64 * PPL, TILE = WAVES * WAVE_PTS, THREADS = 64 * WAVES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = bx * MC;
    __shared__ float4 s_m[MC * 4];
    __shared__ double s_h[MC * 9];               // the FP64 coefficients, for the lanes that need the reference's formula
    __shared__ double s_p[WAVES * PPL * 4 * 64];     // [wave][point of the lane][x1 y1 x2 y2][lane]: every lane's own points in FP64
    for (int i = threadIdx.x; i < MC * 4; i += THREADS) {
        const size_t g = (size_t)m0 * 4 + i;
        s_m[i] = g < (size_t)M * 4 ? reinterpret_cast<const float4*>(H32)[g] : make_float4(0.f, 0.f, 0.f, NAN);
    }
    for (int i = threadIdx.x; i < MC * 9; i += THREADS) {
        const size_t g = (size_t)m0 * 9 + i;
        s_h[i] = g < (size_t)M * 9 ? H[g] : 0.0;
    }
    __syncthreads();
    double* wave_p = s_p + (size_t)wave * (PPL * 4 * 64);           // + (q * 4 + component) * 64 + lane
    double* my_p = wave_p + lane;
    __shared__ unsigned char s_list[WAVES * 64 * PPL];                 // per wave: the (point slot, lane) of the pairs that need FP64
    __shared__ int s_c[WAVES * 64 * PPL];                               // per wave: their costs, on the way back to the owning lane
    unsigned char* my_list = s_list + wave * (64 * PPL);
    int* my_c = s_c + wave * (64 * PPL);
    float vk1;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vk1) : "s"(k1));
    const int beyond = 2 * (int)round(lam * T);
    int cnt = 0;                                 // lane mi of each wave counts model m0 + mi
    for (int base = by * TILE; base < N; base += psplit * TILE) {
        const int base_n = base + wave * WAVE_PTS;
        const int n0 = base_n + lane * PPL;
        float fx[PPL], fy[PPL], gx[PPL], gy[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int n = n0 + q;
            const bool ok = n < N;
            const double px = ok ? x1[n] : 1.0, py = ok ? y1[n] : 1.0, qx = ok ? x2[n] : 1.0, qy = ok ? y2[n] : 1.0;
            my_p[(q * 4 + 0) * 64] = px; my_p[(q * 4 + 1) * 64] = py; my_p[(q * 4 + 2) * 64] = qx; my_p[(q * 4 + 3) * 64] = qy;
            fx[q] = (float)px; fy[q] = (float)py; gx[q] = (float)qx; gy[q] = (float)qy;
        }
        wave_lds_handover();                     // the FP64 copies are read by OTHER lanes of this wave below
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            const int m = m0 + mi;
            if (m >= M) break;
            const float4 ma = s_m[4 * mi], mb = s_m[4 * mi + 1], mc = s_m[4 * mi + 2], md = s_m[4 * mi + 3];
            const float h0 = ma.x, h1 = ma.y, h2 = ma.z, h3 = ma.w, h4 = mb.x, h5 = mb.y, h6 = mb.z, h7 = mb.w, h8 = mc.x;
            const float tau = mc.w, a25 = md.x;
            int c[PPL];
            unsigned long long nearq[PPL], any_near = 0ull;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const float s = __builtin_fmaf(h6, fx[q], __builtin_fmaf(h7, fy[q], h8));
                const float nx = __builtin_fmaf(h0, fx[q], __builtin_fmaf(h1, fy[q], h2));
                const float ny = __builtin_fmaf(h3, fx[q], __builtin_fmaf(h4, fy[q], h5));
                const float wx = __builtin_fmaf(gx[q], s, -nx), wy = __builtin_fmaf(gy[q], s, -ny);
                const float W = fmaxf(fabsf(wx), fabsf(wy));
                nearq[q] = ~(__builtin_amdgcn_ballot_w64(fabsf(s) >= tau) &
                             __builtin_amdgcn_ballot_w64(W >= fmaxf(vk1 * fabsf(s), a25)));
                any_near |= nearq[q];
                c[q] = beyond;
            }
            int c_m = 0;
            if (any_near) {
                // The near pairs of the wave's 256 (typically a few dozen, spread over all four point slots) are packed into
                // a list in LDS and evaluated 64 at a time by whichever lanes come first — a lane works on other lanes'
                // points, which is why every lane's FP64 copies live in LDS — instead of one masked pass per point slot.
                asm volatile("; cost32: FP64 pairs");
                int npairs = 0;
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    if ((nearq[q] >> lane) & 1ull) {
                        const int pos = npairs + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(nearq[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nearq[q], 0u));
                        my_list[pos] = (unsigned char)(q * 64 + lane);
                    }
                    npairs += __builtin_popcountll(nearq[q]);
                }
                wave_lds_handover();             // the list is read across lanes
                const double* h = s_h + 9 * mi;
                const double h0d = h[0], h1d = h[1], h2d = h[2], h3d = h[3], h4d = h[4], h5d = h[5], h6d = h[6], h7d = h[7], h8d = h[8];
                for (int p0 = 0; p0 < npairs; p0 += 64) {
                    bool in64 = false;
                    if (p0 + lane < npairs) {
                        const int id = my_list[p0 + lane], q2 = id >> 6, l2 = id & 63;
                        const double* pp = wave_p + (q2 * 4) * 64 + l2;
                        const double d2 = fwd_d2(h0d, h1d, h2d, h3d, h4d, h5d, h6d, h7d, h8d, pp[0], pp[64], pp[128], pp[192]);
                        my_c[id] = d2 < T ? (int)round(lam * (1.0 - (d2 / T))) : beyond;
                        in64 = (base_n + l2 * PPL + q2 < N) && d2 < thr2;
                    }
                    c_m += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in64));
                }
                wave_lds_handover();             // the costs go back to the lanes that own the pairs
#pragma unroll
                for (int q = 0; q < PPL; ++q)
                    if ((nearq[q] >> lane) & 1ull) c[q] = my_c[q * 64 + lane];
                wave_lds_handover();             // (the next model's list and costs overwrite these)
            }
            int* dst = C + (size_t)m * ldc + n0;
            if (n0 + 3 < N) {
                typedef int i4v __attribute__((ext_vector_type(4)));
                const i4v v = { c[0], c[1], c[2], c[3] };
                __builtin_nontemporal_store(v, reinterpret_cast<i4v*>(dst));
            }
            else
                for (int q = 0; q < PPL; ++q) if (n0 + q < N) dst[q] = c[q];
            if (any_near) {
                const int c_new = __builtin_amdgcn_readlane(cnt, mi) + c_m;
                asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(cnt) : "s"(c_new), "s"(mi) : "m0");
            }
        }
    }
    __shared__ int s_cnt[WAVES][MC];
    if (lane < MC) s_cnt[wave][lane] = cnt;
    __syncthreads();
    if (threadIdx.x < MC && m0 + (int)threadIdx.x < M) {
        const int t = threadIdx.x;
        int cc = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) cc += s_cnt[w][t];
        if (psplit == 1) counts[m0 + t] = cc;
        else atomicAdd(&counts[m0 + t], cc);
    }
}

// cost32_wg_batched (r04 EXPERIMENT, mh_set_tuning key 28 = 1; not the default): the same matrix with the near pairs of
// SEVERAL models evaluated together.  Measured at 50k x 100k: the arithmetic side gains what was expected (4.13 -> 3.56 ms
// when the near pairs' costs are computed but not delivered), the delivery loses more: 4.45 ms with plain stores of the
// constant and 5.95 ms with non-temporal ones — a 4-byte store into a line that has already left L2 is a read-modify-write
// in memory, 150 million of them per launch.  Delivering through LDS to the owning lane before ITS store (the default
// form) needs the costs of all pending models in LDS, which the 64 KB of FP64 point copies leave hardly any room for: a
// second version (up to eight models pending, their lane masks and 272 list entries in the 1 376 bytes per wave that two
// workgroups per compute unit leave, rows written once with the costs fetched by rank in the mask) ran 4.36 ms — the
// bookkeeping of the pending rows costs what the fuller passes save.  In cost32_wg every
// wave x model iteration that contains a near pair (42 % of them on a DLT batch) pays a pass through the IEEE formula with,
// typically, a few dozen of its 64 lanes at work.  Here a wave writes the constant row segment at once, appends its near
// pairs — (model, point slot, lane) — to a list in LDS, and runs the formula only when 64 entries have gathered (and once at
// the end of a tile): full lanes, one eighth of the passes.  The cost of a near pair then goes straight to C[m][n], over
// the constant written before: the wave drains its stores (s_waitcnt vmcnt(0)) before the first such store of a batch, so
// the two writes of an address arrive in order.  Inlier counts go through an LDS counter per (wave, model).
template <int MC, int WAVES>
__device__ __forceinline__ void
cost32_wg_batched(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
                  const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
                  double lam, double T, double thr2, float k1, int* __restrict__ C, long long ldc, int* __restrict__ counts, int psplit,
                  const int bx, const int by)
{
    constexpr int PPL = 4, WAVE_PTS = 64 * PPL, TILE = WAVES * WAVE_PTS, THREADS = 64 * WAVES, LCAP = 64 + 64 * PPL;
    static_assert(MC <= 64, "a list entry carries the model in 6 bits");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = bx * MC;
    __shared__ float4 s_m[MC * 4];
    __shared__ double s_h[MC * 9];
    __shared__ double s_p[WAVES * PPL * 4 * 64];          // [wave][point of the lane][x1 y1 x2 y2][lane]
    __shared__ unsigned short s_list[WAVES * LCAP];       // per wave: (model << 8 | point slot << 6 | lane) of the pending near pairs
    __shared__ int s_cnt[WAVES][MC];
    for (int i = threadIdx.x; i < MC * 4; i += THREADS) {
        const size_t g = (size_t)m0 * 4 + i;
        s_m[i] = g < (size_t)M * 4 ? reinterpret_cast<const float4*>(H32)[g] : make_float4(0.f, 0.f, 0.f, NAN);
    }
    for (int i = threadIdx.x; i < MC * 9; i += THREADS) {
        const size_t g = (size_t)m0 * 9 + i;
        s_h[i] = g < (size_t)M * 9 ? H[g] : 0.0;
    }
    for (int i = threadIdx.x; i < WAVES * MC; i += THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    double* wave_p = s_p + (size_t)wave * (PPL * 4 * 64);
    double* my_p = wave_p + lane;
    unsigned short* my_list = s_list + wave * LCAP;
    float vk1;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vk1) : "s"(k1));
    const int beyond = 2 * (int)round(lam * T);
    typedef int i4v __attribute__((ext_vector_type(4)));
    const i4v vbeyond = { beyond, beyond, beyond, beyond };
    for (int base = by * TILE; base < N; base += psplit * TILE) {
        const int base_n = base + wave * WAVE_PTS;
        const int n0 = base_n + lane * PPL;
        float fx[PPL], fy[PPL], gx[PPL], gy[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int n = n0 + q;
            const bool ok = n < N;
            const double px = ok ? x1[n] : 1.0, py = ok ? y1[n] : 1.0, qx = ok ? x2[n] : 1.0, qy = ok ? y2[n] : 1.0;
            my_p[(q * 4 + 0) * 64] = px; my_p[(q * 4 + 1) * 64] = py; my_p[(q * 4 + 2) * 64] = qx; my_p[(q * 4 + 3) * 64] = qy;
            fx[q] = (float)px; fy[q] = (float)py; gx[q] = (float)qx; gy[q] = (float)qy;
        }
        wave_lds_handover();                     // the FP64 copies are read by OTHER lanes of this wave below
        // One pass of the IEEE formula over the list entries [first, first + 64) (those below `count`).
        auto evaluate = [&](int first, int count) {
            if (first + lane < count) {
                const int id = my_list[first + lane], mi2 = id >> 8, q2 = (id >> 6) & 3, l2 = id & 63;
                const double* h = s_h + 9 * mi2;
                const double* pp = wave_p + (q2 * 4) * 64 + l2;
                const double d2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], pp[0], pp[64], pp[128], pp[192]);
                const int cost = d2 < T ? (int)round(lam * (1.0 - (d2 / T))) : beyond;
                const int n = base_n + l2 * PPL + q2;
                if (n < N) {
                    C[(size_t)(m0 + mi2) * ldc + n] = cost;
                    if (d2 < thr2) atomicAdd(&s_cnt[wave][mi2], 1);
                }
            }
        };
        int nlist = 0;                           // wave-uniform
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            const int m = m0 + mi;
            if (m >= M) break;
            const float4 ma = s_m[4 * mi], mb = s_m[4 * mi + 1], mc = s_m[4 * mi + 2], md = s_m[4 * mi + 3];
            const float h0 = ma.x, h1 = ma.y, h2 = ma.z, h3 = ma.w, h4 = mb.x, h5 = mb.y, h6 = mb.z, h7 = mb.w, h8 = mc.x;
            const float tau = mc.w, a25 = md.x;
            unsigned long long nearq[PPL], any_near = 0ull;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const float s = __builtin_fmaf(h6, fx[q], __builtin_fmaf(h7, fy[q], h8));
                const float nx = __builtin_fmaf(h0, fx[q], __builtin_fmaf(h1, fy[q], h2));
                const float ny = __builtin_fmaf(h3, fx[q], __builtin_fmaf(h4, fy[q], h5));
                const float wx = __builtin_fmaf(gx[q], s, -nx), wy = __builtin_fmaf(gy[q], s, -ny);
                const float W = fmaxf(fabsf(wx), fabsf(wy));
                nearq[q] = ~(__builtin_amdgcn_ballot_w64(fabsf(s) >= tau) &
                             __builtin_amdgcn_ballot_w64(W >= fmaxf(vk1 * fabsf(s), a25)));
                any_near |= nearq[q];
            }
            int* dst = C + (size_t)m * ldc + n0;
            if (n0 + 3 < N) *reinterpret_cast<i4v*>(dst) = vbeyond;      // (a plain store: the line stays in L2 for the near pairs' costs)
            else
                for (int q = 0; q < PPL; ++q) if (n0 + q < N) dst[q] = beyond;
            if (any_near) {
                asm volatile("; cost32: near pairs to the list");
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    if ((nearq[q] >> lane) & 1ull) {
                        const int pos = nlist + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(nearq[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nearq[q], 0u));
                        my_list[pos] = (unsigned short)((mi << 8) | (q << 6) | lane);
                    }
                    nlist += __builtin_popcountll(nearq[q]);
                }
                if (nlist >= 64) {
                    wave_lds_handover();         // the list is read across lanes
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the constants these pairs overwrite have landed
                    do {
                        nlist -= 64;
                        evaluate(nlist, nlist + 64);
                    } while (nlist >= 64);
                    wave_lds_handover();         // (the next entries overwrite what was just read)
                }
            }
        }
        if (nlist > 0) {
            wave_lds_handover();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int p0 = 0; p0 < nlist; p0 += 64) evaluate(p0, nlist);
        }
        wave_lds_handover();                     // (the next tile's points and list overwrite these)
    }
    __syncthreads();
    if (threadIdx.x < MC && m0 + (int)threadIdx.x < M) {
        const int t = threadIdx.x;
        int cc = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) cc += s_cnt[w][t];
        if (psplit == 1) counts[m0 + t] = cc;
        else atomicAdd(&counts[m0 + t], cc);
    }
}

template <int MC, int WAVES, bool BATCH>
__global__ void __launch_bounds__(64 * WAVES)
k_cost32(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
         const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
         double lam, double T, double thr2, float k1, int* __restrict__ C, long long ldc, int* __restrict__ counts, int psplit)
{
    if (BATCH) cost32_wg_batched<MC, WAVES>(x1, y1, x2, y2, N, H, H32, M, lam, T, thr2, k1, C, ldc, counts, psplit, blockIdx.x, blockIdx.y);
    else cost32_wg<MC, WAVES>(x1, y1, x2, y2, N, H, H32, M, lam, T, thr2, k1, C, ldc, counts, psplit, blockIdx.x, blockIdx.y);
}

// The same work items walked by a resident grid that hands them out through a counter (as k_residual_resident, residual.hip).
template <int MC, int WAVES, bool BATCH>
__global__ void __launch_bounds__(64 * WAVES)
k_cost32_resident(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
                  const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
                  double lam, double T, double thr2, float k1, int* __restrict__ C, long long ldc, int* __restrict__ counts, int psplit,
                  int gx, int nitems, int* __restrict__ ctl, int slice_major)
{
    __shared__ int s_item;
#pragma unroll 1
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&ctl[0], 1);
        __syncthreads();
        const int item = s_item;
        if (item >= nitems) break;
        // slice_major: consecutive items are the point slices of one model block — the workgroups at work write a compact
        // window of C (tools/ubench/store_order.hip: the store stream alone gains 9 % from that order)
        int bx, by;
        if (slice_major) { bx = item / psplit; by = item - bx * psplit; }
        else { by = item / gx; bx = item - by * gx; }
        if (BATCH) cost32_wg_batched<MC, WAVES>(x1, y1, x2, y2, N, H, H32, M, lam, T, thr2, k1, C, ldc, counts, psplit, bx, by);
        else cost32_wg<MC, WAVES>(x1, y1, x2, y2, N, H, H32, M, lam, T, thr2, k1, C, ldc, counts, psplit, bx, by);
        __syncthreads();
    }
    if (threadIdx.x == 0 && atomicAdd(&ctl[1], 1) == (int)gridDim.x - 1) {
        ctl[1] = 0;
        __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// H32: the table launch_model32 made for these models with the same Cmax.  thr2 in [2^-40, 2^40], coordinates below 2^20.
template <bool BATCH>
static hipError_t launch_cost32_t(const Points& p, const double* H, const float* H32, int M, double lambda, double thr2, double Cmax,
                                  int* C, long long ldc, int* counts, hipStream_t s, int* resident_ctl, int cu_count, int psplit_override, int slice_major,
                                  int* occ_cache)
{
    if (M <= 0 || p.n <= 0) return hipSuccess;
    constexpr int MC = 32;
    // 512 threads: the FP64 copies of a wave's points take 8 KB of LDS, so two such workgroups (16 waves) fill a CU's 160 KB;
    // 256-thread workgroups got 12 waves onto a CU (4.7 ms), one 1 024-thread workgroup 16 again but 4.8 ms; this: 4.2 ms
    constexpr int WAVES = 8;
    const int gx = (M + MC - 1) / MC, ntiles = (p.n + 256 * WAVES - 1) / (256 * WAVES);
    int psplit = gx < 1024 ? (2048 + gx - 1) / gx : (ntiles >= 8 ? 2 : 1);
    if (psplit > ntiles) psplit = ntiles;
    if (psplit < 1) psplit = 1;
    if (psplit > 1) {
        hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)M, s);
        if (e != hipSuccess) return e;
    }
    // far = beyond T = (9/4 thr)^2 with a 2 % margin: the cheap test's k1 with 1.12 x 9/4 thr in place of 2.5 thr
    const float k1 = (float)(std::fmax(1.12 * 2.25 * std::sqrt(std::fabs(thr2)), 25.4 * 5.9604644775390625e-08 * Cmax) * (1.0 + 1e-6)) + 1e-30f;
    if (resident_ctl) {
        // resident grid: as many workgroups as the chip holds, ~37 500 items (r04 experiment: mh_set_tuning key 23)
        // workgroups a compute unit holds: asked once per ENGINE (the caller's cache; a function-local static would be shared
        // by every engine, device and host thread, and would keep a failed query for ever — r04 advisor finding)
        int per_cu = occ_cache ? *occ_cache : -1;
        if (per_cu < 0) {
            int q = 0;
            per_cu = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, (const void*)k_cost32_resident<MC, WAVES, BATCH>, 64 * WAVES, 0) == hipSuccess ? q : 0;
            if (per_cu > 0 && occ_cache) *occ_cache = per_cu;
        }
        const int grid = per_cu * cu_count;
        int ps = psplit_override > 0 ? psplit_override : (37500 + gx - 1) / gx;
        if (ps > ntiles) ps = ntiles;
        if (ps < 1) ps = 1;
        if (grid > 0 && gx * ps > grid) {
            hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)M, s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_cost32_resident<MC, WAVES, BATCH>), dim3(grid), dim3(64 * WAVES), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, H32, M,
                               100.0 / lambda, thr2 * 81.0 / 16.0, thr2, k1, C, ldc, counts, ps, gx, gx * ps, resident_ctl, slice_major);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((k_cost32<MC, WAVES, BATCH>), dim3(gx, psplit), dim3(64 * WAVES), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, H32, M, 100.0 / lambda,
                       thr2 * 81.0 / 16.0, thr2, k1, C, ldc, counts, psplit);
    return hipGetLastError();
}

hipError_t launch_cost32(const Points& p, const double* H, const float* H32, int M, double lambda, double thr2, double Cmax,
                         int* C, long long ldc, int* counts, hipStream_t s, int* resident_ctl, int cu_count, int psplit_override, int slice_major,
                         int batched, int* occ_cache)
{
    // batched: the experiment above (mh_set_tuning key 28) instead of the default form, in which every wave x model iteration
    // with a near pair runs the IEEE formula at once and hands the costs to the owning lanes through LDS
#ifdef MH_TUNING
    // (a measured-and-rejected variant: compiled into measurement libraries only, mh_set_tuning key 28; so are the
    // slice-major item order, key 27, and the other tilings of the score kernel below, key 16)
    if (batched) return launch_cost32_t<true>(p, H, H32, M, lambda, thr2, Cmax, C, ldc, counts, s, resident_ctl, cu_count, psplit_override, slice_major, nullptr);
#else
    (void)batched;
    slice_major = 0;
#endif
    return launch_cost32_t<false>(p, H, H32, M, lambda, thr2, Cmax, C, ldc, counts, s, resident_ctl, cu_count, psplit_override, slice_major, occ_cache);
}

hipError_t launch_model32(const double* H, int M, double X, double Y, double Cmax, float* H32, hipStream_t s)
{
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_model32, dim3((M + 255) / 256), dim3(256), 0, s, H, M, X, Y, Cmax, H32);
    return hipGetLastError();
}

template <int PPL, int MC, int MINW = 1>
static hipError_t launch_score32_t(const Points& p, const double* H, const float* H32, int M, double thr2, double Cmax, const unsigned char* mask,
                                   int* counts, unsigned long long* fallback_pairs, hipStream_t s, int* resident_ctl = nullptr,
                                   int cu_count = 256, int resident_slices = 0, int* occ_cache = nullptr)
{
    const int gx = (M + MC - 1) / MC, ntiles = (p.n + 256 * PPL - 1) / (256 * PPL);
    int psplit = gx < 1024 ? (2048 + gx - 1) / gx : (ntiles >= 16 ? 4 : 1);
    int resident = 0;
    if (resident_ctl && resident_slices != 0 && !mask) {
        int per_cu = occ_cache ? *occ_cache : -1;              // per engine, and a failed query is not kept (see launch_cost32_t)
        if (per_cu < 0) {
            int q = 0;
            per_cu = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, (const void*)k_score32_resident<PPL, MC, false, MINW>, 256, 0) == hipSuccess ? q : 0;
            if (per_cu > 0 && occ_cache) *occ_cache = per_cu;
        }
        int ps = resident_slices > 0 ? resident_slices : (37500 + gx - 1) / gx;
        if (ps > ntiles) ps = ntiles;
        if (ps < 1) ps = 1;
        if (per_cu * cu_count > 0 && gx * ps > per_cu * cu_count) { resident = per_cu * cu_count; psplit = ps; }
    }
    if (psplit > ntiles) psplit = ntiles;
    if (psplit < 1) psplit = 1;
    if (psplit > 1) {
        hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)M, s);
        if (e != hipSuccess) return e;
    }
    // the threshold in FP32 and the constant part of the bound: the rounding of the threshold itself (one ulp covers it
    // either way) plus the 2.2u d2 term for d2 up to thr^2 (1 + 3u) (see the kernel)
    const float tf = (float)thr2;
    const float c_thr = (float)(std::fabs((double)tf - thr2) * 1.01 + 3.5 * 5.9604644775390625e-08 * std::fabs(thr2) * 1.01) + 1e-45f;
    // the cheap test's k1 = max(1.12 thr, 25.4 u Cmax), rounded up (the product k1 sigma is rounded once more in the kernel)
    const float k1 = (float)(std::fmax(1.12 * std::sqrt(std::fabs(thr2)), 25.4 * 5.9604644775390625e-08 * Cmax) * (1.0 + 1e-6)) + 1e-30f;
    if (resident > 0) {
        hipLaunchKernelGGL((k_score32_resident<PPL, MC, false, MINW>), dim3(resident), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, H32, M, thr2, tf,
                           c_thr, k1, counts, mask, psplit, fallback_pairs, gx, gx * psplit, resident_ctl);
        return hipGetLastError();
    }
    if (mask) hipLaunchKernelGGL((k_score32<PPL, MC, true, MINW>), dim3(gx, psplit), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, H32, M, thr2, tf, c_thr, k1, counts, mask, psplit, fallback_pairs);
    else hipLaunchKernelGGL((k_score32<PPL, MC, false, MINW>), dim3(gx, psplit), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, H32, M, thr2, tf, c_thr, k1, counts, mask, psplit, fallback_pairs);
    return hipGetLastError();
}

// H32: the table launch_model32 made for these M models.  fallback_pairs (nullable): device counter of the pairs decided in
// FP64.  tiling: points per lane / models per workgroup (a schedule choice; the counts do not depend on it).
hipError_t launch_score32(const Points& p, const double* H, const float* H32, int M, double thr2, double Cmax, const unsigned char* mask,
                          int* counts, unsigned long long* fallback_pairs, int tiling, hipStream_t s, int* resident_ctl, int cu_count,
                          int resident_slices, int* occ_cache)
{
    if (M <= 0 || p.n <= 0) return hipSuccess;
    if (tiling == 0 && resident_ctl && resident_slices != 0)
        return launch_score32_t<4, 64, 6>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s, resident_ctl, cu_count, resident_slices, occ_cache);
#ifdef MH_TUNING
    switch (tiling) {
    case 1: return launch_score32_t<4, 16>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 2: return launch_score32_t<8, 32>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 3: return launch_score32_t<4, 32>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 4: return launch_score32_t<8, 64>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 5: return launch_score32_t<6, 32>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 6: return launch_score32_t<8, 16>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 7: return launch_score32_t<4, 64>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 8: return launch_score32_t<4, 32, 5>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 9: return launch_score32_t<4, 32, 6>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 10: return launch_score32_t<4, 32, 8>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 11: return launch_score32_t<2, 32, 8>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 12: return launch_score32_t<4, 64, 6>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 13: return launch_score32_t<4, 32>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 14: return launch_score32_t<4, 64, 5>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 15: return launch_score32_t<4, 64, 8>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    case 16: return launch_score32_t<2, 64, 8>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
    default: break;
    }
#endif
    // 4 points per lane, 64 models per workgroup, registers capped at 80 for six waves per SIMD (2.20 ms; <4, 32, 5> 2.26, uncapped 2.51)
    return launch_score32_t<4, 64, 6>(p, H, H32, M, thr2, Cmax, mask, counts, fallback_pairs, s);
}

} // namespace mh
