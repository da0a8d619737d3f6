// score32.hip — inlier scoring with an FP32 pre-test, gfx950.  Counts are the reference's, bit for bit.
//
// A score needs, per (point, model) pair, only the DECISION  d2 < thr^2  of the reference's FP64 formula
// (M/MultiH.cpp:434-443), never d2 itself.  This kernel evaluates the forward transfer error in FP32 (fused
// multiply-adds, hardware reciprocal: about a third of the FP64 sweep's issue cycles per pair) together with a RIGOROUS
// bound B on |d2_fp32 - d2_fp64|, decides the pairs for which the bound leaves no doubt
//         d2_fp32 + B <  thr^2   ->  inlier            d2_fp32 - B >= thr^2   ->  not an inlier
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
// finite, or with E_s or E_n below 2^-80 gets tau = +inf and all its pairs go to FP64; the launcher uses this kernel
// only when every coordinate is finite and below 2^20 in magnitude.  A NaN anywhere makes both comparisons false, which
// also sends the pair to FP64.
//
// Work split as k_residual: a 256-thread workgroup owns MC = 16 models and sweeps a slice of the points, a lane holds
// PPL = 4 points; the per-model constants come through the scalar unit (one s_load_dwordx16 per model).
#include "mh_kernels.hpp"
#include "mh_device.hpp"

namespace mh {

constexpr float U32 = 5.9604644775390625e-08f;       // 2^-24

// per model: 9 coefficients in FP32, then 1.1 E_s, 1.1 E_n, tau = 64 E_s (or +inf: not eligible), 4 pad
__global__ void __launch_bounds__(256)
k_model32(const double* __restrict__ H, int M, double X, double Y, float* __restrict__ out)
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
    ok = ok && es >= 0x1p-80 && en >= 0x1p-80 && es < 0x1p100 && en < 0x1p100;
    const double up = 1.0 + 0x1p-22;                  // round the bounds UP on their way to FP32
    o[9] = (float)(1.1 * es * up);
    o[10] = (float)(1.1 * en * up);
    o[11] = ok ? (float)(64.0 * es * up) : INFINITY;
    o[12] = o[13] = o[14] = o[15] = 0.f;
}

template <int PPL, int MC, bool MASK>
__global__ void __launch_bounds__(256)
k_score32(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
          const double* __restrict__ y2, int N, const double* __restrict__ H, const float* __restrict__ H32, int M,
          double thr2, float thr2_lo, float thr2_hi, int* __restrict__ counts, const unsigned char* __restrict__ mask,
          int psplit, unsigned long long* __restrict__ fallback_pairs)
{
    constexpr int WAVE_PTS = 64 * PPL, TILE = 4 * WAVE_PTS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * MC;
    int cnt = 0;                                 // lane mi of each wave counts model m0 + mi
    unsigned long long fb = 0;                   // pairs this lane sent to FP64 (diagnostic)
    for (int base = blockIdx.y * TILE; base < N; base += psplit * TILE) {
        const int n0 = base + wave * WAVE_PTS + lane * PPL;
        double px[PPL], py[PPL], qx[PPL], qy[PPL];
        float fx[PPL], fy[PPL], gx[PPL], gy[PPL], cx[PPL];
        unsigned long long okm[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int n = n0 + q;
            bool ok = n < N;
            px[q] = ok ? x1[n] : 1.0; py[q] = ok ? y1[n] : 1.0; qx[q] = ok ? x2[n] : 1.0; qy[q] = ok ? y2[n] : 1.0;
            if (MASK && ok) ok = mask[n] != 0;
            okm[q] = __builtin_amdgcn_ballot_w64(ok);
            fx[q] = (float)px[q]; fy[q] = (float)py[q]; gx[q] = (float)qx[q]; gy[q] = (float)qy[q];
            cx[q] = U32 * fmaxf(fabsf(gx[q]), fabsf(gy[q])) * 1.0000002f;
        }
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            const int m = m0 + mi;
            if (m >= M) break;
            const float* hf = H32 + 16 * (size_t)m;          // uniform address: scalar loads
            const float h0 = hf[0], h1 = hf[1], h2 = hf[2], h3 = hf[3], h4 = hf[4], h5 = hf[5], h6 = hf[6], h7 = hf[7], h8 = hf[8];
            const float es = hf[9], en = hf[10], tau = hf[11];
            int c_m = 0;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const float s = __builtin_fmaf(h6, fx[q], __builtin_fmaf(h7, fy[q], h8));
                const float nx = __builtin_fmaf(h0, fx[q], __builtin_fmaf(h1, fy[q], h2));
                const float ny = __builtin_fmaf(h3, fx[q], __builtin_fmaf(h4, fy[q], h5));
                const float r = __builtin_amdgcn_rcpf(s);
                const float uu = nx * r, vv = ny * r;
                const float dx = gx[q] - uu, dy = gy[q] - vv;
                const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                // the bound (every term >= 0)
                const float mm = fmaxf(fabsf(uu), fabsf(vv));
                const float eq = __builtin_fmaf(__builtin_fmaf(mm, es, en), fabsf(r), (5.0f * U32) * mm);
                const float w = fmaxf(fabsf(dx), fabsf(dy));
                const float E = __builtin_fmaf(1.01f * U32, w, eq + cx[q]);
                const float B = __builtin_fmaf(2.2f * U32, d2, (2.02f * E) * __builtin_fmaf(2.0f, w, E));
                const bool trust = fabsf(s) >= tau;
                const bool in32 = trust && (d2 + B < thr2_lo);          // thr2_lo <= thr^2 <= thr2_hi: the threshold rounded down / up
                const bool out32 = trust && (d2 - B >= thr2_hi);
                bool inl = in32;
                const bool need = !(in32 || out32);
                if (__builtin_amdgcn_ballot_w64(need) != 0ull) {          // rare: some lane's pair is too close to call
                    if (need) {
                        const double* h = H + 9 * (size_t)m;
                        const double e2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], px[q], py[q], qx[q], qy[q]);
                        inl = e2 < thr2;
                        ++fb;
                    }
                }
                c_m += __builtin_popcountll(__builtin_amdgcn_ballot_w64(inl) & okm[q]);
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

hipError_t launch_model32(const double* H, int M, double X, double Y, float* H32, hipStream_t s)
{
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_model32, dim3((M + 255) / 256), dim3(256), 0, s, H, M, X, Y, H32);
    return hipGetLastError();
}

// H32: the table launch_model32 made for these M models.  fallback_pairs (nullable): device counter of the pairs decided in FP64.
hipError_t launch_score32(const Points& p, const double* H, const float* H32, int M, double thr2, const unsigned char* mask,
                          int* counts, unsigned long long* fallback_pairs, hipStream_t s)
{
    if (M <= 0 || p.n <= 0) return hipSuccess;
    constexpr int PPL = 4, MC = 16;
    const int gx = (M + MC - 1) / MC, ntiles = (p.n + 256 * PPL - 1) / (256 * PPL);
    int psplit = gx < 1024 ? (2048 + gx - 1) / gx : (ntiles >= 32 ? 4 : 1);
    if (psplit > ntiles) psplit = ntiles;
    if (psplit < 1) psplit = 1;
    if (psplit > 1) {
        hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)M, s);
        if (e != hipSuccess) return e;
    }
    // the threshold in FP32, rounded towards the side that keeps each test conservative
    float lo = (float)thr2, hi = lo;
    if ((double)lo > thr2) lo = nextafterf(lo, -INFINITY);
    if ((double)hi < thr2) hi = nextafterf(hi, INFINITY);
    if (mask) hipLaunchKernelGGL((k_score32<PPL, MC, true>), dim3(gx, psplit), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, H32, M, thr2, lo, hi, counts, mask, psplit, fallback_pairs);
    else hipLaunchKernelGGL((k_score32<PPL, MC, false>), dim3(gx, psplit), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, H32, M, thr2, lo, hi, counts, mask, psplit, fallback_pairs);
    return hipGetLastError();
}

} // namespace mh
