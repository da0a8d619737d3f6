// residual.hip — forward-transfer residual matrix, inlier scoring, single-model
// labelling and the collinearity moments, for gfx950 (MI355X, wave64).
//
// Arithmetic (bit-exact with the reference's FP64 expression, compiled with
// -ffp-contract=off so every operation rounds once, in the reference's
// association order, M/MultiH.cpp:434-441 / :491-498 / :755-762):
//     s  = (h6*x + h7*y) + h8
//     u  = ((h0*x + h1*y) + h2) / s        v = ((h3*x + h4*y) + h5) / s
//     d2 = (x2-u)^2 + (y2-v)^2             inlier  <=>  d2 < thr^2   (strict)
//
// k_residual — the HBM-bound kernel of the roofline run.
//   Work split: one workgroup (256 threads = 4 waves) owns MC consecutive
//   models and sweeps ALL points, so the per-model inlier count is finished
//   inside the workgroup (ballot + s_bcnt1 per wave, 4-way LDS add at the end)
//   with no global atomics.  Each lane holds PPL points in registers; for every
//   model the wave reads the 9 coefficients from LDS (staged once per
//   workgroup, broadcast reads), evaluates PPL residuals per lane and issues 16-B
//   stores so that one wave-instruction writes 1 KiB of one row of R, eight
//   whole 128-B lines.  Rows start 128-B aligned (ld = N rounded up to 16).
//   The correspondence array (32 B/point, 1.6 MB at N = 50k) is re-read once
//   per workgroup from the XCD's 4 MiB L2, where it stays resident on every
//   XCD; HBM sees the 8 B/pair store stream only.
//   Algorithmic bytes per launch: 8*N*M (R) + 32*N + 72*M + 4*M.
// k_score — same sweep without the stores (FP64-VALU/division bound).

#include <type_traits>

#include "mh_kernels.hpp"
#include "mh_device.hpp"

namespace mh {

// A double in global memory, spelt out: the row pointer of the sweep is rebuilt from an integer held in scalar registers, and
// a plain double* made that way would be a flat pointer (flat_store instead of global_store).
typedef double __attribute__((address_space(1))) gdouble;
typedef char __attribute__((address_space(1))) gchar;

// PPL = points per lane (even), MC = models per workgroup.
// WRITE_R: materialise the matrix.  MASK: per-point activity mask (score only).
// NT: non-temporal stores for the R stream.  FAST: shared-reciprocal division (mh_device.hpp).
// LEAN: per tile and wave, a wave-uniform test — every lane holds real points (a full tile), every point meets the
// fast division's precondition, no point is masked out — selects, for the models whose own preconditions hold
// (model_pre, model_far), a sweep without any per-pair bookkeeping: no fallback branch, no exec masking around the stores,
// no validity mask on the inlier ballot.  Same arithmetic, same bits; what goes is scalar and branch work.
// TILED (tuning): R stored tile-major — [model block][point tile][MC][TILE] — so that a workgroup writes one contiguous
// 128-KiB block per tile instead of MC row segments.
// residual_wg: the work of ONE workgroup — model block bx (MC models), point slice by.
template <int PPL, int MC, bool WRITE_R, bool MASK, bool NT, bool FAST, bool CALIB, bool HSGPR,
          bool SYM, bool CONTRACT, bool LEAN, bool TILED, int SF, bool SEMI>
__device__ __forceinline__ void
residual_wg(const double* __restrict__ x1, const double* __restrict__ y1,
            const double* __restrict__ x2, const double* __restrict__ y2, int N,
            const double* __restrict__ H, int M, double thr2, double* __restrict__ R,
            long long ldr, int* __restrict__ counts, const unsigned char* __restrict__ mask,
            int psplit, double bx0, double bx1, double by0, double by1, const int bx, const int by)
{
    constexpr int CH = PPL / 2;                 // 16-B chunks per lane
    constexpr int WAVE_PTS = 64 * PPL;          // points per wave per tile
    constexpr int TILE = 4 * WAVE_PTS;          // points per workgroup per tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (tells the compiler it is wave-uniform)
    const int m0 = bx * MC;

    // Stage this workgroup's MC x 9 coefficients in LDS once; the sweep reads them
    // back with wave-uniform (broadcast) ds_reads.  Keeping them in SGPRs instead
    // makes the compiler hoist 18*MC scalar registers out of the loop and spill.
    __shared__ double s_h[MC * 9];
    __shared__ double s_a[SYM ? MC * 9 : 1];    // adj(H): the backward transfer of the symmetric mode
    for (int i = threadIdx.x; i < MC * 9; i += 256) {
        const size_t g = (size_t)m0 * 9 + i;
        s_h[i] = (g < (size_t)M * 9) ? H[g] : 0.0;
    }
    __shared__ int s_hok[MC];                   // per-model precondition of the shared-reciprocal division
    __shared__ int s_far[MC];                   // per-model proof that |s| >= 2^-255 on the whole point set (model_far)
    __shared__ int s_aok[SYM ? MC : 1];
    __syncthreads();
    if (threadIdx.x < MC) {
        const double* h = s_h + 9 * threadIdx.x;
        s_hok[threadIdx.x] = model_pre(h);
        s_far[threadIdx.x] = model_far(h, bx0, bx1, by0, by1);
        if (SYM) {
            // H^-1 up to scale = adjugate; each entry is (mul, mul, sub), rounded once per operation
            double* a = s_a + 9 * threadIdx.x;
            a[0] = h[4] * h[8] - h[5] * h[7]; a[1] = h[2] * h[7] - h[1] * h[8]; a[2] = h[1] * h[5] - h[2] * h[4];
            a[3] = h[5] * h[6] - h[3] * h[8]; a[4] = h[0] * h[8] - h[2] * h[6]; a[5] = h[2] * h[3] - h[0] * h[5];
            a[6] = h[3] * h[7] - h[4] * h[6]; a[7] = h[1] * h[6] - h[0] * h[7]; a[8] = h[0] * h[4] - h[1] * h[3];
            s_aok[threadIdx.x] = model_pre(a);
        }
    }
    __syncthreads();

    // The per-model flags as wave-uniform bit masks: the model loop tests them on the scalar unit (one LDS read, two
    // readfirstlanes and a mask round trip through a VGPR per model less than reading s_hok[mi] / s_far[mi] there).
    const unsigned long long hok_bits = __builtin_amdgcn_ballot_w64(lane < MC && s_hok[lane < MC ? lane : 0] != 0);
    const unsigned long long far_bits = __builtin_amdgcn_ballot_w64(lane < MC && s_far[lane < MC ? lane : 0] != 0);

    unsigned lane_bytes = (unsigned)lane * 16u;    // (not const: laundered in place inside the model loop)

    int cnt = 0;                                // lane mi of each wave counts model m0+mi
    static_assert(MC <= 64, "one counting lane per model");

    // psplit > 0: slice y takes tiles y, y+psplit, ...; psplit < 0 (tuning variant): -psplit contiguous slices
    const int nslices = psplit > 0 ? psplit : -psplit;
    const int ntiles_all = (N + TILE - 1) / TILE;
    const int per_slice = (ntiles_all + nslices - 1) / nslices;
    const int base0 = psplit > 0 ? by * TILE : by * per_slice * TILE;
    const int base_end = psplit > 0 ? N : min(N, (by + 1) * per_slice * TILE);
    const int base_step = psplit > 0 ? psplit * TILE : TILE;
    for (int base = base0; base < base_end; base += base_step) {
        double px[PPL], py[PPL], qx[PPL], qy[PPL];
        bool ok[PPL], pok[PPL], pokb[SYM ? PPL : 1];
        unsigned long long okm[PPL];            // ok[] as wave masks: the inlier ballot is (d2 < thr2) & okm
        const int wbase = base + wave * WAVE_PTS;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int n = wbase + c * 128 + lane * 2;
            if (n + 1 < N) {
                const double2 a = *reinterpret_cast<const double2*>(x1 + n);
                const double2 b = *reinterpret_cast<const double2*>(y1 + n);
                const double2 cc = *reinterpret_cast<const double2*>(x2 + n);
                const double2 d = *reinterpret_cast<const double2*>(y2 + n);
                px[2 * c] = a.x; px[2 * c + 1] = a.y;
                py[2 * c] = b.x; py[2 * c + 1] = b.y;
                qx[2 * c] = cc.x; qx[2 * c + 1] = cc.y;
                qy[2 * c] = d.x; qy[2 * c + 1] = d.y;
                ok[2 * c] = true; ok[2 * c + 1] = true;
            } else if (n < N) {
                px[2 * c] = x1[n]; py[2 * c] = y1[n]; qx[2 * c] = x2[n]; qy[2 * c] = y2[n];
                px[2 * c + 1] = 1.0; py[2 * c + 1] = 1.0; qx[2 * c + 1] = 1.0; qy[2 * c + 1] = 1.0;   // padding stays on the fast path
                ok[2 * c] = true; ok[2 * c + 1] = false;
            } else {
                px[2 * c] = 1.0; py[2 * c] = 1.0; qx[2 * c] = 1.0; qy[2 * c] = 1.0;
                px[2 * c + 1] = 1.0; py[2 * c + 1] = 1.0; qx[2 * c + 1] = 1.0; qy[2 * c + 1] = 1.0;
                ok[2 * c] = false; ok[2 * c + 1] = false;
            }
            if (MASK) {
                if (ok[2 * c]) ok[2 * c] = mask[n] != 0;
                if (ok[2 * c + 1]) ok[2 * c + 1] = mask[n + 1] != 0;
            }
            okm[2 * c] = __builtin_amdgcn_ballot_w64(ok[2 * c]);
            okm[2 * c + 1] = __builtin_amdgcn_ballot_w64(ok[2 * c + 1]);
            // per-point precondition of the shared-reciprocal division (see mh_device.hpp)
            pok[2 * c] = point_pre(px[2 * c], py[2 * c], qx[2 * c], qy[2 * c]);
            pok[2 * c + 1] = point_pre(px[2 * c + 1], py[2 * c + 1], qx[2 * c + 1], qy[2 * c + 1]);
            if (SYM) {
                pokb[2 * c] = point_pre(qx[2 * c], qy[2 * c], px[2 * c], py[2 * c]);
                pokb[2 * c + 1] = point_pre(qx[2 * c + 1], qy[2 * c + 1], px[2 * c + 1], py[2 * c + 1]);
            }
        }

        bool tile_lean = false;                 // wave-uniform
        if (LEAN) {
            bool lane_ok = true;
#pragma unroll
            for (int q = 0; q < PPL; ++q) lane_ok = lane_ok && ok[q] && pok[q];
            tile_lean = (wbase + WAVE_PTS <= N) && (__builtin_amdgcn_ballot_w64(lane_ok) == ~0ull);
        }
        const int tile_idx = base / TILE;       // TILED: which point tile this is

        // Not unrolled on purpose: an unrolled model loop lets LICM hoist all MC*9
        // coefficients out of the point sweep (288 registers at MC = 16).
#pragma unroll 1
        for (int mi = 0; mi < MC; ++mi) {
            const int m = m0 + mi;
            if (m < M) {                                   // wave-uniform
                // HSGPR (the product): coefficients through the scalar unit (uniform address -> s_load_dwordx16 + x2 per
                // model), so the twelve linear-form operations read one operand from SGPRs instead of VGPRs.
                const double* h = HSGPR ? (H + 9 * (size_t)m) : (s_h + 9 * mi);
                const double h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5],
                             h6 = h[6], h7 = h[7], h8 = h[8];
                // wave-uniform by construction (every lane reads the same LDS word); readfirstlane tells the compiler so,
                // which keeps the choice of sweep a scalar branch instead of an exec-mask dance
                const bool hok = ((hok_bits >> mi) & 1ull) != 0;
                const bool far = FAST && !CONTRACT && ((far_bits >> mi) & 1ull) != 0;
                const bool aok = SYM ? (s_aok[mi] != 0) : false;
                int c_m = 0;
                // where row m (or, TILED, this workgroup's block) starts for this wave's first chunk
                auto row_ptr = [&](int n) -> double* {
                    if (TILED) return R + ((size_t)bx * ntiles_all + tile_idx) * ((size_t)MC * TILE) + (size_t)mi * TILE + (n - base);
                    return R + (size_t)m * ldr + n;
                };
                // ... as a scalar register pair; the lane's 16 bytes are a 32-bit offset, so the stores address as saddr + voffset and
                // the model loop advances the row on the scalar unit instead of with a 64-bit vector add per model
                gchar* row_base = (gchar*)row_ptr(wbase);
                asm volatile("" : "+v"(lane_bytes));        // (keeps the zero-extension next to the address, where the saddr form is matched)
                asm volatile("" : "+s"(row_base));          // (opaque: keeps the strength reducer from turning it back into a per-lane pointer)
                // the chunk loop, once with the per-pair |s| compare and once without (the model's horizon is provably far
                // from every point): one VALU instruction per pair less on the common path
                auto sweep = [&](auto schk) {
                    constexpr bool SCHK = decltype(schk)::value;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (CALIB) {       // store-bandwidth calibration build: no residual arithmetic
                            const int n = wbase + c * 128 + lane * 2;
                            if (n + 1 < N)
                                *reinterpret_cast<double2*>(R + (size_t)m * ldr + n) =
                                    make_double2(px[2 * c] + h0, py[2 * c] + h0);
                            continue;
                        }
                        double d0s, d1s;
                        const double d0 = CONTRACT ? fwd_d2_contracted(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[2 * c],
                                                                       py[2 * c], qx[2 * c], qy[2 * c])
                                          : FAST ? fwd_d2_fast<SCHK>(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[2 * c],
                                                                   py[2 * c], qx[2 * c], qy[2 * c], pok[2 * c] && hok)
                                               : fwd_d2(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[2 * c],
                                                        py[2 * c], qx[2 * c], qy[2 * c]);
                        const double d1 = CONTRACT ? fwd_d2_contracted(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[2 * c + 1],
                                                                       py[2 * c + 1], qx[2 * c + 1], qy[2 * c + 1])
                                          : FAST ? fwd_d2_fast<SCHK>(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[2 * c + 1],
                                                                   py[2 * c + 1], qx[2 * c + 1], qy[2 * c + 1], pok[2 * c + 1] && hok)
                                               : fwd_d2(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[2 * c + 1],
                                                        py[2 * c + 1], qx[2 * c + 1], qy[2 * c + 1]);
                        d0s = d0; d1s = d1;
                        if (SYM) {       // + ||H^-1 p2 - p1||^2 (north_star's symmetric transfer; no reference oracle)
                            const double* a = s_a + 9 * mi;
                            const double b0 = fwd_d2_fast<true>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8],
                                                                qx[2 * c], qy[2 * c], px[2 * c], py[2 * c], pokb[2 * c] && aok);
                            const double b1 = fwd_d2_fast<true>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8],
                                                                qx[2 * c + 1], qy[2 * c + 1], px[2 * c + 1], py[2 * c + 1],
                                                                pokb[2 * c + 1] && aok);
                            d0s = d0 + b0;
                            d1s = d1 + b1;
                        }
                        if (WRITE_R) {
                            const int n = wbase + c * 128 + lane * 2;
                            gdouble* dstp = reinterpret_cast<gdouble*>(row_base + lane_bytes) + c * 128;
                            if (n + 1 < N) {
                                if (NT) {
                                    __builtin_nontemporal_store(d0s, dstp);
                                    __builtin_nontemporal_store(d1s, dstp + 1);
                                } else {
                                    dstp[0] = d0s; dstp[1] = d1s;
                                }
                            } else if (n < N) {
                                dstp[0] = d0s;
                            }
                        }
                        c_m += __builtin_popcountll(__builtin_amdgcn_ballot_w64(d0s < thr2) & okm[2 * c]);
                        c_m += __builtin_popcountll(__builtin_amdgcn_ballot_w64(d1s < thr2) & okm[2 * c + 1]);
                    }
                };
                // The lean sweep: all PPL residuals first, then the stores and the counts.  `far` models (the horizon provably
                // clear of every point) need no check at all; for the others ONE wave-wide test per model — did any of the
                // PPL x 64 denominators leave the fast division's range? — replaces the per-pair branch, and the rare wave
                // that says yes redoes this model through the checked sweep (nothing has been stored or counted yet).
                bool lean_done = false;
                if (LEAN && FAST && !SYM && !CONTRACT && !CALIB && tile_lean && hok && (SEMI || far)) {
                    double dl[PPL], sq[PPL];
                    unsigned long long bad = 0ull;          // lanes whose denominator is out of range (or NaN), any of the PPL pairs
#pragma unroll
                    for (int q = 0; q < PPL; ++q)
                        dl[q] = fwd_d2_lean(h0, h1, h2, h3, h4, h5, h6, h7, h8, px[q], py[q], qx[q], qy[q], sq[q]);
                    if (!far) {                             // one scalar branch per model, not one per pair
#pragma unroll
                        for (int q = 0; q < PPL; ++q) bad |= __builtin_amdgcn_ballot_w64(!(__builtin_fabs(sq[q]) >= 0x1p-255));
                    }
                    if (bad == 0ull) {
                        lean_done = true;
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const double d0 = dl[2 * c], d1 = dl[2 * c + 1];
                            if (WRITE_R) {
                                gdouble* dstp = reinterpret_cast<gdouble*>(row_base + lane_bytes) + c * 128;
                                if (SF != 0) {          // measurement builds: cache-policy bits of the store spelt out
                                    typedef double d2v __attribute__((ext_vector_type(2)));
                                    const d2v val = { d0, d1 };
                                    if (SF == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dstp), "v"(val) : "memory");
                                    else if (SF == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dstp), "v"(val) : "memory");
                                    else if (SF == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(dstp), "v"(val) : "memory");
                                    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(dstp), "v"(val) : "memory");
                                } else if (NT) {
                                    __builtin_nontemporal_store(d0, dstp);
                                    __builtin_nontemporal_store(d1, dstp + 1);
                                } else {
                                    dstp[0] = d0; dstp[1] = d1;
                                }
                            }
                            c_m += __builtin_popcountll(__builtin_amdgcn_ballot_w64(d0 < thr2));
                            c_m += __builtin_popcountll(__builtin_amdgcn_ballot_w64(d1 < thr2));
                        }
                    }
                }
                if (lean_done) {}
                else if (far) sweep(std::false_type{});
                else sweep(std::true_type{});
                // lane mi accumulates model mi: read-modify-write of that one lane through the scalar unit
                const int c_new = __builtin_amdgcn_readlane(cnt, mi) + c_m;
                asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(cnt) : "s"(c_new), "s"(mi) : "m0");
            }
        }
    }

    __shared__ int s_cnt[4][MC];
    if (lane < MC) s_cnt[wave][lane] = cnt;
    __syncthreads();
    if (threadIdx.x < MC && m0 + (int)threadIdx.x < M) {
        const int t = threadIdx.x;
        const int c = s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
        if (nslices == 1) counts[m0 + t] = c;
        else atomicAdd(&counts[m0 + t], c);           // integer: order-independent
    }
}

// One workgroup per (model block, point slice), placed by the hardware dispatcher.
template <int PPL, int MC, bool WRITE_R, bool MASK, bool NT, bool FAST, bool CALIB = false, bool HSGPR = false,
          bool SYM = false, bool CONTRACT = false, bool LEAN = false, bool TILED = false, int SF = 0, bool SEMI = true, int MINW = 1>
__global__ void __launch_bounds__(256, MINW)
k_residual(const double* __restrict__ x1, const double* __restrict__ y1,
           const double* __restrict__ x2, const double* __restrict__ y2, int N,
           const double* __restrict__ H, int M, double thr2, double* __restrict__ R,
           long long ldr, int* __restrict__ counts, const unsigned char* __restrict__ mask,
           int psplit, int swapxy, double bx0, double bx1, double by0, double by1)
{
    // swapxy (tuning): slice index in blockIdx.x, so that consecutive workgroups write neighbouring
    // chunks of the same rows of R
    const int bx = swapxy ? blockIdx.y : blockIdx.x;
    const int by = swapxy ? blockIdx.x : blockIdx.y;
    residual_wg<PPL, MC, WRITE_R, MASK, NT, FAST, CALIB, HSGPR, SYM, CONTRACT, LEAN, TILED, SF, SEMI>(
        x1, y1, x2, y2, N, H, M, thr2, R, ldr, counts, mask, psplit, bx0, bx1, by0, by1, bx, by);
}

// The same work items walked by a RESIDENT grid (r04): as many workgroups as the chip holds at this kernel's occupancy
// (81 registers, 30 scalar registers spilt into one of them -> 88 allocated: five waves per SIMD, 1 280 workgroups on 256
// CUs) hand themselves the gx x psplit items — item -> (model block item % gx, slice item / gx) — through a counter.
// Measured against one hardware-dispatched workgroup per item at 50k x 100k: 7.27-7.31 ms vs 7.48-7.70 (the launch and
// retirement of 37 500 workgroups, and a dispatcher that refills every slot the moment it frees, cost more than the
// counter).  Holding the kernel to 80 registers for a sixth wave does not work: the scalar spills need the 81st.
template <int PPL, int MC, bool WRITE_R, bool MASK, bool NT, bool FAST, bool CALIB = false, bool HSGPR = false,
          bool SYM = false, bool CONTRACT = false, bool LEAN = false, bool TILED = false, int SF = 0, bool SEMI = true, int MINW = 1>
__global__ void __launch_bounds__(256, MINW) __attribute__((amdgpu_num_vgpr(88)))      // 5 waves per SIMD and 72 registers left for k_dlt4_lds
k_residual_resident(const double* __restrict__ x1, const double* __restrict__ y1,
                    const double* __restrict__ x2, const double* __restrict__ y2, int N,
                    const double* __restrict__ H, int M, double thr2, double* __restrict__ R,
                    long long ldr, int* __restrict__ counts, const unsigned char* __restrict__ mask,
                    int psplit, int gx, int nitems, int* __restrict__ ctl, double bx0, double bx1, double by0, double by1, int slice_major)
{
    // Items are handed out through one counter, first come first served — the workgroups of a resident grid do not run at
    // one speed (a compute unit that holds six of them serves each more slowly than one that holds five, and the memory
    // channels are not equally busy), and a static split waits for the slowest: 8.55 ms against 7.70 at 50k x 100k.
    // ctl[0] = next item, ctl[1] = workgroups that have left; the last one to leave clears both for the next launch.
    __shared__ int s_item;
#pragma unroll 1
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&ctl[0], 1);
        __syncthreads();
        const int item = s_item;
        if (item >= nitems) break;
        // slice_major: consecutive items are the point slices of ONE model block, so the workgroups at work at any moment
        // write a compact window of R (a few dozen model blocks, every row of them along its whole length) instead of one
        // slice of every block in turn
        int bx, by;
        if (slice_major) { bx = item / psplit; by = item - bx * psplit; }
        else { by = item / gx; bx = item - by * gx; }
        residual_wg<PPL, MC, WRITE_R, MASK, NT, FAST, CALIB, HSGPR, SYM, CONTRACT, LEAN, TILED, SF, SEMI>(
            x1, y1, x2, y2, N, H, M, thr2, R, ldr, counts, mask, psplit, bx0, bx1, by0, by1, bx, by);
        __syncthreads();                        // the item's LDS (coefficients, flags, counts) and s_item are rewritten by the next one
    }
    if (threadIdx.x == 0 && atomicAdd(&ctl[1], 1) == (int)gridDim.x - 1) {
        ctl[1] = 0;
        __hip_atomic_store(&ctl[0], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int PPL, int MC, bool WRITE_R, bool MASK, bool NT, bool FAST = true, bool CALIB = false, bool HSGPR = false,
          bool SYM = false, bool CONTRACT = false, bool LEAN = false, bool TILED = false, int SF = 0, bool SEMI = true, int MINW = 1>
static hipError_t launch_rs(const Points& p, const double* H, int M, double thr2, double* R,
                            long long ldr, int* counts, const unsigned char* mask, hipStream_t s,
                            int force_psplit = 0, int swapxy = 0, bool counts_zeroed = false, int resident_grid = 0,
                            int* resident_ctl = nullptr, int slice_major = 0)
{
    if (M <= 0 || p.n <= 0) return hipSuccess;
    const int gx = (M + MC - 1) / MC;
    const int tile = 256 * PPL;
    const int ntiles = (p.n + tile - 1) / tile;
    int psplit = 1;
    if (WRITE_R) {
        // The materialising sweep: aim at ~37 500 workgroups whatever the batch size (at least 4 point slices for a long
        // sweep).  r04, tools/shard_proxy.py on the per-rank shards of BASELINE configs[3]: with the r03 rule (2 048
        // workgroups below 1 024 model blocks, else 4 slices) the 12 500-hypothesis shard of an 8-GPU run ran 2 346
        // workgroups of 16 tiles — 1.5 rounds of the chip's 1 536 resident workgroups, a quarter of the launch in its
        // tail — and cost 1.06 ms against 7.63 / 8 = 0.95; 1.049 / 1.025 / 1.006 / 0.998 / 0.992 ms at 4 / 8 / 12 / 16 / 32
        // slices; 25 000: 2.039 -> 1.984 (24), 50 000: 4.044 -> 3.980 (12); 100 000: flat (7.59 at 4, 7.57 at 8).
        psplit = (37500 + gx - 1) / gx;
        if (psplit < 4 && ntiles >= 32) psplit = 4;
        if (psplit > ntiles) psplit = ntiles;
        if (psplit < 1) psplit = 1;
    } else if (gx < 1024) {                // few models: split the point sweep to fill the chip
        psplit = (2048 + gx - 1) / gx;
        if (psplit > ntiles) psplit = ntiles;
        if (psplit < 1) psplit = 1;
    } else if (ntiles >= 32) {
        // many models and a long sweep: a workgroup that walks all N points runs for milliseconds and
        // the last round of workgroups leaves CUs idle; four shorter slices per model block trim
        // that tail (measured 7.95 -> 7.79 ms at 50k x 100k, tools/kernel_sweep.py RV=104)
        psplit = 4;
    }
    if (force_psplit > 0) psplit = force_psplit < ntiles ? force_psplit : ntiles;
    bool contiguous = false;
    if (force_psplit < 0) { psplit = -force_psplit < ntiles ? -force_psplit : ntiles; contiguous = true; }
    if (psplit > 1 && !counts_zeroed) {
        hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)M, s);
        if (e != hipSuccess) return e;
    }
    constexpr bool PRODUCT_SWEEP = WRITE_R && !MASK && NT && FAST && !CALIB && HSGPR && !SYM && !CONTRACT && LEAN && !TILED && SF == 0 && SEMI && MINW == 1 && PPL == 4 && (MC == 16 || MC == 32 || MC == 64);
    if constexpr (PRODUCT_SWEEP)
    if (resident_grid > 0 && resident_ctl && !contiguous && !swapxy && gx * psplit > resident_grid) {
        hipLaunchKernelGGL((k_residual_resident<PPL, MC, WRITE_R, MASK, NT, FAST, CALIB, HSGPR, SYM, CONTRACT, LEAN, TILED, SF, SEMI, MINW>),
                           dim3(resident_grid), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, M, thr2, R, ldr, counts, mask, psplit, gx,
                           gx * psplit, resident_ctl, p.xmin, p.xmax, p.ymin, p.ymax, slice_major);
        return hipGetLastError();
    }
    dim3 grid(gx, psplit);
    if (swapxy) grid = dim3(psplit, gx);
    hipLaunchKernelGGL((k_residual<PPL, MC, WRITE_R, MASK, NT, FAST, CALIB, HSGPR, SYM, CONTRACT, LEAN, TILED, SF, SEMI, MINW>), grid, dim3(256), 0, s, p.x1, p.y1,
                       p.x2, p.y2, p.n, H, M, thr2, R, ldr, counts, mask, contiguous ? -psplit : psplit, swapxy,
                       p.xmin, p.xmax, p.ymin, p.ymax);
    return hipGetLastError();
}

// variant: 0 = the reference's forward transfer error, -1 = north_star's symmetric transfer error (extension).
// Everything else is a tuning / measurement build of the same kernel (other PPL / MC, nt stores, compiler division,
// coefficients in SGPRs, store-only calibration, forced slices, fused multiply-adds — the last one NOT bit-exact) and
// exists only in libraries compiled with -DMH_TUNING (multi-h_amd/build.py --tuning) for tools/kernel_sweep.py.
hipError_t launch_residual(const Points& p, const double* H, int M, double thr2, double* R,
                           long long ldr, int* counts, int variant, hipStream_t s, bool counts_zeroed, int resident_grid,
                           int* resident_ctl, int slices, int slice_major)
{
    // PPL 4, MC 64 (r05; 16 before: same-box A/B 7.21 / 7.11 / 7.06 ms for 16 / 32 / 64 models per work item — the per-tile work is
    // shared by more models), the lean sweep wherever a tile and a model allow it, non-temporal 16-B stores (r03: the kernel runs at
    // the board's power cap, its time is its energy; nt stores — nothing of R is ever re-read — cost 2.7 % less energy
    // per launch than plain ones, profiles/archive/r03_energy.json)
    // ... and the nine coefficients of the current model through the scalar unit (s_load from H, uniform address) instead
    // of LDS broadcasts into VGPRs: the twelve linear-form operations then read one operand from SGPRs; 2.5 % less energy.
    if (variant == 0) return launch_rs<4, 64, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, slices, 0, counts_zeroed, resident_grid, resident_ctl, slice_major);
    if (variant == -1) return launch_rs<4, 16, true, false, true, true, false, true, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // nt stores, forward coefficients through the scalar unit
#ifdef MH_TUNING
    if (variant == -2)          // symmetric mode at PPL 2 (PPL 4 measured 3 % faster)
        return launch_rs<2, 16, true, false, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);
    if (variant >= 700)         // 700 + psplit: store-only calibration with nt stores (the product's store instruction) and a forced point split
        return launch_rs<4, 16, true, false, true, true, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, variant - 700);
    if (variant >= 600)         // 600 + psplit: fused multiply-adds (NOT bit-exact) with a forced point split
        return launch_rs<4, 16, true, false, false, true, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, variant - 600);
    if (variant >= 500)         // 500 + psplit: store-only calibration (plain stores) with a forced point split
        return launch_rs<4, 16, true, false, false, true, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, variant - 500);
    if (variant >= 400)         // 400 + psplit: the product kernel with a forced point split (tools/shard_proxy.py)
        return launch_rs<4, 16, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, variant - 400);
    if (variant >= 300)         // 300 + s: s interleaved slices with the slice index as the fastest grid dimension
        return launch_rs<4, 16, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s, variant - 300, 1);
    if (variant >= 200)         // 200 + s: s contiguous point slices instead of interleaved tiles
        return launch_rs<4, 16, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s, -(variant - 200));
    if (variant >= 100)         // 100 + psplit: default kernel with a forced point split
        return launch_rs<4, 16, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s, variant - 100);
    switch (variant) {
    // 50 / 51 / 52: the product sweep with 16 / 32 / 64 models per work item (the product: 64 since r05), resident grid and all
    case 50: return launch_rs<4, 16, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, slices, 0, counts_zeroed, resident_grid, resident_ctl, slice_major);
    case 51: return launch_rs<4, 32, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, slices, 0, counts_zeroed, resident_grid, resident_ctl, slice_major);
    case 52: return launch_rs<4, 64, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, slices, 0, counts_zeroed, resident_grid, resident_ctl, slice_major);
    case 1: return launch_rs<2, 16, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);                // PPL 2
    case 2: return launch_rs<4, 16, true, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);                 // nt stores
    case 3: return launch_rs<4, 16, true, false, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);         // compiler IEEE division
    case 4: return launch_rs<4, 16, true, false, false, true, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);  // coefficients in SGPRs
    case 5: return launch_rs<4, 8, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);                 // MC 8
    case 6: return launch_rs<4, 32, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);                // MC 32
    case 7: return launch_rs<4, 16, true, false, false, true, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);    // store-only calibration
    case 8: return launch_rs<8, 16, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);                // PPL 8
    case 9: return launch_rs<6, 16, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);                // PPL 6
    case 10: return launch_rs<4, 16, true, false, false, true, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // fused multiply-adds: NOT bit-exact
    case 20: return launch_rs<4, 16, true, false, false, true, false, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);         // lean sweep on clean tiles, plain stores
    case 32: return launch_rs<4, 16, true, false, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);                                                 // the r02 product kernel: checked sweep everywhere
    case 21: return launch_rs<4, 16, true, false, false, true, false, false, false, false, true, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // lean + tile-major R
    case 22: return launch_rs<4, 16, true, false, true, true, false, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);          // lean + nt stores, coefficients from LDS
    case 23: return launch_rs<4, 16, true, false, false, true, false, false, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);  // tile-major R alone
    case 24: return launch_rs<4, 16, true, false, false, true, true, false, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // store-only calibration, tile-major R
    case 25: return launch_rs<6, 16, true, false, false, true, false, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);         // lean, PPL 6
    case 26: return launch_rs<8, 16, true, false, false, true, false, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);         // lean, PPL 8
    case 28: return launch_rs<4, 16, true, false, false, true, false, false, false, false, true, false, 2>(p, H, M, thr2, R, ldr, counts, nullptr, s);  // lean, sc1 stores
    case 29: return launch_rs<4, 16, true, false, false, true, false, false, false, false, true, false, 3>(p, H, M, thr2, R, ldr, counts, nullptr, s);  // lean, sc0 sc1 stores
    case 30: return launch_rs<4, 16, true, false, false, true, false, false, false, false, true, false, 4>(p, H, M, thr2, R, ldr, counts, nullptr, s);  // lean, sc1 nt stores
    case 31: return launch_rs<4, 16, true, false, false, true, false, false, false, false, true, false, 5>(p, H, M, thr2, R, ldr, counts, nullptr, s);  // lean, sc0 sc1 nt stores
    case 33: return launch_rs<4, 16, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);           // lean + nt, coefficients through the scalar unit (= the product since r03)
    case 34: return launch_rs<4, 16, true, false, true, true, false, true, false, false, true, false, 0, false>(p, H, M, thr2, R, ldr, counts, nullptr, s);  // as the product, but models that are not `far` take the checked sweep
    case 35: return launch_rs<4, 16, true, false, true, true, false, true, false, false, true, false, 0, true, 8>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // product, registers capped for 8 waves per SIMD
    case 36: return launch_rs<4, 16, true, false, true, true, false, true, false, false, true, false, 0, true, 7>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // ... 7 waves per SIMD
    case 37: return launch_rs<6, 16, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);                     // product at PPL 6
    case 38: return launch_rs<2, 16, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);                     // product at PPL 2
    case 39: return launch_rs<4, 32, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);                     // product at MC 32
    case 40: return launch_rs<4, 8, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);                      // product at MC 8
    case 41: return launch_rs<4, 32, true, false, true, true, false, true, false, false, true, false, 0, true, 7>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // MC 32, 7 waves per SIMD
    case 42: return launch_rs<4, 64, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);                      // MC 64
    case 43: return launch_rs<6, 32, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);                      // MC 32, PPL 6
    case 44: return launch_rs<4, 64, true, false, true, true, false, true, false, false, true, false, 0, true, 7>(p, H, M, thr2, R, ldr, counts, nullptr, s);   // MC 64, 7 waves per SIMD
    case 45: return launch_rs<4, 32, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, 8);                   // MC 32, 8 point slices
    case 46: return launch_rs<4, 32, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, 2);                   // MC 32, 2 point slices
    case 47: return launch_rs<4, 64, true, false, true, true, false, true, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s, 8);                   // MC 64, 8 point slices
    case 27: return launch_rs<2, 16, true, false, false, true, false, false, false, false, true>(p, H, M, thr2, R, ldr, counts, nullptr, s);         // lean, PPL 2
    default: break;
    }
#endif
    return hipErrorInvalidValue;
}

int residual_workgroups_per_cu()
{
    int n = 0;
    const hipError_t he = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &n, (const void*)k_residual_resident<4, 64, true, false, true, true, false, true, false, false, true>, 256, 0);
    return he == hipSuccess && n > 0 ? n : 0;
}

hipError_t launch_score(const Points& p, const double* H, int M, double thr2,
                        const unsigned char* mask, int* counts, int variant, hipStream_t s)
{
    if (variant == -1) {
        if (mask) return launch_rs<4, 16, false, true, false, true, false, false, true>(p, H, M, thr2, nullptr, 0, counts, mask, s);
        return launch_rs<4, 16, false, false, false, true, false, false, true>(p, H, M, thr2, nullptr, 0, counts, nullptr, s);
    }
    if (mask) return launch_rs<4, 16, false, true, false, true, false, true, false, false, true>(p, H, M, thr2, nullptr, 0, counts, mask, s);
    if (variant == 0) return launch_rs<4, 16, false, false, false, true, false, true, false, false, true>(p, H, M, thr2, nullptr, 0, counts, nullptr, s);
#ifdef MH_TUNING
    if (variant == 1) return launch_rs<2, 16, false, false, false>(p, H, M, thr2, nullptr, 0, counts, nullptr, s);           // PPL 2
    if (variant == 32) return launch_rs<4, 16, false, false, false>(p, H, M, thr2, nullptr, 0, counts, nullptr, s);          // the r02 score kernel
    if (variant == 20) return launch_rs<4, 16, false, false, false, true, false, false, false, false, true>(p, H, M, thr2, nullptr, 0, counts, nullptr, s);   // lean, coefficients from LDS
    if (variant == 3) return launch_rs<4, 16, false, false, false, false>(p, H, M, thr2, nullptr, 0, counts, nullptr, s);    // compiler IEEE division
#endif
    return hipErrorInvalidValue;
}

// ComputeInliersOfHomography, M/MultiH.cpp:743-768.
__global__ void __launch_bounds__(256)
k_inliers_of_model(const double* __restrict__ x1, const double* __restrict__ y1,
                   const double* __restrict__ x2, const double* __restrict__ y2, int N,
                   const double* __restrict__ H, int idx, double thr2, int label_value,
                   int* __restrict__ labels)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double* h = H + 9 * (size_t)idx;
    const double d2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], x1[n], y1[n],
                             x2[n], y2[n]);
    if (d2 < thr2) labels[n] = label_value;
}

hipError_t launch_inliers_of_model(const Points& p, const double* H, int idx, double thr2,
                                   int label_value, int* labels, hipStream_t s)
{
    hipLaunchKernelGGL(k_inliers_of_model, dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1,
                       p.x2, p.y2, p.n, H, idx, thr2, label_value, labels);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Collinearity moments (M/MultiH.cpp:446-463).  One workgroup per model.  The
// FP64 sums use the engine's deterministic order: lane t of 256 adds its points
// t, t+256, ... in increasing order, then a binary tree v[t] += v[t+s],
// s = 128..1.  Thread 0 finishes with a cyclic-Jacobi 3x3 eigen solve.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_moments(const double* __restrict__ x1, const double* __restrict__ y1,
          const double* __restrict__ x2, const double* __restrict__ y2, int N,
          const double* __restrict__ H, double thr2, double* __restrict__ moments,
          double* __restrict__ min_eig)
{
    const int m = blockIdx.x;
    const int t = threadIdx.x;
    const double* h = H + 9 * (size_t)m;
    const double h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6],
                 h7 = h[7], h8 = h[8];
    double acc[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
    int cnt = 0;
    // (latency bound — a trip to memory per point and lane —: the coordinates of eight of the lane's points are fetched
    // together; the additions keep their order)
    constexpr int UNROLL = 8;
    for (int n0 = t; n0 < N; n0 += 256 * UNROLL) {
        double px[UNROLL], py[UNROLL], qx[UNROLL], qy[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int n = n0 + 256 * j, mm = n < N ? n : 0;
            px[j] = x1[mm]; py[j] = y1[mm]; qx[j] = x2[mm]; qy[j] = y2[mm];
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            if (n0 + 256 * j >= N) continue;
            const double x = px[j], y = py[j];
            const double d2 = fwd_d2(h0, h1, h2, h3, h4, h5, h6, h7, h8, x, y, qx[j], qy[j]);
            if (d2 < thr2) {
                acc[0] = acc[0] + x;
                acc[1] = acc[1] + y;
                acc[2] = acc[2] + x * x;
                acc[3] = acc[3] + x * y;
                acc[4] = acc[4] + y * y;
                ++cnt;
            }
        }
    }
    __shared__ double sv[256][5];
    __shared__ int sc[256];
#pragma unroll
    for (int k = 0; k < 5; ++k) sv[t][k] = acc[k];
    sc[t] = cnt;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int k = 0; k < 5; ++k) sv[t][k] = sv[t][k] + sv[t + s][k];
            sc[t] += sc[t + s];
        }
        __syncthreads();
    }
    if (t == 0) {
        double* mo = moments + 6 * (size_t)m;
        const double c = (double)sc[0];
        mo[0] = c; mo[1] = sv[0][0]; mo[2] = sv[0][1]; mo[3] = sv[0][2]; mo[4] = sv[0][3];
        mo[5] = sv[0][4];
        if (min_eig) {
            double a[9] = { sv[0][2], sv[0][3], sv[0][0], sv[0][3], sv[0][4], sv[0][1],
                            sv[0][0], sv[0][1], c };
            double v[9], d[3];
            jacobi_sym_dev(3, a, v, d);
            double mn = d[0];
            if (d[1] < mn) mn = d[1];
            if (d[2] < mn) mn = d[2];
            min_eig[m] = mn;
        }
    }
}

hipError_t launch_moments(const Points& p, const double* H, int M, double thr2, double* moments,
                          double* min_eig, hipStream_t s)
{
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_moments, dim3(M), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, H, thr2,
                       moments, min_eig);
    return hipGetLastError();
}

} // namespace mh
