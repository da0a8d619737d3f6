// select.hip — greedy model selection over a resident hypothesis batch, entirely on the device.
//
// The sequential-RANSAC scheme of the dead M/MultipleHomographies.h:146-175 (take the best-supported hypothesis,
// take its inliers out of the support set, score again) as the engine runs it behind MultiH::ProposeModels:
// per round
//   launch_score     inlier counts of the candidate hypotheses over the points still in the support mask
//   k_sel_argmax     this rank's best candidate: highest count, lowest GLOBAL hypothesis counter on ties (one 64-bit
//                    atomicMax per workgroup on key = count << 32 | ~counter)
//   k_sel_record     the rank's OFFER: an 88-byte record {key, H[9], error word}
//   (exchange)       sharded batches only: the ranks all-gather their records — 88 bytes per rank — on the engine's
//                    stream (RCCL); in the FIRST round also their whole int32 score vectors (north_star's exchange:
//                    every rank then holds every hypothesis' score; the winner it implies is cross-checked against
//                    the records')
//   k_sel_compact    the candidates that can still win — count >= need, counts only fall as points leave the mask —
//                    are copied to the next round's list (the round's winner is not one of them)
//   k_sel_claim      the winner (largest key over the records, identical on every rank) joins the output list and its
//                    inliers leave the mask
//   k_sel_publish    five control words for the host (mapped pinned memory): the round's best count decides whether
//                    there is another round; nothing else crosses the bus
// After the first round the candidate list is a small fraction of the batch, so later rounds are short.  Keys carry the
// hypothesis' position in the WHOLE batch (shard offset + local index), so the order of selection — and with it every
// output, the counters included — is the single-GPU one for any number of ranks.
#include "mh_device.hpp"
#include "mh_kernels.hpp"

namespace mh {

__device__ __forceinline__ unsigned long long sel_key(int count, unsigned int counter)
{
    return ((unsigned long long)(unsigned int)count << 32) | (unsigned long long)(0xffffffffu - counter);
}

__device__ __forceinline__ unsigned long long wg_max_u64(unsigned long long k)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(k, m, 64); k = o > k ? o : k; }
    __shared__ unsigned long long s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = k;
    __syncthreads();
    unsigned long long b = s[0];
    for (int w = 1; w < 4; ++w) b = s[w] > b ? s[w] : b;
    return b;
}

// counts[c] of candidate c whose position in this rank's batch is orig[c] (identity when null); my_off = position of this
// rank's hypothesis 0 in the whole batch.  scores_full (nullable): scores_full[orig] = count (the vector that is
// all-gathered in the first round; entries never written stay -1).
__global__ void __launch_bounds__(256)
k_sel_argmax(const int* __restrict__ counts, const int* __restrict__ orig, int Mc, unsigned int my_off,
             unsigned long long* __restrict__ key, int* __restrict__ scores_full)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    unsigned long long k = 0;
    if (c < Mc) {
        const int o = orig ? orig[c] : c;
        const int cnt = counts[c];
        if (cnt >= 0) k = sel_key(cnt, my_off + (unsigned int)o);
        if (scores_full) scores_full[o] = cnt;
    }
    const unsigned long long b = wg_max_u64(k);
    if (threadIdx.x == 0 && b) atomicMax(key, b);
}

// gathered: world x longest scores in rank order (-1 = padding); entry (r, j) is hypothesis r * base + min(r, rem) + j of
// the whole batch (contiguous shards, the first `rem` one longer).
__global__ void __launch_bounds__(256)
k_sel_argmax_gathered(const int* __restrict__ gathered, int world, int longest, int base, int rem,
                      unsigned long long* __restrict__ key)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    unsigned long long k = 0;
    if (c < world * longest && gathered[c] >= 0) {
        const int r = c / longest, j = c - r * longest;
        k = sel_key(gathered[c], (unsigned int)(r * base + (r < rem ? r : rem) + j));
    }
    const unsigned long long b = wg_max_u64(k);
    if (threadIdx.x == 0 && b) atomicMax(key, b);
}

// This rank's offer: the candidate whose key is the rank's best.  `record` must have been cleared (key 0 = no offer).
__global__ void __launch_bounds__(256)
k_sel_record(const int* __restrict__ counts, const int* __restrict__ orig, const double* __restrict__ Hs, int Mc,
             unsigned int my_off, const unsigned long long* __restrict__ key_local, int err, int mode, SelRecord* __restrict__ record)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long kl = *key_local;
    if (c == 0) { record->err = err; record->mode = mode; }
    if (c >= Mc || !kl) return;
    const int o = orig ? orig[c] : c;
    if (sel_key(counts[c], my_off + (unsigned int)o) != kl) return;
    record->key = kl;
    const double* h = Hs + 9 * (size_t)c;
    for (int q = 0; q < 9; ++q) record->H[q] = h[q];
}

// the round's winner: the largest key over the ranks' records (every rank computes the same)
__device__ __forceinline__ unsigned long long sel_winner(const SelRecord* __restrict__ records, int world, int* rank_out)
{
    unsigned long long kg = 0;
    int wr = 0;
    for (int r = 0; r < world; ++r) { const unsigned long long k = records[r].key; if (k > kg) { kg = k; wr = r; } }
    if (rank_out) *rank_out = wr;
    return kg;
}

// rec: [0] best count of the round, [1] its global counter, [2] next candidate count, [3] models selected so far, [5] points that
// left the support set in this round's claim,
// [4] sticky error (a rank reported one, or the two ways of finding the first round's winner disagree).
__global__ void __launch_bounds__(256)
k_sel_compact(const int* __restrict__ counts, const int* __restrict__ orig, const double* __restrict__ Hs, int Mc, int need,
              const SelRecord* __restrict__ records, int world, unsigned int my_off, int* __restrict__ next_orig,
              double* __restrict__ next_H, int* __restrict__ rec, int* __restrict__ next_counts)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long kg = sel_winner(records, world, nullptr);
    bool keep = false;
    int o = 0, cnt_c = 0;
    if (c < Mc) {
        o = orig ? orig[c] : c;
        const int cnt = counts[c];
        cnt_c = cnt;
        const bool is_winner = kg && cnt >= 0 && sel_key(cnt, my_off + (unsigned int)o) == kg;
        keep = cnt >= need && !is_winner;
    }
    __shared__ int s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int off = 0;
    if (keep) off = atomicAdd(&s_cnt, 1);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt > 0) s_base = atomicAdd(&rec[2], s_cnt);
    __syncthreads();
    if (keep) {
        const int pos = s_base + off;
        next_orig[pos] = o;
        if (next_counts) next_counts[pos] = cnt_c;       // what the candidate counts on the support set of THIS round: the next round subtracts what leaves
        const double* h = Hs + 9 * (size_t)c;
        for (int q = 0; q < 9; ++q) next_H[9 * (size_t)pos + q] = h[q];
    }
}

// The winner's inliers leave the support mask; its H joins the output list (one thread).  key_check (nullable): the
// winner as the all-gathered score vector of the first round gives it — must equal the records' winner.
__global__ void __launch_bounds__(256)
k_sel_claim(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
            const double* __restrict__ y2, int N, const SelRecord* __restrict__ records, int world,
            const unsigned long long* __restrict__ key_check, double thr2, int need, unsigned char* __restrict__ mask,
            int* __restrict__ rec, double* __restrict__ sel_H, long long* __restrict__ sel_counter, int max_models, int symmetric,
            const double* __restrict__ refit /* nullable: 9 doubles + the refit's inlier count as a double */,
            double* __restrict__ cx1, double* __restrict__ cy1, double* __restrict__ cx2, double* __restrict__ cy2 /* nullable: the points that leave, packed */)
{
    int wr = 0;
    const unsigned long long kg = sel_winner(records, world, &wr);
    const int best = (int)(kg >> 32);
    const unsigned int pos = 0xffffffffu - (unsigned int)(kg & 0xffffffffull);
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0) {
        rec[0] = kg ? best : -1;
        rec[1] = (int)pos;
        int err = 0;
        for (int r = 0; r < world; ++r) if (records[r].err) err = records[r].err;
        if (key_check && *key_check != kg) err = 2;
        for (int r = 1; r < world; ++r) if (records[r].mode != records[0].mode) err = 3;      // forward on one rank, symmetric on another; or winners refitted on one only
        if (err) rec[4] = err;
    }
    if (!kg || best < need) return;
    const int sel = rec[3];
    if (sel >= max_models) return;
    const double* h = records[wr].H;
    // r05: the winner refitted to its own inliers (k_sel_winner_labels -> k_haf_reestimate -> k_sel_count) takes its place
    // when the refit is finite and explains at least as many points of the support set as the hypothesis did
    if (refit && refit[9] >= (double)best) {
        bool finite = true;
        for (int q = 0; q < 9; ++q) finite = finite && fabs(refit[q]) < 0x1p1000;
        if (finite) h = refit;
    }
    if (n == 0) {
        for (int q = 0; q < 9; ++q) sel_H[9 * (size_t)sel + q] = h[q];
        sel_counter[sel] = (long long)pos;
    }
    bool leaves = false;
    if (n < N && mask[n]) {
        double d2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], x1[n], y1[n], x2[n], y2[n]);
        if (symmetric) {
            // + the backward transfer through adj(H), every entry (mul, mul, sub) — the arithmetic of the scoring kernel's
            // symmetric mode (residual.hip), so the points that leave are the points that were counted
            const double a0 = h[4] * h[8] - h[5] * h[7], a1 = h[2] * h[7] - h[1] * h[8], a2 = h[1] * h[5] - h[2] * h[4];
            const double a3 = h[5] * h[6] - h[3] * h[8], a4 = h[0] * h[8] - h[2] * h[6], a5 = h[2] * h[3] - h[0] * h[5];
            const double a6 = h[3] * h[7] - h[4] * h[6], a7 = h[1] * h[6] - h[0] * h[7], a8 = h[0] * h[4] - h[1] * h[3];
            d2 = d2 + fwd_d2(a0, a1, a2, a3, a4, a5, a6, a7, a8, x2[n], y2[n], x1[n], y1[n]);
        }
        leaves = d2 < thr2;
        if (leaves) mask[n] = 0;
    }
    // rec[5]: how many points left the support set (the winner's count — or its refit's: the host keeps the size of the set);
    // the points themselves are packed (any order) for the next round, which subtracts what every candidate counted on them
    const unsigned long long lm = __ballot(leaves);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && lm) base = atomicAdd(&rec[5], (int)__popcll(lm));
    base = __shfl(base, 0, 64);
    if (leaves && cx1) {
        const int at = base + (int)__popcll(lm & ((1ull << lane) - 1ull));
        cx1[at] = x1[n]; cy1[at] = y1[n]; cx2[at] = x2[n]; cy2[at] = y2[n];
    }
}

// counts[c] = carried[c] - left[c]: what candidate c counts on the support set once the points of the last claim are gone
__global__ void __launch_bounds__(256)
k_sel_subtract(const int* __restrict__ carried, const int* __restrict__ left, int Mc, int* __restrict__ counts)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < Mc) counts[c] = carried[c] - left[c];
}

// ---- the winner refitted to its inliers before it claims them (r05, mh_set_tuning key 30) --------------------------------
// A hypothesis fitted to four matches explains 60-70 % of its plane at the inlier threshold; what it leaves behind is
// raw material for a later winner that sits BETWEEN two planes.  With the refit the winner of a round is
// re-estimated from the points of the support set it explains — the per-label HAF least squares of the loop
// (k_haf_reestimate, reestimate.hip, M/MultiH.cpp:913-989) with one label — and the refit takes the hypothesis' place if
// it explains at least as many points.
// labels[n] = 0 for the winner's inliers in the support set, -1 elsewhere; refit[0..9) = the winner's H (what a label
// without points keeps), refit[9] = 0.
__global__ void __launch_bounds__(256)
k_sel_winner_labels(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
                    const double* __restrict__ y2, int N, const SelRecord* __restrict__ records, int world, double thr2, int need,
                    const unsigned char* __restrict__ mask, int* __restrict__ labels, double* __restrict__ refit, int symmetric)
{
    int wr = 0;
    const unsigned long long kg = sel_winner(records, world, &wr);
    const int n = blockIdx.x * 256 + threadIdx.x;
    const double* h = records[wr].H;
    if (n == 0) {
        for (int q = 0; q < 9; ++q) refit[q] = h[q];
        refit[9] = 0.0;
    }
    if (n >= N) return;
    int lab = -1;
    if (kg && (int)(kg >> 32) >= need && mask[n]) {
        double d2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], x1[n], y1[n], x2[n], y2[n]);
        if (symmetric) {
            const double a0 = h[4] * h[8] - h[5] * h[7], a1 = h[2] * h[7] - h[1] * h[8], a2 = h[1] * h[5] - h[2] * h[4];
            const double a3 = h[5] * h[6] - h[3] * h[8], a4 = h[0] * h[8] - h[2] * h[6], a5 = h[2] * h[3] - h[0] * h[5];
            const double a6 = h[3] * h[7] - h[4] * h[6], a7 = h[1] * h[6] - h[0] * h[7], a8 = h[0] * h[4] - h[1] * h[3];
            d2 = d2 + fwd_d2(a0, a1, a2, a3, a4, a5, a6, a7, a8, x2[n], y2[n], x1[n], y1[n]);
        }
        if (d2 < thr2) lab = 0;
    }
    labels[n] = lab;
}

// refit[9] = number of points of the support set inside thr of the refit (an integer count kept in a double: exact)
__global__ void __launch_bounds__(256)
k_sel_count(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
            const double* __restrict__ y2, int N, double thr2, const unsigned char* __restrict__ mask, double* __restrict__ refit,
            int* __restrict__ counter, int symmetric)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const double* h = refit;
    bool in = false;
    if (n < N && mask[n]) {
        double d2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], x1[n], y1[n], x2[n], y2[n]);
        if (symmetric) {
            const double a0 = h[4] * h[8] - h[5] * h[7], a1 = h[2] * h[7] - h[1] * h[8], a2 = h[1] * h[5] - h[2] * h[4];
            const double a3 = h[5] * h[6] - h[3] * h[8], a4 = h[0] * h[8] - h[2] * h[6], a5 = h[2] * h[3] - h[0] * h[5];
            const double a6 = h[3] * h[7] - h[4] * h[6], a7 = h[1] * h[6] - h[0] * h[7], a8 = h[0] * h[4] - h[1] * h[3];
            d2 = d2 + fwd_d2(a0, a1, a2, a3, a4, a5, a6, a7, a8, x2[n], y2[n], x1[n], y1[n]);
        }
        in = d2 < thr2;
    }
    const unsigned long long m = __ballot(in);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (int)__popcll(m));
}

__global__ void k_sel_count_finish(const int* __restrict__ counter, double* __restrict__ refit)
{
    if (threadIdx.x == 0) refit[9] = (double)*counter;
}

__global__ void k_sel_publish(int* __restrict__ rec, unsigned long long* __restrict__ keys, SelRecord* __restrict__ my_record,
                              int need, int* __restrict__ h_rec)
{
    if (threadIdx.x != 0) return;
    h_rec[0] = rec[0]; h_rec[1] = rec[1]; h_rec[2] = rec[2]; h_rec[4] = rec[4]; h_rec[5] = rec[5];
    if (rec[0] >= need) rec[3] += 1;
    h_rec[3] = rec[3];
    rec[2] = 0;
    rec[5] = 0;
    keys[0] = 0; keys[1] = 0;
    my_record->key = 0;
}

// Best model of a scored batch (the step bench.py times): arg-max of the resident counts — or of the all-gathered
// vector — then one word pair for the host.
__global__ void k_best_publish(unsigned long long* __restrict__ key, int* __restrict__ h_best)
{
    if (threadIdx.x != 0) return;
    const unsigned long long k = *key;
    h_best[0] = k ? (int)(k >> 32) : -1;
    h_best[1] = (int)(0xffffffffu - (unsigned int)(k & 0xffffffffull));
    h_best[2] += 1;                                   // sequence number: the host can tell a fresh result from an old one
    *key = 0;
}

// Arg-max + publication + clearing in ONE launch (r04): what follows a sweep (or the all-gather of a sharded batch) on the
// exchange's stream.  Kernels dispatched while the next sweep is resident do not run before it ends, so the
// fewer dispatches an exchange consists of, the more of it fits into the gap between two sweeps.  One workgroup: the best
// key over `scores` — plain (world = 1: entry j is hypothesis j) or gathered (world x longest, -1 = padding, entry (r, j) =
// hypothesis r * base + min(r, rem) + j) —, the three words for the host, then `clear` (nullable; may be `scores` itself)
// is zeroed for the sweep after next.
__global__ void __launch_bounds__(1024)
k_best_fused(int* scores, int world, int longest, int base, int rem, int* __restrict__ h_best, int* clear, int clear_count)
{
    // (`scores` and `clear` may be the same buffer — no __restrict__ on either; every score is read before the barrier,
    // nothing is cleared before it.)  An entry below -1 is a rank's ERROR MARKER (mh_select_best: a rank-local failure
    // travels through the collective instead of leaving it): word 3 tells every rank's host.
    const int total = world * longest;
    unsigned long long k = 0;
    int err = 0;
    for (int c = threadIdx.x; c < total; c += 1024) {
        const int v = scores[c];
        if (v < 0) { err |= v < -1 ? 1 : 0; continue; }
        const int r = world > 1 ? c / longest : 0, j = c - r * longest;
        const unsigned long long kk = sel_key(v, (unsigned int)(r * base + (r < rem ? r : rem) + j));
        k = kk > k ? kk : k;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(k, m, 64); k = o > k ? o : k; }
    err = __ballot(err != 0) != 0ull ? 1 : 0;
    __shared__ unsigned long long s_k[16];
    __shared__ int s_err[16];
    if ((threadIdx.x & 63) == 0) { s_k[threadIdx.x >> 6] = k; s_err[threadIdx.x >> 6] = err; }
    __syncthreads();                                  // (also: every score has been read before anything is cleared)
    if (threadIdx.x == 0) {
        unsigned long long b = 0;
        int any = 0;
        for (int w = 0; w < 16; ++w) { b = s_k[w] > b ? s_k[w] : b; any |= s_err[w]; }
        h_best[0] = b ? (int)(b >> 32) : -1;
        h_best[1] = (int)(0xffffffffu - (unsigned int)(b & 0xffffffffull));
        h_best[3] = any;
        h_best[2] += 1;                               // sequence number: the host can tell a fresh result from an old one
    }
    if (clear) for (int c = threadIdx.x; c < clear_count; c += 1024) clear[c] = 0;
}

// scores[0, m) <- counts, scores[m, longest) <- -1: the send buffer of the score all-gather
__global__ void __launch_bounds__(256)
k_pad_scores(const int* __restrict__ counts, int m, int longest, int* __restrict__ scores)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < longest) scores[c] = c < m ? counts[c] : -1;
}

// The points that are still in the support set, packed: the re-scoring of a round then sweeps only them (the score
// kernel is bound by FP64 work per pair, so a pass costs what the points it sees cost).  The order of the packed
// points is whatever the atomics make it — an inlier COUNT does not depend on it.  `count` must be zero on entry.
__global__ void __launch_bounds__(256)
k_sel_pack_points(Points p, const unsigned char* __restrict__ mask, double* __restrict__ cx1, double* __restrict__ cy1,
                  double* __restrict__ cx2, double* __restrict__ cy2, int* __restrict__ count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool on = i < p.n && mask[i] != 0;
    const unsigned long long m = __ballot(on);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(count, (int)__popcll(m));
    base = __shfl(base, (int)__builtin_ctzll(m));
    if (on) {
        const int pos = base + (int)__popcll(m & ((1ull << lane) - 1ull));
        cx1[pos] = p.x1[i]; cy1[pos] = p.y1[i]; cx2[pos] = p.x2[i]; cy2[pos] = p.y2[i];
    }
}

hipError_t launch_sel_pack_points(const Points& p, const unsigned char* mask, double* cx1, double* cy1, double* cx2, double* cy2,
                                  int* count, hipStream_t s)
{
    hipError_t he = hipMemsetAsync(count, 0, sizeof(int), s);
    if (he != hipSuccess) return he;
    hipLaunchKernelGGL(k_sel_pack_points, dim3((p.n + 255) / 256), dim3(256), 0, s, p, mask, cx1, cy1, cx2, cy2, count);
    return hipGetLastError();
}

hipError_t launch_sel_argmax(const int* counts, const int* orig, int Mc, unsigned int my_off, unsigned long long* key,
                             int* scores_full, hipStream_t s)
{
    if (Mc <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sel_argmax, dim3((Mc + 255) / 256), dim3(256), 0, s, counts, orig, Mc, my_off, key, scores_full);
    return hipGetLastError();
}

hipError_t launch_sel_argmax_gathered(const int* gathered, int world, int longest, int base, int rem, unsigned long long* key,
                                      hipStream_t s)
{
    if (world * longest <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sel_argmax_gathered, dim3((world * longest + 255) / 256), dim3(256), 0, s, gathered, world, longest,
                       base, rem, key);
    return hipGetLastError();
}

hipError_t launch_sel_record(const int* counts, const int* orig, const double* Hs, int Mc, unsigned int my_off,
                             const unsigned long long* key_local, int err, int mode, SelRecord* record, hipStream_t s)
{
    hipLaunchKernelGGL(k_sel_record, dim3(Mc > 0 ? (Mc + 255) / 256 : 1), dim3(256), 0, s, counts, orig, Hs, Mc, my_off, key_local,
                       err, mode, record);
    return hipGetLastError();
}

hipError_t launch_sel_compact(const int* counts, const int* orig, const double* Hs, int Mc, int need, const SelRecord* records,
                              int world, unsigned int my_off, int* next_orig, double* next_H, int* rec, int* next_counts, hipStream_t s)
{
    if (Mc <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sel_compact, dim3((Mc + 255) / 256), dim3(256), 0, s, counts, orig, Hs, Mc, need, records, world, my_off,
                       next_orig, next_H, rec, next_counts);
    return hipGetLastError();
}

hipError_t launch_sel_subtract(const int* carried, const int* left, int Mc, int* counts, hipStream_t s)
{
    if (Mc <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sel_subtract, dim3((Mc + 255) / 256), dim3(256), 0, s, carried, left, Mc, counts);
    return hipGetLastError();
}

hipError_t launch_sel_claim(const Points& p, const SelRecord* records, int world, const unsigned long long* key_check, double thr2,
                            int need, unsigned char* mask, int* rec, double* sel_H, long long* sel_counter, int max_models,
                            hipStream_t s, int symmetric, const double* refit, double* cx1, double* cy1, double* cx2, double* cy2)
{
    hipLaunchKernelGGL(k_sel_claim, dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, records, world,
                       key_check, thr2, need, mask, rec, sel_H, sel_counter, max_models, symmetric, refit, cx1, cy1, cx2, cy2);
    return hipGetLastError();
}

// the round's winner refitted to its inliers: labels (n ints), refit (10 doubles), counter (1 int) are scratch of the caller
hipError_t launch_sel_refit(const Points& p, const Affines& a, const Epipolar& ep, const SelRecord* records, int world, double thr2,
                            int need, const unsigned char* mask, int* labels, double* refit, int* counter, int* label_count,
                            hipStream_t s, int symmetric)
{
    const dim3 grid((p.n + 255) / 256);
    hipLaunchKernelGGL(k_sel_winner_labels, grid, dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, records, world, thr2, need, mask,
                       labels, refit, symmetric);
    hipError_t he = launch_reestimate(p, a, labels, 1, ep, refit, label_count, s);
    if (he != hipSuccess) return he;
    he = hipMemsetAsync(counter, 0, sizeof(int), s);
    if (he != hipSuccess) return he;
    hipLaunchKernelGGL(k_sel_count, grid, dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, thr2, mask, refit, counter, symmetric);
    hipLaunchKernelGGL(k_sel_count_finish, dim3(1), dim3(64), 0, s, counter, refit);
    return hipGetLastError();
}

hipError_t launch_sel_publish(int* rec, unsigned long long* keys, SelRecord* my_record, int need, int* h_rec_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_sel_publish, dim3(1), dim3(64), 0, s, rec, keys, my_record, need, h_rec_dev);
    return hipGetLastError();
}

hipError_t launch_best_publish(unsigned long long* key, int* h_best_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_best_publish, dim3(1), dim3(64), 0, s, key, h_best_dev);
    return hipGetLastError();
}

hipError_t launch_best_fused(int* scores, int world, int longest, int base, int rem, int* h_best_dev, int* clear,
                             int clear_count, hipStream_t s)
{
    hipLaunchKernelGGL(k_best_fused, dim3(1), dim3(1024), 0, s, scores, world, longest, base, rem, h_best_dev, clear, clear_count);
    return hipGetLastError();
}

hipError_t launch_pad_scores(const int* counts, int m, int longest, int* scores, hipStream_t s)
{
    if (longest <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pad_scores, dim3((longest + 255) / 256), dim3(256), 0, s, counts, m, longest, scores);
    return hipGetLastError();
}

} // namespace mh
