// select.hip — greedy model selection over a resident hypothesis batch, entirely on the device.
//
// The sequential-RANSAC scheme of the dead M/MultipleHomographies.h:146-175 (take the best-supported hypothesis,
// take its inliers out of the support set, score again) as the engine runs it behind MultiH::ProposeModels:
// per round
//   launch_score     inlier counts of the candidate hypotheses over the points still in the support mask
//   k_sel_argmax     best candidate: highest count, lowest hypothesis counter on ties (one 64-bit atomicMax per
//                    workgroup on key = count << 32 | ~counter)
//   k_sel_compact    the winner's H goes to the output list; the candidates that can still win — count >= need,
//                    counts only fall as points leave the mask — are copied to the next round's list
//   k_sel_claim      the winner's inliers leave the mask
//   k_sel_publish    three control words for the host (mapped pinned memory): the round's best count decides whether
//                    there is another round; nothing else crosses the bus
// After the first round the candidate list is a small fraction of the batch, so later rounds are short.
// Sharded batches (one process per GPU): between argmax and compact the ranks all-gather their int32 score vectors
// (device pointers; RCCL on a real node) and the 72-byte H of their local best, and every rank picks the same winner.
#include "mh_device.hpp"
#include "mh_kernels.hpp"

namespace mh {

__device__ __forceinline__ unsigned long long sel_key(int count, unsigned int counter)
{
    return ((unsigned long long)(unsigned int)count << 32) | (unsigned long long)(0xffffffffu - counter);
}

// counts[c] of candidate c whose hypothesis counter (position in its rank's batch) is orig[c] (identity when null).
// counter_base: added to the counter (sharded: the rank's offset is NOT added here; see k_sel_argmax_gathered).
__global__ void __launch_bounds__(256)
k_sel_argmax(const int* __restrict__ counts, const int* __restrict__ orig, int Mc, unsigned long long* __restrict__ key,
             int* __restrict__ scores_full /* nullable: scores_full[orig] = count */)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    unsigned long long k = 0;
    if (c < Mc) {
        const int o = orig ? orig[c] : c;
        const int cnt = counts[c];
        if (cnt >= 0) k = sel_key(cnt, (unsigned int)o);
        if (scores_full) scores_full[o] = cnt;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(k, m, 64); k = o > k ? o : k; }
    __shared__ unsigned long long s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = s[0];
        for (int w = 1; w < 4; ++w) b = s[w] > b ? s[w] : b;
        if (b) atomicMax(key, b);
    }
}

// gathered: world x longest scores in rank order (-1 = padding or pruned); counter of entry (r, j) = r * longest + j,
// which orders entries like the single-rank batch orders its hypotheses (the ranks own contiguous ascending ranges).
__global__ void __launch_bounds__(256)
k_sel_argmax_gathered(const int* __restrict__ gathered, int total, unsigned long long* __restrict__ key)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    unsigned long long k = 0;
    if (c < total && gathered[c] >= 0) k = sel_key(gathered[c], (unsigned int)c);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(k, m, 64); k = o > k ? o : k; }
    __shared__ unsigned long long s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = s[0];
        for (int w = 1; w < 4; ++w) b = s[w] > b ? s[w] : b;
        if (b) atomicMax(key, b);
    }
}

// rec: [0] best count of the round (global), [1] its counter, [2] next candidate count, [3] models selected so far.
// key_local: this rank's best; key_global: the round's winner (== key_local on one rank).  my_pos0: counter of this
// rank's hypothesis 0 in the gathered numbering (rank * longest), or 0.
__global__ void __launch_bounds__(256)
k_sel_compact(const int* __restrict__ counts, const int* __restrict__ orig, const double* __restrict__ Hs, int Mc, int need,
              const unsigned long long* __restrict__ key_local, const unsigned long long* __restrict__ key_global,
              unsigned int my_pos0, int* __restrict__ next_orig, double* __restrict__ next_H, int* __restrict__ rec,
              double* __restrict__ my_best_H, int* __restrict__ scores_full)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long kl = *key_local, kg = *key_global;
    bool keep = false;
    int o = 0;
    if (c < Mc) {
        o = orig ? orig[c] : c;
        const int cnt = counts[c];
        const double* h = Hs + 9 * (size_t)c;
        if (kl && sel_key(cnt, (unsigned int)o) == kl)
            for (int q = 0; q < 9; ++q) my_best_H[q] = h[q];                    // what this rank offers
        const bool is_winner = kg && cnt >= 0 && sel_key(cnt, my_pos0 + (unsigned int)o) == kg;
        keep = cnt >= need && !is_winner;
        if (!keep && scores_full) scores_full[o] = -1;                           // can never win again
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
        const double* h = Hs + 9 * (size_t)c;
        for (int q = 0; q < 9; ++q) next_H[9 * (size_t)pos + q] = h[q];
    }
}

// The winner's inliers leave the support mask; its H joins the output list (one thread).  all_H: world x 9 (the ranks'
// offers, rank order) or this rank's own offer when world == 1.
__global__ void __launch_bounds__(256)
k_sel_claim(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
            const double* __restrict__ y2, int N, const double* __restrict__ all_H, int longest,
            const unsigned long long* __restrict__ key_global, double thr2, int need, unsigned char* __restrict__ mask,
            int* __restrict__ rec, double* __restrict__ sel_H, long long* __restrict__ sel_counter, int max_models)
{
    const unsigned long long kg = *key_global;
    const int best = (int)(kg >> 32);
    const unsigned int pos = 0xffffffffu - (unsigned int)(kg & 0xffffffffull);
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0) { rec[0] = kg ? best : -1; rec[1] = (int)pos; }
    if (!kg || best < need) return;
    const int sel = rec[3];
    if (sel >= max_models) return;
    const double* h = all_H + 9 * (size_t)(longest > 0 ? pos / (unsigned int)longest : 0u);
    if (n == 0) {
        for (int q = 0; q < 9; ++q) sel_H[9 * (size_t)sel + q] = h[q];
        sel_counter[sel] = (long long)pos;
    }
    if (n >= N || !mask[n]) return;
    const double d2 = fwd_d2(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], x1[n], y1[n], x2[n], y2[n]);
    if (d2 < thr2) mask[n] = 0;
}

__global__ void k_sel_publish(int* __restrict__ rec, unsigned long long* __restrict__ keys, int need, int* __restrict__ h_rec)
{
    if (threadIdx.x != 0) return;
    h_rec[0] = rec[0]; h_rec[1] = rec[1]; h_rec[2] = rec[2];
    if (rec[0] >= need) rec[3] += 1;
    h_rec[3] = rec[3];
    rec[2] = 0;
    keys[0] = 0; keys[1] = 0;
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

hipError_t launch_sel_argmax(const int* counts, const int* orig, int Mc, unsigned long long* key, int* scores_full, hipStream_t s)
{
    if (Mc <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sel_argmax, dim3((Mc + 255) / 256), dim3(256), 0, s, counts, orig, Mc, key, scores_full);
    return hipGetLastError();
}

hipError_t launch_sel_argmax_gathered(const int* gathered, int total, unsigned long long* key, hipStream_t s)
{
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sel_argmax_gathered, dim3((total + 255) / 256), dim3(256), 0, s, gathered, total, key);
    return hipGetLastError();
}

hipError_t launch_sel_compact(const int* counts, const int* orig, const double* Hs, int Mc, int need,
                              const unsigned long long* key_local, const unsigned long long* key_global, unsigned int my_pos0,
                              int* next_orig, double* next_H, int* rec, double* my_best_H, int* scores_full, hipStream_t s)
{
    if (Mc <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sel_compact, dim3((Mc + 255) / 256), dim3(256), 0, s, counts, orig, Hs, Mc, need, key_local,
                       key_global, my_pos0, next_orig, next_H, rec, my_best_H, scores_full);
    return hipGetLastError();
}

hipError_t launch_sel_claim(const Points& p, const double* all_H, int longest, const unsigned long long* key_global, double thr2,
                            int need, unsigned char* mask, int* rec, double* sel_H, long long* sel_counter, int max_models,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_sel_claim, dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, all_H, longest,
                       key_global, thr2, need, mask, rec, sel_H, sel_counter, max_models);
    return hipGetLastError();
}

hipError_t launch_sel_publish(int* rec, unsigned long long* keys, int need, int* h_rec_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_sel_publish, dim3(1), dim3(64), 0, s, rec, keys, need, h_rec_dev);
    return hipGetLastError();
}

} // namespace mh
