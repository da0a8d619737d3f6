// knn.hip — exact k-nearest-neighbour hit list in the reference's 4-D float32
// space, gfx950.  SURVEY §8(f) row 1: replaces the FLANN radius search of
// ClusterMergingAndLabeling (M/MultiH.cpp:233-253), which is approximate,
// RNG-dependent and allocates N x N matrices inside OpenCV.  DEVIATION (stated
// in DESIGN.md): exact kNN instead of approximate radius search; the hit list
// has the same shape (directed hits per query) and feeds mh_set_neighbors_csr
// semantics unchanged.
//
// Point vectors are (float)x1, (float)y1, (float)x2, (float)y2 (:242-246).
// Distance = ((dx*dx + dy*dy) + dz*dz) + dw*dw in float32, rounding once per op;
// ties are broken by the smaller index, so the result is a pure function of
// the input.  One thread per query; candidates stream through LDS in tiles of
// 256 (16 B each, conflict-free broadcast reads); the running top-K is a
// sorted register array with a fully unrolled insertion.  One thread per query
// alone leaves the chip empty (50k queries = 196 workgroups on 256 CUs, one
// wave per SIMD), so the candidate range is cut into `splits` slices
// (blockIdx.y), each slice keeps its own top-K per query, and k_knn_merge takes
// the K smallest (distance, index) pairs of the slices' lists: the same set in
// the same order as one pass over all candidates.

#include "mh_kernels.hpp"

#include <algorithm>
#include <cmath>

namespace mh {

template <int K>
__global__ void __launch_bounds__(256)
k_knn(const double* __restrict__ x1, const double* __restrict__ y1,
      const double* __restrict__ x2, const double* __restrict__ y2, int N, int k, int slice,
      int* __restrict__ out, float* __restrict__ part_d, int* __restrict__ part_i)
{
    __shared__ float4 tile[256];
    const int q = blockIdx.x * 256 + threadIdx.x;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < N) me = make_float4((float)x1[q], (float)y1[q], (float)x2[q], (float)y2[q]);
    float bd[K];
    int bi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = 0x7fffffff; }

    const int first = blockIdx.y * slice;                       // my slice of the candidates (a multiple of 256 long)
    const int last = (first + slice) < N ? (first + slice) : N;
    for (int base = first; base < last; base += 256) {
        const int c = base + threadIdx.x;
        __syncthreads();
        if (c < N) tile[threadIdx.x] = make_float4((float)x1[c], (float)y1[c], (float)x2[c], (float)y2[c]);
        __syncthreads();
        const int lim = (last - base) < 256 ? (last - base) : 256;
        for (int t = 0; t < lim; ++t) {
            const int j = base + t;
            const float4 o = tile[t];
            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z, dw = me.w - o.w;
            const float d = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            if (j == q) continue;
            // (d, j) < (bd[K-1], bi[K-1]) lexicographically; j increases, so on equal d the
            // incumbent (smaller index) stays.
            if (d < bd[K - 1]) {
                float cd = d;
                int ci = j;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const bool lt = (cd < bd[i]) || (cd == bd[i] && ci < bi[i]);
                    const float td = bd[i];
                    const int ti = bi[i];
                    bd[i] = lt ? cd : td;
                    bi[i] = lt ? ci : ti;
                    cd = lt ? td : cd;
                    ci = lt ? ti : ci;
                }
            }
        }
    }
    if (q >= N) return;
    if (gridDim.y == 1) {
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (i < k) out[(size_t)q * k + i] = bi[i];
    } else {
        const size_t o = ((size_t)blockIdx.y * N + q) * K;
#pragma unroll
        for (int i = 0; i < K; ++i) { part_d[o + i] = bd[i]; part_i[o + i] = bi[i]; }
    }
}

// The K smallest (distance, index) pairs of `splits` sorted lists per query: a K-step merge by list heads.
template <int K>
__global__ void __launch_bounds__(256)
k_knn_merge(int N, int k, int splits, const float* __restrict__ part_d, const int* __restrict__ part_i,
            int* __restrict__ out)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= N) return;
    int head[KNN_MAX_SPLITS];
    for (int s = 0; s < splits; ++s) head[s] = 0;
    for (int i = 0; i < k; ++i) {
        float bd = __builtin_inff();
        int bi = 0x7fffffff, bs = 0;
        for (int s = 0; s < splits; ++s) {
            if (head[s] >= K) continue;
            const size_t o = ((size_t)s * N + q) * K + head[s];
            const float d = part_d[o];
            const int j = part_i[o];
            if (d < bd || (d == bd && j < bi)) { bd = d; bi = j; bs = s; }
        }
        out[(size_t)q * k + i] = bi;
        ++head[bs];
    }
}

// ---- the same k nearest hits through a grid over the source image (r05) ------------------------------------------------
// The pass above compares every query with every point: 2.5 10^9 pairs and 4.5 ms at 50 000 points, growing with N^2.  A
// query's k nearest lie within a few dozen pixels, so the points are binned by (x1, y1) into G x G cells (about three points
// per cell, G <= 256), sorted by cell — a run of cells in one grid row is then ONE contiguous range of the sorted array —
// and a query walks the square rings of cells around its own until no unexamined point can enter its list: a point r + 1
// or more cells away in x or in y is at least (r - 0.01) c away in the plane (0.01 c covers the float32 rounding of the
// cell index, 5 10^-5 cells at most), hence in the 4-D space, so once the k-th best squared distance is below
// ((r - 0.01) c)^2 (1 - 10^-5) the walk ends.  Same float32 distance, same (distance, index) order, same self-exclusion:
// the same table as k_knn, entry for entry (tests: grid against the exhaustive pass on ties, duplicates, clusters).
struct KnnGrid { float x0, y0, inv_c, c; int G; };

__device__ __forceinline__ int grid_coord(float v, float v0, float inv_c, int G)
{
    const int i = (int)((v - v0) * inv_c);
    return i < 0 ? 0 : (i >= G ? G - 1 : i);
}

__global__ void __launch_bounds__(256)
k_grid_count(const double* __restrict__ x1, const double* __restrict__ y1, int N, KnnGrid g, int* __restrict__ cell_of,
             int* __restrict__ count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int c = grid_coord((float)y1[i], g.y0, g.inv_c, g.G) * g.G + grid_coord((float)x1[i], g.x0, g.inv_c, g.G);
    cell_of[i] = c;
    atomicAdd(&count[c], 1);
}

// start[0 .. cells] = exclusive prefix sums of count[0 .. cells); count is zeroed for its second life as the scatter's cursor
__global__ void __launch_bounds__(1024)
k_grid_scan(int* __restrict__ count, int* __restrict__ start, int cells)
{
    __shared__ int s_part[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int b = 0; b < cells; b += 1024) {
        const int i = b + threadIdx.x;
        const int v = i < cells ? count[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += s_part[w];
        if (i < cells) { start[i] = before + incl - v; count[i] = 0; }
        __syncthreads();
        if (threadIdx.x == 1023) s_base = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[cells] = s_base;
}

__global__ void __launch_bounds__(256)
k_grid_scatter(const double* __restrict__ x1, const double* __restrict__ y1, const double* __restrict__ x2,
               const double* __restrict__ y2, int N, const int* __restrict__ cell_of, const int* __restrict__ start,
               int* __restrict__ cursor, float4* __restrict__ P, int* __restrict__ orig)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int c = cell_of[i];
    const int pos = start[c] + atomicAdd(&cursor[c], 1);       // any order inside a cell: the list below does not depend on it
    P[pos] = make_float4((float)x1[i], (float)y1[i], (float)x2[i], (float)y2[i]);
    orig[pos] = i;
}

// One WAVE per query: the 64 lanes examine 64 candidates of a range at once, the list lives in lanes 0 .. K-1 (one entry
// per lane, sorted) and a candidate that beats its last entry is inserted with one comparison per lane and a shift by one
// lane.  (One thread per query — the exhaustive pass's form — left the chip empty here: 50 000 threads of dependent loads,
// an outlier's long walk holding its whole wave: 3.0 ms against 4.4 ms exhaustive; this form 1.4 ms — the neighbourhood build of 50 000
// points 4.96 -> 1.86 ms, `tools/knn_probe.py`.  What is left is the walk of the 25 % gross outliers, whose 16 nearest in the 4-D
// space lie a dozen rings out: some 300 ranges per query, most of them single cells on the ring's sides.)
template <int K>
__global__ void __launch_bounds__(256)
k_knn_grid(int N, int k, KnnGrid g, const int* __restrict__ start, const float4* __restrict__ P, const int* __restrict__ orig,
           int* __restrict__ out)
{
    static_assert(K <= 64, "the list has one entry per lane");
    const int lane = threadIdx.x & 63;
    const int G = g.G;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < N; s += gridDim.x * 4) {      // queries in cell order
        const float4 me = P[s];
        const int q = orig[s];
        const int cx = grid_coord(me.x, g.x0, g.inv_c, G), cy = grid_coord(me.y, g.y0, g.inv_c, G);
        float bd = __builtin_inff();                            // lanes >= K stay at (+inf, max): never read
        int bi = 0x7fffffff;
        float kd = __builtin_inff();                            // the list's last entry (wave-uniform)
        int ki = 0x7fffffff;
        auto range = [&](int first, int last) {                 // sorted positions [first, last)
            for (int t0 = first; t0 < last; t0 += 64) {
                const int t = t0 + lane;
                const bool valid = t < last && t != s;
                const float4 o = P[valid ? t : s];
                const int j = orig[valid ? t : s];
                const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z, dw = me.w - o.w;
                const float d = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
                unsigned long long pass = __ballot(valid && (d < kd || (d == kd && j < ki)));
                while (pass) {
                    const int b = (int)__builtin_ctzll(pass);
                    pass &= pass - 1;
                    const float cd = __shfl(d, b, 64);
                    const int cj = __shfl(j, b, 64);
                    if (!(cd < kd || (cd == kd && cj < ki))) continue;          // the list has moved on since the ballot
                    // position of the newcomer = entries (among the K) that stay in front of it, lexicographically by (d, j)
                    const bool front = lane < K && ((bd < cd) || (bd == cd && bi < cj));
                    const int pos = (int)__popcll(__ballot(front));
                    const float ud = __shfl_up(bd, 1, 64);
                    const int ui = __shfl_up(bi, 1, 64);
                    if (lane == pos) { bd = cd; bi = cj; }
                    else if (lane > pos && lane < K) { bd = ud; bi = ui; }
                    kd = __shfl(bd, K - 1, 64);
                    ki = __shfl(bi, K - 1, 64);
                }
            }
        };
        for (int r = 0; r < G; ++r) {
            const int xlo = cx - r, xhi = cx + r, ylo = cy - r, yhi = cy + r;
            const int xa = xlo < 0 ? 0 : xlo, xb = xhi >= G ? G - 1 : xhi;
            for (int yy = (ylo < 0 ? 0 : ylo); yy <= (yhi >= G ? G - 1 : yhi); ++yy) {
                if (yy == ylo || yy == yhi) {
                    range(start[yy * G + xa], start[yy * G + xb + 1]);      // a whole row of the ring: one contiguous range
                } else {
                    if (xlo >= 0) range(start[yy * G + xlo], start[yy * G + xlo + 1]);
                    if (xhi < G) range(start[yy * G + xhi], start[yy * G + xhi + 1]);
                }
            }
            if (xlo <= 0 && ylo <= 0 && xhi >= G - 1 && yhi >= G - 1) break;        // the whole grid has been seen
            if (r >= 1) {
                const float reach = ((float)r - 0.01f) * g.c;               // no unexamined point is closer than this
                // r06 (advisor): the walk ends on the k-th entry — the last one that is output — not on the K-th of the list
                // (K = 8 / 16 / 32 >= k: for k = 9 .. 15 or 17 .. 31 it ran several rings longer than the table needs)
                const float kdk = __shfl(bd, k - 1, 64);
                if (kdk < reach * reach * (1.0f - 1e-5f)) break;
            }
        }
        if (lane < k) out[(size_t)q * k + lane] = bi;
    }
}

template <int K>
static hipError_t launch_knn_grid_k(const Points& p, int k, int* nbr_out, const KnnGrid& g, int* cell_of, int* count, int* start,
                                    float4* P, int* orig, hipStream_t s)
{
    const int blocks = (p.n + 255) / 256, cells = g.G * g.G;
    hipError_t he = hipMemsetAsync(count, 0, sizeof(int) * (size_t)cells, s);
    if (he != hipSuccess) return he;
    hipLaunchKernelGGL(k_grid_count, dim3(blocks), dim3(256), 0, s, p.x1, p.y1, p.n, g, cell_of, count);
    hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, s, count, start, cells);
    hipLaunchKernelGGL(k_grid_scatter, dim3(blocks), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, cell_of, start, count, P, orig);
    const int waves_blocks = std::min((p.n + 3) / 4, 4096);      // a wave per query, queries strided over the resident workgroups
    hipLaunchKernelGGL((k_knn_grid<K>), dim3(waves_blocks), dim3(256), 0, s, p.n, k, g, start, P, orig, nbr_out);
    return hipGetLastError();
}

int knn_grid_cells(int n)
{
    int G = (int)std::ceil(std::sqrt((double)n / 3.0));
    G = G < 1 ? 1 : (G > 256 ? 256 : G);
    return G;
}

// scratch: cell_of n ints, count / start G*G (+1) ints, P n float4 (as 4n floats), orig n ints; the source points' bounding
// box (p.xmin ..) must be finite.  Same table as launch_knn.
hipError_t launch_knn_grid(const Points& p, int k, int* nbr_out, int* cell_of, int* count, int* start, float* P4, int* orig,
                           hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    const int G = knn_grid_cells(p.n);
    const double range = std::fmax(p.xmax - p.xmin, p.ymax - p.ymin);
    // r06 (advisor): no grid when all source points coincide (every point in cell 0, every query still walking up to G empty
    // rings) or when the cell size does not fit float32 (c = inf, inv_c = 0: the same); hipErrorInvalidValue sends the caller
    // to the exhaustive pass
    if (!(range > 0.0) || !std::isfinite(range) || !std::isfinite(p.xmin) || !std::isfinite(p.ymin) ||
        !std::isfinite((float)(range / G)) || !((float)(G / range) > 0.0f) || !std::isfinite((float)(G / range)))
        return hipErrorInvalidValue;
    KnnGrid g;
    g.G = G;
    g.c = (float)(range / G);
    g.inv_c = (float)(G / range);
    g.x0 = (float)p.xmin;
    g.y0 = (float)p.ymin;
    float4* P = reinterpret_cast<float4*>(P4);
    if (k <= 8) return launch_knn_grid_k<8>(p, k, nbr_out, g, cell_of, count, start, P, orig, s);
    if (k <= 16) return launch_knn_grid_k<16>(p, k, nbr_out, g, cell_of, count, start, P, orig, s);
    if (k <= 32) return launch_knn_grid_k<32>(p, k, nbr_out, g, cell_of, count, start, P, orig, s);
    return hipErrorInvalidValue;
}

// Exact radius search — the reference's own neighbourhood rule (M/MultiH.cpp:252-253: radiusMatch with
// maxDistance = 1/locality in the same float32 4-D space; FLANN answers it approximately, this is the
// exact set).  Hit  <=>  ((dx^2+dy^2)+dz^2)+dw^2 <= r2 in float32, the query itself included (FLANN
// returns it at distance 0; LabelingStep skips it, :537).  Two passes over the same tiles: FILL =
// false counts the hits of each query, FILL = true writes them in increasing index order at
// rowptr[q] (the host does the prefix sum in between).
template <bool FILL>
__global__ void __launch_bounds__(256)
k_radius(const double* __restrict__ x1, const double* __restrict__ y1,
         const double* __restrict__ x2, const double* __restrict__ y2, int N, float r2,
         int* __restrict__ counts, const int* __restrict__ rowptr, int* __restrict__ col)
{
    __shared__ float4 tile[256];
    const int q = blockIdx.x * 256 + threadIdx.x;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < N) me = make_float4((float)x1[q], (float)y1[q], (float)x2[q], (float)y2[q]);
    int cnt = 0;
    int* dst = (FILL && q < N) ? col + rowptr[q] : nullptr;
    for (int base = 0; base < N; base += 256) {
        const int c = base + threadIdx.x;
        __syncthreads();
        if (c < N) tile[threadIdx.x] = make_float4((float)x1[c], (float)y1[c], (float)x2[c], (float)y2[c]);
        __syncthreads();
        const int lim = (N - base) < 256 ? (N - base) : 256;
        for (int t = 0; t < lim; ++t) {
            const float4 o = tile[t];
            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z, dw = me.w - o.w;
            const float d = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            if (d <= r2) {
                if (FILL && q < N) dst[cnt] = base + t;
                ++cnt;
            }
        }
    }
    if (!FILL && q < N) counts[q] = cnt;
}

hipError_t launch_radius_count(const Points& p, float r2, int* counts, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL((k_radius<false>), dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, r2,
                       counts, nullptr, nullptr);
    return hipGetLastError();
}

hipError_t launch_radius_fill(const Points& p, float r2, const int* rowptr, int* col, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL((k_radius<true>), dim3((p.n + 255) / 256), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, r2,
                       nullptr, rowptr, col);
    return hipGetLastError();
}

template <int K>
static hipError_t launch_knn_k(const Points& p, int k, int* nbr_out, int splits, float* part_d, int* part_i, hipStream_t s)
{
    const int blocks = (p.n + 255) / 256;
    int slice = ((p.n + splits - 1) / splits + 255) / 256 * 256;
    if (slice < 256) slice = 256;
    hipLaunchKernelGGL((k_knn<K>), dim3(blocks, splits), dim3(256), 0, s, p.x1, p.y1, p.x2, p.y2, p.n, k, slice, nbr_out,
                       part_d, part_i);
    if (splits > 1)
        hipLaunchKernelGGL((k_knn_merge<K>), dim3(blocks), dim3(256), 0, s, p.n, k, splits, part_d, part_i, nbr_out);
    return hipGetLastError();
}

// splits > 1 needs scratch for the slices' lists: splits x n x K floats and ints (K = 8, 16 or 32, the smallest >= k)
hipError_t launch_knn(const Points& p, int k, int* nbr_out, int splits, float* part_d, int* part_i, hipStream_t s)
{
    if (p.n <= 0) return hipSuccess;
    if (splits < 1 || splits > KNN_MAX_SPLITS || (splits > 1 && (!part_d || !part_i))) return hipErrorInvalidValue;
    if (k <= 8) return launch_knn_k<8>(p, k, nbr_out, splits, part_d, part_i, s);
    if (k <= 16) return launch_knn_k<16>(p, k, nbr_out, splits, part_d, part_i, s);
    if (k <= 32) return launch_knn_k<32>(p, k, nbr_out, splits, part_d, part_i, s);
    return hipErrorInvalidValue;
}

} // namespace mh
