// approx_neighbours.cpp — the reference's neighbourhood as FLANN answers it, re-enacted on the host.
//
// MultiH::ClusterMergingAndLabeling asks cv::FlannBasedMatcher::radiusMatch for every correspondence within
// 1 / locality_lambda of each correspondence in the float32 (x1, y1, x2, y2) space (M/MultiH.cpp:233-253).  FLANN's default
// index (OpenCV 3.1.0 / FLANN 1.6, outside /root/reference: restated from the published algorithm, parity unpinned) is
// KDTreeIndexParams(4): four randomised KD-trees — one point per leaf, the cut dimension drawn among the five dimensions of
// largest variance estimated on the first 100 points of the node, the cut value their mean in it — searched best-bin-first
// over all trees with ONE priority queue and SearchParams(32): a query EXAMINES at most 32 points and reports those of
// them inside the radius.  The answer is a few dozen approximate nearest neighbours, found one-way (a hit i -> j does not
// imply j -> i), never the ball itself.  This file builds that answer with the engine's counter RNG (splitmix64) in place
// of FLANN's rand(): MultiH::SetNeighbourApprox(trees, checks, seed).  tools/neighbourhood_sweep.py holds the same
// algorithm in Python (sequential sums, the same draws); tests/test_host_cpu.py compares the two hit for hit.
#include "merge_step.h"

#include <algorithm>
#include <cstdint>
#include <queue>
#include <vector>

namespace multih {

namespace {

struct Rng {
    uint64_t s;
    uint64_t below(uint64_t n)
    {
        s += 1;
        uint64_t z = s + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        return z % n;
    }
};

struct Node { int dim; double val; int lo, hi, point; };      // leaf: dim = -1, point = the row

// r06 (advisor): built with an explicit stack.  A tree split at the mean can be as deep as it has points (one gross outlier peeled
// off per level), and the recursive form needed a host stack frame per level.  Same tree, node for node: a node is numbered and
// draws its cut dimension when it is taken off the stack, and the left child is taken before anything of the right one
// (pre-order — the order of the recursion, and of the Python twin in tools/neighbourhood_sweep.py).
int build(const double* pv, std::vector<int>& idx, int first0, int count0, Rng& rng, std::vector<Node>& nodes)
{
    struct Todo { int first, count, parent; bool is_hi; };
    std::vector<Todo> todo;
    todo.push_back(Todo{ first0, count0, -1, false });
    const int root = (int)nodes.size();
    std::vector<int> left, right;
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        const int first = t.first, count = t.count;
        const int me = (int)nodes.size();
        nodes.push_back(Node{ -1, 0.0, -1, -1, -1 });
        if (t.parent >= 0) (t.is_hi ? nodes[t.parent].hi : nodes[t.parent].lo) = me;
        if (count == 1) { nodes[me].point = idx[first]; continue; }
        const int ns = std::min(count, 100);
        double mean[4], var[4];
        for (int d = 0; d < 4; ++d) {
            double s = 0.0;
            for (int k = 0; k < ns; ++k) s = s + pv[4 * (size_t)idx[first + k] + d];
            mean[d] = s / (double)ns;
            double v = 0.0;
            for (int k = 0; k < ns; ++k) { const double x = pv[4 * (size_t)idx[first + k] + d] - mean[d]; v = v + x * x; }
            var[d] = v / (double)ns;
        }
        int order[4] = { 0, 1, 2, 3 };
        std::stable_sort(order, order + 4, [&](int a, int b) { return var[a] > var[b]; });      // largest variance first
        const int dim = order[rng.below(4)];                  // "among the five of largest variance": all four here
        const double val = mean[dim];
        left.clear();
        right.clear();
        for (int k = 0; k < count; ++k) {
            const int p = idx[first + k];
            (pv[4 * (size_t)p + dim] < val ? left : right).push_back(p);
        }
        if (left.empty() || right.empty()) {                  // every point equal in that dimension: halve
            left.assign(idx.begin() + first, idx.begin() + first + count / 2);
            right.assign(idx.begin() + first + count / 2, idx.begin() + first + count);
        }
        std::copy(left.begin(), left.end(), idx.begin() + first);
        std::copy(right.begin(), right.end(), idx.begin() + first + (int)left.size());
        nodes[me].dim = dim;
        nodes[me].val = val;
        const int nl = (int)left.size();
        todo.push_back(Todo{ first + nl, count - nl, me, true });      // taken after the whole left subtree
        todo.push_back(Todo{ first, nl, me, false });
    }
    return root;
}

} // namespace

// pv: n x 4 doubles holding the float32-rounded (x1, y1, x2, y2).  hits[i] = ascending rows j != i among the at most
// `checks` points the search examined that lie within `radius` of row i.
void ApproxNeighbourHits(const double* pv, int n, int trees, int checks, double radius, uint64_t seed,
                         std::vector<std::vector<int>>& hits)
{
    hits.assign((size_t)std::max(n, 0), {});
    if (n <= 1 || trees <= 0 || checks <= 0) return;
    Rng rng{ seed };
    std::vector<std::vector<Node>> forest((size_t)trees);
    std::vector<int> roots((size_t)trees);
    for (int t = 0; t < trees; ++t) {
        std::vector<int> perm((size_t)n);
        for (int i = 0; i < n; ++i) perm[i] = i;
        for (int i = n - 1; i > 0; --i) std::swap(perm[i], perm[rng.below((uint64_t)i + 1)]);     // FLANN shuffles the rows per tree
        forest[t].reserve(2 * (size_t)n);
        roots[t] = build(pv, perm, 0, n, rng, forest[t]);
    }
    const double r2 = radius * radius;
    // r06 (advisor): a far branch whose lower bound already exceeds the radius cannot hold a hit and is not queued (FLANN prunes
    // against its worst distance).  The hit lists do not change: such branches sat behind every branch that can hold one and
    // only used up what was left of the `checks` budget on points that are no hits.
    struct Branch { double mind; long long order; int tree, node; };
    auto later = [](const Branch& a, const Branch& b) { return a.mind > b.mind || (a.mind == b.mind && a.order > b.order); };
    std::vector<int> stamp((size_t)n, -1);
    for (int q = 0; q < n; ++q) {
        const double* v = pv + 4 * (size_t)q;
        std::priority_queue<Branch, std::vector<Branch>, decltype(later)> heap(later);
        long long pushed = 0;
        int count = 0;
        std::vector<int>& found = hits[q];
        auto descend = [&](int t, int node, double mind) {
            const std::vector<Node>& nd = forest[t];
            while (nd[node].dim >= 0) {
                const double diff = v[nd[node].dim] - nd[node].val;
                const int near = diff < 0 ? nd[node].lo : nd[node].hi, far = diff < 0 ? nd[node].hi : nd[node].lo;
                if (mind + diff * diff <= r2) heap.push(Branch{ mind + diff * diff, pushed++, t, far });
                node = near;
            }
            const int p = nd[node].point;
            if (stamp[p] == q || count >= checks) return;
            stamp[p] = q;
            ++count;
            double d2 = 0.0;
            for (int d = 0; d < 4; ++d) { const double x = pv[4 * (size_t)p + d] - v[d]; d2 = d2 + x * x; }
            if (d2 <= r2 && p != q) found.push_back(p);
        };
        for (int t = 0; t < trees; ++t) descend(t, roots[t], 0.0);
        while (!heap.empty() && count < checks) {
            const Branch b = heap.top();
            heap.pop();
            descend(b.tree, b.node, b.mind);
        }
        std::sort(found.begin(), found.end());
    }
}

} // namespace multih

// C entry for tools and tests: src / dst as the class takes them; rowptr: n + 1 ints, col: capacity `cap` ints.  Returns the
// number of hits (or -1 when `cap` is too small).
extern "C" __attribute__((visibility("default")))
int mhh_approx_neighbour_hits(const double* src_xy, const double* dst_xy, int n, int trees, int checks, double radius,
                              unsigned long long seed, int* rowptr, int* col, int cap)
{
    std::vector<double> pv(4 * (size_t)std::max(n, 0));
    for (int i = 0; i < n; ++i) {
        pv[4 * (size_t)i] = (double)(float)src_xy[2 * i];
        pv[4 * (size_t)i + 1] = (double)(float)src_xy[2 * i + 1];
        pv[4 * (size_t)i + 2] = (double)(float)dst_xy[2 * i];
        pv[4 * (size_t)i + 3] = (double)(float)dst_xy[2 * i + 1];
    }
    std::vector<std::vector<int>> hits;
    multih::ApproxNeighbourHits(pv.data(), n, trees, checks, radius, seed, hits);
    int total = 0;
    rowptr[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (total + (int)hits[i].size() > cap) return -1;
        std::copy(hits[i].begin(), hits[i].end(), col + total);
        total += (int)hits[i].size();
        rowptr[i + 1] = total;
    }
    return total;
}
