#!/usr/bin/env python3
"""Which neighbourhood reproduces the reference's recorded barrsmith result (5 planes)?  VERDICT r04 item 3.

The reference builds its neighbourhood with cv::FlannBasedMatcher::radiusMatch in the float32 (x1, y1, x2, y2) space,
radius 1 / locality = 200 px (M/MultiH.cpp:233-253).  FLANN's default index is 4 randomised KD-trees searched
best-bin-first with 32 checks: a query EXAMINES at most 32 points and reports those of them inside the radius — a few dozen
approximate nearest neighbours, found one-way.  Every directed hit becomes a setNeighbors call, so a pair found from both
sides carries two Potts terms (SURVEY A-2).  OpenCV/FLANN are not in /root/reference or the image; this tool re-enacts the
published algorithm (randomised KD-trees: split dimension drawn from the five of largest variance on a 100-point sample,
split at the sample mean, one point per leaf; best-bin-first over all trees with one heap, `checks` leaf visits) in
numpy with the engine's counter RNG, and runs the reference's recorded 1 094 correspondences through Process() with
 * the exact k nearest hits inside the radius (the class default is k = 16), k = 8 ... 32,
 * the same with every unordered pair listed ONCE (no double Potts term for mutual pairs),
 * the FLANN re-enactment with 16 / 32 / 64 checks, several tree seeds,
on both initialisation routes and three seeds; planes, ARI against the reference's labels (its non-outliers), median.
Output is kept under profiles/ and summarised in BASELINE.md 1a."""
import ctypes as C, heapq, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
mh = importlib.import_module("multi-h_amd")
import barrsmith_agreement as BA
from scipy.spatial import cKDTree


def splitmix(z):
    z = (z + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


class Rng:
    def __init__(self, seed): self.s = seed & 0xFFFFFFFFFFFFFFFF
    def below(self, n):
        self.s = (self.s + 1) & 0xFFFFFFFFFFFFFFFF
        return splitmix(self.s) % n


def build_tree(pv, idx, rng):
    """FLANN KDTreeIndex::divideTree (the algorithm of multi-h_amd/host/approx_neighbours.cpp, operation for operation):
    leaves hold ONE point; the cut dimension is drawn among the 5 dimensions of largest variance (here all 4) estimated on
    the first 100 points of the node — sequential sums —, the cut value is their mean in it."""
    if len(idx) == 1:
        return ("leaf", int(idx[0]))
    sample = [pv[int(i)] for i in idx[:100]]
    ns = len(sample)
    mean, var = [], []
    for d in range(4):
        s = 0.0
        for r in sample: s = s + float(r[d])
        m = s / ns
        v = 0.0
        for r in sample:
            x = float(r[d]) - m
            v = v + x * x
        mean.append(m); var.append(v / ns)
    order = sorted(range(4), key=lambda d: -var[d])                 # stable: largest variance first
    dim = order[rng.below(4)]
    val = mean[dim]
    left = [int(i) for i in idx if pv[int(i), dim] < val]
    right = [int(i) for i in idx if not (pv[int(i), dim] < val)]
    if len(left) == 0 or len(right) == 0:                  # all equal in that dimension: halve
        half = len(idx) // 2
        left, right = [int(i) for i in idx[:half]], [int(i) for i in idx[half:]]
    return ("node", dim, val, build_tree(pv, left, rng), build_tree(pv, right, rng))


def forest_hits(pv, trees_n, checks, radius, seed):
    """Directed hits of every point against the forest: best-bin-first with `checks` examined points, inside `radius`."""
    n = pv.shape[0]
    rng = Rng(seed)
    trees = []
    for t in range(trees_n):
        perm = list(range(n))
        for i in range(n - 1, 0, -1):                      # FLANN shuffles the point order per tree
            j = rng.below(i + 1); perm[i], perm[j] = perm[j], perm[i]
        trees.append(build_tree(pv, perm, rng))
    r2 = radius * radius
    rows = []
    for q in range(n):
        v = pv[q]
        heap, checked, found = [], set(), []
        state = {"count": 0, "pushed": 0}

        def descend(node, mind):
            while node[0] == "node":
                _, dim, val, lo, hi = node
                diff = float(v[dim]) - val
                near, far = (lo, hi) if diff < 0 else (hi, lo)
                heapq.heappush(heap, (mind + diff * diff, state["pushed"], far))     # ties: first pushed, first out
                state["pushed"] += 1
                node = near
            p = node[1]
            if p in checked or state["count"] >= checks:
                return
            checked.add(p); state["count"] += 1
            d2 = 0.0
            for d in range(4):
                x = float(pv[p, d]) - float(v[d])
                d2 = d2 + x * x
            if d2 <= r2 and p != q:
                found.append(p)

        for t in trees:
            descend(t, 0.0)
        while heap and state["count"] < checks:
            mind, _, node = heapq.heappop(heap)
            descend(node, mind)
        rows.append(sorted(found))
    return rows


def knn_hits(pv, k, radius, once):
    d, idx = cKDTree(pv).query(pv, k=k + 1)
    rows = []
    for i in range(pv.shape[0]):
        r = [int(j) for j, dd in zip(idx[i], d[i]) if j != i and dd <= radius][:k]
        rows.append(r)
    if once:                                               # every unordered pair once: keep i -> j only for i < j, add missing
        pairs = {(min(i, j), max(i, j)) for i, r in enumerate(rows) for j in r}
        rows = [[] for _ in rows]
        for a, b in sorted(pairs): rows[a].append(b)
    return rows


def csr(rows):
    rowptr = np.zeros(len(rows) + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.array([j for r in rows for j in r], dtype=np.int32)
    return rowptr, col


def main():
    corr, ref, matched, total = BA.kept_correspondences()
    e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
    e.set_correspondences(corr[:, 0:2], corr[:, 2:4], corr[:, 4:8])
    F, e2, mask, inl = e.estimate_fundamental(1234 ^ 0xf00d, 4000, 2.6)
    e.close()
    host = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
    pv = corr[:, 0:4].astype(np.float32).astype(np.float64)
    radius = 1.0 / 0.005
    modes = []
    for k in (8, 12, 16, 24, 32):
        modes.append((f"exact {k}-NN in radius", knn_hits(pv, k, radius, False)))
    for k in (16, 32):
        modes.append((f"exact {k}-NN, each pair once", knn_hits(pv, k, radius, True)))
    for checks in (16, 32, 64):
        for tseed in (1, 2):
            modes.append((f"FLANN re-enactment 4 trees / {checks} checks (trees {tseed})", forest_hits(pv, 4, checks, radius, 1000 + tseed)))
    out = []
    for name, rows in modes:
        rowptr, col = csr(rows)
        deg = np.diff(rowptr)
        hit = {(i, j) for i, r in enumerate(rows) for j in r}
        mutual = sum(1 for (i, j) in hit if (j, i) in hit) / max(len(hit), 1)
        host.mhh_set_neighbour_hits(rowptr.ctypes.data_as(C.POINTER(C.c_int)), col.ctypes.data_as(C.POINTER(C.c_int)), len(rows))
        rec = {"mode": name, "hits_per_point_mean": float(deg.mean()), "hits_per_point_max": int(deg.max()), "mutual_share": mutual, "routes": {}}
        for route in ("stable_sets", "dlt"):
            runs = []
            for seed in (1234, 7, 99):
                k, labels, it, en = BA.run(route, corr, F, e2, seed=seed)
                a = BA.agreement(labels, ref) if k > 0 else {"planes": int(k), "ari_reference_inliers": float("nan"), "ari_all": float("nan"), "ours_histogram": []}
                runs.append({"seed": seed, "planes": int(k), "ari_reference_inliers": a["ari_reference_inliers"], "ari_all": a["ari_all"],
                             "histogram": a["ours_histogram"]})
            rec["routes"][route] = {"runs": runs, "planes": [r["planes"] for r in runs],
                                    "median_ari_reference_inliers": float(np.median([r["ari_reference_inliers"] for r in runs]))}
        out.append(rec)
        print(f"{name:58s} hits/pt {deg.mean():5.1f} mutual {mutual:.2f} | stable: planes {rec['routes']['stable_sets']['planes']} "
              f"ARI med {rec['routes']['stable_sets']['median_ari_reference_inliers']:.3f} | dlt: planes {rec['routes']['dlt']['planes']} "
              f"ARI med {rec['routes']['dlt']['median_ari_reference_inliers']:.3f}", flush=True)
    host.mhh_set_neighbour_hits(None, None, 0)
    print(json.dumps({"reference_histogram": np.bincount(ref + 1).tolist(), "modes": out}))


if __name__ == "__main__":
    main()
