"""Seeded synthetic two-view scenes for tests and bench.

K planes seen by two cameras that share ONE relative pose, so a single
fundamental matrix F and epipole e2 exist and the HAF re-estimation
(reference M/MultiH.cpp:913-989) is meaningful:

    H_k = K (R + t n_k^T / d_k) K^-1 ,   F = K^-T [t]_x R K^-1 ,   e2 ~ K t

Points are uniform in image 1 (1000 x 1000), assigned to planes by a Voronoi
region mask (spatial coherence for the Potts term), dst = H_k src + N(0, sigma),
a fraction of gross outliers gets a uniform dst.  Affinities are the Jacobian of
H_k at src (+1 % noise) in the reference's order a11 a12 a21 a22
(M/main.cpp:394, M/MultiH.cpp:928-931).  Neighbour hits are exact kNN in the
float32 4-D vectors (x1,y1,x2,y2) the reference feeds to FLANN
(M/MultiH.cpp:233-253), returned as a DIRECTED hit list in CSR form (the
reference's `neighbours[i][j].trainIdx`), optionally symmetric-closed.

Everything is numpy float64 / int32; nothing here touches the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

IMG = 1000.0


@dataclass
class Scene:
    src: np.ndarray        # (N,2) f64
    dst: np.ndarray        # (N,2) f64
    aff: np.ndarray        # (N,4) f64  a11 a12 a21 a22
    gt_label: np.ndarray   # (N,) i32, -1 = outlier
    H_true: np.ndarray     # (K,9) f64 row-major, h33 = 1
    F: np.ndarray          # (9,) f64 row-major
    e2: np.ndarray         # (2,) f64 epipole in image 2 (x, y), third coord 1
    hit_rowptr: np.ndarray  # (N+1,) i32
    hit_col: np.ndarray    # (nnz,) i32

    @property
    def n(self) -> int:
        return int(self.src.shape[0])


def _rot(ax: float, ay: float, az: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def apply_h(H9: np.ndarray, pts: np.ndarray) -> np.ndarray:
    h = np.asarray(H9, dtype=np.float64).reshape(3, 3)
    p = np.concatenate([pts, np.ones((pts.shape[0], 1))], axis=1) @ h.T
    return p[:, :2] / p[:, 2:3]


def jacobian_h(H9: np.ndarray, pts: np.ndarray) -> np.ndarray:
    h = np.asarray(H9, dtype=np.float64).reshape(9)
    x, y = pts[:, 0], pts[:, 1]
    s = h[6] * x + h[7] * y + h[8]
    u = (h[0] * x + h[1] * y + h[2]) / s
    v = (h[3] * x + h[4] * y + h[5]) / s
    return np.stack([(h[0] - h[6] * u) / s, (h[1] - h[7] * u) / s,
                     (h[3] - h[6] * v) / s, (h[4] - h[7] * v) / s], axis=1)


def knn_hits(src: np.ndarray, dst: np.ndarray, k: int, symmetric: bool) -> tuple[np.ndarray, np.ndarray]:
    """Directed kNN hit list (CSR) over float32 (x1,y1,x2,y2); self excluded."""
    from scipy.spatial import cKDTree

    n = src.shape[0]
    k = min(k, n - 1)
    pv = np.concatenate([src, dst], axis=1).astype(np.float32).astype(np.float64)
    _, idx = cKDTree(pv).query(pv, k=k + 1)
    rows = np.repeat(np.arange(n), k + 1)
    cols = idx.reshape(-1)
    keep = rows != cols
    # a point coincident with another may not be its own first hit; keep k per row at most
    rows, cols = rows[keep], cols[keep]
    if symmetric:
        pairs = np.unique(np.concatenate([np.stack([rows, cols], 1), np.stack([cols, rows], 1)]), axis=0)
        rows, cols = pairs[:, 0], pairs[:, 1]
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    return rowptr, cols.astype(np.int32)


_PLANE_CACHE: dict = {}


def _plane_homography(K, Kinv, R, t, nrm, d):
    Hk = K @ (R + np.outer(t, nrm) / d) @ Kinv
    return (Hk / Hk[2, 2]).reshape(9)


def _feature6(H9: np.ndarray) -> np.ndarray:
    """The reference's 6-D merge feature: the images of (0,0), (1,0), (0,1) (M/MultiH.cpp:364-390)."""
    h = H9
    return np.array([h[2] / h[8], h[5] / h[8], (h[0] + h[2]) / (h[6] + h[8]), (h[3] + h[5]) / (h[6] + h[8]),
                     (h[1] + h[2]) / (h[7] + h[8]), (h[4] + h[5]) / (h[7] + h[8])])


def _separated_planes(rng, n_planes: int, seeds: np.ndarray, K, Kinv, R, t, sep: float, feat_sep: float) -> np.ndarray:
    """K plane homographies of ONE relative pose that are distinguishable where they are observed: every point of the
    Voronoi cell of plane k is transferred at least `sep` pixels away by every other plane's homography (and the other
    way round), and the 6-D merge features of any two planes differ by at least `feat_sep` in L1.  `sep` has to exceed TWO
    truncation radii (2 x 4.95 px at the harness defaults): the reference's data cost falls towards the truncation
    threshold (M/MultiH.cpp:501-502: 200 at a perfect fit, 0 at the threshold), so a homography half-way between two planes
    less than that apart explains both at a LOWER cost than their own models — with 8 px the loop merged four planes
    into one model that way (profiles/r05_loop_confusion_sep8.txt).  Without any separation the
    scene is K planes in name only: with normals and depths drawn independently, 40-90 % of a plane's correspondences
    lie within the reference's truncation threshold (4.95 px) of ANOTHER plane's homography, and PEARL's labeling
    cannot (and, with the reference's data cost growing towards the threshold, M/MultiH.cpp:501-502, will not) keep the
    planes apart — tools/plane_trace.py, profiles/r05_plane_trace_*.txt.  Rejection sampling on a 41 x 41 grid of image 1;
    deterministic in rng."""
    g = np.stack(np.meshgrid(np.linspace(0.0, IMG, 41), np.linspace(0.0, IMG, 41)), -1).reshape(-1, 2)
    reg = np.argmin(((g[:, None, :] - seeds[None, :, :]) ** 2).sum(-1), axis=1)
    gh = np.concatenate([g, np.ones((g.shape[0], 1))], axis=1)
    B = 128                                               # candidates per draw

    def transfer(Hs):                                     # [b, 9] -> [b, G, 2]
        p = np.einsum("bij,gj->bgi", Hs.reshape(-1, 3, 3), gh)
        return p[..., :2] / p[..., 2:3]

    def features(Hs):                                     # [b, 9] -> [b, 6]
        return np.stack([_feature6(h) for h in Hs])

    def draw(b):
        nrm = np.stack([rng.uniform(-0.6, 0.6, b), rng.uniform(-0.6, 0.6, b), np.ones(b)], axis=1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        d = 1.0 / rng.uniform(1.0 / 12.0, 1.0 / 2.5, b)  # uniform in inverse depth = uniform in parallax
        return np.stack([_plane_homography(K, Kinv, R, t, nrm[i], d[i]) for i in range(b)])

    def conflicts(k, ct, cf, H_t, H_f):
        """[b, n_planes] — does candidate i for slot k collide with plane l?"""
        out = np.zeros((ct.shape[0], n_planes), dtype=bool)
        for l in range(n_planes):
            if l == k:
                continue
            bad = np.abs(cf - H_f[l]).sum(1) < feat_sep
            m = (reg == k) | (reg == l)
            if m.any():
                bad |= np.sqrt(((ct[:, m, :] - H_t[l][m]) ** 2).sum(-1)).min(axis=1) < sep
            out[:, l] = bad
        return out

    # min-conflicts search: start from any K planes, then redraw one colliding plane at a time, keeping the candidate
    # that collides with the fewest others (a greedy one-after-the-other placement gets stuck on the last planes)
    H = draw(n_planes)
    H_t, H_f = transfer(H), features(H)
    C = np.zeros((n_planes, n_planes), dtype=bool)
    for k in range(n_planes):
        C[k] = conflicts(k, H_t[k:k + 1], H_f[k:k + 1], H_t, H_f)[0]
    C |= C.T
    for _ in range(4000):
        bad = np.flatnonzero(C.any(axis=1))
        if bad.size == 0:
            return H
        k = int(bad[rng.integers(bad.size)])
        cand = draw(B)
        ct, cf = transfer(cand), features(cand)
        cc = conflicts(k, ct, cf, H_t, H_f)
        i = int(np.argmin(cc.sum(axis=1)))
        if cc[i].sum() <= C[k].sum():
            H[k], H_t[k], H_f[k] = cand[i], ct[i], cf[i]
            C[k, :] = cc[i]
            C[:, k] = cc[i]
    raise RuntimeError(f"make_scene: no set of {n_planes} planes {sep} px apart found")


def make_scene(n_points: int, n_planes: int, seed: int = 1234, outlier_frac: float = 0.25,
               noise: float = 0.5, knn: int = 16, symmetric: bool = True,
               with_neighbours: bool = True, legacy_r04: bool = False,
               plane_separation: float = 13.0, feature_separation: float = 15.0) -> Scene:
    """legacy_r04: the generator as it stood until round 4 (normals and depths of the planes drawn independently of
    each other — planes that cannot be told apart inside the truncation threshold, see _separated_planes)."""
    rng = np.random.default_rng(seed)
    K = np.array([[1000.0, 0, 500.0], [0, 1000.0, 500.0], [0, 0, 1.0]])
    Kinv = np.linalg.inv(K)
    R = _rot(0.03, -0.08, 0.02)
    # The baseline: twice the r04 one in the current generator.  Same epipole and F (both are scale-free in t), twice the
    # parallax — room for ten planes that stay more than two truncation radii (2 x 4.95 px) apart where they are observed.
    t = np.array([0.40, 0.05, 0.08]) * (1.0 if legacy_r04 else 2.0)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = Kinv.T @ tx @ R @ Kinv
    F = F / np.linalg.norm(F)
    e2h = K @ t
    e2 = e2h[:2] / e2h[2]

    if legacy_r04:
        H_true = np.zeros((n_planes, 9))
        for k in range(n_planes):
            nrm = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), 1.0])
            nrm /= np.linalg.norm(nrm)
            d = rng.uniform(4.0, 9.0)
            H_true[k] = _plane_homography(K, Kinv, R, t, nrm, d)
        src = rng.uniform(0.0, IMG, size=(n_points, 2))
        seeds = rng.uniform(0.0, IMG, size=(n_planes, 2))
    else:
        seeds = rng.uniform(0.0, IMG, size=(n_planes, 2))
        # the planes come from a generator of their own: the number of rejected candidates does not move the other draws,
        # and the set is remembered per (seed, planes, separations) — the search takes seconds for ten planes
        key = (seed, n_planes, float(plane_separation), float(feature_separation))
        if key not in _PLANE_CACHE:
            _PLANE_CACHE[key] = _separated_planes(np.random.default_rng([seed, 0x9E3779B9]), n_planes, seeds, K, Kinv, R, t,
                                                  plane_separation, feature_separation)
        H_true = _PLANE_CACHE[key].copy()
        src = rng.uniform(0.0, IMG, size=(n_points, 2))

    region = np.argmin(((src[:, None, :] - seeds[None, :, :]) ** 2).sum(-1), axis=1).astype(np.int32)
    dst = np.empty_like(src)
    aff = np.empty((n_points, 4))
    for k in range(n_planes):
        m = region == k
        dst[m] = apply_h(H_true[k], src[m])
        aff[m] = jacobian_h(H_true[k], src[m])
    dst += rng.normal(0.0, noise, size=dst.shape)
    aff *= 1.0 + rng.normal(0.0, 0.01, size=aff.shape)

    gt = region.copy()
    out = rng.random(n_points) < outlier_frac
    n_out = int(out.sum())
    dst[out] = rng.uniform(0.0, IMG, size=(n_out, 2))
    aff[out] = np.array([1.0, 0.0, 0.0, 1.0]) + rng.normal(0.0, 0.2, size=(n_out, 4))
    gt[out] = -1

    if with_neighbours:
        rowptr, col = knn_hits(src, dst, knn, symmetric)
    else:
        rowptr, col = np.zeros(n_points + 1, dtype=np.int32), np.zeros(0, dtype=np.int32)
    return Scene(src=src, dst=dst, aff=aff, gt_label=gt, H_true=H_true,
                 F=F.reshape(9).copy(), e2=e2.copy(), hit_rowptr=rowptr, hit_col=col)


def adjusted_rand_index(a: np.ndarray, b: np.ndarray) -> float:
    """Adjusted Rand index of two labelings (Hubert & Arabie), plain numpy."""
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    table = np.zeros((ai.max() + 1, bi.max() + 1), dtype=np.int64)
    np.add.at(table, (ai, bi), 1)
    c2 = lambda x: x.astype(np.float64) * (x.astype(np.float64) - 1.0) / 2.0
    s_ij, s_a, s_b, total = c2(table).sum(), c2(table.sum(1)).sum(), c2(table.sum(0)).sum(), c2(np.array([a.size]))[0]
    expected = s_a * s_b / total if total > 0 else 0.0
    denom = 0.5 * (s_a + s_b) - expected
    return float((s_ij - expected) / denom) if denom != 0 else 1.0


def agreement(gt_label: np.ndarray, labels: np.ndarray, share: float = 0.8) -> dict:
    """Result quality against the generator's ground truth: planes recovered (one label holds >= `share` of the plane's
    inlier correspondences, and no label is counted for two planes), ARI over all correspondences with the outliers as a
    class of their own, outliers labelled vs generated."""
    gt_label, labels = np.asarray(gt_label), np.asarray(labels)
    planes = int(gt_label.max()) + 1 if gt_label.size else 0
    taken, recovered = set(), 0
    for p in range(planes):
        ip = np.flatnonzero(gt_label == p)
        lp = labels[ip]
        lp = lp[lp >= 0]
        if lp.size == 0:
            continue
        cnt = np.bincount(lp)
        best = int(np.argmax(cnt))
        if cnt[best] >= share * ip.size and best not in taken:
            taken.add(best)
            recovered += 1
    return {"planes_recovered": recovered, "planes": planes, "ari": adjusted_rand_index(gt_label, labels),
            "outliers_labelled": int((labels < 0).sum()), "outliers_generated": int((gt_label < 0).sum())}
