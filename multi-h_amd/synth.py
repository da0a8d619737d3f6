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


def make_scene(n_points: int, n_planes: int, seed: int = 1234, outlier_frac: float = 0.25,
               noise: float = 0.5, knn: int = 16, symmetric: bool = True,
               with_neighbours: bool = True) -> Scene:
    rng = np.random.default_rng(seed)
    K = np.array([[1000.0, 0, 500.0], [0, 1000.0, 500.0], [0, 0, 1.0]])
    Kinv = np.linalg.inv(K)
    R = _rot(0.03, -0.08, 0.02)
    t = np.array([0.40, 0.05, 0.08])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = Kinv.T @ tx @ R @ Kinv
    F = F / np.linalg.norm(F)
    e2h = K @ t
    e2 = e2h[:2] / e2h[2]

    H_true = np.zeros((n_planes, 9))
    for k in range(n_planes):
        nrm = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), 1.0])
        nrm /= np.linalg.norm(nrm)
        d = rng.uniform(4.0, 9.0)
        Hk = K @ (R + np.outer(t, nrm) / d) @ Kinv
        H_true[k] = (Hk / Hk[2, 2]).reshape(9)

    src = rng.uniform(0.0, IMG, size=(n_points, 2))
    seeds = rng.uniform(0.0, IMG, size=(n_planes, 2))
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
