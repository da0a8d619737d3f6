"""numpy stand-in (TESTS ONLY) for the OpenCV front half the reference runs before its hot loop:
normalised 8-point fundamental matrix + RANSAC (cv::findFundamentalMat, M/MultiH.cpp:775) and the
epipole in image 2 (M/MultiH.cpp:786-793).  The product takes F and e2 as inputs."""
import numpy as np


def _normalise(p):
    c = p.mean(0)
    q = p - c
    s = np.sqrt(2) / np.mean(np.linalg.norm(q, axis=1))
    T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
    return q * s, T


def eight_point(src, dst):
    a, T1 = _normalise(src)
    b, T2 = _normalise(dst)
    A = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                  a[:, 0], a[:, 1], np.ones(len(a))], axis=1)
    F = np.linalg.svd(A)[2][-1].reshape(3, 3)
    u, s, vt = np.linalg.svd(F)
    F = u @ np.diag([s[0], s[1], 0]) @ vt
    F = T2.T @ F @ T1
    return F / np.linalg.norm(F)


def sampson(F, src, dst):
    x1 = np.concatenate([src, np.ones((len(src), 1))], 1)
    x2 = np.concatenate([dst, np.ones((len(dst), 1))], 1)
    Fx1 = x1 @ F.T
    Ftx2 = x2 @ F
    num = np.sum(x2 * Fx1, axis=1) ** 2
    return num / (Fx1[:, 0] ** 2 + Fx1[:, 1] ** 2 + Ftx2[:, 0] ** 2 + Ftx2[:, 1] ** 2)


def fundamental_ransac(src, dst, thr=2.0, iters=2000, seed=0):
    rng = np.random.default_rng(seed)
    best, best_in = None, None
    n = len(src)
    for _ in range(iters):
        idx = rng.choice(n, 8, replace=False)
        F = eight_point(src[idx], dst[idx])
        inl = sampson(F, src, dst) < thr * thr
        if best_in is None or inl.sum() > best_in.sum():
            best, best_in = F, inl
    F = eight_point(src[best_in], dst[best_in])
    inl = sampson(F, src, dst) < thr * thr
    F = eight_point(src[inl], dst[inl])
    return F, sampson(F, src, dst) < thr * thr


def epipole2(F):
    """eigenvector of F F^T with the smallest eigenvalue, third coordinate 1 (M/MultiH.cpp:789-793)."""
    w, v = np.linalg.eigh(F @ F.T)
    e = v[:, 0]
    return e[:2] / e[2]
