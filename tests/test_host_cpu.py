"""CPU tests of the host-side MergingStep pieces (multi-h_amd/host/merge_step.cpp) against
numpy restatements of the reference code they mirror."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def host(mh, engine_lib):
    return C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))


def _splitmix64(z):
    m = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def np_mean_shift(data, bw, seed):
    """numpy restatement of MeanShiftClustering<double>::Cluster (MeanShiftClustering.h:23-157)
    with the explicit seed RNG of merge_step.h."""
    n, d = data.shape
    band_sq, stop = bw * bw, 1e-3 * bw
    visited = np.zeros(n, bool)
    init = list(range(n))
    cent, votes = [], []
    c = 0
    while init:
        rnd = (_splitmix64(seed + c) >> 11) * (1.0 / 9007199254740992.0)
        c += 1
        st = init[int(np.floor(rnd * (len(init) - 1) + 0.5))]
        mean = data[st].copy()
        my = np.zeros(n, int)
        while True:
            old = mean.copy()
            dist = np.sqrt((old[None, :] - data) ** 2).sum(axis=1)       # L1 via sqrt of squares
            inl = dist < band_sq
            my += inl
            visited |= inl
            if inl.sum() == 0:
                visited[st] = True
                break
            acc = np.zeros(d)
            for i in np.flatnonzero(inl):
                acc = acc + data[i]
            mean = acc * (1.0 / inl.sum())
            if np.sqrt(((mean - old) ** 2).sum()) < stop:
                mw = -1
                for k, cc in enumerate(cent):
                    if np.sqrt(((mean - cc) ** 2).sum()) < bw / 2:
                        mw = k
                        break
                if mw >= 0:
                    cent[mw] = 0.5 * (cent[mw] + mean)
                    votes[mw] = votes[mw] + my
                else:
                    cent.append(mean)
                    votes.append(my.copy())
                break
        init = [i for i in range(n) if not visited[i]]
    best_v, best_i = np.zeros(n, int), np.full(n, -1)
    for r, v in enumerate(votes):
        upd = best_v < v
        best_v[upd] = v[upd]
        best_i[upd] = r
    return np.array(cent), best_i, c


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_mean_shift_matches_numpy_restatement(host, seed):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-50, 50, size=(4, 6))
    data = np.concatenate([c + rng.normal(0, 0.15, size=(12, 6)) for c in centres] + [rng.uniform(-50, 50, size=(5, 6))])
    data = np.ascontiguousarray(data)
    n = data.shape[0]
    modes = np.zeros((n, 6))
    assign = np.zeros(n, dtype=np.int32)
    draws = C.c_ulonglong(0)
    k = host.mhh_mean_shift(data.ctypes.data_as(_dp), n, 6, C.c_double(2.2), C.c_ulonglong(seed),
                            modes.ctypes.data_as(_dp), n, assign.ctypes.data_as(C.POINTER(C.c_int)), C.byref(draws))
    cent, best_i, c = np_mean_shift(data, 2.2, seed)
    assert k == len(cent) and draws.value == c
    assert np.allclose(modes[:k], cent, rtol=0, atol=1e-12)
    assert np.array_equal(assign, best_i)
    # the four tight clusters are each one mode
    for j in range(4):
        assert len(set(assign[12 * j: 12 * (j + 1)])) == 1


@pytest.mark.parametrize("seed,n_clusters,per,strays", [(4, 9, 30, 40), (5, 3, 120, 10), (6, 40, 8, 30)])
def test_mean_shift_through_its_index_matches_the_full_scan(host, seed, n_clusters, per, strays):
    """r06: from 64 rows on the host's mean shift takes the rows of a window on one coordinate (sorted once), puts them back
    into row order and applies the reference's own test to them, picks its seeds through a Fenwick tree over the unvisited
    rows and keeps the votes sparse.  Same modes, same assignment (ties included: duplicated rows), same number of draws as
    the plain restatement above, which scans every row in every iteration."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-80, 80, size=(n_clusters, 6))
    data = np.concatenate([c + rng.normal(0, 0.2, size=(per, 6)) for c in centres] + [rng.uniform(-80, 80, size=(strays, 6))])
    data[5] = data[4]; data[per + 1] = data[per]                     # duplicated rows: equal votes, the first mode wins
    data = np.ascontiguousarray(data[rng.permutation(len(data))])
    n = data.shape[0]
    assert n >= 64
    modes = np.zeros((n, 6))
    assign = np.zeros(n, dtype=np.int32)
    draws = C.c_ulonglong(0)
    k = host.mhh_mean_shift(data.ctypes.data_as(_dp), n, 6, C.c_double(2.2), C.c_ulonglong(seed),
                            modes.ctypes.data_as(_dp), n, assign.ctypes.data_as(C.POINTER(C.c_int)), C.byref(draws))
    cent, best_i, c = np_mean_shift(data, 2.2, seed)
    assert k == len(cent) and draws.value == c
    assert np.allclose(modes[:k], cent, rtol=0, atol=1e-12)
    assert np.array_equal(assign, best_i)
    assert k < n - strays, "the clusters should merge into modes"


def test_homography_features(host):
    rng = np.random.default_rng(0)
    H = rng.normal(size=(5, 9))
    feat = np.zeros((5, 6))
    host.mhh_homography_features(H.ctypes.data_as(_dp), 5, feat.ctypes.data_as(_dp))
    for i in range(5):
        h = H[i].reshape(3, 3)
        exp = []
        for p in ([0, 0, 1], [1, 0, 1], [0, 1, 1]):
            q = h @ np.array(p, float)
            exp += [q[0] / q[2], q[1] / q[2]]
        assert np.allclose(feat[i], exp, rtol=1e-13)


def test_homography_3pt_linear(host, synth):
    """GetHomography3PT's linear part (M/MultiH.cpp:995-1050): three exact correspondences of a
    plane compatible with F reproduce that plane's homography."""
    sc = synth.make_scene(50, 3, seed=4, noise=0.0, outlier_frac=0.0, with_neighbours=False)
    pts1 = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])            # the canonical points of MergingStep
    for k in range(3):
        pts2 = np.ascontiguousarray(synth.apply_h(sc.H_true[k], pts1))
        H = np.zeros(9)
        F = np.ascontiguousarray(sc.F)
        ok = host.mhh_homography_3pt(pts1.ctypes.data_as(_dp), pts2.ctypes.data_as(_dp), 3, F.ctypes.data_as(_dp),
                                     H.ctypes.data_as(_dp))
        assert ok == 1
        a, b = H / H[8], sc.H_true[k] / sc.H_true[k][8]
        assert np.max(np.abs(a - b) / np.maximum(1e-6, np.abs(b))) < 1e-6
        # and it maps the scene's own points of that plane
        m = sc.gt_label == k
        assert np.max(np.abs(synth.apply_h(H, sc.src[m]) - sc.dst[m])) < 1e-6


def _h3pt(host, fn, pts1, pts2, F):
    H = np.zeros(9)
    it = C.c_int(0)
    pts1, pts2, F = (np.ascontiguousarray(a, dtype=np.float64) for a in (pts1, pts2, F))
    if fn == "lin":
        ok = host.mhh_homography_3pt(pts1.ctypes.data_as(_dp), pts2.ctypes.data_as(_dp), len(pts1),
                                     F.ctypes.data_as(_dp), H.ctypes.data_as(_dp))
    else:
        ok = host.mhh_homography_3pt_refined(pts1.ctypes.data_as(_dp), pts2.ctypes.data_as(_dp), len(pts1),
                                             F.ctypes.data_as(_dp), H.ctypes.data_as(_dp), C.byref(it))
    assert ok == 1
    return H, it.value


def test_homography_3pt_lm_refinement(host, synth):
    """RefineHomography3PT (LM on the third row of H, Homography_Refine3PTCallback.h + M/Utilities.hpp
    LMSolverImpl): on exact data it must leave the linear solution in place; on noisy data it must
    not increase the reprojection error it minimises."""
    sc = synth.make_scene(400, 2, seed=8, noise=0.0, outlier_frac=0.0, with_neighbours=False)
    canon = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    H_lin, _ = _h3pt(host, "lin", canon, synth.apply_h(sc.H_true[0], canon), sc.F)
    H_ref, it = _h3pt(host, "ref", canon, synth.apply_h(sc.H_true[0], canon), sc.F)
    assert it >= 1
    assert np.max(np.abs(H_ref / H_ref[8] - H_lin / H_lin[8])) < 1e-6
    rng = np.random.default_rng(0)
    m = sc.gt_label == 1
    src = sc.src[m][:40]
    dst = sc.dst[m][:40] + rng.normal(0, 0.7, size=(40, 2))
    H_lin, _ = _h3pt(host, "lin", src, dst, sc.F)
    H_ref, it = _h3pt(host, "ref", src, dst, sc.F)
    sse = lambda H: float(((synth.apply_h(H, src) - dst) ** 2).sum())
    assert sse(H_ref) <= sse(H_lin) * (1 + 1e-9)
    assert sse(H_ref) < 40 * 2 * 0.7 ** 2 * 3        # about the noise level


def test_compatibility_check(host, synth):
    """HomographyCompatibilityCheck (M/MultiH.cpp:100-222): a cluster that is one plane survives, a
    cluster of scrambled matches and a cluster below min_inliers are removed, labels are compacted."""
    sc = synth.make_scene(3000, 3, seed=3, noise=0.3, outlier_frac=0.0, with_neighbours=False)
    labels = sc.gt_label.copy().astype(np.int32)          # clusters 0,1,2 = planes
    # cluster 3: 300 points whose destination is scrambled (no single homography explains them)
    rng = np.random.default_rng(1)
    pick = np.flatnonzero((labels == 0) | (labels == 1))
    mixed = rng.choice(pick, size=300, replace=False)
    labels[mixed] = 3
    sc.dst[mixed] = rng.uniform(0, 1000, size=(300, 2))
    # cluster 4: too small (5 < min_inliers)
    labels[np.flatnonzero(labels == 2)[:5]] = 4
    H = np.concatenate([sc.H_true, sc.H_true[:1], sc.H_true[1:2]], axis=0).copy()
    src, dst, F = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.F))
    lab = labels.copy()
    kept = host.mhh_compatibility_check(src.ctypes.data_as(_dp), dst.ctypes.data_as(_dp), sc.n,
                                        lab.ctypes.data_as(C.POINTER(C.c_int)), H.ctypes.data_as(_dp), 5,
                                        F.ctypes.data_as(_dp), C.c_double(2.2 ** 2), 20, C.c_ulonglong(7))
    assert kept == 3
    assert np.array_equal(H[:3], sc.H_true)
    assert (lab[labels == 3] == -1).all() and (lab[labels == 4] == -1).all()
    for k in range(3):
        assert (lab[labels == k] == k).all()


def _literal_cluster_median(host, s, d, F, seed, counter):
    """HomographyCompatibilityCheck's trial loop as the reference writes it (M/MultiH.cpp:128-194):
    vectors with erase/append, an N-entry distance buffer whose last three entries go stale, a full
    sort per trial.  The host library does the same with selection and threads; this is the checker."""
    s = [tuple(p) for p in s]
    d = [tuple(p) for p in d]
    n = len(s)
    rest = n - 3
    dist = np.zeros(n)
    medians = []
    for _ in range(501):
        ms, md = [], []
        for _j in range(3):
            u = (_splitmix64((seed + counter) & ((1 << 64) - 1)) >> 11) * (1.0 / 9007199254740992.0)
            counter += 1
            idx = int((len(s) - 1) * u)
            ms.append(s.pop(idx)); md.append(d.pop(idx))
        H = np.zeros(9)
        p1 = np.ascontiguousarray(np.array(ms, dtype=np.float64)); p2 = np.ascontiguousarray(np.array(md, dtype=np.float64))
        ok = host.mhh_homography_3pt(p1.ctypes.data_as(_dp), p2.ctypes.data_as(_dp), 3, F.ctypes.data_as(_dp),
                                     H.ctypes.data_as(_dp))
        S = np.array(s); D = np.array(d)
        with np.errstate(all="ignore"):
            if ok:
                ss = H[6] * S[:, 0] + H[7] * S[:, 1] + H[8]
                x1 = (H[0] * S[:, 0] + H[1] * S[:, 1] + H[2]) / ss
                y1 = (H[3] * S[:, 0] + H[4] * S[:, 1] + H[5]) / ss
                dx = D[:, 0] - x1; dy = D[:, 1] - y1
                d2 = dx * dx + dy * dy
            else:
                d2 = np.full(rest, np.nan)
        dist[:rest] = np.where(np.isnan(d2), 1e300, d2)
        dist.sort()
        medians.append(dist[rest // 2] if rest % 2 else 0.5 * (dist[rest // 2] + dist[rest // 2 + 1]))
        s += [None] * 3; d += [None] * 3
        for j in range(3):
            s[n - j - 1] = ms[j]; d[n - j - 1] = md[j]
    medians.sort()
    return medians[250], counter


@pytest.mark.parametrize("sizes", [(700, 523, 64), (10, 21, 19)])
def test_compatibility_medians_equal_the_literal_trial_loop(host, synth, sizes):
    """The selection-based, threaded trial loop of host/merge_step.cpp gives, bit for bit, the
    median-of-medians of the reference's literal loop (full sorts, stale buffer entries), for odd
    and even cluster sizes, a cluster with scrambled matches and the tiny-cluster path."""
    sc = synth.make_scene(2500, 3, seed=11, noise=0.4, outlier_frac=0.0, with_neighbours=False)
    rng = np.random.default_rng(5)
    labels = np.full(sc.n, -1, dtype=np.int32)
    for c, sz in enumerate(sizes):
        members = np.flatnonzero(sc.gt_label == c)[:sz]
        labels[members] = c
    scr = np.flatnonzero(labels == 2)
    sc.dst[scr[::2]] = rng.uniform(0, 1000, size=(scr[::2].size, 2))           # cluster 2: half scrambled
    src, dst, F = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.F))
    H = sc.H_true.copy()
    lab = labels.copy()
    med = np.zeros(3)
    seed = 99
    host.mhh_compatibility_medians(src.ctypes.data_as(_dp), dst.ctypes.data_as(_dp), sc.n,
                                   lab.ctypes.data_as(C.POINTER(C.c_int)), H.ctypes.data_as(_dp), 3,
                                   F.ctypes.data_as(_dp), C.c_double(2.2 ** 2), 4, C.c_ulonglong(seed),
                                   med.ctypes.data_as(_dp))
    counter = 0
    for c in range(3):
        m = labels == c
        want, counter = _literal_cluster_median(host, src[m], dst[m], F, seed, counter)
        assert med[c] == want, f"cluster {c} (n={m.sum()}): {med[c]!r} != {want!r}"


def test_host_3pt_solver_agrees_with_the_oracles_independent_restatement(host, synth, oracle):
    """Two restatements of GetHomography3PT + RefineHomography3PT written separately from the reference text (the
    product's host/merge_step.cpp and oracle/mh_oracle.cpp section 11) agree to 1e-9 on exact and on noisy inputs,
    with and without the LM refinement.  (Bitwise agreement is not expected: the OpenCV primitives underneath are
    not under /root/reference and each side defines them for itself.)"""
    sc = synth.make_scene(300, 3, seed=21, noise=0.0, outlier_frac=0.0, with_neighbours=False)
    rng = np.random.default_rng(1)
    canon = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    for k in range(3):
        for noise in (0.0, 0.3):
            for pts1 in (canon, sc.src[sc.gt_label == k][:12]):
                pts2 = synth.apply_h(sc.H_true[k], pts1) + rng.normal(0, noise, size=pts1.shape)
                for fn, refine in (("lin", False), ("lm", True)):
                    H_host, _ = _h3pt(host, fn, pts1, pts2, sc.F)
                    H_or, ok = oracle.homography_3pt(pts1, pts2, sc.F, refine=refine)
                    assert ok
                    a, b = H_host / H_host[8], H_or / H_or[8]
                    assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(b))), (k, noise, fn)


@pytest.mark.parametrize("sizes,min_inliers", [((700, 523, 64, 17), 20), ((10, 21, 19), 4), ((40, 33, 5), 0)])
def test_compatibility_check_equals_the_oracles_restatement(host, synth, oracle, sizes, min_inliers):
    """The product's post-filter (host/merge_step.cpp: selection instead of sorting, threads, stale entries threaded
    through afterwards) against the ORACLE's literal restatement of M/MultiH.cpp:100-222 (oracle/mh_oracle.cpp
    section 12: erase/append on the cluster vectors, N-entry buffer, one full sort per trial), which runs on the
    oracle's OWN 3-point solver.  Every decision must be equal — which clusters go, the compacted labels, the kept
    homographies — and the median-of-medians agree to 1e-6 relative (the two 3-point solvers differ in the last bits).
    Covers a cluster below min_inliers (:199-200), one with scrambled matches (median test, :195), the tiny-cluster
    path and min_inliers = 0 with a cluster of 5 (tested, as N >= 4)."""
    planes = len(sizes)
    sc = synth.make_scene(6000, planes, seed=11 + planes, noise=0.4, outlier_frac=0.0, with_neighbours=False)
    rng = np.random.default_rng(5)
    labels = np.full(sc.n, -1, dtype=np.int32)
    for c, sz in enumerate(sizes):
        members = np.flatnonzero(sc.gt_label == c)[:sz]
        assert members.size == sz
        labels[members] = c
    scr = np.flatnonzero(labels == 1)
    sc.dst[scr[::2]] = rng.uniform(0, 1000, size=(scr[::2].size, 2))           # cluster 1: half scrambled -> removed by the median test
    src, dst, F = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.F))
    seed = 4242
    H = sc.H_true.copy()
    lab = labels.copy()
    med = np.zeros(planes)
    kept = host.mhh_compatibility_medians(src.ctypes.data_as(_dp), dst.ctypes.data_as(_dp), sc.n,
                                          lab.ctypes.data_as(C.POINTER(C.c_int)), H.ctypes.data_as(_dp), planes,
                                          F.ctypes.data_as(_dp), C.c_double(2.2 ** 2), min_inliers, C.c_ulonglong(seed),
                                          med.ctypes.data_as(_dp))
    lab_o, H_o, med_o = oracle.compatibility_check(src, dst, labels, sc.H_true, F, 2.2 ** 2, min_inliers, seed)
    assert kept == H_o.shape[0]
    assert np.array_equal(lab, lab_o)
    assert np.array_equal(H[:kept], H_o)
    assert np.array_equal(np.isnan(med), np.isnan(med_o))
    t = ~np.isnan(med_o)
    assert t.any() and np.max(np.abs(med[t] - med_o[t]) / med_o[t]) <= 1e-6
    assert kept < planes, "the case should remove a cluster"
    assert (lab_o[labels == 1] == -1).all()


def test_oracle_greedy_selection_known_answer(synth, oracle):
    """mho_select_greedy on hypotheses whose supports are known: the ground-truth planes (each supported by its own
    region) among duplicates and junk.  Picks come in order of support, a duplicate of a picked plane finds its inliers
    gone, junk never reaches `need`, ties go to the lower index."""
    sc = synth.make_scene(2000, 3, seed=3, noise=0.3, outlier_frac=0.2, with_neighbours=False)
    junk = np.eye(3).reshape(1, 9) * np.array([[1, 1, 1, 1, 1, 1, 1, 1, 1.0]])
    H = np.concatenate([junk, sc.H_true[1:2], sc.H_true, sc.H_true[0:1]], axis=0)        # 0 junk, 1 = plane 1, 2..4 planes, 5 = plane 0
    Hs, idx, cnt, mask = oracle.select_greedy(sc.src, sc.dst, H, 2.2 ** 2, 8, 10)
    support = np.array([int((sc.gt_label == k).sum()) for k in range(3)])
    first_of_plane = {0: 2, 1: 1, 2: 4}                                                 # lowest index carrying each plane
    order = [first_of_plane[k] for k in np.argsort(-support, kind="stable")]
    assert idx.tolist() == order
    assert (np.diff(cnt) <= 0).all() and cnt[-1] >= 8
    assert mask.sum() == sc.n - cnt.sum()
    full = oracle.score(sc.src, sc.dst, H, 2.2 ** 2)
    assert cnt[0] == full[idx[0]]


def _merge_scene(synth, seed):
    """Models of a MergingStep in which the mean shift has something to merge: each true plane in 1..7 slightly perturbed
    copies (clusters of 3, 5, 6, 7 members: sizes for which x / n and x * (1 / n) round differently), shuffled."""
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 7))
    sc = synth.make_scene(600, K, seed=seed, with_neighbours=False)
    Hs = []
    for k in range(K):
        for _ in range(int(rng.integers(1, 8))):
            Hs.append(sc.H_true[k] * (1 + rng.normal(0, 2e-5, 9)))
    H = np.ascontiguousarray(np.array(Hs))
    return sc, np.ascontiguousarray(H[rng.permutation(len(H))])


def _host_merge_candidates(host, H, F, thr_h, seed):
    nh = H.shape[0]
    feat, modes, cand = np.zeros((nh, 6)), np.zeros((nh, 6)), np.zeros((nh, 9))
    cand_mode = np.zeros(nh, np.int32)
    k, draws = C.c_int(0), C.c_ulonglong(0)
    nc = host.mhh_merge_candidates(H.ctypes.data_as(_dp), nh, F.ctypes.data_as(_dp), C.c_double(thr_h), C.c_ulonglong(seed),
                                   feat.ctypes.data_as(_dp), modes.ctypes.data_as(_dp), C.byref(k), cand.ctypes.data_as(_dp),
                                   cand_mode.ctypes.data_as(C.POINTER(C.c_int)), C.byref(draws))
    return feat, modes[:k.value].copy(), cand[:nc].copy(), cand_mode[:nc].copy(), int(draws.value)


def test_merging_step_candidates_equal_the_oracles_bit_for_bit(host, oracle, synth):
    """the host half of MergingStep (M/MultiH.cpp:352-428) — 6-D features, mean-shift modes
    in the reference's summation order with `myMean / inInds.size()` as OpenCV evaluates it (a scale by the reciprocal,
    MeanShiftClustering.h:96), one LM-refined 3-point homography per mode in the operation order of GetHomography3PT —
    is the same arithmetic in the product (host/merge_step.cpp) and in the oracle (oracle/mh_oracle.cpp section 11): every
    double equal, on scenes whose clusters have sizes for which the division and the scale differ."""
    differing = 0
    for seed in range(24):
        sc, H = _merge_scene(synth, seed)
        F = np.ascontiguousarray(sc.F)
        want = oracle.merge_candidates(H, F, 2.2, 7 * seed + 1)
        got = _host_merge_candidates(host, H, F, 2.2, 7 * seed + 1)
        for name, a, b in zip(("features", "modes", "candidates"), want[:3], got[:3]):
            assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)), (seed, name)
        assert np.array_equal(want[3], got[3]) and want[4] == got[4], seed
        assert len(want[1]) < len(H), "the scene should merge something"
        # does this scene tell the two roundings of :96 apart?  (members of a mode in row order, as :85-96 adds them)
        feat = want[0]
        assign = np.zeros(len(H), np.int32)
        modes = np.zeros((len(H), 6))
        host.mhh_mean_shift(feat.ctypes.data_as(_dp), len(H), 6, C.c_double(2.2), C.c_ulonglong(7 * seed + 1),
                            modes.ctypes.data_as(_dp), len(H), assign.ctypes.data_as(C.POINTER(C.c_int)), None)
        for c in range(len(want[1])):
            rows = feat[assign == c]
            if len(rows) == 0:
                continue
            s = np.zeros(6)
            for r in rows:
                s = s + r
            differing += int(np.any(s / len(rows) != s * (1.0 / len(rows))))
    assert differing >= 5, "the scenes must contain clusters for which x / n and x * (1 / n) differ"


def test_approximate_neighbourhood_equals_its_python_restatement(mh):
    """r05 (VERDICT r04 missing 4): MultiH::SetNeighbourApprox — the reference's radiusMatch as FLANN's default search answers it
    (4 randomised KD-trees, best-bin-first, 32 examined points per query), built on the host with the engine's counter RNG
    (host/approx_neighbours.cpp).  Hit for hit the lists of tools/neighbourhood_sweep.py's Python restatement of the same
    algorithm (sequential sums, the same draws) — on a scene with exact duplicates and coordinates shared by many points —
    and characterised: at most `checks` hits per point, most of the exact 8 nearest among them, a share of the hits one-way."""
    import importlib.util
    from scipy.spatial import cKDTree
    spec = importlib.util.spec_from_file_location("neighbourhood_sweep", os.path.join(ROOT, "tools", "neighbourhood_sweep.py"))
    ns = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ns)
    lib = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
    sc = mh.synth.make_scene(700, 3, seed=5, with_neighbours=False)
    src, dst = sc.src.copy(), sc.dst.copy()
    src[50:60] = src[50]; dst[50:60] = dst[50]              # exact duplicates
    src[100:140, 0] = 123.0                                 # a coordinate shared by many points
    for trees, checks, seed in ((4, 32, 1001), (1, 8, 7), (4, 64, 0x464c414e4e)):
        rowptr = np.zeros(sc.n + 1, dtype=np.int32)
        col = np.zeros(sc.n * checks, dtype=np.int32)
        total = lib.mhh_approx_neighbour_hits(src.ctypes.data_as(_dp), dst.ctypes.data_as(_dp), sc.n, trees, checks, C.c_double(200.0),
                                              C.c_ulonglong(seed), rowptr.ctypes.data_as(C.POINTER(C.c_int)),
                                              col.ctypes.data_as(C.POINTER(C.c_int)), col.size)
        assert total == rowptr[-1] > 0
        pv = np.concatenate([src, dst], axis=1).astype(np.float32).astype(np.float64)
        want = ns.forest_hits(pv, trees, checks, 200.0, seed)
        got = [col[rowptr[i]:rowptr[i + 1]].tolist() for i in range(sc.n)]
        assert got == want, (trees, checks, seed)
        deg = np.diff(rowptr)
        assert deg.max() <= checks - 1 or (deg.max() <= checks)      # the query itself is among the examined points nearly always
        if (trees, checks) == (4, 32):
            _, idx = cKDTree(pv).query(pv, k=9)
            recall = np.mean([len(set(got[i]) & set(idx[i, 1:].tolist())) / 8.0 for i in range(sc.n)])
            hit = {(i, j) for i in range(sc.n) for j in got[i]}
            mutual = sum((j, i) in hit for (i, j) in hit) / len(hit)
            # (700 points in 4-D: many of the 32 examined points lie outside the 200-px ball; on barrsmith's 1 094: 29 hits per point,
            # 98 % of the exact 8 nearest, 67 % of the hits mutual — BASELINE.md 1a)
            assert recall > 0.7 and 0.4 < mutual < 0.95 and 8 < deg.mean() < 32, (recall, mutual, deg.mean())
