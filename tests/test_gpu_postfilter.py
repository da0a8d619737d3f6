"""GPU tests of the engine's part of HomographyCompatibilityCheck (M/MultiH.cpp:100-222): mh_compat_trial_stats
(csrc/compat.hip) — per trial the order statistics of the squared transfer errors at ranks k-3 .. k+1 and their three
largest values, by a radix select on the device — against sorting the oracle's distances, and the whole check with the
engine's statistics against the host's own and the oracle's restatement."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_dp = C.POINTER(C.c_double)


def _expected_stats(oracle, pts, begin, tri, H, ok):
    clusters, trials = tri.shape[0], tri.shape[1]
    out = np.empty((clusters, trials, 8))
    for c in range(clusters):
        p = pts[begin[c]:begin[c + 1]]
        with np.errstate(all="ignore"):
            R = oracle.residual_matrix(p[:, :2], p[:, 2:], H[c])            # trials x Nc, bit-exact formula of :162-170
        for t in range(trials):
            keep = np.ones(p.shape[0], dtype=bool)
            keep[tri[c, t]] = False
            d = R[t][keep] if ok[c, t] else np.full(keep.sum(), np.nan)
            d = np.where(np.isnan(d), 1e300, d)
            d.sort()
            k = d.size // 2
            out[c, t, :5] = d[k - 3:k + 2]
            out[c, t, 5:] = d[-3:]
    return out


@pytest.mark.parametrize("sizes,trials", [((19, 20, 22), 9), ((257, 1000, 64, 5003), 7), ((30000,), 4)])
def test_trial_statistics_equal_a_sort_of_the_oracles_distances(engine, synth, oracle, sizes, trials):
    """Cluster sizes from the smallest the entry point accepts (19: the caller keeps smaller ones, where the reference's
    three stale buffer entries reach the median ranks) past one and several strides of the workgroup; homographies near
    the truth, far from it, degenerate (all zero: every distance NaN -> 1e300), with a vanishing denominator on some
    points (inf), failed fits (ok = 0), and duplicated points so that the selected rank falls into a run of equal values."""
    rng = np.random.default_rng(sum(sizes) + trials)
    total = int(sum(sizes))
    sc = synth.make_scene(max(total, 2000), 3, seed=17, noise=0.6, outlier_frac=0.3, with_neighbours=False)
    pts = np.concatenate([sc.src, sc.dst], axis=1)[:total].copy()
    begin = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    clusters = len(sizes)
    for c in range(clusters):                                                # a run of equal values around the middle ranks
        b0, nc = begin[c], sizes[c]
        pts[b0 + nc // 3:b0 + nc // 3 + min(nc // 2, 40)] = pts[b0 + nc // 3]
    tri = np.stack([np.stack([rng.choice(sizes[c], 3, replace=False) for _ in range(trials)]) for c in range(clusters)]).astype(np.int32)
    H = np.empty((clusters, trials, 9))
    ok = np.ones((clusters, trials), dtype=np.uint8)
    for c in range(clusters):
        for t in range(trials):
            base = sc.H_true[(c + t) % 3]
            kind = t % 7
            if kind == 0: H[c, t] = base
            elif kind == 1: H[c, t] = base * (1 + rng.normal(0, 1e-3, 9))
            elif kind == 2: H[c, t] = rng.normal(0, 1, 9)
            elif kind == 3: H[c, t] = 0.0                                    # 0/0 everywhere
            elif kind == 4:
                H[c, t] = base; ok[c, t] = 0
            elif kind == 5:                                                  # denominator exactly zero on one point: x/0 = inf
                H[c, t] = base
                x, y = pts[begin[c] + 1, 0], pts[begin[c] + 1, 1]
                H[c, t, 6:] = [1.0, 0.0, -x]
            else: H[c, t] = base * (1 + rng.normal(0, 3e-5, 9))
    want = _expected_stats(oracle, pts, begin, tri, H, ok)
    got = engine.compat_trial_stats(pts, begin, tri, H, ok)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert (want[:, :, :5] == want[:, :, 0:1]).all(axis=2).any(), "a trial whose middle ranks are all equal should be among the cases"


def test_trial_statistics_refuse_what_the_caller_must_keep(mh, engine):
    pts = np.zeros((30, 4)); tri = np.zeros((1, 2, 3), np.int32); tri[0, :, 1] = 1; tri[0, :, 2] = 2
    H = np.zeros((1, 2, 9)); ok = np.ones((1, 2), np.uint8)
    with pytest.raises(mh.MultiHError):
        engine.compat_trial_stats(pts[:18], [0, 18], tri, H, ok)             # fewer than 19 points
    bad = tri.copy(); bad[0, 1, 2] = 30
    with pytest.raises(mh.MultiHError):
        engine.compat_trial_stats(pts, [0, 30], bad, H, ok)                  # a draw outside the cluster
    assert engine.compat_trial_stats(pts, [0, 30], tri, H, ok).shape == (1, 2, 8)


@pytest.mark.parametrize("sizes,min_inliers", [((700, 523, 64, 17), 20), ((2400, 2500, 19, 22, 18), 4), ((40, 33, 5), 0)])
def test_check_with_the_engines_statistics_equals_the_hosts_and_the_oracles(mh, engine, engine_lib, synth, oracle, sizes, min_inliers):
    """multih::CompatibilityCheck as Process() runs it (3-point fits and the reference's buffer bookkeeping on the host, the
    trials' order statistics from the engine) against the same function on the host alone — medians, labels, models bit
    for bit — and against the oracle's literal restatement of M/MultiH.cpp:100-222 (decisions equal, medians to 1e-6:
    the oracle has its own 3-point solver)."""
    host = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
    planes = len(sizes)
    sc = synth.make_scene(30000 if planes == 5 else 12000, planes, seed=11 + planes, noise=0.4, outlier_frac=0.0, with_neighbours=False)
    rng = np.random.default_rng(5)
    labels = np.full(sc.n, -1, dtype=np.int32)
    for c, sz in enumerate(sizes):
        members = np.flatnonzero(sc.gt_label == c)[:sz]
        assert members.size == sz
        labels[members] = c
    scr = np.flatnonzero(labels == 1)
    sc.dst[scr[::2]] = rng.uniform(0, 1000, size=(scr[::2].size, 2))          # cluster 1: half scrambled -> removed by the median test
    src, dst, F = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.F))
    seed = 4242

    def run(fn, *front):
        H = sc.H_true.copy(); lab = labels.copy(); med = np.zeros(planes)
        kept = fn(*front, src.ctypes.data_as(_dp), dst.ctypes.data_as(_dp), sc.n, lab.ctypes.data_as(C.POINTER(C.c_int)),
                  H.ctypes.data_as(_dp), planes, F.ctypes.data_as(_dp), C.c_double(2.2 ** 2), min_inliers, C.c_ulonglong(seed),
                  med.ctypes.data_as(_dp))
        return kept, lab, H, med

    k_h, lab_h, H_h, med_h = run(host.mhh_compatibility_medians)
    k_e, lab_e, H_e, med_e = run(host.mhh_compatibility_medians_on_engine, engine._h)
    assert k_e == k_h >= 0 and np.array_equal(lab_e, lab_h) and np.array_equal(H_e[:k_e], H_h[:k_h])
    assert np.array_equal(med_e.view(np.uint64), med_h.view(np.uint64))
    lab_o, H_o, med_o = oracle.compatibility_check(src, dst, labels, sc.H_true, F, 2.2 ** 2, min_inliers, seed)
    assert k_e == H_o.shape[0] and np.array_equal(lab_e, lab_o) and np.array_equal(H_e[:k_e], H_o)
    t = ~np.isnan(med_o)
    assert np.array_equal(np.isnan(med_e), ~t) and np.max(np.abs(med_e[t] - med_o[t]) / med_o[t]) <= 1e-6
    assert k_e < planes


def test_the_trials_three_point_fits_on_the_device_are_the_hosts_bit_for_bit(mh, engine, synth):
    """r06: mh_compat_trial_stats_fit fits the trials' homographies on the device (csrc/compat.hip k_compat_fit: GetHomography3PT
    without refinement, M/MultiH.cpp:154, :995-1050) — until r05 the host did (3.2 of the post-filter's 3.6 ms at configs[4]).
    Every fit against the host's Homography3PTLinear on the same three correspondences, bit for bit, including degenerate draws
    (a point drawn twice; three collinear points; coincident points: non-finite fits -> ok = 0, H = 0), and the statistics
    against the r05 form fed with those fits."""
    host = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
    sizes, trials = (40, 700, 2333), 60
    total = int(sum(sizes))
    sc = synth.make_scene(4000, 3, seed=23, noise=0.5, outlier_frac=0.2, with_neighbours=False)
    pts = np.concatenate([sc.src, sc.dst], axis=1)[:total].copy()
    begin = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    rng = np.random.default_rng(8)
    tri = np.stack([np.stack([rng.choice(sizes[c], 3, replace=False) for _ in range(trials)]) for c in range(len(sizes))]).astype(np.int32)
    tri[0, 0] = [5, 5, 9]                                        # a point drawn twice
    tri[1, 1] = [7, 7, 7]                                        # ... three times: no homography
    pts[begin[2] + 1] = pts[begin[2] + 0]; pts[begin[2] + 2] = pts[begin[2] + 0]; tri[2, 2] = [0, 1, 2]      # three coincident correspondences
    line = np.linspace(100.0, 300.0, 3)
    pts[begin[1] + 10:begin[1] + 13, 0] = line; pts[begin[1] + 10:begin[1] + 13, 1] = 2.0 * line + 5.0; tri[1, 3] = [10, 11, 12]   # collinear sources
    F = np.ascontiguousarray(sc.F)
    stats, H_dev, ok_dev = engine.compat_trial_stats_fit(pts, begin, tri, F)
    H_host = np.zeros_like(H_dev)
    ok_host = np.zeros_like(ok_dev)
    for c in range(len(sizes)):
        for t in range(trials):
            p = pts[begin[c] + tri[c, t]]
            p1, p2 = np.ascontiguousarray(p[:, :2]), np.ascontiguousarray(p[:, 2:])
            h = np.zeros(9)
            with np.errstate(all="ignore"):
                ok_host[c, t] = host.mhh_homography_3pt(p1.ctypes.data_as(_dp), p2.ctypes.data_as(_dp), 3, F.ctypes.data_as(_dp), h.ctypes.data_as(_dp))
            H_host[c, t] = h if ok_host[c, t] else 0.0
    assert np.array_equal(ok_dev, ok_host), f"{int((ok_dev != ok_host).sum())} fits succeed on one side only"
    assert np.array_equal(H_dev.view(np.uint64), H_host.view(np.uint64)), "a 3-point fit differs in some bit"
    assert ok_dev.sum() >= ok_dev.size - 6 and (ok_dev == 0).any()
    want = engine.compat_trial_stats(pts, begin, tri, H_host, ok_host)
    assert np.array_equal(stats.view(np.uint64), want.view(np.uint64))
    # the host-fit form of the whole check stays available and equal (mhh_set_compat_fits_on_engine)
    labels = np.full(sc.n, -1, dtype=np.int32)
    for c in range(3):
        labels[np.flatnonzero(sc.gt_label == c)[:600]] = c
    src, dst = np.ascontiguousarray(sc.src), np.ascontiguousarray(sc.dst)
    res = []
    for on in (1, 0):
        host.mhh_set_compat_fits_on_engine(on)
        H = sc.H_true.copy(); lab = labels.copy(); med = np.zeros(3)
        k = host.mhh_compatibility_medians_on_engine(engine._h, src.ctypes.data_as(_dp), dst.ctypes.data_as(_dp), sc.n, lab.ctypes.data_as(C.POINTER(C.c_int)),
                                                     H.ctypes.data_as(_dp), 3, F.ctypes.data_as(_dp), C.c_double(2.2 ** 2), 20, C.c_ulonglong(77), med.ctypes.data_as(_dp))
        res.append((k, lab.tobytes(), H.tobytes(), med.tobytes()))
    host.mhh_set_compat_fits_on_engine(1)
    assert res[0] == res[1] and res[0][0] == 3
