#!/usr/bin/env python3
"""Quantitative agreement with the ONE output the reference ships: Executable/results/barrsmith/
result_barrsmith.txt holds the 1 094 correspondences the reference kept of the 2 903 in barrsmith_points_with_no_annotation.txt
(both stored as arrays in tests/golden/barrsmith.npz) and the label it gave each: -1 (182), planes 0..4 (33, 514, 128, 83,
154).  The result file carries x1 y1 twice (quirk A-9), so each row is matched back to the input by (x1, y1) — and, where
several input matches share that source point, by the closest affinity — to recover x2 y2 and the input affinity.  EXACTLY
those correspondences then go through Process() with the harness defaults (M/main.cpp:55-59: 2.6 / 2.2 / 0.005 / 0.5 / 20):
F by the engine's own 8-point RANSAC on them (cv::findFundamentalMat is outside /root/reference), then
  (i)  the reference's own route: INIT_STABLE_SETS (per-point HAF homographies, mean shift, 3-point fits), and
  (ii) the default route: DLT proposals + greedy selection,
each with the post-filter on, as the harness runs it.  Reported: number of planes, adjusted Rand index against the
reference's labels (all points with -1 as a class of its own; and on the reference's non-outliers only), per-plane purity
(share of a reference plane's points under our dominant label for it) and the outlier agreement.  The reference's run
is not bit-reproducible (OpenCV RANSAC + FLANN + MSVC rand()), so this is agreement, not parity."""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")


def kept_correspondences(with_rows=False):
    """(x1 y1 x2 y2 a11 a12 a21 a22) of the reference's 1 094 kept correspondences, in result-file order, and their labels."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "barrsmith.npz"))
    pts, res = g["points"], g["result"]
    by_key = {}
    for i, (a, b) in enumerate(pts[:, :2]):
        by_key.setdefault((round(a, 3), round(b, 3)), []).append(i)
    used, rows = set(), []
    for r in res:
        cand = [i for i in by_key.get((round(r[0], 3), round(r[1], 3)), []) if i not in used]
        if not cand:
            rows.append(-1)
            continue
        best = min(cand, key=lambda i: float(np.abs(pts[i, 4:8] - r[4:8]).sum()))
        used.add(best)
        rows.append(best)
    rows = np.asarray(rows)
    ok = rows >= 0
    if with_rows:
        return pts, rows[ok], res[ok, 8].astype(int)
    return pts[rows[ok]], res[ok, 8].astype(int), int(ok.sum()), int(len(res))


def adjusted_rand(a, b):
    a, b = np.asarray(a), np.asarray(b)
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    n = a.size
    cont = np.zeros((ai.max() + 1, bi.max() + 1), dtype=np.int64)
    np.add.at(cont, (ai, bi), 1)
    comb = lambda x: x * (x - 1) // 2
    s_ij, s_a, s_b = comb(cont).sum(), comb(cont.sum(1)).sum(), comb(cont.sum(0)).sum()
    exp = s_a * s_b / comb(n)
    mx = 0.5 * (s_a + s_b)
    return float((s_ij - exp) / (mx - exp)) if mx != exp else 1.0


def agreement(ours, ref):
    out = {"planes": int(ours.max() + 1), "ours_histogram": np.bincount(ours + 1).tolist(), "ari_all": adjusted_rand(ours, ref)}
    inl = ref >= 0
    out["ari_reference_inliers"] = adjusted_rand(ours[inl], ref[inl])
    both = inl & (ours >= 0)
    out["ari_points_both_assign_to_a_plane"] = adjusted_rand(ours[both], ref[both])
    pur = {}
    for p in np.unique(ref[inl]):
        mine = ours[ref == p]
        vals, cnts = np.unique(mine, return_counts=True)
        pur[int(p)] = {"points": int(mine.size), "dominant_label": int(vals[np.argmax(cnts)]), "purity": float(cnts.max() / mine.size)}
    out["per_reference_plane"] = pur
    out["outliers_agreeing"] = float(((ours == -1) & (ref == -1)).sum() / max((ref == -1).sum(), 1))
    out["reference_inliers_we_call_outliers"] = float(((ours == -1) & inl).sum() / inl.sum())
    return out


def _host():
    return C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))


def run(route, corr, F, e2, seed=1234, hypotheses=20000, knn=0, approx=None):
    host = _host()
    host.mhh_set_neighbourhood(int(knn), C.c_double(0.0))      # 0 = the class default (16 nearest hits within 1 / locality)
    # approx = (trees, checks): MultiH::SetNeighbourApprox — the reference's radiusMatch as FLANN's default search answers it
    host.mhh_set_neighbourhood_approx(int(approx[0]) if approx else 0, int(approx[1]) if approx else 32, C.c_ulonglong(0x464c414e4e + seed))
    dp = C.POINTER(C.c_double)
    src, dst, aff = (np.ascontiguousarray(corr[:, a:b]) for a, b in ((0, 2), (2, 4), (4, 8)))
    n = len(src)
    labels = np.full(n, -7, dtype=np.int32)
    Hout = np.zeros((256, 9))
    it, en = C.c_int(0), C.c_double(0)
    Fc, e2c = np.ascontiguousarray(F.reshape(9)), np.ascontiguousarray(e2)
    host.mhh_set_post_filter(1)
    k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), n, Fc.ctypes.data_as(dp),
                             e2c.ctypes.data_as(dp), C.c_double(2.6), C.c_double(2.2), C.c_double(0.005), C.c_double(0.5), 20,
                             C.c_ulonglong(seed), hypotheses, 32, 0, None, 0, labels.ctypes.data_as(C.POINTER(C.c_int)),
                             Hout.ctypes.data_as(dp), 256, C.byref(it), C.byref(en), None, 0, -1 if route == "stable_sets" else 4)
    C.CDLL(None).fflush(None)
    return k, labels, it.value, en.value


def front_half(pts, seed=1234):
    """The reference's GetFundamentalMatrixAndRefineData (M/MultiH.cpp:770-848) on ALL input correspondences through the
    engine's pieces — F by 8-point RANSAC, epipoles, Hartley-Sturm correction + affine consistency filter + optimal affinity
    — exactly what Process() does without a given F (tests/test_gpu_alternation.py holds that decomposition equal to it).
    Returns (indices of the kept rows, their REFINED correspondences, F, e2)."""
    e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
    e.set_correspondences(pts[:, 0:2], pts[:, 2:4], pts[:, 4:8])
    F, e2, mask, inl = e.estimate_fundamental(seed ^ 0xf00d, 4000, 2.6)
    e1, e2 = e.epipoles(F)
    keep, refined = e.refine_correspondences(F, e1, e2, mask)
    e.close()
    idx = np.flatnonzero(keep)
    return idx, np.ascontiguousarray(refined[idx]), F, e2


def harness_route(pts, route, seed, load_filter=2.0, metric=1, hypotheses=20000):
    """The reference's caller on the RAW rows, stage by stage, as multih_harness runs it since r06 (host/main.cpp):
    LoadPointsFromFile's filter (M/main.cpp:399-409: F-RANSAC at `load_filter` px, 0 = off) through
    multih::FilterCorrespondencesByEpipolarGeometry, then Process() WITHOUT a given F (its own RANSAC at 2.6 px,
    OptimalTriangulation, distanceError <= 1: M/MultiH.cpp:770-848), both with `metric` (1 = the point-to-epipolar-line
    distance cv::findFundamentalMat thresholds, 0 = Sampson).  Returns (rows of `pts` the loop saw, their labels, planes,
    the stage table)."""
    host = _host()
    dp = C.POINTER(C.c_double)
    n0 = len(pts)
    src, dst = (np.ascontiguousarray(pts[:, a:b]) for a, b in ((0, 2), (2, 4)))
    mask = np.ones(n0, dtype=np.uint8)
    if load_filter > 0:
        k0 = host.mhh_filter_correspondences(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), n0, C.c_double(load_filter),
                                             C.c_ulonglong(seed ^ 0x10adf117e4), 4000, int(metric), 0, mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
        assert k0 >= 8, "the load filter failed"
    rows1 = np.flatnonzero(mask)
    sub = np.ascontiguousarray(pts[rows1])
    # Process()'s own front half, decomposed with the engine's pieces to learn WHICH rows it keeps (the class returns labels
    # for the kept rows in order; tests/test_gpu_alternation.py holds this decomposition equal to what the class does)
    e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
    e.set_fundamental_metric(metric)
    e.set_correspondences(sub[:, 0:2], sub[:, 2:4], sub[:, 4:8])
    F, e2, m, inl = e.estimate_fundamental(seed ^ 0xf00d, 4000, 2.6)
    e1, e2b = e.epipoles(F)
    keep, _ = e.refine_correspondences(F, e1, e2b, m)
    reason = e.refine_reasons()
    e.close()
    rows2 = rows1[np.flatnonzero(keep)]
    host.mhh_set_neighbourhood(0, C.c_double(0.0))
    host.mhh_set_neighbourhood_approx(0, 32, C.c_ulonglong(0))
    host.mhh_set_post_filter(1)
    host.mhh_set_fundamental_metric(int(metric))
    s2, d2, a2 = (np.ascontiguousarray(sub[:, a:b]) for a, b in ((0, 2), (2, 4), (4, 8)))
    labels = np.full(len(sub), -7, dtype=np.int32)
    Hout = np.zeros((256, 9))
    it, en = C.c_int(0), C.c_double(0)
    k = host.mhh_run_process(s2.ctypes.data_as(dp), d2.ctypes.data_as(dp), a2.ctypes.data_as(dp), len(sub), None, None,
                             C.c_double(2.6), C.c_double(2.2), C.c_double(0.005), C.c_double(0.5), 20, C.c_ulonglong(seed), hypotheses, 32, 0,
                             None, 0, labels.ctypes.data_as(C.POINTER(C.c_int)), Hout.ctypes.data_as(dp), 256, C.byref(it), C.byref(en),
                             None, 0, -1 if route == "stable_sets" else 4)
    host.mhh_set_fundamental_metric(-1)
    st = (C.c_int * 4)()
    host.mhh_get_front_stages(st)
    C.CDLL(None).fflush(None)
    stages = {"loaded": int(n0), "after_load_filter": int(len(rows1)), "in_ransac_mask": int(st[1]), "after_optimal_triangulation": int(st[2]),
              "after_distance_error": int(st[3])}
    assert st[3] == len(rows2) and st[1] == int((reason != 1).sum()), "the class and its decomposition keep different rows"
    return rows2, labels[:len(rows2)].copy(), int(k), stages


def raw_route(seeds=(1234, 7, 99), configs=((2.0, 1), (0.0, 0))):
    """From the RAW input file (2 903 rows) as the reference's harness runs it; compared with the reference's labels on the
    rows BOTH kept.  configs: (load-filter threshold, metric) — (2.0, 1) is the harness since r06, (0.0, 0) what it did until r05."""
    pts, ref_rows, ref_labels = kept_correspondences(with_rows=True)
    out = {}
    for load_filter, metric in configs:
        tag = f"load filter {load_filter:g} px, {'point-to-line' if metric else 'Sampson'} distance"
        out[tag] = {}
        for route in ("dlt", "stable_sets"):
            runs = []
            for seed in seeds:
                rows, labels, k, stages = harness_route(pts, route, seed, load_filter, metric)
                full = np.full(len(pts), -2, dtype=int)              # -2: dropped by OUR front half
                full[rows] = labels
                ours = full[ref_rows]
                both = ours > -2
                a = agreement(ours[both], ref_labels[both]) if k > 0 else {"planes": int(k)}
                a.update(seed=seed, stages=stages, kept_by_both=int(both.sum()), kept_by_reference=int(len(ref_rows)))
                runs.append(a)
                print(f"raw input [{tag}], {route:12s} seed {seed:5d}: {stages['loaded']} -> {stages['after_load_filter']} -> {stages['in_ransac_mask']} -> "
                      f"{stages['after_optimal_triangulation']} -> {stages['after_distance_error']} (the reference kept {len(ref_rows)}, {int(both.sum())} in common): "
                      f"{k} planes, ARI on the reference's inliers {a.get('ari_reference_inliers', float('nan')):.3f}, all {a.get('ari_all', float('nan')):.3f}, "
                      f"histogram {a.get('ours_histogram')}", flush=True)
            aris = sorted(r.get("ari_reference_inliers", float("nan")) for r in runs)
            print(f"   => {route}: planes {[r['planes'] for r in runs]}, median ARI on the reference's inliers {aris[len(aris) // 2]:.3f}, min {aris[0]:.3f}", flush=True)
            out[tag][route] = runs
    return out


def main():
    corr, ref, matched, total = kept_correspondences()
    e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
    e.set_correspondences(corr[:, 0:2], corr[:, 2:4], corr[:, 4:8])
    F, e2, mask, inl = e.estimate_fundamental(1234 ^ 0xf00d, 4000, 2.6)
    e.close()
    rec = {"matched_rows": matched, "result_rows": total, "reference_histogram": np.bincount(ref + 1).tolist(),
           "F_inliers_at_2.6px": int(inl), "routes": {}}
    for route in ("stable_sets", "dlt"):
        k, labels, it, en = run(route, corr, F, e2)
        a = agreement(labels, ref) if k > 0 else {"planes": int(k)}
        a.update(iterations=it, energy=en)
        rec["routes"][route] = a
        print(f"{route:12s}: {k} planes, ARI all {a.get('ari_all', float('nan')):.3f}, on the reference's inliers {a.get('ari_reference_inliers', float('nan')):.3f}, "
              f"where both assign a plane {a.get('ari_points_both_assign_to_a_plane', float('nan')):.3f}; purity per reference plane "
              + ", ".join(f"{p}:{v['purity']:.2f}" for p, v in a.get("per_reference_plane", {}).items())
              + f"; reference outliers we also reject {a.get('outliers_agreeing', float('nan')):.2f}", flush=True)
    # sensitivity (printed, not part of the record's headline): neighbourhood size and seeds
    if os.environ.get("SWEEP"):
        for route in ("stable_sets", "dlt"):
            for knn in (16, 32, 64):
                for seed in (1234, 7, 99):
                    k, labels, it, en = run(route, corr, F, e2, seed=seed, knn=knn)
                    a = agreement(labels, ref)
                    print(f"  sweep {route:12s} k-NN {knn:3d} seed {seed:5d}: {k} planes, ARI inliers {a['ari_reference_inliers']:.3f}, all {a['ari_all']:.3f}, "
                          f"histogram {a['ours_histogram']}", flush=True)
        run("dlt", corr, F, e2, knn=0)
    # the reference's own neighbourhood rule as FLANN answers it (MultiH::SetNeighbourApprox, 4 trees / 32 checks), three seeds
    rec["approx_neighbourhood"] = {}
    for route in ("stable_sets", "dlt"):
        runs = []
        for seed in (1234, 7, 99):
            k, labels, it, en = run(route, corr, F, e2, seed=seed, approx=(4, 32))
            a = agreement(labels, ref) if k > 0 else {"planes": int(k), "ari_reference_inliers": float("nan"), "ari_all": float("nan")}
            runs.append({"seed": seed, "planes": int(k), "ari_reference_inliers": a["ari_reference_inliers"], "ari_all": a["ari_all"]})
        rec["approx_neighbourhood"][route] = runs
        print(f"approximate neighbourhood (4 trees / 32 checks), {route:12s}: planes {[r['planes'] for r in runs]}, ARI on the reference's inliers "
              + ", ".join(f"{r['ari_reference_inliers']:.3f}" for r in runs), flush=True)
    run("dlt", corr, F, e2)                                  # (leaves the default neighbourhood set)
    if os.environ.get("RAW", "1") != "0":
        seeds = tuple(int(x) for x in os.environ.get("SEEDS", "1234,7,99").split(","))
        rec["from_the_raw_input_file"] = raw_route(seeds)
    print(json.dumps(rec))
    return rec


if __name__ == "__main__":
    main()
