#!/usr/bin/env python3
"""Stress of mh_select_greedy with refitted winners (mh_set_tuning key 30) against the oracle's sequential restatement
(mho_select_greedy_refit): random scenes and batch sizes, both residual modes' forward path, degenerate inputs (duplicate
points, collinear clusters, a support mask with holes).  Every selected model bit for bit, positions, counts, masks."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mh = importlib.import_module("multi-h_amd")
import oracle_lib as O
CASES = int(os.environ.get("CASES", 40))
rng = np.random.default_rng(int(os.environ.get("SEED", 5)))
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
bad = 0
for case in range(CASES):
    n = int(rng.integers(60, 6000)); planes = int(rng.integers(1, 6)); m = int(rng.integers(50, 3000))
    sc = mh.synth.make_scene(n, planes, seed=int(rng.integers(1, 10 ** 6)), outlier_frac=float(rng.uniform(0, 0.5)),
                             noise=float(rng.uniform(0.1, 1.5)), with_neighbours=False, legacy_r04=bool(case % 2))
    src, dst, aff = sc.src.copy(), sc.dst.copy(), sc.aff.copy()
    if case % 5 == 0 and n > 100:                         # duplicates and a collinear cluster
        src[10:30] = src[10]; dst[10:30] = dst[10]
        src[40:80, 1] = src[40, 1]
    thr2 = float(rng.choice([2.2, 1.0, 4.0])) ** 2
    need, maxm = int(rng.integers(8, 40)), int(rng.integers(1, 12))
    mask = (rng.random(n) > (0.2 if case % 3 == 0 else 0.0)).astype(np.uint8)
    e.set_correspondences(src, dst, aff)
    e.set_epipolar(sc.F, sc.e2)
    e.propose_dlt4(int(rng.integers(1, 10 ** 6)), 0, m)
    H = e.get_models()
    e.set_tuning(30, 1)
    try:
        Hs, idx, cnt, mk = e.select_greedy(thr2, need, maxm, mask)
    finally:
        e.set_tuning(30, 0)
    with np.errstate(all="ignore"):
        Hr, ir, cr, mr = O.select_greedy_refit(src, dst, aff, sc.F, sc.e2, H, thr2, need, maxm, mask)
    ok = (np.array_equal(idx, ir) and np.array_equal(cnt, cr) and np.array_equal(mk, mr) and
          np.array_equal(Hs.view(np.uint64), Hr.view(np.uint64)))
    if not ok:
        bad += 1
        print(f"case {case}: MISMATCH n {n} planes {planes} m {m} thr2 {thr2} need {need} max {maxm}: idx {idx.tolist()} vs {ir.tolist()}, counts {cnt.tolist()} vs {cr.tolist()}, "
              f"mask differs at {int((mk != mr).sum())}, H equal {np.array_equal(Hs.view(np.uint64), Hr.view(np.uint64)) if Hs.shape == Hr.shape else 'shape'}", flush=True)
e.close()
print(f"stress_select_refit: {CASES} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
