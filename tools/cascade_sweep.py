#!/usr/bin/env python3
"""LabelingStep time against the cap on the dominance cascade's passes inside the solver launch (mh_set_tuning key 17)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10))
sc = mh.synth.make_scene(N, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
ref = None
for cap in [int(x) for x in os.environ.get("CAPS", "2,0,2,1,0").split(",")]:
    e.set_tuning(17, cap)
    ts = []
    for rep in range(4):
        e.set_models(H)
        t0 = time.time(); lab, en, cyc = e.labeling_step(False, np.full(N, -1, np.int32)); ts.append((time.time() - t0) * 1e3)
    st = e.expand_stats()
    if ref is None: ref = (lab.copy(), en)
    same = bool(np.array_equal(lab, ref[0]) and en == ref[1])
    print(f"cascade cap {cap}: LabelingStep {min(ts[1:]):.2f} ms (min of 3), solver {st['solve_us'] / 1e3:.2f} ms, barriers {st['barriers']}, relabels {st['relabels']}, core sites {st['core_sites']}, labels as uncapped: {same}", flush=True)
