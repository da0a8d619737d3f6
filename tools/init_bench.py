#!/usr/bin/env python3
"""Times the reference-style initialisation pieces (per-point HAF, mean shift over N points) on the GPU."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10))
sc = mh.synth.make_scene(N, K, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2)
t0 = time.time(); H, feat = e.local_homographies(0.005); t1 = time.time() - t0
feat = np.where(np.isfinite(feat), feat, 1e300)
if "MS_BATCH" in os.environ: e.set_tuning(7, int(os.environ["MS_BATCH"]))
t0 = time.time(); modes, assign, k = e.mean_shift(feat, 2.2, 99); t2 = time.time() - t0
sizes = np.bincount(assign[assign >= 0], minlength=k)
print(f"N={N}: local homographies {t1*1e3:.1f} ms; mean shift {t2:.2f} s -> {k} modes, {int((sizes>=3).sum())} with >=3 points, largest {np.sort(sizes)[-5:]}")
