#!/usr/bin/env python3
"""BASELINE configs[4]-style run on ONE GPU: the full propose -> {merge, label, re-estimate} loop
through the host class MultiH (libmultih_host.so) with a fixed number of iterations.
Prints wall time of the loop and label agreement with the synthetic ground truth."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10))
ITERS, HYP = int(os.environ.get("ITERS", 20)), int(os.environ.get("HYP", 100000))
ITER_HYP = int(os.environ.get("ITER_HYP", 0))
host = C.CDLL(os.path.join(ROOT, "multi-h_amd", "libmultih_host.so"))
sc = mh.synth.make_scene(N, K, seed=1234, with_neighbours=False)
dp = C.POINTER(C.c_double)
labels = np.full(N, -7, dtype=np.int32); Hout = np.zeros((256, 9))
it, en, secs = C.c_int(0), C.c_double(0), C.c_double(0)
src, dst, aff, F, e2 = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff, sc.F, sc.e2))
t0 = time.time()
k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), N,
                         F.ctypes.data_as(dp), e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(2.2),
                         C.c_double(0.005), C.c_double(0.5), 20, C.c_ulonglong(1234), HYP, 32, ITERS,
                         None, 0, labels.ctypes.data_as(C.POINTER(C.c_int)), Hout.ctypes.data_as(dp), 256,
                         C.byref(it), C.byref(en), C.byref(secs), ITER_HYP, 4)
wall = time.time() - t0
agree = 0
for p in range(K):
    lp = labels[sc.gt_label == p]; lp = lp[lp >= 0]
    if lp.size: agree += np.bincount(lp).max()
print(f"N={N} planes={K} hypotheses={HYP}: clusters={k} iterations={it.value} energy={en.value:.0f} "
      f"loop={secs.value:.2f}s total={wall:.2f}s  inlier-agreement={agree/(sc.gt_label>=0).sum():.3f} "
      f"outliers labelled -1: {(labels[sc.gt_label<0]==-1).mean():.3f}")
