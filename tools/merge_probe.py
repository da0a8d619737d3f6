#!/usr/bin/env python3
"""Where a MergingStep with hundreds of models spends its time (M/MultiH.cpp:352-471): the host part (features, mean shift, one
3-point fit with LM per mode: mhh_merge_candidates) against the whole step on the engine (mhh_merging_step: + upload of the
candidates, k_moments over all points, the moments back).  Models: the stable sets of the scene (mh_local_homographies +
mh_mean_shift + 3-point fits through the class would need the class; here DLT hypotheses + perturbed planes stand in).
Env: N (50000), MODELS (539)."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, MODELS = int(os.environ.get("N", 50000)), int(os.environ.get("MODELS", 539))
host = C.CDLL(os.path.join(ROOT, "multi-h_amd", "libmultih_host.so"))
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2)
e.propose_dlt4(7, 0, MODELS - 100)
rng = np.random.default_rng(1)
H = np.ascontiguousarray(np.concatenate([sc.H_true[rng.integers(0, 10, 100)] * (1 + rng.normal(0, 3e-3, (100, 9))), e.get_models()]))
nh = H.shape[0]
dp = C.POINTER(C.c_double)
F = np.ascontiguousarray(sc.F)
feat = np.zeros((nh, 6)); modes = np.zeros((nh, 6)); cand = np.zeros((nh, 9)); cm = np.zeros(nh, np.int32)
nm = C.c_int(0); draws = C.c_ulonglong(0)
kept = np.zeros((nh, 9)); changed = C.c_int(0)
for rep in range(4):
    t0 = time.perf_counter()
    nc = host.mhh_merge_candidates(H.ctypes.data_as(dp), nh, F.ctypes.data_as(dp), C.c_double(2.2), C.c_ulonglong(99), feat.ctypes.data_as(dp),
                                   modes.ctypes.data_as(dp), C.byref(nm), cand.ctypes.data_as(dp), cm.ctypes.data_as(C.POINTER(C.c_int)), C.byref(draws))
    t1 = time.perf_counter()
    k = host.mhh_merging_step(e._h, H.ctypes.data_as(dp), nh, F.ctypes.data_as(dp), C.c_double(2.2), C.c_double(0.005), C.c_ulonglong(99),
                              kept.ctypes.data_as(dp), C.byref(changed), C.byref(draws))
    t2 = time.perf_counter()
    e.set_models(cand[:nc])
    t3 = time.perf_counter()
    mom = e.inlier_moments(2.2 * 2.2) if hasattr(e, "inlier_moments") else None
    t4 = time.perf_counter()
    print(f"{nh} models on {N} points: host candidates {1e3 * (t1 - t0):.2f} ms ({nm.value} modes, {nc} candidates); whole MergingStep on the engine {1e3 * (t2 - t1):.2f} ms "
          f"({k} kept); mh_set_models({nc}) {1e3 * (t3 - t2):.2f} ms; mh_inlier_moments {1e3 * (t4 - t3):.2f} ms")
e.close()
