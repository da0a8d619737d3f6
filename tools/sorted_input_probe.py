import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
mh = importlib.import_module("multi-h_amd")
sc = mh.synth.make_scene(50000, 10, seed=1234)
H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
for name, perm in (("as generated", np.arange(sc.n)), ("sorted by x", np.argsort(sc.src[:, 0], kind="stable"))):
    inv = np.empty(sc.n, np.int64); inv[perm] = np.arange(sc.n)
    src, dst, aff = sc.src[perm], sc.dst[perm], sc.aff[perm]
    e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
    e.set_correspondences(src, dst, aff); e.set_epipolar(sc.F, sc.e2); e.build_neighbors_knn(16)
    e.set_models(H); e.data_cost(fetch=False); e.expand()
    t0 = time.time(); lab, en, cyc = e.expand(); dt = time.time() - t0
    st = e.expand_stats()
    print(f"{name:14s}: expansion {dt*1e3:6.1f} ms, solver {st['solve_us']/1e3:6.1f} ms, energy {en}, relabels {st['relabels']}", flush=True)
    e.close()
