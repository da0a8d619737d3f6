#!/usr/bin/env python3
"""Sweeps the alpha-expansion schedule knobs (relax rounds/launch, relax launches/check, push cycles/launch,
push launches/round) on one labeling problem; prints ms per LabelingStep.  Diagnostic."""
import importlib, itertools, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10))
sc = mh.synth.make_scene(N, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
ref = None
CFGS = [(8,3,16,8),(16,2,32,8),(16,2,32,16),(16,2,64,8),(16,2,48,8),(12,2,32,8),(16,2,32,12),(24,2,32,8),(16,2,64,4),(16,2,32,8),(8,3,16,8)]
if os.environ.get("CFGS"): CFGS = [tuple(int(x) for x in c.split(',')) for c in os.environ["CFGS"].split(';')]
for cfg in CFGS:
    for k, v in enumerate(cfg): e.set_tuning(2 + k, v)
    e.set_models(H); e.data_cost(fetch=False)
    t0 = time.time(); lab, en, cyc = e.expand(); dt = time.time() - t0
    if ref is None: ref = (lab.copy(), en)
    ok = np.array_equal(lab, ref[0]) and en == ref[1]
    st = e.expand_stats()
    print(f"cfg {cfg}: {dt*1e3:7.1f} ms  ok={ok} launches pr={st['pr_launches']} bfs={st['bfs_launches']} syncs={st['host_syncs']}", flush=True)
