#!/usr/bin/env python3
"""Experiment (r04): the FP32 pre-test score (k_score32) as a resident grid — mh_set_tuning key 24: 0 hardware dispatch, -1 resident
with ~37 500 items, n resident with n point slices.  Same counts."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, M = 50000, 100000
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
thr2 = 2.2 ** 2
ref = None
for v in (0, 0, -1, 4, 8, 12, 16, 24, 49, 0):
    e.set_tuning(24, v)
    cnt = e.score(thr2)
    e.profile_reset(); e.profile_enable(True)
    for _ in range(8):
        e.score(thr2, fetch=False)
    e.synchronize(); n, ms = e.profile_get(2); e.profile_enable(False)
    if ref is None: ref = cnt
    print(f"key 24 = {v:3d}: {ms / n:.4f} ms; counts {'equal' if np.array_equal(cnt, ref) else 'DIFFERENT'}", flush=True)
e.close()
