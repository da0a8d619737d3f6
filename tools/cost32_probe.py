#!/usr/bin/env python3
"""Experiment (r04): the int32 cost matrix (k_cost32) as a resident grid — mh_set_tuning key 23: 0 hardware dispatch, -1 resident
with ~37 500 items, n resident with n point slices.  Same matrix (checked on sample rows and counts)."""
import os as _os
# r05: these schedule variants live in the measurement library only (python multi-h_amd/build.py --tuning)
_os.environ.setdefault("MH_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multi-h_amd", "libmultih_hip_tuning.so"))
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, M = 50000, 100000
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
ref = None
for v, sm, pm in ((8, 0, 0), (8, 0, 1), (8, 0, 0), (8, 0, 1), (0, 0, 0), (4, 0, 0), (12, 0, 0), (25, 0, 0), (8, 1, 0)):
    e.set_tuning(23, v)
    e.set_tuning(27, sm)
    e.set_tuning(28, pm)
    _, cnt = e.cost_matrix(fetch_C=False)
    e.profile_reset(); e.profile_enable(True)
    for _ in range(6):
        e.cost_matrix(fetch_C=False, fetch_counts=False)
    e.synchronize(); n, ms = e.profile_get(6); e.profile_enable(False)
    if ref is None: ref = cnt
    print(f"key 23 = {v:3d} key 27 (slice-major) = {sm} key 28 (near pairs of several models batched) = {pm}: {ms / n:.4f} ms  = {(4.0 * N * M) / (ms / n) / 1e6 / 8000:.4f} of the HBM peak; counts {'equal' if np.array_equal(cnt, ref) else 'DIFFERENT'}", flush=True)
e.close()
