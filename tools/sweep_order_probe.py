#!/usr/bin/env python3
"""Experiment (r04): the resident residual sweep taking its items SLICE-major (mh_set_tuning key 26 = number of point
slices; consecutive items are the slices of one model block, so the workgroups at work write a compact window of R)
against the launcher's rule (key 26 = 0: model block fastest, ~37 500 items).  Same counts, same sample rows."""
import os as _os
# r05: these schedule variants live in the measurement library only (python multi-h_amd/build.py --tuning)
_os.environ.setdefault("MH_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multi-h_amd", "libmultih_hip_tuning.so"))
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N = 50000
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
thr2 = 2.2 ** 2
for M in [int(v) for v in os.environ.get("SIZES", "100000,12500").split(",")]:
    e.propose_dlt4(1234, 0, M)
    ref = rows = None
    for v in [int(x) for x in os.environ.get("SLICES", "0,6,12,16,24,32,49,0").split(",")]:
        e.set_tuning(26, v)
        _, cnt = e.residual_matrix(thr2, fetch_R=False)
        r = e.get_residual_rows(M // 3, 2)
        e.profile_reset(); e.profile_enable(True)
        for _ in range(int(os.environ.get("REPS", "12"))):
            e.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
        e.synchronize(); n, ms = e.profile_get(1); e.profile_enable(False)
        if ref is None: ref, rows = cnt, r
        ok = np.array_equal(cnt, ref) and np.array_equal(r.view(np.uint64), rows.view(np.uint64))
        print(f"M = {M:6d}  key 26 = {v:3d}: {ms / n:.4f} ms  = {(8.0 * N * M) / (ms / n) / 1e6 / 8000:.4f} of the HBM peak; counts and sample rows {'equal' if ok else 'DIFFERENT'}", flush=True)
e.set_tuning(26, 0)
e.close()
