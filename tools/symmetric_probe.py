#!/usr/bin/env python3
"""Time of the symmetric-transfer residual sweep (MH_RESIDUAL_SYMMETRIC, north_star's extension) at 50k x 100k, with a hash of
the counts and of two rows (a change of schedule must not change a bit)."""
import hashlib, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, M = 50000, 100000
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
e.set_residual_mode(True)
thr2 = 2.2 ** 2
_, cnt = e.residual_matrix(thr2, fetch_R=False)
rows = e.get_residual_rows(M // 3, 2)
e.profile_reset(); e.profile_enable(True)
for _ in range(10):
    e.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
e.synchronize(); n, ms = e.profile_get(1); e.profile_enable(False)
print(f"symmetric sweep: {ms / n:.3f} ms = {8.0 * N * M / (ms / n) / 1e6 / 8000:.3f} of the HBM peak; counts {hashlib.sha256(cnt.tobytes()).hexdigest()[:12]} rows {hashlib.sha256(rows.tobytes()).hexdigest()[:12]}")
e.close()
