#!/usr/bin/env python3
"""Experiment (r04): the DLT proposer with the nine columns of W in registers, handed round with DPP row shifts (mh_set_tuning
key 25 = 2; the default for mh_propose_dlt4) against the LDS-staged form of r01-r04 (key 25 = 1; the default for mh_prefetch_dlt4).  Same bits (checked on all M models), time per
launch of M hypotheses alone on the device."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N = 50000
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
SLOT = int(os.environ.get("SLOT", "0"))          # profile slot of MH_K_DLT4
for M in (100000, 12500):
    ref = None
    for v in (1, 2, 1, 2):
        e.set_tuning(25, v)
        e.propose_dlt4(1234, 0, M)
        H = e.get_models()
        e.profile_reset(); e.profile_enable(True)
        for _ in range(10):
            e.propose_dlt4(1234, 0, M)
        e.synchronize(); n, ms = e.profile_get(SLOT); e.profile_enable(False)
        if ref is None: ref = H
        same = np.array_equal(H.view(np.uint64), ref.view(np.uint64))
        print(f"M = {M:6d}  key 25 = {v} ({'registers + DPP' if v == 2 else 'LDS'}): {ms / n:.4f} ms per launch, models {'bit-identical' if same else 'DIFFERENT'}", flush=True)
e.close()
