#!/usr/bin/env python3
"""Time of mh_select_greedy on one scene: 100 000 DLT hypotheses, refitted winners as class MultiH sets them (key 30), with the
decremental rounds of r05 (key 36 = 1: a round counts its candidates on the points the last claim took away and subtracts)
and without (0: it counts them again on the packed support set).  The selection must be the same."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, P, M = int(os.environ.get("N", 50000)), int(os.environ.get("PLANES", 10)), int(os.environ.get("M", 100000))
sc = mh.synth.make_scene(N, P, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2)
thr2 = 2.2 ** 2
ref = None
for refit in (1, 0):
    e.set_tuning(30, refit)
    for dec in (0, 1, 0, 1):
        e.set_tuning(36, dec)
        best = 1e9
        for _ in range(3):
            e.propose_dlt4(1234, 0, M)
            e.synchronize()
            t0 = time.perf_counter()
            H, counters, counts, _ = e.select_greedy(thr2, max(8, N // 200), 64)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        key = (refit, H.tobytes(), counters.tobytes(), counts.tobytes())
        if dec == 0 and (ref is None or ref[0] != refit): ref = key
        assert key == ref, "the decremental rounds selected something else"
        print(f"N {N}, {M} hypotheses, refit {refit}, decremental rounds {dec}: {len(H)} models selected, counts {counts[:12].tolist()} ..., {best:7.2f} ms", flush=True)
e.close()
