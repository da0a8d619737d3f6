#!/usr/bin/env python3
"""Times the propose stage of Process() at BASELINE configs[2] size: 100 000 DLT hypotheses on 50 000 correspondences,
then the greedy selection of up to 32 models on the device (mh_select_greedy).  Diagnostic."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K, M = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10)), int(os.environ.get("M", 100000))
sc = mh.synth.make_scene(N, K, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
for rep in range(3):
    t0 = time.time(); e.propose_dlt4(1234, rep * M, M); e.synchronize(); t1 = time.time()
    H, counters, counts, _ = e.select_greedy(2.2 ** 2, 20, 32)
    t2 = time.time()
    print(f"propose {1e3 * (t1 - t0):.2f} ms, greedy selection {1e3 * (t2 - t1):.2f} ms -> {len(counts)} models, counts {counts.tolist()[:12]}...", flush=True)
