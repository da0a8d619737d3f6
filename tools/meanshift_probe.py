#!/usr/bin/env python3
"""Time of mh_mean_shift on the 10-D features of the reference-style initialisation (EstablishStablePointSets,
M/MultiH.cpp:604-694) for scenes of several sizes; MH_LIB selects the library, so two builds can be compared on one box.
Prints modes, a hash of the assignment (the result must not depend on the build) and the best of three calls."""
import hashlib, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
print("library:", mh.LIB_PATH)
SIZES = {500: 2, 2000: 3, 5000: 3, 20000: 6, 50000: 10}
for n, planes in ((int(v), SIZES[int(v)]) for v in os.environ.get("SIZES", "500,2000,5000,20000,50000").split(",")):
    sc = mh.synth.make_scene(n, planes, seed=1234, with_neighbours=False)
    e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
    e.set_correspondences(sc.src, sc.dst, sc.aff)
    e.set_epipolar(sc.F, sc.e2)
    _, feat = e.local_homographies(0.005)
    feat = np.where(np.isfinite(feat), feat, 1e300)
    if os.environ.get("MS_ITERS"):
        e.set_tuning(7, int(os.environ["MS_ITERS"]))
    if os.environ.get("MS_PERSIST"):                       # key 29: 0 = a launch per iteration throughout (the r04 schedule)
        e.set_tuning(29, int(os.environ["MS_PERSIST"]))
    if os.environ.get("MS_INDEXED"):                       # key 32: 0 = the launched / persistent schedule (no index)
        e.set_tuning(32, int(os.environ["MS_INDEXED"]))
    if os.environ.get("MS_DENSE"):                         # key 33: members per iteration beyond which an indexed climb is handed on
        e.set_tuning(33, int(os.environ["MS_DENSE"]))
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        modes, assign, k = e.mean_shift(feat, 2.2, 77)
        best = min(best, time.perf_counter() - t)
    print(f"N = {n:6d}: {k:5d} modes, assignment {hashlib.sha256(assign.tobytes()).hexdigest()[:12]}, {best * 1e3:8.2f} ms", flush=True)
    e.close()
