#!/usr/bin/env python3
"""How the resident residual sweep's time depends on the number of work items (tuning library, key 26: slice-major with this
many point slices; 0 = the product's ~37 500 items, model block fastest).  The last items of a launch end one item's duration
apart: is that tail worth smaller items?  python multi-h_amd/build.py --tuning; MH_LIB=multi-h_amd/libmultih_hip_tuning.so"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, M = int(os.environ.get("N", 50000)), int(os.environ.get("M", 100000))
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
thr2 = 2.2 ** 2
def run(name, reps=10):
    f = lambda: e.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
    f(); f(); e.synchronize(); e.profile_reset(); e.profile_enable(True)
    for _ in range(reps): f()
    e.synchronize(); n, ms = e.profile_get(1); e.profile_enable(False)
    ms /= max(n, 1)
    print(f"{name:40s} {ms:8.3f} ms   {8.0 * N * M / ms / 1e6:8.1f} GB/s", flush=True)
if os.environ.get("MODELS_PER_ITEM"):                 # variants 50 / 51 / 52 of the tuning library: the product sweep with 16 / 32 / 64 models per item
    for rep in range(int(os.environ.get("REPS", 3))):
        for v, mc in ((50, 16), (51, 32), (52, 64)):
            e.set_tuning(0, v)
            run(f"{mc} models per work item")
    sys.exit(0)
for rep in range(2):
    for s in [int(x) for x in os.environ.get("SLICES", "0,3,6,12,24,49").split(",")]:
        e.set_tuning(26, s)
        run(f"slices {s} ({'product order' if s == 0 else str((M + 15) // 16 * s) + ' items, slice-major'})")
