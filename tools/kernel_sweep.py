#!/usr/bin/env python3
"""Times the residual / score kernel variants on one GPU (HIP events through the C ABI).  The variants exist only in the
measurement library: python multi-h_amd/build.py --tuning; MH_LIB=multi-h_amd/libmultih_hip_tuning.so python tools/kernel_sweep.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, M = int(os.environ.get("N", 50000)), int(os.environ.get("M", 100000))
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
thr2 = 2.2 ** 2
bytes_ = 8.0 * N * M
def run(name, fn, kid, reps=6):
    fn(); e.synchronize(); e.profile_reset(); e.profile_enable(True)
    for _ in range(reps): fn()
    e.synchronize(); n, ms = e.profile_get(kid); e.profile_enable(False)
    ms /= max(n, 1)
    print(f"{name:36s} {ms:8.3f} ms   {bytes_/ms/1e6:8.1f} GB/s-equivalent   {N*M/ms/1e6:8.1f} Gpair/s", flush=True)
for v in [int(x) for x in os.environ.get("RV", "0,1,2,3,4,5,6,7").split(",")]:
    e.set_tuning(0, v)
    run(f"residual variant {v}", lambda: e.residual_matrix(thr2, fetch_R=False, fetch_counts=False), 1)
for v in [0, 1, 3]:
    e.set_tuning(1, v)
    run(f"score variant {v}", lambda: e.score(thr2, fetch=False), 2)
