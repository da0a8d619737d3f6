#!/usr/bin/env python3
"""Row pitch of the residual matrix vs kernel time (tuning library; key 13).  Does the DRAM mapping care?"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, M = int(os.environ.get("N", 50000)), int(os.environ.get("M", 100000))
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
thr2 = 2.2 ** 2
for variant in [int(x) for x in os.environ.get("RV", "0,7").split(",")]:
    e.set_tuning(0, variant)
    for ld in [0, 50048, 50176, 50688, 51200, 52224, 53248, 57344, 65536]:
        e.set_tuning(13, ld)
        f = lambda: e.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
        f(); e.synchronize(); e.profile_reset(); e.profile_enable(True)
        for _ in range(8): f()
        e.synchronize(); n, ms = e.profile_get(1); e.profile_enable(False)
        ms /= max(n, 1)
        print(f"variant {variant:3d}  ld {ld or 50000:6d} ({(ld or 50000) * 8 / 4096:8.2f} x 4 KiB)  {ms:7.3f} ms  {8.0 * N * M / ms / 1e6:7.1f} GB/s", flush=True)
