#!/usr/bin/env python3
"""Fused score pass, 50k points x 100k DLT hypotheses: FP32 pre-test (csrc/score32.hip) vs the FP64 sweep; then the propose +
greedy selection stage that uses it."""
import os as _os
# r05: these schedule variants live in the measurement library only (python multi-h_amd/build.py --tuning)
_os.environ.setdefault("MH_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multi-h_amd", "libmultih_hip_tuning.so"))
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, M = int(os.environ.get("N", 50000)), int(os.environ.get("M", 100000))
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
for tiling in [int(x) for x in os.environ.get("TILINGS", "0,1,3,7,9,12,13").split(",")]:
    e.set_tuning(15, 1); e.set_tuning(16, tiling)
    e.score(2.2 ** 2, fetch=False); e.synchronize()
    e.profile_reset(); e.profile_enable(True)
    for _ in range(10): e.score(2.2 ** 2, fetch=False)
    e.synchronize(); n, ms = e.profile_get(2); e.profile_enable(False)
    print(f"tiling {tiling}: {ms / n:7.3f} ms", flush=True)
e.set_tuning(16, 0)
res = {}
for mode in (1, 0, 1, 0):
    e.set_tuning(15, mode)
    e.score(2.2 ** 2, fetch=False); e.synchronize(); e.score_stats(reset=True)
    e.profile_reset(); e.profile_enable(True)
    for _ in range(10): e.score(2.2 ** 2, fetch=False)
    e.synchronize(); n, ms = e.profile_get(2); e.profile_enable(False)
    pairs, p64 = e.score_stats(reset=True)
    print(f"{'FP32 pre-test' if mode else 'FP64 sweep   '} {ms / n:7.3f} ms per {N} x {M} score pass = {M / (ms / n) * 1e3:.3e} hypotheses/s"
          + (f"   ({p64} of {pairs} pairs decided in FP64 = {p64 / max(pairs, 1):.2e})" if mode else ""), flush=True)
    res[mode] = e.score(2.2 ** 2)
print("counts equal:", bool(np.array_equal(res[0], res[1])))
for mode in (1, 0):
    e.set_tuning(15, mode)
    for seed in (7, 8, 9):
        t0 = time.time(); e.propose_dlt4(seed, 0, M); H, cnt, cts, _ = e.select_greedy(2.2 ** 2, 20, 32); e.synchronize()
        print(f"{'FP32 pre-test' if mode else 'FP64 sweep   '} propose + greedy selection {time.time() - t0:.4f} s, {len(cnt)} models", flush=True)
