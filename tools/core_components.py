#!/usr/bin/env python3
"""VERDICT r03 item 6, the measurement: connected components of the undecided core of every move of a cold LabelingStep at
BASELINE configs[4] size (50 000 sites, 11 labels, the k = 16 neighbourhood) — would the moves' flow problems fall apart into
pieces that one workgroup each could solve without a grid barrier?  mh_set_tuning key 21 + mh_get_core_components
(csrc/expand.hip k_core_components; diagnostic only).  Env: N PLANES SEED."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K, SEED = int(os.environ.get("N", 50000)), int(os.environ.get("PLANES", 10)), int(os.environ.get("SEED", 1234))
sc = mh.synth.make_scene(N, K, seed=SEED)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.set_epipolar(sc.F, sc.e2)
e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
MOVES = 96
for label, init in (("cold (all sites start as outliers)", np.full(N, -1, np.int32)), ("warm (from the previous labeling)", None)):
    e.set_models(H)
    e.set_tuning(21, 0)
    if init is None:
        init = lab
        warm = True
    else:
        warm = False
    e.labeling_step(warm, init)                               # untimed: what the diagnostic costs is not the solver's time
    e.set_models(H)
    e.set_tuning(21, MOVES)
    lab, energy, cycles = e.labeling_step(warm, init)
    st = e.expand_stats()
    cc = e.core_components(MOVES)
    e.set_tuning(21, 0)
    live = cc[cc[:, 0] > 0]
    print(f"== {label}: {N} sites, {H.shape[0] + 1} labels, {cycles} cycles, {st['moves_solved']} moves with a core, energy {int(energy)}")
    print(f"{'move':>4} {'core':>6} {'comps':>6} {'largest':>8} {'2nd':>6} | sites in components of <=64 <=256 <=1024 <=2048 <=8192 | rounds")
    for t in range(MOVES):
        r = cc[t]
        if r[0] > 0:
            print(f"{t:4d} {r[0]:6d} {r[1]:6d} {r[2]:8d} {r[3]:6d} | {r[4]:6d} {r[5]:6d} {r[6]:6d} {r[7]:6d} {r[8]:6d} | {r[14]:4d}")
    tot = live[:, 0].sum()
    print(f"total core sites {tot}; in components of <= 64: {live[:, 4].sum() / tot:.3f}, <= 256: {live[:, 5].sum() / tot:.3f}, "
          f"<= 1024: {live[:, 6].sum() / tot:.3f}, <= 2048: {live[:, 7].sum() / tot:.3f}, <= 8192: {live[:, 8].sum() / tot:.3f}; "
          f"the largest component holds {live[:, 2].sum() / tot:.3f} of the core on average; components per move "
          f"{live[:, 1].mean():.1f} (median {np.median(live[:, 1]):.0f})")
e.close()
