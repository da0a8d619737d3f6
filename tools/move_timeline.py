#!/usr/bin/env python3
"""Relabel-by-relabel timeline of a few moves of one LabelingStep at 50k sites / 11 labels (mh_set_tuning keys 8 and 9): sites
still holding excess after each exact relabel, the relabel's depth, microseconds since the launch began."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
mh = importlib.import_module("multi-h_amd")
N, K = 50000, 10
sc = mh.synth.make_scene(N, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
e.set_tuning(8, 64)
for mv in (3, 5, 14, 16, 25):
    e.set_tuning(9, mv)
    e.set_models(H)
    lab, en, cyc = e.labeling_step(False, np.full(N, -1, np.int32))
    tr = e.expand_trace(64 + 200)
    row = tr[mv]
    print(f"move {mv}: core {row[0]} wgs {row[1]} relabels {row[2]} intervals {row[3]} push phases {row[4]} barriers {row[5]} us {row[6]/100:.0f} barrier us {row[7]/100:.0f}")
    d = tr[64:].reshape(-1, 4)
    prev = 0
    for i in range(row[2]):
        nact, hmax, iv, ticks = d[i]
        print(f"    relabel {i}: active {nact} hmax {hmax} intervals so far {iv} t={ticks/100:.0f} us (+{(ticks-prev)/100:.0f})")
        prev = ticks
