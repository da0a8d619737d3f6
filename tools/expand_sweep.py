#!/usr/bin/env python3
"""Sweeps the alpha-expansion solver's schedule knobs (relaxation rounds per barrier interval, push cycles per
phase, push phases per global relabel, solver workgroups) on one labeling problem; prints ms per expansion and
where the solver launches spend their time.  Diagnostic."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10))
sc = mh.synth.make_scene(N, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
ref = None
TRACE = int(os.environ.get("TRACE", "0"))
if "RECYCLE" in os.environ: e.set_tuning(11, int(os.environ["RECYCLE"]))
if TRACE: e.set_tuning(8, 64); e.set_tuning(9, int(os.environ.get("DETAIL", "20")))
CFGS = [(128, 256, 2, 256, 2), (128, 256, 1, 256, 2), (128, 256, 1, 256, 3), (128, 256, 1, 256, 4), (128, 256, 2, 256, 1), (128, 256, 2, 256, 3),
        (128, 256, 3, 256, 2), (128, 256, 2, 256, 4), (128, 256, 1, 256, 6), (128, 256, 2, 256, 2)]
if os.environ.get("CFGS"): CFGS = [tuple(int(x) for x in c.split(',')) for c in os.environ["CFGS"].split(';')]
for cfg in CFGS:
    for k, v in enumerate(cfg[:4]): e.set_tuning(2 + k, v)
    if len(cfg) > 4: e.set_tuning(10, cfg[4])
    e.set_models(H); e.data_cost(fetch=False)
    e.expand()
    t0 = time.time(); lab, en, cyc = e.expand(); dt = time.time() - t0
    if ref is None: ref = (lab.copy(), en)
    ok = np.array_equal(lab, ref[0]) and en == ref[1]
    st = e.expand_stats()
    print(f"cfg {cfg}: {dt*1e3:7.1f} ms ok={ok} solve={st['solve_us']/1e3:.1f} ms (barriers {st['barrier_us']/1e3:.1f}, relabel {st['relax_us']/1e3:.1f}, "
          f"push {st['push_us']/1e3:.1f}) relabels={st['relabels']} relax_int={st['relax_intervals']} push_ph={st['push_phases']} "
          f"barriers={st['barriers']} moves_run={st['moves_run']} solved={st['moves_solved']} core={st['core_sites']}/{st['core_max']}", flush=True)
    if TRACE:
        tr = e.expand_trace(64 + 1024)
        det = tr[64:].reshape(-1, 4)
        prev = 0
        for i, d in enumerate(det):
            if d[3] == 0: break
            print(f"      relabel {i:3d}: active {d[0]:5d} hmax {d[1]:4d} intervals {d[2]:4d} t {d[3] / 100:8.1f} us (+{(d[3] - prev) / 100:6.1f})"); prev = d[3]
        for t, row in enumerate(tr[:64]):
            if row[0]: print(f"   move {t:2d} alpha {t % (H.shape[0] + 1):2d}: core {row[0]:6d} wgs {row[1]:3d} relabels {row[2]:3d} intervals {row[3]:3d} push {row[4]:3d} "
                             f"barriers {row[5]:4d} us {row[6] / 100:8.1f} in-barrier {row[7] / 100:8.1f}")
        TRACE = 0
