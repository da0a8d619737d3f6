#!/usr/bin/env python3
"""Writes the problems of tools/dirty_label_study.cpp (CPU only, through the oracle): a many-label set on a synthetic scene —
(a) true planes + perturbed copies + 4-point fits, as tools/batch_probe.py; (b) the reference's own initial models
(EstablishStablePointSets through the oracle) — with the data costs of the oracle.
   python tools/dirty_label_study.py OUTDIR [N] [K]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
synth = importlib.import_module("multi-h_amd.synth")
import oracle_lib as O
out = sys.argv[1]
N, K = int(sys.argv[2]) if len(sys.argv) > 2 else 8000, int(sys.argv[3]) if len(sys.argv) > 3 else 4
EXTRA = int(os.environ.get("EXTRA", 200))
sc = synth.make_scene(N, K, seed=1234)
THR2, LAM = 2.2 ** 2, 0.5


def dump(name, H):
    cost = O.data_cost(sc.src, sc.dst, H, LAM, THR2)
    hdr = np.array([sc.n, cost.shape[1], O.potts(LAM), sc.hit_col.size], np.int32)
    with open(os.path.join(out, name), "wb") as f:
        f.write(hdr.tobytes()); f.write(np.ascontiguousarray(cost, np.int32).tobytes())
        f.write(np.ascontiguousarray(sc.hit_rowptr, np.int32).tobytes()); f.write(np.ascontiguousarray(sc.hit_col, np.int32).tobytes())
    print(name, "sites", sc.n, "labels", cost.shape[1])


rng = np.random.default_rng(1)
idx = O.sample4(7, 0, EXTRA, sc.n)
Hd = O.dlt4(sc.src, sc.dst, idx)
Hd = Hd[0] if isinstance(Hd, tuple) else Hd
H = np.concatenate([sc.H_true, sc.H_true[rng.integers(0, K, EXTRA // 3)] * (1 + rng.normal(0, 3e-3, (EXTRA // 3, 9))), Hd.reshape(-1, 9)])
dump("many_labels.bin", np.ascontiguousarray(H))
if os.environ.get("STABLE", "1") == "1":
    r = O.establish_stable_point_sets(sc.src, sc.dst, sc.aff, sc.F, sc.e2, 0.005, 2.2, 1234)
    Hs = r[0] if isinstance(r, tuple) else r
    dump("stable_sets.bin", np.ascontiguousarray(np.asarray(Hs).reshape(-1, 9)))
