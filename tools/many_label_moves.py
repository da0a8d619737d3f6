#!/usr/bin/env python3
"""The alpha-expansion with HUNDREDS of labels (the reference's own route starts the loop with several hundred stable-set
models): one LabelingStep at N points from the models of mh_local_homographies + mean shift + ... is expensive to set up, so
the label set here is K true planes plus EXTRA perturbed copies and DLT hypotheses — the per-move statistics of the
solver launches by core size (mh_set_tuning key 8): how many moves, microseconds inside k_solve, barriers, relabels."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, K, EXTRA = int(os.environ.get("N", 20000)), int(os.environ.get("K", 6)), int(os.environ.get("EXTRA", 300))
sc = mh.synth.make_scene(N, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
e.propose_dlt4(7, 0, EXTRA)
rng = np.random.default_rng(1)
H = np.concatenate([sc.H_true, sc.H_true[rng.integers(0, K, EXTRA // 3)] * (1 + rng.normal(0, 3e-3, (EXTRA // 3, 9))), e.get_models()])
L = H.shape[0] + 1
e.set_tuning(8, 4 * L)
e.set_models(H)
e.labeling_step(False, np.full(N, -1, np.int32))
e.set_models(H)
t0 = time.perf_counter()
lab, en, cyc = e.labeling_step(False, np.full(N, -1, np.int32))
ms = (time.perf_counter() - t0) * 1e3
st = e.expand_stats()
tr = e.expand_trace(4 * L)
rows = tr[tr[:, 1] > 0]
print(f"N {N}, {L} labels: LabelingStep {ms:.1f} ms, {cyc} cycles, {st['moves']} moves of which {len(rows)} launched the solver on a non-empty core; "
      f"inside the solver {rows[:, 6].sum() / 100 / 1e3:.1f} ms")
for lo, hi in ((1, 64), (65, 256), (257, 1024), (1025, 4096), (4097, 1 << 30)):
    r = rows[(rows[:, 0] >= lo) & (rows[:, 0] <= hi)]
    if len(r):
        print(f"  cores of {lo:5d}..{min(hi, 999999):6d} sites: {len(r):5d} moves, workgroups {r[:, 1].mean():6.1f}, {r[:, 6].mean() / 100:7.1f} us inside the solver "
              f"(of which at barriers {r[:, 7].mean() / 100:6.1f}), relabels {r[:, 2].mean():4.1f}, intervals {r[:, 3].mean():5.1f}, push phases {r[:, 4].mean():4.1f}, barriers {r[:, 5].mean():5.1f}")
e.close()
