#!/usr/bin/env python3
"""r06: per-move solver times (mh_set_tuning key 8) of the SAME moves, one after the other against 8 per batch, on the many-label
case of tools/batch_probe.py: does a move take longer inside the solver when seven others are solved beside it, and how does a
batch's solver launch compare with the longest of its moves?"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
NL, EXTRA, CTX = int(os.environ.get("NL", 20000)), int(os.environ.get("EXTRA", 400)), int(os.environ.get("CTX", 8))
K = 6
sc = mh.synth.make_scene(NL, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
e.propose_dlt4(7, 0, EXTRA)
rng = np.random.default_rng(1)
H = np.ascontiguousarray(np.concatenate([sc.H_true, sc.H_true[rng.integers(0, K, EXTRA // 3)] * (1 + rng.normal(0, 3e-3, (EXTRA // 3, 9))), e.get_models()]))
L = H.shape[0] + 1
e.set_tuning(8, 4 * L)
out = {}
for ctx in (1, CTX):
    e.set_tuning(37, ctx)
    for _ in range(2):
        e.set_models(H)
        t0 = time.perf_counter()
        lab, en, cyc = e.labeling_step(False, np.full(sc.n, -1, np.int32))
        ms = (time.perf_counter() - t0) * 1e3
    out[ctx] = (ms, e.expand_trace(4 * L).copy(), e.expand_stats(), e.expand_batch_stats())
seq, bat = out[1][1], out[CTX][1]
both = (seq[:, 1] > 0) & (bat[:, 1] > 0)
print(f"{L} labels, {sc.n} sites: sequential {out[1][0]:.1f} ms, {CTX} per batch {out[CTX][0]:.1f} ms; moves with a solver run in both forms: {int(both.sum())} "
      f"(sequential {int((seq[:, 1] > 0).sum())}, batched {int((bat[:, 1] > 0).sum())})")
same_core = both & (seq[:, 0] == bat[:, 0])
print(f"  of them with the same core size in both forms: {int(same_core.sum())}")
for lo, hi in ((1, 64), (65, 1024), (1025, 4096), (4097, 1 << 30)):
    m = same_core & (seq[:, 0] >= lo) & (seq[:, 0] <= hi)
    if m.any():
        print(f"  cores of {lo:5d}..{min(hi, 99999):5d} sites: {int(m.sum()):4d} moves; inside the solver {seq[m, 6].mean() / 100:7.1f} us alone, {bat[m, 6].mean() / 100:7.1f} us beside others "
              f"(at barriers {seq[m, 7].mean() / 100:6.1f} / {bat[m, 7].mean() / 100:6.1f} us; barriers {seq[m, 5].mean():5.1f} / {bat[m, 5].mean():5.1f}; workgroups {seq[m, 1].mean():5.1f} / {bat[m, 1].mean():5.1f})")
print(f"  solver time summed over the moves: alone {seq[seq[:, 1] > 0, 6].sum() / 1e5:.1f} ms, beside others {bat[bat[:, 1] > 0, 6].sum() / 1e5:.1f} ms")
# a batch's solver launch lasts as long as its longest move: sum over batches of the max (moves t in [b*CTX, (b+1)*CTX) is only approximately a batch: failed validations shift the boundaries)
T = bat.shape[0]
mx = sum(bat[i:i + CTX, 6].max() for i in range(0, T, CTX)) / 1e5
print(f"  sum over aligned groups of {CTX} moves of the longest solver time: {mx:.1f} ms")
print("batch stats:", out[CTX][3])
e.close()
