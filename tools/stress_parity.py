#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress: many seeds/sizes for the alpha-expansion (labels, energies),
residual matrix and DLT.  Exits non-zero on the first mismatch.  Run on the GPU box."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mh = importlib.import_module("multi-h_amd")
import oracle_lib as O
budget = float(os.environ.get("SECONDS", 120))
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
t0 = time.time(); runs = 0
while time.time() - t0 < budget:
    n = int(rng.integers(50, 6000)); k = int(rng.integers(1, 8)); seed = int(rng.integers(0, 1 << 30))
    knn = int(rng.integers(2, 24)); sym = bool(rng.integers(0, 2)); lam = float(rng.choice([0.1, 0.5, 1.0, 3.0]))
    thr = float(rng.choice([1.0, 2.2, 4.0]))
    sc = mh.synth.make_scene(n, k, seed=seed, knn=min(knn, n - 1), symmetric=sym, noise=float(rng.uniform(0.1, 2.0)),
                             outlier_frac=float(rng.uniform(0, 0.6)))
    if n <= 2500 and rng.integers(0, 4) == 0:
        # the reference's own rule — every correspondence within a radius — gives degrees far above the 48 arcs a solver
        # row keeps in registers: the arcs-in-memory path of k_solve
        pv = np.concatenate([sc.src, sc.dst], axis=1).astype(np.float32)
        sq = (pv[:, None, :] - pv[None, :, :]) ** 2
        hit = ((sq[..., 0] + sq[..., 1]) + sq[..., 2]) + sq[..., 3] <= np.float32(rng.uniform(80.0, 400.0)) ** 2
        sc.hit_rowptr = np.concatenate([[0], np.cumsum(hit.sum(axis=1))]).astype(np.int32)
        sc.hit_col = np.nonzero(hit)[1].astype(np.int32)
        lam = float(rng.choice([0.01, 0.05, 0.5]))
    e.set_params(2.6, thr, 0.005, lam, 20)
    e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
    e.set_tuning(5, int(rng.choice([256, 256, 32, 8, 2])))    # solver workgroups: few of them give every solver row several sites
    e.set_tuning(11, int(rng.integers(0, 2)))                 # flow recycling on / off
    e.set_tuning(17, int(rng.choice([0, 1, 2, 2, 4])))        # passes of the dominance cascade inside the solver launch
    e.set_tuning(37, int(rng.choice([1, 2, 3, 5, 8, 16, 16])))     # r06: alpha-moves solved together ...
    e.set_tuning(38, int(rng.choice([0, 0, 4, 16])))               # ... from the first cycle on for label sets of at least this size
    e.set_tuning(39, int(rng.choice([0, 16, 32, 64])))             # ... sites per wave in a batch's setup / reduction launches
    extra = int(rng.integers(0, 6)) if rng.integers(0, 3) else int(rng.integers(6, 60))      # (r06: now and then dozens of labels — the batches' case)
    H = np.concatenate([sc.H_true] + [sc.H_true[rng.integers(0, k)][None] * (1 + rng.normal(0, 3e-4, (1, 9))) for _ in range(extra)])
    e.set_models(H)
    cost = e.data_cost()
    assert np.array_equal(cost, O.data_cost(sc.src, sc.dst, H, lam, thr * thr)), ("cost", n, k, seed)
    init = None if rng.integers(0, 2) else rng.integers(0, H.shape[0] + 1, size=n).astype(np.int32)
    lab, en, cyc = e.expand(init)
    lab_o, en_o, cyc_o, _ = O.expand(cost, sc.hit_rowptr, sc.hit_col, O.potts(lam), init_labels=init)
    if not (en == en_o and cyc == cyc_o and np.array_equal(lab, lab_o)):
        print("MISMATCH expand", dict(n=n, k=k, seed=seed, knn=knn, sym=sym, lam=lam, thr=thr, en=en, en_o=en_o, cyc=cyc, cyc_o=cyc_o,
                                      diff=int((lab != lab_o).sum())))
        sys.exit(1)
    m = int(rng.integers(1, 200))
    e.propose_dlt4(seed, int(rng.integers(0, 1000)), m)
    Hm = e.get_models(); idx = e.get_samples()
    Ho, wit, _ = O.dlt4(sc.src, sc.dst, idx)
    good = wit > 1e-6
    assert np.max(np.abs(Hm[good] - Ho[good]), initial=0) <= 1e-6, ("dlt", n, seed)
    with np.errstate(all="ignore"):
        R, cnt = e.residual_matrix(thr * thr); Rr = O.residual_matrix(sc.src, sc.dst, Hm)
    nan = np.isnan(Rr)
    assert np.array_equal(np.isnan(R), nan) and np.array_equal(R[~nan].view(np.uint64), Rr[~nan].view(np.uint64)), ("R", n, seed)
    runs += 1
print(f"stress ok: {runs} random problems in {time.time()-t0:.0f} s")
