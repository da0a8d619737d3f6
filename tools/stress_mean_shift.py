#!/usr/bin/env python3
"""Stress of mh_mean_shift (indexed climbs, launched rounds, the persistent tail: r05) against the oracle's restatement: random
sizes (now and then more rows than the definition has slots), dimensions 6 and 10 (the two with an indexed / persistent form)
and others, cluster densities from singletons to blobs that put hundreds of members into every group, one coordinate stretched
or squeezed (which coordinate the index bins), rows parked at 1e300, exact duplicates; the schedule keys drawn at random.
Modes bit for bit, assignments equal."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mh = importlib.import_module("multi-h_amd")
import oracle_lib as O
CASES = int(os.environ.get("CASES", 30))
rng = np.random.default_rng(int(os.environ.get("SEED", 3)))
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
bad = 0
for case in range(CASES):
    d = int(rng.choice([6, 10, 10, 6, 3, 12]))
    n = int(rng.integers(20, 5000))
    nc = int(rng.integers(1, 40))
    per = max(1, int(rng.integers(1, max(2, n // (2 * nc)))))
    spread = float(rng.uniform(0.05, 0.8))
    centres = rng.uniform(-80, 80, size=(nc, d))
    data = np.concatenate([c + rng.normal(0, spread, size=(per, d)) for c in centres])[:n]
    if len(data) < n:
        data = np.concatenate([data, rng.uniform(-80, 80, size=(n - len(data), d))])
    if case % 5 == 3:                                      # a blob: dense groups, the split and the dense walk of k_ms_indexed
        n = int(rng.choice([3000, 7000, 12000, 22000, 36000]))
        blob = int(n * rng.uniform(0.3, 0.95))
        data = np.concatenate([rng.uniform(-40, 40, size=(1, d)) + rng.normal(0, rng.uniform(0.01, 0.3), size=(blob, d)),
                               rng.uniform(-80, 80, size=(n - blob, d))])
        data = data[rng.permutation(n)]
    if case % 3 == 1:
        data[:, int(rng.integers(0, d))] *= float(rng.choice([1e-3, 30.0, 1e4]))        # the coordinate the index bins / must not bin
    if case % 4 == 0 and n > 50:
        data[5:15] = data[5]                               # duplicates
        data[20:24] = 1e300                                # parked rows
    bw = float(rng.choice([2.2, 1.0, 3.5]))
    persist, per_round = int(rng.choice([0, 1, 4, 12, 64])), int(rng.choice([1, 3, 6, 9]))
    indexed, dense = int(rng.choice([1, 1, 1, 0])), int(rng.choice([0, 8, 8, 40, 1 << 20]))
    e.set_tuning(29, persist); e.set_tuning(7, per_round); e.set_tuning(32, indexed); e.set_tuning(33, dense)
    seed = int(rng.integers(1, 10 ** 6))
    modes, assign, k = e.mean_shift(data, bw, seed)
    with np.errstate(all="ignore"):
        mo, ao, ko = O.mean_shift(data, bw, seed)
    ok = k == ko and np.array_equal(assign, ao) and np.array_equal(modes.view(np.uint64), mo.view(np.uint64))
    if not ok:
        bad += 1
        print(f"case {case}: MISMATCH n {n} d {d} clusters {nc} x {per} spread {spread:.2f} bw {bw} persist {persist} per-round {per_round} indexed {indexed} dense {dense}: modes {k} vs {ko}, "
              f"assignments differing {int((assign != ao).sum()) if assign.shape == ao.shape else 'shape'}", flush=True)
e.close()
print(f"stress_mean_shift: {CASES} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
