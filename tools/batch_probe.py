#!/usr/bin/env python3
"""r06: the concurrent alpha-moves (mh_set_tuning key 37, csrc/expand.hip k_commit) against the sequential form (key 37 = 1):
one LabelingStep each on (a) the bench's separated scene, (b) the r04 scene (cores of thousands of sites), (c) a label set of
hundreds (the reference's own route hands the loop several hundred stable-set models: true planes + perturbed copies + DLT
hypotheses, as tools/many_label_moves.py), with the batch statistics — batches, moves kept out of batches, failed validations,
moves the host never launched — and labels / energy / cycle count compared between the two forms.
Env: N (50000), NL (20000: points of the many-label case), EXTRA (400), CTX (8), MINL (16), REPS (3)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
N, NL, EXTRA = int(os.environ.get("N", 50000)), int(os.environ.get("NL", 20000)), int(os.environ.get("EXTRA", 400))
CTX, MINL, REPS = int(os.environ.get("CTX", 8)), int(os.environ.get("MINL", 16)), int(os.environ.get("REPS", 3))


def step(e, H, n, ctx, warm_labels=None):
    e.set_tuning(37, ctx)
    e.set_tuning(38, MINL)
    best = None
    for _ in range(REPS):
        e.set_models(H)
        t0 = time.perf_counter()
        lab, en, cyc = e.labeling_step(warm_labels is not None, np.full(n, -1, np.int32) if warm_labels is None else warm_labels)
        ms = (time.perf_counter() - t0) * 1e3
        if best is None or ms < best[0]:
            best = (ms, lab, en, cyc, e.expand_stats(), e.expand_batch_stats())
    return best


def case(name, sc, H):
    e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
    e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
    seq = step(e, H, sc.n, 1)
    bat = step(e, H, sc.n, CTX)
    same = bool(np.array_equal(seq[1], bat[1]) and seq[2] == bat[2] and seq[3] == bat[3])
    b = bat[5]
    print(f"{name}: {sc.n} sites, {H.shape[0] + 1} labels, {seq[3]} cycles, energy {int(seq[2])}: sequential {seq[0]:.2f} ms ({seq[4]['launches']} launches, "
          f"{seq[4]['moves_solved']} moves solved) | {CTX} moves per batch {bat[0]:.2f} ms ({bat[4]['launches']} launches): {b['batches']} batches, "
          f"{b['batch_committed']} moves kept out of them, {b['batch_invalid']} batches cut short by a failed test, {b['solo_moves']} moves alone, {b['host_skipped']} never launched "
          f"| labels, energy, cycles {'EQUAL' if same else 'DIFFERENT'}", flush=True)
    # a warm second step (what the loop's later iterations are): from the labels of the first
    H2 = e.get_models()
    seq2 = step(e, H2, sc.n, 1, warm_labels=seq[1])
    bat2 = step(e, H2, sc.n, CTX, warm_labels=seq[1])
    same2 = bool(np.array_equal(seq2[1], bat2[1]) and seq2[2] == bat2[2] and seq2[3] == bat2[3])
    b = bat2[5]
    print(f"   warm step from those labels (re-estimated models): {seq2[3]} cycles: sequential {seq2[0]:.2f} ms | batched {bat2[0]:.2f} ms: {b['batches']} batches, "
          f"{b['batch_committed']} kept, {b['batch_invalid']} failed, {b['solo_moves']} alone, {b['host_skipped']} never launched | {'EQUAL' if same2 else 'DIFFERENT'}", flush=True)
    e.close()
    return same and same2


ok = True
CASES = os.environ.get("CASES", "separated,r04,many").split(",")
rng = np.random.default_rng(0)
if "separated" in CASES:
    sc = mh.synth.make_scene(N, 10, seed=1234)
    ok &= case("separated scene", sc, sc.H_true * (1.0 + rng.normal(0, 1e-4, size=sc.H_true.shape)))
if "r04" in CASES:
    sc = mh.synth.make_scene(N, 10, seed=1234, legacy_r04=True)
    ok &= case("r04 scene", sc, sc.H_true * (1.0 + rng.normal(0, 1e-4, size=sc.H_true.shape)))
if "many" not in CASES:
    print("BATCH PROBE:", "EQUAL" if ok else "DIFFERENT")
    sys.exit(0 if ok else 1)
K = 6
sc = mh.synth.make_scene(NL, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2)
e.propose_dlt4(7, 0, EXTRA)
rng = np.random.default_rng(1)
H = np.concatenate([sc.H_true, sc.H_true[rng.integers(0, K, EXTRA // 3)] * (1 + rng.normal(0, 3e-3, (EXTRA // 3, 9))), e.get_models()])
e.close()
ok &= case("many labels", sc, np.ascontiguousarray(H))
print("BATCH PROBE:", "EQUAL" if ok else "DIFFERENT")
sys.exit(0 if ok else 1)
