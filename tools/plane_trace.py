#!/usr/bin/env python3
"""Where do the planes go?  A step-by-step trace of the merge <-> label alternation (M/MultiH.cpp:263-311) on a synthetic
scene WITH ground truth, through the ORACLE (tests/oracle_lib.py: the restatement of MergingStep / LabelingStep with the
reference's own GCO inside) on the CPU — no GPU, no product code.  After every MergingStep and every LabelingStep it says,
for each ground-truth plane, which model (if any) covers it — a model COVERS a plane when >= 80 % of the plane's inlier
correspondences lie within the inlier threshold of it — and, when a plane loses its model, which rule took it:

  merge:<planes>   mean shift on the 6-D feature (images of (0,0), (1,0), (0,1), M/MultiH.cpp:364-390; window
                   sum_j |delta_j| < thr^2, MeanShiftClustering.h:78-85) put the plane's model into one mode with the model of
                   ANOTHER plane; the 3-point model fitted to the mode (M/MultiH.cpp:407-409) covers neither or only one
  straight         the mode's model was dropped by the straightness / < 3 inliers test (M/MultiH.cpp:446-463)
  label            the model survived MergingStep but after alpha-expansion + HAF re-estimation (M/MultiH.cpp:513-602) it no
                   longer covers the plane (its points were taken by a neighbouring model, or re-estimation moved it)

Env: N PLANES SEED ITERS KNN; INIT=truth (perturbed ground truth + near-copies + strays, as tools/at_size_alternation.py)
or INIT=exact (the ground-truth homographies themselves); SCENE=r04 (the generator as it stood until round 4) to reproduce
the lost planes.  Prints a table and one JSON line; kept under profiles/."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mh = importlib.import_module("multi-h_amd")
import oracle_lib as O
from scipy.spatial import cKDTree

N, K, SEED, ITERS, KNN = (int(os.environ.get(k, d)) for k, d in (("N", 50000), ("PLANES", 10), ("SEED", 1234), ("ITERS", 20), ("KNN", 16)))
INIT = os.environ.get("INIT", "truth")
THR, LAM, LOC = 2.2, 0.5, 0.005
thr2 = THR * THR
kw = {"legacy_r04": True} if os.environ.get("SCENE") == "r04" else {}
sc = mh.synth.make_scene(N, K, seed=SEED, with_neighbours=False, **kw)
rng = np.random.default_rng(SEED)
if INIT == "exact":
    H = sc.H_true.copy()
else:
    H0 = [sc.H_true * (1.0 + rng.normal(0, 1e-4, size=sc.H_true.shape))]
    for _ in range(5):
        k = rng.integers(0, K); H0.append(sc.H_true[k:k + 1] * (1.0 + rng.normal(0, 2e-4, size=(1, 9))))
    for _ in range(2):
        H0.append((np.eye(3) + rng.normal(0, 0.05, size=(3, 3))).reshape(1, 9))
    H = np.ascontiguousarray(np.concatenate(H0))

# the class's default neighbourhood: the KNN nearest hits of the float32 (x1, y1, x2, y2) vectors inside 1 / locality
pv = np.concatenate([sc.src, sc.dst], axis=1).astype(np.float32).astype(np.float64)
dist, idx = cKDTree(pv).query(pv, k=KNN + 1)
rows = np.repeat(np.arange(N), KNN + 1); cols = idx.reshape(-1)
keep = (rows != cols) & (dist.reshape(-1) <= 1.0 / LOC)
rows, cols = rows[keep], cols[keep]
order = np.lexsort((cols, rows)); rows, cols = rows[order], cols[order]
rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=N))]).astype(np.int32)
col = cols.astype(np.int32)

gt = sc.gt_label
plane_pts = [np.flatnonzero(gt == p) for p in range(K)]


def coverage(Hs):
    """cover[p] = (best model, fraction of plane p's inliers inside thr of it)."""
    out = []
    for p in range(K):
        ip = plane_pts[p]
        if Hs.shape[0] == 0 or ip.size == 0:
            out.append((-1, 0.0)); continue
        with np.errstate(all="ignore"):
            R = O.residual_matrix(sc.src[ip], sc.dst[ip], Hs)          # [models, points]
        fr = (R < thr2).mean(axis=1)
        m = int(np.argmax(fr))
        out.append((m, float(fr[m])))
    return out


def planes_of(cov):
    return {p for p, (m, f) in enumerate(cov) if f >= 0.8}


def feature(h):
    return np.array([h[2] / h[8], h[5] / h[8], (h[0] + h[2]) / (h[6] + h[8]), (h[3] + h[5]) / (h[6] + h[8]),
                     (h[1] + h[2]) / (h[7] + h[8]), (h[4] + h[5]) / (h[7] + h[8])])


print(f"scene: {N} correspondences, {K} planes ({'r04 generator' if kw else 'current generator'}), seed {SEED}, {H.shape[0]} initial models "
      f"({INIT}), {col.size} directed neighbour hits (k = {KNN}), {ITERS} iterations at most")
ft = np.array([feature(h) for h in sc.H_true])
l1 = np.abs(ft[:, None, :] - ft[None, :, :]).sum(-1) + np.eye(K) * 1e9
close = [(int(a), int(b), float(l1[a, b])) for a in range(K) for b in range(a + 1, K) if l1[a, b] < 3 * thr2]
print(f"ground truth: pairs of planes whose 6-D features lie within 3 x the mean-shift window ({thr2:.2f}, L1): "
      + (", ".join(f"{a}-{b} ({d:.2f})" for a, b, d in close) if close else "none") + f"; smallest L1 distance {l1.min():.2f}")

cov = coverage(H)
have = planes_of(cov)
print(f"start: {H.shape[0]} models cover planes {sorted(have)}")
events, labeling, last_energy, not_changed = [], np.full(N, -1, np.int32), float(2**31 - 1), 0
history = [{"step": "start", "models": int(H.shape[0]), "planes_covered": len(have)}]
t0 = time.time()
for it in range(1, ITERS + 1):
    owner_before = {p: cov[p][0] for p in have}
    feat, modes, cand, cand_mode, _ = O.merge_candidates(H, sc.F, THR, SEED + it)
    # the mode each model climbs into: the nearest mode in the window's own metric
    mode_of = np.array([int(np.argmin(np.abs(modes - f).sum(1))) for f in feat]) if modes.shape[0] else np.zeros(0, int)
    Hm, changed, _ = O.merging_step(sc.src, sc.dst, H, sc.F, THR, SEED + it)
    if changed:
        H = Hm
    cov = coverage(H)
    now = planes_of(cov)
    for p in sorted(have - now):
        m = owner_before[p]
        mates = sorted({q for q in have if q != p and mode_of[owner_before[q]] == mode_of[m]})
        kept_modes = set(cand_mode.tolist())
        why = ("merge:" + ",".join(map(str, mates))) if mates else ("straight" if mode_of[m] not in kept_modes or Hm.shape[0] < cand.shape[0] else "merge:copies")
        events.append({"iteration": it, "step": "merge", "plane": p, "why": why, "best_fraction_after": round(cov[p][1], 3)})
    have = now
    history.append({"step": f"merge {it}", "models": int(H.shape[0]), "changed": bool(changed), "modes": int(modes.shape[0]), "planes_covered": len(have)})
    if H.shape[0] <= 1:
        break
    owner_before = {p: cov[p][0] for p in have}
    labeling, H, energy, cycles = O.labeling_step(sc.src, sc.dst, sc.aff, H, LAM, thr2, rowptr, col, not changed, sc.F, sc.e2, labeling)
    cov = coverage(H)
    now = planes_of(cov)
    for p in sorted(have - now):
        m = owner_before[p]
        held = int((labeling[plane_pts[p]] == m).sum())
        events.append({"iteration": it, "step": "label", "plane": p, "why": "label", "points_of_the_plane_on_its_model": held,
                       "of": int(plane_pts[p].size), "best_fraction_after": round(cov[p][1], 3)})
    have = now
    history.append({"step": f"label {it}", "models": int(H.shape[0]), "energy": int(energy), "cycles": cycles, "planes_covered": len(have),
                    "labelled_outlier": int((labeling < 0).sum())})
    print(f"  iteration {it:2d}: merge -> {history[-2]['modes']} modes, {history[-2]['models']} models ({'changed' if changed else 'kept'}), "
          f"{history[-2]['planes_covered']} planes | label -> energy {int(energy)}, {len(have)} planes, {(labeling < 0).sum()} outliers", flush=True)
    not_changed = 0 if changed else not_changed + 1
    if (not changed and abs(last_energy - energy) < 1e-2) or not_changed > 10:
        break
    last_energy = energy

from sklearn.metrics import adjusted_rand_score
inl = gt >= 0
ari = float(adjusted_rand_score(gt, labeling))
rec_planes = 0
for p in range(K):
    lp = labeling[plane_pts[p]]; lp = lp[lp >= 0]
    if lp.size and np.bincount(lp).max() >= 0.8 * plane_pts[p].size:
        rec_planes += 1
print(f"end: {H.shape[0]} models, {len(have)} of {K} planes covered by a model, {rec_planes} recovered in the labels (>= 80 % of a plane's "
      f"inliers on one label), ARI {ari:.3f}, outliers labelled {(labeling < 0).sum()} of {(~inl).sum()} generated, {time.time() - t0:.0f} s")
for e in events:
    print("  lost:", json.dumps(e))
print(json.dumps({"points": N, "planes": K, "seed": SEED, "generator": "r04" if kw else "current", "init": INIT, "models_end": int(H.shape[0]),
                  "planes_covered_end": len(have), "planes_recovered_in_labels": rec_planes, "ari": ari,
                  "outliers_labelled": int((labeling < 0).sum()), "outliers_generated": int((~inl).sum()),
                  "close_feature_pairs": close, "events": events, "history": history}))
