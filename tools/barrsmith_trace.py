#!/usr/bin/env python3
"""Where do the reference's five barrsmith planes go?    A step-by-step trace of Process()
on the reference's only data set through the ORACLE on the CPU (tests/oracle_lib.py — the restatement of MergingStep /
LabelingStep / the post-filter, M/MultiH.cpp:100-222, 352-471, 513-602; no GPU, no product code), following the same seeds
mho_process / class MultiH use, so that each trajectory IS the one tools/barrsmith_agreement.py reports an ARI for.

"Ground truth" = the labels of Executable/results/barrsmith/result_barrsmith.txt (5 planes: 33 / 514 / 128 / 83 / 154
correspondences, 182 outliers).  A model COVERS a reference plane when >= COVER (0.6) of the plane's correspondences lie
within the inlier threshold of it.  After the initialisation, every MergingStep, every LabelingStep and the post-filter the
trace says how many models there are and which reference planes are covered, and when a plane loses its cover, by which rule:

  merge:<planes>  mean shift on the 6-D feature (M/MultiH.cpp:364-390; window sum_j |delta_j| < thr^2, quirk A-8) put the
                  plane's model into one mode with the model of ANOTHER covered plane
  merge:copies    ... into a mode with models that cover no other plane (near-copies, fragments); the mode's 3-point model
                  (:407-409) no longer covers it
  refit           the plane's model was ALONE in its mode, the mode's 3-point model (rebuilt from the images of (0,0), (1,0),
                  (0,1) — a one-pixel triangle in the image corner, :395-409) was kept, and it no longer covers the plane:
                  whenever the number of models changes, EVERY model is replaced by such a reconstruction (:468-470)
  straight        the mode's model was dropped by the straightness / < 3 inliers test (:446-463)
  label           the model survived MergingStep, but after alpha-expansion + HAF re-estimation (:513-602) it no longer covers
                  the plane (its correspondences went to a neighbouring model, or the re-estimation moved it)
  filter          HomographyCompatibilityCheck removed the cluster (:100-222: fewer than min_inliers members, or the median
                  of 501 three-point cross-validation medians above 81/16 thr^2)

At the end: per reference plane the share of its correspondences under our dominant label (purity), and whether the plane is
SPLIT (two of our labels each hold >= 25 % of it).

Inputs: the reference's 1 094 kept rows with F estimated from them (ROWS=reference, what tests/test_gpu_barrsmith.py and
DESIGN 6a quote), or ROWS=raw: the harness route from the 2 903 input rows (load filter at 2.0 px, RANSAC at 2.6 px,
OptimalTriangulation, distanceError <= 1; METRIC=1 the point-to-epipolar-line distance, 0 Sampson).
Env: ROUTES=dlt,stable_sets  SEEDS=1234,7,99,...  ROWS=reference|raw  METRIC=1  KNN=16  HYP=20000  VERBOSE=0.
Prints one line per event, a per-run summary, a table over the seeds and one JSON line; kept under profiles/."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_lib as O  # noqa: E402

ba = importlib.import_module("barrsmith_agreement")
synth = importlib.import_module("multi-h_amd.synth")

THR, LAM, LOC, MIN_INL, THR_F = 2.2, 0.5, 0.005, 20, 2.6
thr2 = THR * THR
COVER = 0.6
ROUTES = os.environ.get("ROUTES", "dlt,stable_sets").split(",")
SEEDS = [int(x) for x in os.environ.get("SEEDS", "1234,7,99,1,2,3,4,5").split(",")]
ROWS = os.environ.get("ROWS", "reference")
METRIC = int(os.environ.get("METRIC", "1"))
KNN = int(os.environ.get("KNN", "16"))
HYP = int(os.environ.get("HYP", "20000"))
VERBOSE = os.environ.get("VERBOSE", "0") != "0"


def knn_radius_hits(src, dst, k, radius):
    """the class's default neighbourhood: the k nearest hits of the float32 (x1, y1, x2, y2) vectors inside 1 / locality"""
    from scipy.spatial import cKDTree
    n = len(src)
    pv = np.concatenate([src, dst], axis=1).astype(np.float32).astype(np.float64)
    dist, idx = cKDTree(pv).query(pv, k=min(k, n - 1) + 1)
    rows = np.repeat(np.arange(n), idx.shape[1])
    cols = idx.reshape(-1)
    keep = (rows != cols) & (dist.reshape(-1) <= radius)
    rows, cols = rows[keep], cols[keep]
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    return rowptr, cols.astype(np.int32)


def feature(h):
    return np.array([h[2] / h[8], h[5] / h[8], (h[0] + h[2]) / (h[6] + h[8]), (h[3] + h[5]) / (h[6] + h[8]),
                     (h[1] + h[2]) / (h[7] + h[8]), (h[4] + h[5]) / (h[7] + h[8])])


class Run:
    def __init__(self, src, dst, aff, F, e2, ref, seed):
        self.src, self.dst, self.aff, self.F, self.e2, self.ref, self.seed = src, dst, aff, F, e2, ref, seed
        self.n = len(src)
        self.planes = [int(p) for p in np.unique(ref[ref >= 0])]
        self.plane_pts = {p: np.flatnonzero(ref == p) for p in self.planes}
        self.rowptr, self.col = knn_radius_hits(src, dst, KNN, 1.0 / LOC)
        self.events, self.history = [], []

    def coverage(self, Hs):
        out = {}
        for p in self.planes:
            ip = self.plane_pts[p]
            if Hs.shape[0] == 0 or ip.size == 0:
                out[p] = (-1, 0.0)
                continue
            with np.errstate(all="ignore"):
                R = O.residual_matrix(self.src[ip], self.dst[ip], Hs)
            fr = (R < thr2).mean(axis=1)
            m = int(np.argmax(fr))
            out[p] = (m, float(fr[m]))
        return out

    def covered(self, cov):
        return {p for p, (m, f) in cov.items() if f >= COVER}

    def log(self, msg):
        if VERBOSE:
            print("      " + msg, flush=True)

    def loop(self, H):
        """ClusterMergingAndLabeling (M/MultiH.cpp:263-311) as oracle/mh_oracle.cpp section 11 runs it, with the trace."""
        cov = self.coverage(H)
        have = self.covered(cov)
        self.history.append({"step": "start", "models": int(H.shape[0]), "covered": sorted(have)})
        labeling = np.full(self.n, -1, np.int32)
        last_energy, not_changed, step, iteration = float(2 ** 31 - 1), 0, 0, 0
        final_energy = 0.0
        while iteration < 500:
            iteration += 1
            owner = {p: cov[p][0] for p in have}
            mseed = self.seed ^ 0x4d53 ^ (step << 20)
            feat, modes, cand, cand_mode, _ = O.merge_candidates(H, self.F, THR, mseed)
            mode_of = np.array([int(np.argmin(np.abs(modes - f).sum(1))) for f in feat]) if modes.shape[0] else np.zeros(0, int)
            Hm, changed, _ = O.merging_step(self.src, self.dst, H, self.F, THR, mseed)
            step += 1
            if changed:
                H = Hm
            cov = self.coverage(H)
            now = self.covered(cov)
            for p in sorted(have - now):
                m = owner[p]
                mates = sorted({q for q in have if q != p and mode_of[owner[q]] == mode_of[m]})
                in_mode = int((mode_of == mode_of[m]).sum())
                kept_modes = set(cand_mode.tolist())
                if mates:
                    why = "merge:" + ",".join(map(str, mates))
                elif mode_of[m] not in kept_modes:
                    why = "straight"
                elif in_mode > 1:
                    why = "merge:copies"
                else:
                    why = "refit"
                self.events.append({"iteration": iteration, "step": "merge", "plane": p, "why": why, "models_in_its_mode": in_mode,
                                    "best_fraction_after": round(cov[p][1], 3)})
                self.log(f"iteration {iteration}: plane {p} loses its cover in MergingStep ({why}; {in_mode} models in its mode; best model now holds {cov[p][1]:.2f})")
            have = now
            not_changed = 0 if changed else not_changed + 1
            nh = H.shape[0]
            self.history.append({"step": f"merge {iteration}", "models": int(nh), "changed": bool(changed), "covered": sorted(have)})
            if nh <= 1:
                if nh == 1:
                    with np.errstate(all="ignore"):
                        labeling = np.where(O.residual_matrix(self.src, self.dst, H)[0] < thr2, 0, -1).astype(np.int32)
                break
            owner = {p: cov[p][0] for p in have}
            labeling, H, energy, cycles = O.labeling_step(self.src, self.dst, self.aff, H, LAM, thr2, self.rowptr, self.col,
                                                          not changed, self.F, self.e2, labeling)
            cov = self.coverage(H)
            now = self.covered(cov)
            for p in sorted(have - now):
                m = owner[p]
                held = int((labeling[self.plane_pts[p]] == m).sum())
                self.events.append({"iteration": iteration, "step": "label", "plane": p, "why": "label", "points_of_the_plane_on_its_model": held,
                                    "of": int(self.plane_pts[p].size), "best_fraction_after": round(cov[p][1], 3)})
                self.log(f"iteration {iteration}: plane {p} loses its cover in LabelingStep ({held} of {self.plane_pts[p].size} of its points stayed on its model; "
                         f"best model now holds {cov[p][1]:.2f})")
            for p in sorted(now - have):
                self.log(f"iteration {iteration}: plane {p} is covered again after LabelingStep")
            have = now
            self.history.append({"step": f"label {iteration}", "models": int(nh), "energy": int(energy), "covered": sorted(have),
                                 "labelled_outlier": int((labeling < 0).sum())})
            if (not changed and abs(last_energy - energy) < 1e-5) or not_changed > 10:
                final_energy = float(energy)
                break
            last_energy = energy
        return labeling, H, iteration - 1 if iteration >= 500 else iteration, final_energy, have

    def post_filter(self, labeling, H, have):
        cov = self.coverage(H)
        owner = {p: cov[p][0] for p in have}
        members = np.bincount(labeling[labeling >= 0], minlength=H.shape[0])
        lab2, H2, med = O.compatibility_check(self.src, self.dst, labeling, H, self.F, thr2, MIN_INL, self.seed ^ 0xc0117a7)
        removed = [i for i in range(H.shape[0]) if members[i] < MIN_INL or (np.isfinite(med[i]) and med[i] > thr2 * 81.0 / 16.0)]
        cov2 = self.coverage(H2)
        now = self.covered(cov2)
        for p in sorted(have - now):
            m = owner[p]
            self.events.append({"iteration": -1, "step": "filter", "plane": p, "why": "filter", "members": int(members[m]),
                                "median_of_medians": None if not np.isfinite(med[m]) else round(float(med[m]), 2)})
            self.log(f"post-filter: plane {p} loses its cover (its cluster had {members[m]} members, median of medians {med[m]:.2f} against {thr2 * 81 / 16:.2f})")
        self.history.append({"step": "post-filter", "models": int(H2.shape[0]), "removed": len(removed),
                             "removed_for_size": int(sum(members[i] < MIN_INL for i in removed)), "covered": sorted(now)})
        return lab2, H2, now


def initial_models(run, route):
    if route == "stable_sets":
        return O.establish_stable_point_sets(run.src, run.dst, run.aff, run.F, run.e2, LOC, THR, run.seed ^ 0x57ab1e)
    idx = O.sample4(run.seed, 0, HYP, run.n)
    Hh, _, _ = O.dlt4(run.src, run.dst, idx)
    Hs, _, _, _ = O.select_greedy_refit(run.src, run.dst, run.aff, run.F, run.e2, Hh, thr2, max(MIN_INL, 8), 32)
    return Hs


def inputs(seed):
    pts, ref_rows, ref_labels = ba.kept_correspondences(with_rows=True)
    O.set_fundamental_metric(METRIC)
    try:
        if ROWS == "reference":
            corr = np.ascontiguousarray(pts[ref_rows])
            # F from exactly these rows (the engine's estimate_fundamental = mho_front_half's first part); points as they are
            k, F, e1, e2, keep, refined = O.front_half(corr[:, :2], corr[:, 2:4], corr[:, 4:8], 1234 ^ 0xf00d, 4000, THR_F)
            return corr[:, :2].copy(), corr[:, 2:4].copy(), corr[:, 4:8].copy(), F, e2, ref_labels.copy(), {"rows": int(len(corr))}
        # the harness route from the raw file: load filter (2.0 px), then Process()'s own front half
        src, dst, aff = pts[:, :2].copy(), pts[:, 2:4].copy(), pts[:, 4:8].copy()
        k0, F0, _, _, _, _, reason0 = O.front_half(src, dst, aff, seed ^ 0x10adf117e4, 4000, 2.0, with_reasons=True)
        rows1 = np.flatnonzero(reason0 != 1)                       # the RANSAC mask of the load filter
        k1, F, e1, e2, keep, refined, reason = O.front_half(src[rows1], dst[rows1], aff[rows1], seed ^ 0xf00d, 4000, THR_F, with_reasons=True)
        rows2 = rows1[keep == 1]
        R = refined[keep == 1]
        full = np.full(len(pts), -2)
        full[ref_rows] = ref_labels
        ref_here = full[rows2]                                     # -2: a row the reference did not keep
        stages = {"loaded": int(len(pts)), "after_load_filter": int(len(rows1)), "in_ransac_mask": int((reason != 1).sum()),
                  "after_optimal_triangulation": int(np.isin(reason, (0, 3)).sum()), "after_distance_error": int(len(rows2)),
                  "in_common_with_the_reference": int((ref_here > -2).sum())}
        return R[:, :2].copy(), R[:, 2:4].copy(), R[:, 4:8].copy(), F, e2, ref_here, stages
    finally:
        O.set_fundamental_metric(0)


def one(route, seed):
    src, dst, aff, F, e2, ref, stages = inputs(seed)
    run = Run(src, dst, aff, F, e2, ref, seed)
    t0 = time.time()
    H0 = initial_models(run, route)
    cov0 = run.coverage(H0)
    per_plane_models = {}
    for p in run.planes:
        with np.errstate(all="ignore"):
            R = O.residual_matrix(src[run.plane_pts[p]], dst[run.plane_pts[p]], H0) if H0.shape[0] else np.zeros((0, 1))
        per_plane_models[p] = int(((R < thr2).mean(axis=1) >= COVER).sum()) if H0.shape[0] else 0
    labeling, H, it, energy, have = run.loop(H0)
    n_loop = int(H.shape[0])
    if H.shape[0] > 1:
        labeling, H, have = run.post_filter(labeling, H, have)
    known = ref > -2
    a = ba.agreement(labeling[known], ref[known]) if H.shape[0] else {"planes": 0}
    split = {}
    for p in run.planes:
        lp = labeling[run.plane_pts[p]]
        vals, cnts = np.unique(lp, return_counts=True)
        order = np.argsort(-cnts)
        big = [(int(vals[i]), int(cnts[i])) for i in order if cnts[i] >= 0.25 * lp.size]
        split[p] = {"points": int(lp.size), "labels_holding_a_quarter": big, "split": sum(1 for v, c in big if v >= 0) >= 2,
                    "lost": bool(big and big[0][0] == -1 and len(big) == 1)}
    rec = {"route": route, "seed": seed, "stages": stages, "initial_models": int(H0.shape[0]), "initial_models_covering_each_plane": per_plane_models,
           "models_after_the_loop": n_loop, "models_after_the_filter": int(H.shape[0]), "iterations": it, "energy": energy,
           "planes_covered_at_the_end": sorted(have), "ari_reference_inliers": a.get("ari_reference_inliers"), "ari_all": a.get("ari_all"),
           "events": run.events, "history": run.history, "split": split, "seconds": round(time.time() - t0, 1)}
    lost = [f"{e['plane']}@{e['iteration']}:{e['why']}" for e in run.events]
    print(f"  {route:12s} seed {seed:5d}: {H0.shape[0]:3d} initial models (covering planes: {per_plane_models}) -> {n_loop} after the loop "
          f"({it} iterations) -> {H.shape[0]} after the filter; covered at the end {sorted(have)}; ARI on the reference's inliers "
          f"{a.get('ari_reference_inliers', float('nan')):.3f}; cover lost: {lost if lost else 'never'}; "
          f"split planes {[p for p in run.planes if split[p]['split']]}, lost planes {[p for p in run.planes if split[p]['lost']]}", flush=True)
    return rec


def main():
    print(f"barrsmith through the oracle: rows = {ROWS}, epipolar distance = {'point-to-line' if METRIC else 'Sampson'}, k = {KNN}, "
          f"{HYP} DLT hypotheses, cover = {COVER}, seeds {SEEDS}")
    out = []
    for route in ROUTES:
        for seed in SEEDS:
            out.append(one(route, seed))
    print("\nsummary over the seeds")
    for route in ROUTES:
        rs = [r for r in out if r["route"] == route]
        aris = sorted(r["ari_reference_inliers"] for r in rs if r["ari_reference_inliers"] is not None)
        why = {}
        for r in rs:
            for e in r["events"]:
                key = e["step"] + ":" + e["why"].split(":")[0]
                why[key] = why.get(key, 0) + 1
        n_split = sum(any(v["split"] for v in r["split"].values()) for r in rs)
        n_lost = sum(any(v["lost"] for v in r["split"].values()) for r in rs)
        n_dom_split = sum(r["split"][1]["split"] for r in rs if 1 in r["split"])
        print(f"  {route:12s}: planes after the filter {[r['models_after_the_filter'] for r in rs]} (after the loop {[r['models_after_the_loop'] for r in rs]}, "
              f"initial {[r['initial_models'] for r in rs]}); ARI median {aris[len(aris) // 2]:.3f}, min {aris[0]:.3f}, max {aris[-1]:.3f}; "
              f"runs with a split plane {n_split} of {len(rs)} (the 514-point plane split in {n_dom_split}), with a lost plane {n_lost}; cover-loss events by rule {why}")
    print(json.dumps({"rows": ROWS, "metric": METRIC, "knn": KNN, "runs": [{k: v for k, v in r.items() if k != "history"} for r in out]}))


if __name__ == "__main__":
    main()
