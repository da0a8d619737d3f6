#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the BUILD container only).

What pins what:
  * labels / energies  <- the REFERENCE's own alpha-expansion sources compiled unmodified
                          (oracle/_ref/libmh_ref_gco.so, built in place from /root/reference by
                          oracle/Makefile), driven exactly like MultiH::LabelingStep
                          (M/MultiH.cpp:520-555) with the dataEnergy restatement as callback;
  * cost matrices, residuals, re-estimated homographies, DLT outputs
                       <- the oracle restatement (oracle/mh_oracle.cpp), whose formulas cite the
                          reference line by line; known-answer constants are asserted here.
  * barrsmith.npz      <- the reference's only checked-in data files
                          (Executable/results/barrsmith/*.txt) as numeric arrays (DATA).
Fixtures are inputs + expected outputs only; no reference source text is stored.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

synth = importlib.import_module("multi-h_amd.synth")
THR, LAM = 2.2, 0.5          # harness defaults, M/main.cpp:55-59
THR2 = THR * THR


def models_for(sc, rng, extra):
    H = [sc.H_true]
    for _ in range(extra):
        k = rng.integers(0, sc.H_true.shape[0])
        H.append(sc.H_true[k:k + 1] * (1.0 + rng.normal(0, 2e-4, size=(1, 9))))
    return np.concatenate(H, axis=0)


def labeling_fixture(tag, n, k, seed, sym, extra):
    assert O.ref() is not None, "oracle/_ref is not built (needs /root/reference)"
    sc = synth.make_scene(n, k, seed=seed, symmetric=sym)
    H = models_for(sc, np.random.default_rng(seed), extra)
    cost = O.data_cost(sc.src, sc.dst, H, LAM, THR2)
    assert (cost[:, 0] == 4901).all() and cost.max() <= 9802          # SURVEY §8(c) known answers
    potts = O.potts(LAM)
    assert potts == 50
    # reference GCO, callback data cost (the LabelingStep configuration) and dense table
    lab_ref, e_ref = O.ref_expand_formula(sc.src, sc.dst, H, LAM, THR2, sc.hit_rowptr, sc.hit_col)
    lab_tab, e_tab = O.ref_expand_table(cost, sc.hit_rowptr, sc.hit_col, potts)
    assert e_ref == e_tab and np.array_equal(lab_ref, lab_tab)
    init = (sc.gt_label + 1).astype(np.int32)
    lab_warm, e_warm = O.ref_expand_table(cost, sc.hit_rowptr, sc.hit_col, potts, init_labels=init)
    # the oracle restatement must agree with the reference before anything is written
    lab_o, e_o, cyc_o, en_o = O.expand(cost, sc.hit_rowptr, sc.hit_col, potts)
    assert e_o == e_ref and np.array_equal(lab_o, lab_ref), tag
    lab_ow, e_ow, _, _ = O.expand(cost, sc.hit_rowptr, sc.hit_col, potts, init_labels=init)
    assert e_ow == e_warm and np.array_equal(lab_ow, lab_warm), tag
    H_re, cnt = O.haf_reestimate(sc.src, sc.dst, sc.aff, lab_ref - 1, H, sc.F, sc.e2)
    R = O.residual_matrix(sc.src[:64], sc.dst[:64], H)
    np.savez_compressed(os.path.join(HERE, f"labeling_{tag}.npz"),
                        src=sc.src, dst=sc.dst, aff=sc.aff, F=sc.F, e2=sc.e2, H=H,
                        hit_rowptr=sc.hit_rowptr, hit_col=sc.hit_col, lam=LAM, thr=THR,
                        cost=cost, potts=potts, labels_ref=lab_ref, energy_ref=e_ref,
                        init_warm=init, labels_warm=lab_warm, energy_warm=e_warm,
                        cycles=cyc_o, cycle_energies=en_o, H_reestimated=H_re, label_counts=cnt,
                        residual_first64=R, counts=O.score(sc.src, sc.dst, H, THR2))
    print(f"{tag}: n={n} Nh={H.shape[0]} energy={e_ref} cycles={cyc_o} hist={np.bincount(lab_ref)}")


def dlt_fixture(tag, n, m, seed):
    sc = synth.make_scene(n, 3, seed=seed, with_neighbours=False)
    idx = O.sample4(seed, 5, m, n)
    H, wit, sweeps = O.dlt4(sc.src, sc.dst, idx)
    np.savez_compressed(os.path.join(HERE, f"dlt_{tag}.npz"), src=sc.src, dst=sc.dst, seed=seed, first=5,
                        idx=idx, H=H, witness=wit, sweeps=sweeps, rr=O.rr_schedule())
    print(f"{tag}: {m} hypotheses, sweeps {np.bincount(sweeps)}")


def main():
    labeling_fixture("n64_k2", 64, 2, 1, True, 1)
    labeling_fixture("n1000_k3", 1000, 3, 2, False, 3)
    labeling_fixture("n5000_k3", 5000, 3, 1234, True, 3)      # BASELINE configs[1] scale
    dlt_fixture("n500_m256", 500, 256, 77)
    # The reference's only checked-in data: one cached correspondence file and one result file.
    # Stored as numeric arrays (data, not text): points (2903 x 8), result (1094 x 9, last = label).
    ref_data = "/root/reference/Executable/results/barrsmith"
    pts = np.loadtxt(os.path.join(ref_data, "barrsmith_points_with_no_annotation.txt"))
    res = np.loadtxt(os.path.join(ref_data, "result_barrsmith.txt"))
    assert pts.shape == (2903, 8) and res.shape == (1094, 9)
    np.savez_compressed(os.path.join(HERE, "barrsmith.npz"), points=pts, result=res)
    print("barrsmith:", pts.shape, res.shape, "label histogram", np.unique(res[:, 8], return_counts=True))


if __name__ == "__main__":
    main()
