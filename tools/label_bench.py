#!/usr/bin/env python3
"""Times one LabelingStep (data cost -> alpha-expansion -> re-estimate) on the GPU against the CPU
oracle and, when built, the reference's own GCO (oracle/_ref).  Diagnostic / BASELINE.md B3."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mh = importlib.import_module("multi-h_amd")
import oracle_lib as O
N, K = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10))
# LEGACY=1: the r04 generator (planes inside each other's truncation threshold: the HARD instances of the max-flows)
# SEPARATION=<px>: planes that far apart where they are observed (default 13; 2: inside the truncation threshold — some moves keep a core)
kw = {"plane_separation": float(os.environ["SEPARATION"])} if "SEPARATION" in os.environ else {}
t0 = time.time(); sc = mh.synth.make_scene(N, K, seed=1234, legacy_r04=bool(os.environ.get("LEGACY")), **kw); print(f"scene {time.time()-t0:.1f}s ({'r04 generator' if os.environ.get('LEGACY') else 'current generator'}{', planes ' + os.environ['SEPARATION'] + ' px apart' if kw else ''}), hits {sc.hit_col.size}")
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2)
t0 = time.time(); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col); print(f"graph upload {time.time()-t0:.2f}s")
H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
lab = np.full(N, -1, np.int32)
e.set_models(H)
if "RECYCLE" in os.environ: e.set_tuning(11, int(os.environ["RECYCLE"]))    # flow recycling off (A/B)
if "REDUCE" in os.environ: e.set_tuning(6, int(os.environ["REDUCE"]))     # 0 = no dominance reduction (A/B)
if "CTX" in os.environ: e.set_tuning(37, int(os.environ["CTX"]))         # alpha-moves per batch (1 = one after the other, the form until r05)
for it in range(int(os.environ.get('STEPS', 3))):
    t0 = time.time(); lab_g, en, cyc = e.labeling_step(it > 0, lab); tg = time.time() - t0
    print(f"GPU labeling step {it}: {tg*1e3:.1f} ms, energy {int(en)}, cycles {cyc}, stats {e.expand_stats()}, concurrent moves {e.expand_batch_stats()}")
    if it == 0: lab0, en0 = lab_g.copy(), en
    lab = lab_g
if os.environ.get("CPU", "1") == "1":
    t0 = time.time(); lab_o, H_o, e_o, cyc_o = O.labeling_step(sc.src, sc.dst, sc.aff, H, 0.5, 2.2**2, sc.hit_rowptr, sc.hit_col, False, sc.F, sc.e2, np.full(N, -1, np.int32)); to = time.time() - t0
    print(f"oracle (Dinic) step 0: {to*1e3:.1f} ms energy {e_o} cycles {cyc_o} labels equal {np.array_equal(lab_o, lab0)}")
    if O.ref() is not None:
        t0 = time.time(); lab_r, e_r = O.ref_expand_formula(sc.src, sc.dst, H, 0.5, 2.2**2, sc.hit_rowptr, sc.hit_col); tr = time.time() - t0
        print(f"reference GCO (BK, callback cost) expansion only: {tr*1e3:.1f} ms energy {e_r} labels equal {np.array_equal(lab_r - 1, lab0)}")
