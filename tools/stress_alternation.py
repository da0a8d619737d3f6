#!/usr/bin/env python3
"""Randomised stress of the WHOLE merge <-> label alternation: Process() of the host class (GPU engine) from
SetInitialHomographies against the oracle's independent restatement of the loop (oracle/mh_oracle.cpp section 11,
every alpha-expansion inside it by the reference's own GCO where oracle/_ref is built) on random scenes — labels,
model count, GetIterationNumber() and GetEnergy() must be equal.  tests/test_gpu_alternation.py runs five fixed
scenes of the same comparison.  Run on the GPU box:  SECONDS=600 SEED=1 python tools/stress_alternation.py"""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mh = importlib.import_module("multi-h_amd")
import oracle_lib as O
from test_gpu_alternation import _knn_hits, _initial_models, THR, LAM, LOCALITY
budget = float(os.environ.get("SECONDS", 300))
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
host = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
dp = C.POINTER(C.c_double)
t0 = time.time(); runs = 0; tails = 0; removed = 0; dlt_runs = 0; used_ref_all = True
while time.time() - t0 < budget:
    n = int(rng.integers(600, 4000)); planes = int(rng.integers(2, 6)); seed = int(rng.integers(0, 1 << 30))
    dup, strays = int(rng.integers(0, 5)), int(rng.integers(0, 3))
    sc = mh.synth.make_scene(n, planes, seed=seed, with_neighbours=False, noise=float(rng.uniform(0.2, 1.2)),
                             outlier_frac=float(rng.uniform(0.05, 0.5)))
    H0 = _initial_models(sc, seed, dup, strays)
    rowptr, col = _knn_hits(sc, 16)
    # r03: the whole Process() — post-filter (HomographyCompatibilityCheck) on or off, the degenerate tail included, and
    # either the given initial models or the default route (DLT proposals + greedy selection on the device)
    post = bool(rng.integers(0, 2))
    from_dlt = rng.integers(0, 4) == 0
    hyp = int(rng.integers(500, 3000)) if from_dlt else 0
    want = O.process(sc.src, sc.dst, sc.aff, sc.F, sc.e2, THR, LOCALITY, LAM, 20, seed, rowptr, col,
                     init_H=None if from_dlt else H0, init_mode=2 if from_dlt else 0, hypotheses=hyp, max_propose=16, post_filter=post)
    used_ref_all = used_ref_all and want["used_reference_gco"]
    tails += int(want["degenerate_tail"]); removed += want["removed_by_filter"]; dlt_runs += int(from_dlt)
    H_o, lab_o, it_o, en_o = want["H"], want["labels"], want["iterations"], want["energy"]
    labels = np.full(n, -7, dtype=np.int32); Hout = np.zeros((256, 9)); it, en = C.c_int(-1), C.c_double(-1)
    src, dst, aff, F, e2 = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff, sc.F, sc.e2))
    host.mhh_set_post_filter(1 if post else 0)
    k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), n, F.ctypes.data_as(dp),
                             e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(THR), C.c_double(LOCALITY), C.c_double(LAM), 20,
                             C.c_ulonglong(seed), hyp, 16 if from_dlt else 0, 0, None if from_dlt else H0.ctypes.data_as(dp), 0 if from_dlt else H0.shape[0],
                             labels.ctypes.data_as(C.POINTER(C.c_int)), Hout.ctypes.data_as(dp), 256, C.byref(it), C.byref(en), None, 0, 4)
    ok = k == H_o.shape[0] and it.value == it_o and en.value == en_o and np.array_equal(labels, lab_o)
    if not ok:
        print("MISMATCH", dict(n=n, planes=planes, seed=seed, dup=dup, strays=strays, post=post, from_dlt=from_dlt, hyp=hyp, k=k, k_o=H_o.shape[0],
                               it=it.value, it_o=it_o, en=en.value, en_o=en_o, diff=int((labels != lab_o).sum())))
        sys.exit(1)
    runs += 1
host.mhh_set_post_filter(1)
sys.stdout.flush()
msg = (f"Process() stress ok: {runs} random scenes in {time.time() - t0:.0f} s — {dlt_runs} from DLT proposals + greedy selection, {tails} ended in the degenerate tail, "
       f"the post-filter removed {removed} clusters in all (expansions of the oracle side by the reference GCO: {used_ref_all})")
print(msg, file=sys.stderr, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "stress_alternation.txt"), "a").write(msg + "\n")
