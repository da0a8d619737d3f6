#!/usr/bin/env python3
"""The merge <-> label alternation at BASELINE configs[4] size — 50 000 correspondences, 10 planes, 20 fixed iterations —
through the host class on the GPU against the ORACLE's restatement of the same loop (oracle/mh_oracle.cpp section 11,
every alpha-expansion inside it by the reference's own GCoptimization, oracle/_ref).  Process() starts from
SetInitialHomographies (perturbed ground truth plus near-copies and strays), F given, post-filter off; labels, model
count, iteration number, energy must be EQUAL, homographies equal to 1e-9.  The oracle side takes about a minute on one host
core (the reference's GCO needs 2-5 s per LabelingStep at this size).  tests/test_gpu_at_size_oracle.py runs both routes in the
driver's -m gpu suite; the output is also kept under profiles/.  Env: N PLANES ITERS SEED, and
ROUTE=dlt: the DEFAULT route of Process() instead of given initial models — HYP (100 000) DLT proposals, mh_select_greedy,
then the loop — against mho_process(init_mode = 2): the oracle's own sampling, DLT and sequential selection (its scoring
spread over the host's cores), then the same loop."""
import ctypes as C, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mh = importlib.import_module("multi-h_amd")
import oracle_lib as O
N, K, ITERS, SEED = (int(os.environ.get(k, d)) for k, d in (("N", 50000), ("PLANES", 10), ("ITERS", 20), ("SEED", 1234)))
ROUTE, HYP, MAXM = os.environ.get("ROUTE", "init"), int(os.environ.get("HYP", 100000)), int(os.environ.get("MAX_MODELS", 32))
THR, LAM, LOC = 2.2, 0.5, 0.005
sc = mh.synth.make_scene(N, K, seed=SEED, with_neighbours=False)
rng = np.random.default_rng(SEED)
H0 = [sc.H_true * (1.0 + rng.normal(0, 1e-4, size=sc.H_true.shape))]
for _ in range(5):
    k = rng.integers(0, K); H0.append(sc.H_true[k:k + 1] * (1.0 + rng.normal(0, 2e-4, size=(1, 9))))
for _ in range(2):
    H0.append((np.eye(3) + rng.normal(0, 0.05, size=(3, 3))).reshape(1, 9))
H0 = np.ascontiguousarray(np.concatenate(H0))
# the neighbourhood the class builds by default (16 nearest hits within 1/locality), as a directed hit list for the oracle:
# a pair of weight 2 was found from both sides, a pair of weight 1 from one (which one does not matter, SURVEY A-2)
e = mh.Engine(0, 2.6, THR, LOC, LAM, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.build_neighbors_knn(16, radius=1.0 / LOC)
rp, col, w = e.get_sym_graph()
e.close()
rows = np.repeat(np.arange(N), np.diff(rp))
keep = (w == 2) | ((w == 1) & (rows < col))
hr, hc = rows[keep], col[keep]
order = np.lexsort((hc, hr))
hit_col = hc[order].astype(np.int32)
hit_rowptr = np.concatenate([[0], np.cumsum(np.bincount(hr, minlength=N))]).astype(np.int32)
print(f"scene: {N} correspondences, {K} planes, " + (f"{H0.shape[0]} initial models" if ROUTE == "init" else f"{HYP} DLT proposals (at most {MAXM} selected)")
      + f", {hit_col.size} neighbour hits, {ITERS} fixed iterations", flush=True)

host = C.CDLL(os.path.join(ROOT, "multi-h_amd", "libmultih_host.so"))
dp = C.POINTER(C.c_double)
labels = np.full(N, -7, dtype=np.int32); Hout = np.zeros((256, 9)); it, en, secs = C.c_int(-1), C.c_double(-1), C.c_double(0)
src, dst, aff, F, e2 = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff, sc.F, sc.e2))
host.mhh_set_post_filter(0)
t0 = time.time()
k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), N, F.ctypes.data_as(dp),
                         e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(THR), C.c_double(LOC), C.c_double(LAM), 20,
                         C.c_ulonglong(SEED), 0 if ROUTE == "init" else HYP, 0 if ROUTE == "init" else MAXM, ITERS,
                         H0.ctypes.data_as(dp) if ROUTE == "init" else None, H0.shape[0] if ROUTE == "init" else 0,
                         labels.ctypes.data_as(C.POINTER(C.c_int)), Hout.ctypes.data_as(dp), 256, C.byref(it), C.byref(en), C.byref(secs), 0, 4)
gpu_s = time.time() - t0
C.CDLL(None).fflush(None)
print(f"GPU: {k} models, GetIterationNumber() {it.value}, energy {en.value:.0f}, loop {secs.value:.3f} s, Process() {gpu_s:.2f} s", flush=True)
O.lib().mho_set_fixed_iterations(ITERS)
t0 = time.time()
if ROUTE == "init":
    lab_o, H_o, it_o, en_o, used_ref = O.cluster_merging_and_labeling(sc.src, sc.dst, sc.aff, H0, sc.F, sc.e2, LAM, THR, hit_rowptr, hit_col, SEED)
else:
    w = O.process(sc.src, sc.dst, sc.aff, sc.F, sc.e2, THR, LOC, LAM, 20, SEED, hit_rowptr, hit_col, init_mode=2, hypotheses=HYP,
                  max_propose=MAXM, post_filter=False)
    lab_o, H_o, it_o, en_o, used_ref = w["labels"], w["H"], w["iterations"], w["energy"], w["used_reference_gco"]
cpu_s = time.time() - t0
O.lib().mho_set_fixed_iterations(0)
print(f"oracle ({'reference GCO' if used_ref else 'own expansion'}): {H_o.shape[0]} models, iterations {it_o}, energy {en_o:.0f}, {cpu_s:.1f} s "
      + ("on one core" if ROUTE == "init" else f"(selection scored on {os.cpu_count()} cores, the loop on one)"), flush=True)
same_labels = bool(np.array_equal(labels, lab_o))
hdiff = float(np.max(np.abs(Hout[:k] - H_o) / np.max(np.abs(H_o), axis=1, keepdims=True))) if k == H_o.shape[0] and k > 0 else float("nan")
rec = {"route": ROUTE, "points": N, "planes": K, "fixed_iterations": ITERS, "initial_models": int(H0.shape[0]) if ROUTE == "init" else None,
       "dlt_proposals": HYP if ROUTE != "init" else None, "neighbour_hits": int(hit_col.size),
       "gpu": {"models": int(k), "iterations": it.value, "energy": en.value, "loop_s": secs.value, "process_s": gpu_s},
       "oracle": {"models": int(H_o.shape[0]), "iterations": it_o, "energy": en_o, "seconds_one_core": cpu_s, "reference_gco": bool(used_ref)},
       "labels_identical": same_labels, "labels_differing": int((labels != lab_o).sum()), "max_rel_homography_difference": hdiff,
       "label_histogram": np.bincount(labels + 1).tolist()}
print(json.dumps(rec))
ok = same_labels and k == H_o.shape[0] and it.value == it_o and en.value == en_o and hdiff <= 1e-9
print("AT-SIZE ALTERNATION:", "EQUAL" if ok else "DIFFERENT")
sys.exit(0 if ok else 1)
