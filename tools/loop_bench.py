#!/usr/bin/env python3
"""BASELINE configs[4]: the full propose -> {merge, label, re-estimate} loop through the host class
MultiH (libmultih_host.so) with a fixed number of iterations, on 1..G GPUs of one node.

  python tools/loop_bench.py                                   # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
         --master-port 29511 tools/loop_bench.py               # one process per GPU

Multi-GPU: the propose stage shards (each rank owns M/G hypotheses of every batch; all-gather of
the int32 scores per greedy round); labeling and re-estimation run replicated and
deterministic.  Every rank must end with the same labels — checked here — and the result must not
depend on G.  Env: N K ITERS HYP ITER_HYP; LOOP_BACKEND=gloo LOOP_DEVICE=0 put several ranks on one
GPU (the parity test's setup).  INIT=stable: the reference's own initialisation (per-point homographies, mean shift,
3-point fits) instead of the DLT batch.  CPU_LOOP=1 (one GPU, DLT route): the same loop — ClusterMergingAndLabeling,
M/MultiH.cpp:263-311 — once more through the ORACLE on one host core (oracle/mh_oracle.cpp section 11, every
alpha-expansion by the reference's own GCoptimization of oracle/_ref), from the models the GPU's selection handed to the
loop and on the neighbourhood the engine built: the CPU baseline beside `loop_s` (checker code, reported only)."""
import ctypes as C, hashlib, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, K = int(os.environ.get("N", 50000)), int(os.environ.get("K", 10))
ITERS, HYP = int(os.environ.get("ITERS", 20)), int(os.environ.get("HYP", 100000))
ITER_HYP = int(os.environ.get("ITER_HYP", 0))
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local_rank = int(os.environ.get("LOCAL_RANK", 0))
device = int(os.environ.get("LOOP_DEVICE", local_rank))
if world > 1:
    # torch before the engine: the library then binds to the HIP runtime torch has already loaded
    # (one runtime per process; loaded the other way round the second copy finds no device)
    import torch, torch.distributed as dist
host = C.CDLL(os.path.join(ROOT, "multi-h_amd", "libmultih_host.so"))
host.mhh_set_device(device)
if "RECYCLE" in os.environ: host.mhh_set_engine_tuning(11, int(os.environ["RECYCLE"]))      # alpha-expansion flow recycling A/B
for kv in filter(None, os.environ.get("TUNE", "").split(",")):                                    # e.g. TUNE=3=512,10=6 (mh_set_tuning keys)
    host.mhh_set_engine_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
if "TRACE" in os.environ: host.mhh_set_engine_tuning(8, 64)                                     # per-move solver log (with MULTIH_TIMING=1)
if "KNN" in os.environ: host.mhh_set_neighbourhood(int(os.environ["KNN"]), C.c_double(0.0))
if world > 1 and "LOOP_DEVICE" in os.environ:
    # Rehearsal with several ranks on ONE GPU: the alpha-expansion's solver launch synchronises through a grid barrier
    # and must be resident as a whole, so engines that share a device split its CUs between them (each rank's solver
    # would otherwise wait for workgroups the other rank's launch keeps off the chip until the barrier times out).
    # Results never depend on the solver's grid.  One process per GPU — the deployment — needs none of this.
    host.mhh_set_engine_tuning(5, max(1, 256 // world))
hook = None
if world > 1:
    backend = os.environ.get("LOOP_BACKEND", "nccl")
    torch.cuda.set_device(device)
    dist.init_process_group(backend, rank=rank, world_size=world)
    sharding = importlib.import_module("multi-h_amd.sharding")
    hook = sharding.make_allgather_hook(world, torch.device("cuda", device))
    host.mhh_set_sharding(rank, world, hook, None)
    dist.barrier()
sc = mh.synth.make_scene(N, K, seed=1234, with_neighbours=False)
dp = C.POINTER(C.c_double)
labels = np.full(N, -7, dtype=np.int32); Hout = np.zeros((256, 9))
it, en, secs = C.c_int(0), C.c_double(0), C.c_double(0)
src, dst, aff, F, e2 = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff, sc.F, sc.e2))
t0 = time.time()
k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), N,
                         F.ctypes.data_as(dp), e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(2.2),
                         C.c_double(0.005), C.c_double(0.5), 20, C.c_ulonglong(1234), HYP, 32, ITERS,
                         None, 0, labels.ctypes.data_as(C.POINTER(C.c_int)), Hout.ctypes.data_as(dp), 256,
                         C.byref(it), C.byref(en), C.byref(secs), ITER_HYP, -1 if os.environ.get("INIT") == "stable" else 4)
wall = time.time() - t0
labeling_steps = int(host.mhh_get_labeling_steps())
digest = hashlib.sha256(labels.tobytes() + Hout[:max(k, 0)].tobytes()).hexdigest()[:16]
# REPEAT=1: the same call once more in this process — the first call of a process also pays for the HIP runtime and the code
# objects; the second is what a caller that processes image pair after image pair sees
wall_warm = None
if os.environ.get("REPEAT") and world == 1:
    lab2 = np.full(N, -7, dtype=np.int32); H2 = np.zeros((256, 9)); it2, en2, secs2 = C.c_int(0), C.c_double(0), C.c_double(0)
    t0 = time.time()
    k2 = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), N,
                              F.ctypes.data_as(dp), e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(2.2),
                              C.c_double(0.005), C.c_double(0.5), 20, C.c_ulonglong(1234), HYP, 32, ITERS,
                              None, 0, lab2.ctypes.data_as(C.POINTER(C.c_int)), H2.ctypes.data_as(dp), 256,
                              C.byref(it2), C.byref(en2), C.byref(secs2), ITER_HYP, -1 if os.environ.get("INIT") == "stable" else 4)
    wall_warm = time.time() - t0
    assert k2 == k and np.array_equal(lab2, labels) and np.array_equal(H2, Hout), "the second call of the process gave another result"
same = True
if world > 1:
    walls = [None] * world; digs = [None] * world
    dist.all_gather_object(walls, wall); dist.all_gather_object(digs, digest)
    wall = max(walls); same = len(set(digs)) == 1
# Agreement with the generator's ground truth: a plane counts as RECOVERED when one label holds at least
# 80 % of its inlier correspondences; ARI over all correspondences (outliers = one class of their own, label -1)
quality = mh.synth.agreement(sc.gt_label, labels)
if rank == 0 and os.environ.get("CONFUSION"):
    # planes x labels: where every ground-truth plane's correspondences ended up (column 0 = outlier label)
    tab = np.zeros((K + 1, max(k, 0) + 1), dtype=np.int64)
    np.add.at(tab, (sc.gt_label + 1, labels + 1), 1)
    print("confusion (rows: outliers, plane 0..; columns: label -1, 0..):")
    for r in range(K + 1):
        print(("outl " if r == 0 else f"p{r - 1:<3d} ") + " ".join(f"{v:6d}" for v in tab[r]))
if rank == 0:
    print(f"N={N} planes={K} hypotheses={HYP} gpus={world}: clusters={k} iterations={it.value} energy={en.value:.0f} "
          f"loop={secs.value:.2f}s total={wall:.2f}s  planes recovered {quality['planes_recovered']}/{K}  ARI {quality['ari']:.3f}  "
          f"outliers labelled {quality['outliers_labelled']} (generated {quality['outliers_generated']})")
    cpu_loop = None
    if os.environ.get("CPU_LOOP") and world == 1 and os.environ.get("INIT") != "stable":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        e = mh.Engine(device, 2.6, 2.2, 0.005, 0.5, 20)
        e.set_correspondences(sc.src, sc.dst, sc.aff)
        e.set_epipolar(sc.F, sc.e2)
        e.propose_dlt4(1234, 0, HYP)                                   # MultiH::ProposeModels: the same batch, the same selection
        e.set_tuning(30, 1)
        H0, _, _, _ = e.select_greedy(2.2 * 2.2, 20, 32, mask=np.ones(N, np.uint8))
        e.build_neighbors_knn(16, radius=1.0 / 0.005)
        rp, col, w = e.get_sym_graph()
        e.close()
        rows = np.repeat(np.arange(N), np.diff(rp))
        keepw = (w == 2) | ((w == 1) & (rows < col))                   # a pair of weight 2 was found from both sides (SURVEY A-2)
        hr, hc = rows[keepw], col[keepw]
        order = np.lexsort((hc, hr))
        hit_col = hc[order].astype(np.int32)
        hit_rowptr = np.concatenate([[0], np.cumsum(np.bincount(hr, minlength=N))]).astype(np.int32)
        O.lib().mho_set_fixed_iterations(ITERS)
        t0 = time.time()
        lab_o, H_o, it_o, en_o, used_ref = O.cluster_merging_and_labeling(sc.src, sc.dst, sc.aff, H0, sc.F, sc.e2, 0.5, 2.2, hit_rowptr, hit_col, 1234)
        cpu_s = time.time() - t0
        O.lib().mho_set_fixed_iterations(0)
        cpu_loop = {"value": cpu_s, "unit": "s", "cores": 1, "kind": "port",
                    "sample": f"the whole loop on the same scene from the same {H0.shape[0]} initial models: oracle restatement of ClusterMergingAndLabeling "
                              f"(M/MultiH.cpp:263-311), every alpha-expansion by the reference's GCoptimization compiled unmodified (oracle/_ref): {bool(used_ref)}",
                    "iterations": int(it_o), "energy": float(en_o), "models": int(H_o.shape[0]),
                    "same_iterations_and_energy_as_the_gpu_loop": bool(it_o == it.value and en_o == en.value)}
        print(f"CPU loop (oracle, one core): {cpu_s:.2f} s, iterations {it_o}, energy {en_o:.0f}, {H_o.shape[0]} models")
    print(json.dumps({"workload": "full_loop", "points": N, "planes": K, "hypotheses": HYP, "iterations": it.value,
                      "labeling_steps": labeling_steps, "cpu_loop": cpu_loop,
                      "iter_hypotheses": ITER_HYP, "n_gpus": world, "clusters": k, "energy": en.value,
                      "loop_s": secs.value, "total_s": wall, "total_s_second_call": wall_warm, "digest": digest, "ranks_identical": same,
                      "exchanges": hook.stats["calls"] if hook else 0, **quality}))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
sys.exit(0 if (k >= 0 and same) else 1)
