#!/usr/bin/env python3
"""bench.py — scored homography hypotheses/s + residual-kernel HBM GB/s.

One "step" = one pass of the propose-score hot path over one hypothesis batch:
    propose   M 4-tuples (counter RNG) -> batched 4-point DLT          (k_dlt4)
    score     N x M forward-transfer residual matrix written to HBM,
              inlier counts fused into the same kernel                 (k_residual)
    gather    (N_gpus > 1) RCCL all-gather of the per-model int32 scores, enqueued on the engine's stream by the
              native transport (multi-h_amd/host/rccl_transport.cpp: ncclAllGather; no Python in the exchange)
    select    best model on every rank, identical everywhere (the engine's own arg-max kernel, csrc/select.hip)
Two forms of the same step are timed in every run, EXACTLY K steps each, nothing cached or skipped in either:
  sequential  propose, sweep, arg-max one after the other on one stream — the headline at ONE GPU;
  pipelined   the DLT solve of batch i+2 on the engine's second, high-priority stream beside the residual sweep of
              batch i (mh_prefetch_dlt4 / mh_adopt_prefetched) — the headline at SEVERAL GPUs, where a rank's step is 1 ms.
The residual kernel runs at the board's power cap — its time is its energy (profiles/archive/r03_energy.json) — so a DLT beside
it is not free: it shows as a longer sweep, and at one GPU the two forms step within 0.1 % of each other.  The
roofline's launch time is the kernel's in the headline form; the other form is reported as `pipelined_form` /
`sequential_form`.
Workload at N=1 = BASELINE.json configs[2]: 50 000 correspondences / 10 planes,
100 000 hypotheses (the configuration the metric is quoted on; it fits one GPU:
R is 40 GB of the 288 GB).  Inputs are resident in HBM before the timed region.
Multi-GPU (`--gpus N`): the headline is BASELINE configs[3] — ONE batch of 100 000
hypotheses split across the ranks (`"scaling": "strong"`: the total work per step is the
same for every N), correspondences replicated, the only collective the RCCL all-gather of
the int32 scores.  The same run also times the weak-scaling variant (every rank owns its
own 100 000-hypothesis batch, disjoint RNG counters) and reports it as `"weak_scaling"`.
`--scaling weak` swaps the two.

Launching: the driver starts one process per GPU with torch.distributed.run.  Run by hand
as `python bench.py --gpus N` (no WORLD_SIZE in the environment) the script spawns the N
ranks itself — before torch or the GPU is touched — and exits non-zero if any rank fails.

Prints ONE JSON line (rank 0) with the contract's fields plus `roofline` for
the dominant kernel (k_residual, HBM-write bound) and `cpu_baseline` (the oracle
score loop timed on this box's host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
HBM_WRITE_CEILING_GBPS = 6240.0  # best pure streaming-store kernel on this chip (profiles/archive/r01_store_bw.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--planes", type=int, default=10)
    ap.add_argument("--models", type=int, default=100000, help="hypotheses per GPU per step")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="headline mode for N > 1: strong = one batch of --models split over the ranks (BASELINE "
                         "configs[3]); weak = --models per rank.  The other mode is timed too and reported alongside.")
    ap.add_argument("--cpu-sample", type=int, default=150000, help="hypotheses in the CPU baseline sample (about 13 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="N > 1: time the headline scaling mode only (a rehearsal of 8 ranks on ONE GPU "
                    "cannot hold eight weak-scaling batches of 40 GB each)")
    ap.add_argument("--no-rehearsal", action="store_true", help="skip the one-GPU rehearsal of the strong split's shards (keeps a "
                    "profiler's per-kernel statistics to the full-size launches)")
    return ap.parse_args()


class _DevView:
    """Zero-copy view of an engine buffer for torch (CUDA array interface)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def cpu_baseline(sc, thr2: float, sample: int, seed: int, beside=None):
    """Oracle score loop (restatement of M/MultiH.cpp:430-443) on the host cores.
    Checker code used as a *reported baseline only*; never on the product path.
    The one-core figure is timed on an otherwise idle host (nothing enqueued on the GPU, no second thread).
    beside (optional): called in this thread while a SECOND one-core pass runs in another thread (the C call releases
    the interpreter lock) — bench.py keeps the GPU stepping meanwhile and reports that as the sustained rate; the
    second pass's time is reported as `one_core_beside_the_gpu_loop`, never as the baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import threading

    idx = O.sample4(seed, 0, sample, sc.n)
    H, _, _ = O.dlt4(sc.src, sc.dst, idx)
    t0 = time.perf_counter()
    c1 = O.score(sc.src, sc.dst, H, thr2)
    t1 = time.perf_counter() - t0
    box = {}

    def one_core_again():
        t0 = time.perf_counter()
        box["c"] = O.score(sc.src, sc.dst, H, thr2)
        box["t"] = time.perf_counter() - t0

    if beside is not None:
        th = threading.Thread(target=one_core_again)
        th.start()
        beside(th.is_alive)
        th.join()
        assert (box["c"] == c1).all()
    t0 = time.perf_counter()
    c2, threads = O.score_mt(sc.src, sc.dst, H, thr2)
    t2 = time.perf_counter() - t0
    assert (c1 == c2).all()
    out = {
        "value": sample / t1, "unit": "hypotheses/s", "cores": 1, "kind": "port",
        "sample": f"{sample} DLT hypotheses x {sc.n} points, oracle score loop "
                  f"(restates M/MultiH.cpp:430-443), g++ -O2 -ffp-contract=off, {t1:.1f} s on an idle host",
        "all_cores": {"value": sample / t2, "cores": threads, "seconds": t2},
    }
    if "t" in box:
        out["one_core_beside_the_gpu_loop"] = {"value": sample / box["t"], "seconds": box["t"],
                                               "what": "the same one-core pass repeated while the GPU loop of `sustained` ran (context only)"}
    return out, H, c1


def labeling_extra(mh, eng, a, thr2, lam, legacy=False, plane_separation=None):
    """Context for the label half of the path (not part of `value`): one LabelingStep on the GPU next
    to the REFERENCE's own alpha-expansion (oracle/_ref: GCoptimization + BK compiled unmodified, its
    lazy callback data cost restated) on one host core, same inputs, labels compared.
    Three scenes (r06): the bench's own (planes 13 px apart where they are observed: every move is decided by the dominance
    reduction, NO max-flow is solved — `moves_solved` 0), an intermediate one (planes 2 px apart, inside the truncation
    threshold of 4.95 px: a quarter of the moves keep an undecided core of a few hundred sites and go through the solver,
    GCoptimization.cpp:1212-1274; at 4 px and more none does) and the r04 generator's (planes drawn independently: cores of
    thousands of sites).  Each record says which it is — moves_solved, core_max, barriers — and how the concurrent moves
    fared (csrc/expand.hip k_batch_commit): batches, moves kept out of them, batches cut short by a failed test."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    # legacy: the r04 generator — planes inside each other's truncation threshold, i.e. ambiguous data terms: the HARD
    # instances of the alpha-expansion (cores of thousands of sites, several cycles), kept as the solver's stress figure
    kw = {} if plane_separation is None else {"plane_separation": float(plane_separation)}
    sc = mh.synth.make_scene(a.points, a.planes, seed=a.seed, legacy_r04=legacy, **kw)
    eng.set_correspondences(sc.src, sc.dst, sc.aff)
    eng.set_epipolar(sc.F, sc.e2)
    eng.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
    H = sc.H_true * (1.0 + np.random.default_rng(0).normal(0, 1e-4, size=sc.H_true.shape))
    eng.set_models(H)
    eng.labeling_step(False, np.full(sc.n, -1, np.int32))           # warm-up
    eng.set_models(H)
    t0 = time.perf_counter()
    lab, energy, cycles = eng.labeling_step(False, np.full(sc.n, -1, np.int32))
    gpu_ms = (time.perf_counter() - t0) * 1e3
    st = eng.expand_stats()
    bs = eng.expand_batch_stats()
    out = {"scene": ("the r04 generator (planes inside each other's truncation threshold)" if legacy else
                     f"planes {plane_separation if plane_separation is not None else 13.0:g} px apart where they are observed"),
           "sites": sc.n, "labels": H.shape[0] + 1, "neighbour_hits": int(sc.hit_col.size),
           "gpu_labeling_step_ms": gpu_ms, "energy": int(energy), "cycles": int(cycles),
           "moves_run": int(st["moves_run"]), "moves_solved": int(st["moves_solved"]), "core_max": int(st["core_max"]),
           "barriers": int(st["barriers"]), "solver_ms": st["solve_us"] * 1e-3, "launches": int(st["launches"]),
           "concurrent_moves": {"moves_per_batch": int(bs["moves_per_batch"]), "moves": int(st["moves"]), "batches": int(bs["batches"]),
                                "kept_out_of_batches": int(bs["batch_committed"]), "batches_cut_short_by_a_failed_test": int(bs["batch_invalid"]),
                                "run_alone": int(bs["solo_moves"]), "never_launched_as_idempotent": int(bs["host_skipped"]),
                                "validation_hit_rate": (bs["batch_committed"] / max(bs["batch_committed"] + bs["batch_invalid"], 1))},
           "what": ("NO max-flow solved: every move of this step was decided by the dominance reduction (k_move_setup + k_reduce only)"
                    if int(st["moves_solved"]) == 0 else
                    f"{int(st['moves_solved'])} of {int(st['moves_run'])} moves kept an undecided core (at most {int(st['core_max'])} sites) and went through the "
                    "push-relabel solver (k_solve)")}
    # B4: per-label HAF re-estimation, GPU kernel vs the oracle's restatement of
    # GetHomographyHAFNonminimal (M/MultiH.cpp:913-989) on one host core, same labels
    eng.set_models(H)
    eng.reestimate(lab)
    eng.profile_reset(); eng.profile_enable(True)
    for _ in range(5):
        eng.set_models(H)
        H_gpu = eng.reestimate(lab)
    n_re, ms_re = eng.profile_get(5)
    eng.profile_enable(False)
    t0 = time.perf_counter()
    H_cpu, _ = O.haf_reestimate(sc.src, sc.dst, sc.aff, lab, H, sc.F, sc.e2)
    out["reestimate"] = {"gpu_kernel_ms": ms_re / max(n_re, 1), "cpu_port_ms": (time.perf_counter() - t0) * 1e3, "cores": 1,
                         "identical_bits": bool(np.array_equal(H_gpu.view(np.uint64), H_cpu.view(np.uint64)))}
    if O.ref() is not None:
        t0 = time.perf_counter()
        lab_r, e_r = O.ref_expand_formula(sc.src, sc.dst, H, lam, thr2, sc.hit_rowptr, sc.hit_col)
        out["cpu_reference_expansion_ms"] = (time.perf_counter() - t0) * 1e3
        out["cpu_reference"] = {"kind": "reference", "cores": 1,
                                "what": "GCoptimization::expansion of /root/reference compiled unmodified (oracle/_ref)"}
        out["labels_identical"] = bool(np.array_equal(lab_r - 1, lab) and e_r == int(energy))
    return out


def full_loop_extra(a):
    """Context (not part of `value`): BASELINE configs[4] on this one GPU — the whole Process() of the host class
    (100 000 proposals, greedy selection, then the merge/label/re-estimate loop capped at 20 iterations) through
    tools/loop_bench.py in a child process; r06: the loop's CPU baseline beside it (the oracle's restatement with the
    reference's GCO inside, one core, same scene, same initial models) and the reference's own route (INIT_STABLE_SETS)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(N=str(a.points), K=str(a.planes), HYP=str(a.models), ITERS="20", REPEAT="1")

    def run(iter_hyp: int, **extra):
        e = dict(env, ITER_HYP=str(iter_hyp), **extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loop_bench.py")], env=e, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return None, (r.stdout + r.stderr)[-400:]
        return json.loads(line[-1]), None

    def per_iteration(rec):
        n = max(int(rec.get("labeling_steps") or 0), 1)
        return rec["loop_s"] / n * 1e3

    rec, err = run(0, CPU_LOOP="1")
    if rec is None:
        return {"error": err}
    out = {"workload": "BASELINE configs[4] on 1 GPU: Process() of class MultiH, 100000 DLT proposals + greedy selection, then the merge / label / "
                       "re-estimate loop until the reference's stop rule (M/MultiH.cpp:295) fires, at most 20 iterations",
           "iteration_cap": 20, "iterations_run": rec.get("labeling_steps"), "iterations_reported_by_the_class": rec["iterations"],
           "iterations_note": "iterations_run = LabelingSteps the loop ran (MultiH::GetLabelingStepsRun); GetIterationNumber() is the reference's "
                              "iteration_number - 1 (M/MultiH.cpp:311), one less.  The cap of 20 (SetFixedIterations) is a cap: the reference's own "
                              "stop rule stays in force and ends this loop earlier",
           "clusters": rec["clusters"], "energy": rec["energy"], "loop_s": rec["loop_s"],
           "ms_per_iteration": per_iteration(rec),
           "ground_truth": {"what": "agreement of the labels with the generator's ground truth: a plane is recovered when one label holds >= 80 % of "
                                    "its inlier correspondences; ARI over all correspondences (outliers a class of their own)",
                            "planes_recovered": rec.get("planes_recovered"), "planes": rec.get("planes"), "ari": rec.get("ari"),
                            "outliers_labelled": rec.get("outliers_labelled"), "outliers_generated": rec.get("outliers_generated")},
           "process_s": rec["total_s"], "process_s_second_call": rec.get("total_s_second_call"),
           "note": "process_s is the first Process() of a fresh process (it also pays for the HIP runtime and the code objects); "
                   "process_s_second_call is the same call repeated in that process, identical result",
           "digest": rec["digest"]}
    cl = rec.get("cpu_loop")
    if cl:
        out["cpu_baseline_loop"] = dict(cl, gpu_loop_s=rec["loop_s"], ratio=cl["value"] / max(rec["loop_s"], 1e-9))
    # the reference's OWN route: per-point HAF homographies, mean shift, 3-point fits hand hundreds of models to the loop
    rec3, err3 = run(0, INIT="stable")
    out["reference_route"] = ({"what": "MultiH::INIT_STABLE_SETS (EstablishStablePointSets, M/MultiH.cpp:604-694) instead of the DLT proposals",
                               "iterations_run": rec3.get("labeling_steps"), "clusters": rec3["clusters"], "energy": rec3["energy"],
                               "loop_s": rec3["loop_s"], "ms_per_iteration": per_iteration(rec3), "process_s": rec3["total_s"],
                               "process_s_second_call": rec3.get("total_s_second_call"), "planes_recovered": rec3.get("planes_recovered"),
                               "ari": rec3.get("ari"), "digest": rec3["digest"]}
                              if rec3 is not None else {"error": err3})
    # the same with a fresh batch of proposals in EVERY iteration (PEARL re-proposal on the points left unexplained)
    rec2, err2 = run(a.models)
    out["with_reproposal"] = ({"iter_hypotheses": a.models, "iterations_run": rec2.get("labeling_steps"), "clusters": rec2["clusters"], "energy": rec2["energy"],
                               "loop_s": rec2["loop_s"], "ms_per_iteration": per_iteration(rec2), "process_s": rec2["total_s"], "digest": rec2["digest"],
                               "planes_recovered": rec2.get("planes_recovered"), "ari": rec2.get("ari")}
                              if rec2 is not None else {"error": err2})
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of a
    parent that never touches torch or the GPU (no re-exec of a process that has).  Rank 0 inherits
    stdout, so its JSON line is this command's output.  Any failing rank ends the run non-zero."""
    import socket
    import subprocess

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import torch
    import torch.distributed as dist

    mh = importlib.import_module("multi-h_amd")
    sharding = importlib.import_module("multi-h_amd.sharding")
    if not torch.cuda.is_available() or mh.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # MH_BENCH_DEVICE / MH_BENCH_BACKEND exist only to rehearse the multi-rank path on a box with
    # fewer GPUs than ranks (all ranks on one device, gloo); the driver never sets them.
    if "MH_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MH_BENCH_DEVICE"])
    backend = os.environ.get("MH_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            except TypeError:                            # a torch without the device_id argument
                dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    thr, lam = 2.2, 0.5                      # harness defaults, M/main.cpp:55-59
    thr2 = thr * thr
    sc = mh.synth.make_scene(a.points, a.planes, seed=a.seed, with_neighbours=False)
    N = sc.n

    eng = mh.Engine(local_rank, 2.6, thr, 0.005, lam, 20)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_correspondences(sc.src, sc.dst, sc.aff)
    eng.set_epipolar(sc.F, sc.e2)

    # Multi-GPU transport of the score exchange.  On a real node: RCCL through the engine's native transport
    # (libmultih_rccl.so) — torch.distributed only carries the 128-byte unique id to the other ranks once, outside the
    # timed region.  MH_BENCH_BACKEND=gloo (several ranks rehearsing on ONE GPU, which RCCL refuses): the
    # host-synchronised torch.distributed hook instead.
    transport_kind = None
    keep = []
    native_comm = None
    if world > 1:
        import ctypes
        native_ok = False
        if backend == "nccl" and os.environ.get("MH_BENCH_TRANSPORT", "native") == "native":
            # every rank takes part in every step of the bootstrap, and the ranks agree on its outcome before anyone
            # depends on it: a rank that cannot load the library or create the communicator sends all of them to the hook
            status = torch.zeros(1, dtype=torch.int32, device=dev)
            try:
                rl = ctypes.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_rccl.so"))
                rl.mhr_last_error.restype = ctypes.c_char_p
            except OSError as ex:
                rl = None
                status += 1
                print(f"[bench] rank {rank}: libmultih_rccl.so: {ex}", file=sys.stderr, flush=True)
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0 and rl is not None:
                buf = (ctypes.c_ubyte * 128)()
                if rl.mhr_unique_id(buf) != 0:
                    status += 1
                    print("[bench] mhr_unique_id: " + rl.mhr_last_error().decode(), file=sys.stderr, flush=True)
                uid = torch.tensor(list(buf), dtype=torch.uint8)
            uid = uid.to(dev)
            dist.broadcast(uid, src=0)
            dist.all_reduce(status, op=dist.ReduceOp.MAX)
            if int(status.item()) == 0:
                raw = (ctypes.c_ubyte * 128)(*uid.cpu().tolist())
                comm = ctypes.c_void_p()
                rc = rl.mhr_init(ctypes.byref(comm), rank, world, raw, local_rank)      # collective (ncclCommInitRank)
                if rc != 0:
                    print(f"[bench] rank {rank}: mhr_init: " + rl.mhr_last_error().decode(), file=sys.stderr, flush=True)
                status += 1 if rc != 0 else 0
                dist.all_reduce(status, op=dist.ReduceOp.MAX)
                if int(status.item()) == 0:
                    eng.set_transport(rank, world, stream_fn=rl.mhr_allgather, ctx=comm)
                    keep += [rl, comm]
                    native_comm = (rl, comm)
                    transport_kind = "native RCCL (ncclAllGather on the engine's stream, libmultih_rccl.so)"
                    native_ok = True
        if native_ok:
            pass
        elif backend == "nccl":
            hook = sharding.make_allgather_hook(world, dev)
            eng.set_transport(rank, world, host_fn=hook)
            keep.append(hook)
            transport_kind = "torch.distributed all_gather_into_tensor over RCCL through the host-synchronised hook (native transport unavailable)"
        else:
            hook = sharding.make_allgather_hook(world, dev)
            eng.set_transport(rank, world, host_fn=hook)
            keep.append(hook)
            transport_kind = f"torch.distributed hook over {backend} (rehearsal)"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_mode(scaling: str, steps: int, warmup: int, pipelined: bool = False, profile: bool = True, models: int = 0,
                 mark_steps: bool = True):
        """`warmup` untimed steps, then exactly `steps` timed ones between two fences; max over ranks.
        models: batch size instead of --models (the one-GPU rehearsal of a strong split's per-rank shard)."""
        batch = models if models > 0 else a.models
        if scaling == "weak":
            sizes = [batch] * world                          # every GPU scores its own batch of M
        else:
            sizes = sharding.shard_counts(batch, world)      # one batch of M split across the GPUs (configs[3])
        M = sizes[rank]
        total = sum(sizes)

        def first_of(i: int) -> int:
            if scaling == "weak":
                return sharding.batch_first(i, world, rank, M)     # disjoint RNG counters per (step, rank)
            return i * batch + sharding.shard_range(batch, world, rank)[0]

        def step(i: int, last: bool = False):
            if pipelined:
                eng.adopt_prefetched()                           # batch i (its DLT was dispatched two sweeps ago)
                eng.prefetch_dlt4(a.seed, first_of(i + 2), M)    # batch i+2 on the second stream (the queue holds two batches)
            else:
                eng.propose_dlt4(a.seed, first_of(i), M)
            eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
            return eng.select_best(total, fetch=last)        # (all-gather +) arg-max on the engine's stream

        if pipelined:
            eng.prefetch_dlt4(a.seed, first_of(0), M)
            eng.prefetch_dlt4(a.seed, first_of(1), M)
        for i in range(warmup):
            step(i)
        fence()
        eng.profile_reset()
        eng.profile_enable(profile)              # (two marker packets around every kernel launch, like the step marks below)
        # A timing event at every step boundary is a marker packet on the engine's stream: about 25 us per step
        # (tools/enqueue_probe.py: 0.970 ms per 12 500-hypothesis step without them, 0.995-1.005 with).  Nothing at the 7.4 ms
        # steps of one GPU, 2.5 % at the 1 ms steps of an 8-GPU strong split — so with several ranks only the ends are marked.
        every = 1 if world == 1 and mark_steps else steps
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps // every + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        last = None
        for i in range(steps):
            last = step(warmup + i, last=(i == steps - 1))
            if (i + 1) % every == 0:
                marks[(i + 1) // every].record()
        fence()
        dt = time.perf_counter() - t0
        eng.profile_enable(False)
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) / every for i in range(len(marks) - 1))
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if pipelined:                            # the two batches still queued are not part of any step: drop them
            eng.adopt_prefetched()
            eng.adopt_prefetched()
        n_res, ms_res = eng.profile_get(1)       # MH_K_RESIDUAL
        n_dlt, ms_dlt = eng.profile_get(0)       # MH_K_DLT4 (on the second stream, beside the sweep)
        n_x, ms_x = eng.profile_get(7)           # MH_K_EXCHANGE: all-gather + arg-max on the exchange stream, from the sweep's end
        best, score = last
        import hashlib
        import numpy as np
        if world > 1:
            ptr, _ = eng.device_buffer(5)        # MH_BUF_GATHERED_SCORES: world x longest, -1 padded
            longest = max(sizes)
            g = torch.as_tensor(_DevView(ptr, world * longest, "<i4"), device=dev).cpu().numpy().reshape(world, longest)
            scores = np.concatenate([g[r, :sizes[r]] for r in range(world)])
        else:
            ptr, _ = eng.device_buffer(0)
            scores = torch.as_tensor(_DevView(ptr, M, "<i4"), device=dev).cpu().numpy()
        return {"sizes": sizes, "M": M, "dt": float(tt.item()), "dt_local": dt, "xchg_ms": ms_x / max(n_x, 1), "res_ms": ms_res / max(n_res, 1),
                "dlt_ms": ms_dlt / max(n_dlt, 1), "best_model": int(best), "best_score": int(score),
                "step_ms_median": per_step[len(per_step) // 2], "step_ms_min": per_step[0], "step_ms_max": per_step[-1],
                "scores_sha256": hashlib.sha256(np.ascontiguousarray(scores).tobytes()).hexdigest()[:16]}

    def run_sustained(keep_going, block: int = 100):
        """The pipelined step over and over while keep_going() — tens of seconds, the board in its thermal and power steady
        state — in blocks of `block` steps between host waits.  Reported beside the headline, never instead of it."""
        strong = a.scaling == "strong"
        M = sharding.shard_counts(a.models, world)[rank] if strong else a.models
        # this rank's RNG counters of step i (the same rule as run_mode's first_of)
        first = lambda i: (i * a.models + sharding.shard_range(a.models, world, rank)[0]) if strong else sharding.batch_first(i, world, rank, M)
        eng.prefetch_dlt4(a.seed, first(0), M)
        eng.prefetch_dlt4(a.seed, first(1), M)
        i, t_gpu, per_block = 0, 0.0, []
        while keep_going() or i == 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(block):
                eng.adopt_prefetched()
                eng.prefetch_dlt4(a.seed, first(i + 2), M)
                eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
                eng.select_best(M, fetch=False)
                i += 1
            torch.cuda.synchronize()
            eng.synchronize()
            per_block.append((time.perf_counter() - t0) / block * 1e3)
            t_gpu += per_block[-1] * block * 1e-3
        eng.adopt_prefetched()
        eng.adopt_prefetched()
        return {"steps": i, "seconds": t_gpu, "hypotheses_per_s": i * float(M) / t_gpu, "ms_per_step_first_block": per_block[0],
                "ms_per_step_last_block": per_block[-1], "ms_per_step_mean": t_gpu / i * 1e3,
                "what": "the same pipelined step repeated back to back while the one-core CPU baseline ran (blocks of 100 steps between host "
                        "waits): the rate the board sustains once its power controller has settled"}

    # One GPU: kernel events inside the timed region (the roofline's launch time is measured in the run it describes).  Several
    # ranks: the headline steps carry no timing markers at all, and the kernel times come from a short pass of their own.
    pipe = run_mode(a.scaling, a.steps, a.warmup, pipelined=True, profile=(world == 1))
    kernel_pass = None
    if world > 1:
        kernel_pass = run_mode(a.scaling, min(a.steps, 8), 1, pipelined=True, profile=True)
        pipe["res_ms"], pipe["dlt_ms"], pipe["xchg_ms"] = kernel_pass["res_ms"], kernel_pass["dlt_ms"], kernel_pass["xchg_ms"]
    other = None
    if world > 1 and not a.no_other_mode:
        other = run_mode("weak" if a.scaling == "strong" else "strong", a.steps, a.warmup, pipelined=True, profile=False)
        ko = run_mode("weak" if a.scaling == "strong" else "strong", min(a.steps, 8), 1, pipelined=True, profile=True)
        other["res_ms"], other["dlt_ms"] = ko["res_ms"], ko["dlt_ms"]
    seq = run_mode(a.scaling, a.steps, a.warmup, pipelined=False)
    # Which form is the headline.  One GPU: the SEQUENTIAL form — the sweep runs at the board's power cap, so a DLT beside it
    # is not free: it lengthens the sweep by what it would have cost alone (7.31 + 0.06 ms pipelined against 7.00 + 0.33 +
    # 0.04 ms in sequence, r05), the step is the same to 0.1 %, and in sequence the residual kernel's launch time — the
    # roofline's denominator — is the kernel's own.  Several GPUs: the PIPELINED form — a rank's step is 1 ms, and the
    # hand-over between the DLT and the sweep is 5 % of it.  The other form is timed in the same run and
    # reported beside the headline (`pipelined_form` / `sequential_form`).
    head, other_form = (seq, pipe) if world == 1 else (pipe, seq)
    M, sizes, dt = head["M"], head["sizes"], head["dt"]
    # What the transport is, read back from RCCL itself, and each rank's own figures — gathered ONCE, outside every timed
    # region: a multi-GPU number can then be decomposed into the ranks' sweeps, their waits in the exchange and their steps.
    transport = {"kind": transport_kind if world > 1 else "none (one GPU: no exchange; the arg-max runs on the engine's exchange stream)",
                 "rccl_ranks": None, "rccl_rank_of_rank0": None, "rccl_version": None}
    if native_comm is not None:
        transport["rccl_ranks"] = int(native_comm[0].mhr_count(native_comm[1]))
        transport["rccl_rank_of_rank0"] = int(native_comm[0].mhr_rank(native_comm[1]))
        transport["rccl_version"] = int(native_comm[0].mhr_version())
        if transport["rccl_ranks"] != world:
            raise SystemExit(f"bench.py: RCCL reports a communicator of {transport['rccl_ranks']} ranks, the launcher started {world}")
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "hypotheses_per_step": head["M"], "k_residual_ms": head["res_ms"], "exchange_ms": head.get("xchg_ms"),
                "step_ms": head["dt_local"] / a.steps * 1e3,
                "rccl_rank": int(native_comm[0].mhr_rank(native_comm[1])) if native_comm is not None else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    head_first = (sharding.shard_range(a.models, world, rank)[0] if a.scaling == "strong" else sharding.batch_first(0, world, rank, M))
    # One GPU only: the per-rank shards a strong split of this batch over 2 / 4 / 8 GPUs would hand a rank, stepped the same
    # way (no timing markers inside, 40 steps) — what the split can reach at best before any exchange between real ranks.
    shard_rehearsal = None
    if world == 1 and a.models >= 8 * 1024 and not a.no_rehearsal:
        ref = run_mode("strong", 20, 3, pipelined=True, profile=False, mark_steps=False)
        shard_rehearsal = {"what": "the same pipelined step on 1 GPU with this batch's per-rank shard of a 2 / 4 / 8-GPU strong split (no timing "
                                   "markers inside the timed steps); efficiency = (ms per step of the whole batch / ranks) / ms per step of the shard",
                           "whole_batch_ms_per_step": ref["dt"] / 20 * 1e3, "shards": []}
        for ranks in (2, 4, 8):
            r = run_mode("strong", 40, 5, pipelined=True, profile=False, models=a.models // ranks, mark_steps=False)
            ms = r["dt"] / 40 * 1e3
            shard_rehearsal["shards"].append({"ranks": ranks, "hypotheses": a.models // ranks, "ms_per_step": ms,
                                              "efficiency": ref["dt"] / 20 * 1e3 / ranks / ms})

    # Outside the timed region: the store-free fused score kernel and the s = 4 matrix (SURVEY §8(d) "fused score
    # kernel: not HBM-bound", "also report s = 4"), reported next to the headline for context.  They are timed on a
    # FULL batch proposed here — whatever ran last (the rehearsal's 12 500-hypothesis shard, for one) is not what
    # stays resident — and every figure below divides by the model count READ BACK from the engine.
    eng.propose_dlt4(a.seed, head_first, M)
    M_res = eng.model_count
    if M_res != M:
        raise SystemExit(f"bench.py: {M_res} models resident where the side metrics expect this rank's batch of {M}")

    def time_score():
        eng.score(thr2, fetch=False)
        eng.profile_reset()
        eng.profile_enable(True)
        for _ in range(5):
            eng.score(thr2, fetch=False)
        eng.synchronize()
        n_sc, ms_sc = eng.profile_get(2)     # MH_K_SCORE
        eng.profile_enable(False)
        assert eng.model_count == M_res
        return ms_sc / max(n_sc, 1)

    eng.set_tuning(15, 0)                    # the FP64 sweep for every pair (k_residual without the stores)
    fused_ms = time_score()
    eng.set_tuning(15, 1)                    # the product's score path: FP32 pre-test with a rigorous bound, FP64 for the doubtful pairs
    eng.score_stats(reset=True)
    pretest_ms = time_score()
    pre_pairs, pre_fp64 = eng.score_stats(reset=True)
    # ... and the s = 4 variant of the matrix: the int32 data cost of every hypothesis against every point
    eng.profile_reset()
    eng.profile_enable(True)
    for _ in range(4):
        eng.cost_matrix(fetch_C=False, fetch_counts=False)
    eng.synchronize()
    n_cm, ms_cm = eng.profile_get(6)         # MH_K_COSTMATRIX
    eng.profile_enable(False)
    cost_ms = ms_cm / max(n_cm, 1)
    eng.set_tuning(15, 0)                    # the FP64 formula for every pair (k_cost_matrix)
    try:
        eng.profile_reset()
        eng.profile_enable(True)
        for _ in range(3):
            eng.cost_matrix(fetch_C=False, fetch_counts=False)
        eng.synchronize()
        n_c64, ms_c64 = eng.profile_get(6)
        eng.profile_enable(False)
    finally:
        eng.set_tuning(15, 1)
    assert eng.model_count == M_res
    cost64_ms = ms_c64 / max(n_c64, 1)
    # ... and north_star's SYMMETRIC transfer error (MH_RESIDUAL_SYMMETRIC: d2 = ||H p1 - p2||^2 + ||adj(H) p2 - p1||^2, the same
    # 8-byte matrix; the reference has the forward form only, so this is an extension — tests/test_symmetric_exact.py checks its
    # definition in exact rationals): FP64-issue bound, 57 rounded operations per pair
    eng.set_residual_mode(True)
    try:
        eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
        eng.profile_reset()
        eng.profile_enable(True)
        for _ in range(3):
            eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
        eng.synchronize()
        n_sy, ms_sy = eng.profile_get(1)
        eng.profile_enable(False)
    finally:
        eng.set_residual_mode(False)
    sym_ms = ms_sy / max(n_sy, 1)
    cost_bytes = 4.0 * N * M_res + 32.0 * N + 72.0 * M_res + 4.0 * M_res
    avg_res_ms = head["res_ms"]
    alg_bytes = 8.0 * N * M + 32.0 * N + 72.0 * M + 4.0 * M
    achieved = alg_bytes / (avg_res_ms * 1e-3) / 1e9

    # What this box's memory takes from stores alone, measured in this run: hipMemset over the residual matrix itself (the 40 GB
    # the sweep has just written; a fill kernel of the runtime, no arithmetic).  Context for roofline.frac — the sweep computes
    # 28 FP64 operations per pair at the board's power cap on top of the same store stream.  A reference, not a ceiling.
    memset_GBps = None
    try:
        ptr_R, bytes_R = eng.device_buffer(2)            # MH_BUF_RESIDUALS
        if ptr_R and bytes_R >= 8 * N * M:
            t_R = torch.as_tensor(_DevView(ptr_R, int(bytes_R), "|u1"), device=dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            t_R.zero_()
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(3):
                t_R.zero_()
            ev[1].record()
            torch.cuda.synchronize()
            memset_GBps = float(bytes_R) * 3 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
    except Exception:
        memset_GBps = None

    fractions = {}

    def frac(name: str, value: float) -> float:
        """Every fraction of a peak the line prints goes through here: above 1 the timed kernel did not do the work the
        numerator states (a stale batch, a skipped launch) and the run fails instead of printing it."""
        fractions[name] = value
        if not (0.0 < value <= 1.0):
            raise SystemExit(f"bench.py: {name} = {value:.4f} is not a fraction of a peak: the timed region did not do the stated work")
        return value

    if rank == 0:
        total_hyp = float(sum(sizes)) * a.steps
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "residual_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("points") == N and tj.get("models") == M:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/residual_traffic.json (rocprofv3 --pmc passes of this command, tools/profile_bench.sh; not measured in this run)"
            except Exception:
                traffic = None
        which = ("BASELINE configs[2]" if (N, a.models, a.planes, world) == (50000, 100000, 10, 1) else
                 "BASELINE configs[3]" if (N, a.models, a.planes) == (50000, 100000, 10) and a.scaling == "strong" else
                 "BASELINE configs[1]" if (N, a.models, a.planes, world) == (5000, 10000, 3, 1) else "custom size")
        per = (f"one batch of {a.models} DLT hypotheses per step split over {world} GPU(s) ({M} on rank 0)"
               if a.scaling == "strong" else f"{M} DLT hypotheses per GPU per step")
        out = {
            "metric": "scored homography hypotheses/sec (50k pts x 100k models)",
            "value": total_hyp / dt,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{N} correspondences / {a.planes} planes, {per} ({which}); "
                                   f"propose+residual-matrix+score" + ("+all-gather" if world > 1 else "") + "+argmax",
                       "points": N, "models_per_step": int(sum(sizes)), "models_rank0": M, "planes": a.planes, "thr": thr,
                       "parallelism": f"hypothesis-sharded x{world}"},
            "residual_kernel_GBps": achieved,
            "pair_evals_per_s": total_hyp * N / dt,
            "step_ms": {"median": head["step_ms_median"], "min": head["step_ms_min"], "max": head["step_ms_max"],
                        "mean_wall": dt / a.steps * 1e3, "note": "HIP events on the engine's stream at every step boundary"},
            "form": "sequential (propose, sweep, arg-max on one stream)" if head is seq else "pipelined (the DLT of batch i+2 on the second stream beside sweep i)",
            "kernel_ms": {"k_residual": avg_res_ms, "k_dlt4_span_on_the_second_stream": pipe["dlt_ms"], "k_dlt4_alone": seq["dlt_ms"],
                          "k_score_fused_fp64": fused_ms, "k_score_fp32_pretest": pretest_ms, "k_cost_matrix_int32": cost_ms},
            "step_minus_residual_ms": head["step_ms_median"] - avg_res_ms,
            "kernel_ms_source": ("HIP events around every launch inside the timed region" if world == 1 else
                                 "a separate pass of 8 steps with HIP events around every launch; the headline steps carry no timing markers"),
            ("pipelined_form" if head is seq else "sequential_form"): {
                "what": ("the same steps software-pipelined: the DLT of batch i+2 on the engine's second stream beside sweep i (k_residual_ms is the "
                         "sweep WITH the DLT beside it)" if head is seq else
                         "the same steps with the four stages in sequence on one stream (no second stream)"),
                "value": float(sum(other_form["sizes"])) * a.steps / other_form["dt"], "ms_per_step": other_form["dt"] / a.steps * 1e3,
                "step_ms_median": other_form["step_ms_median"], "k_residual_ms": other_form["res_ms"], "k_dlt4_ms": other_form["dlt_ms"],
                "k_residual_frac_of_hbm_peak": frac("other_form.k_residual_frac_of_hbm_peak", alg_bytes / (other_form["res_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS),
                "scores_identical": other_form["scores_sha256"] == head["scores_sha256"]},
            "transport": transport,
            "per_rank": per_rank,
            "per_rank_note": ("k_residual_ms / exchange_ms: HIP events around every launch in a separate pass of 8 steps (exchange_ms: from the end of the batch's "
                              "sweep to the end of the arg-max behind the all-gather, on the exchange stream = the rank's wait for its peers + the wire time); "
                              "step_ms: this rank's own wall time over the headline steps (the headline takes the MAX over ranks)") if per_rank else None,
            "strong_split_rehearsal_on_one_gpu": shard_rehearsal,
            "fused_score_hypotheses_per_s_per_gpu": M_res / (pretest_ms * 1e-3),
            "fused_score": {"what": "mh_score on the same batch, no matrix written: FP32 pre-test with a rigorous error bound, FP64 formula only for the "
                                    "pairs it cannot decide (csrc/score32.hip); counts identical to the FP64 sweep's",
                            "models_resident": M_res, "ms": pretest_ms, "ms_fp64_sweep": fused_ms, "pairs_decided_in_fp64": pre_fp64 / max(pre_pairs, 1)},
            # fused score kernel, FP64 sweep: FP64-issue bound.  28 rounded FP64 operations per pair (M/MultiH.cpp:434-441 with two IEEE
            # divisions sharing one refined reciprocal) against the chip's FP64 vector issue rate at its 2.4 GHz maximum
            # (256 CUs x 4 SIMDs x 16 lanes per clock = 39.3 T operations/s, i.e. the 78.6 TFLOP/s spec counting an FMA as two)
            "fused_score_fp64": {"models_resident": M_res, "ops_per_pair": 28, "ops_per_s": 28.0 * N * M_res / (fused_ms * 1e-3),
                                 "peak_ops_per_s_at_2.4GHz": 256 * 4 * 16 * 2.4e9,
                                 "utilisation_vs_2.4GHz_peak": frac("fused_score_fp64.utilisation_vs_2.4GHz_peak", 28.0 * N * M_res / (fused_ms * 1e-3) / (256 * 4 * 16 * 2.4e9))},
            "cost_matrix_s4": {"what": "int32 PEARL data cost of every hypothesis against every point, materialised (mh_cost_matrix): "
                                       "the s = 4 variant of SURVEY 8(d), through the FP32 pre-test (csrc/score32.hip k_cost32: the constant for pairs "
                                       "proved beyond the truncation threshold, the reference's FP64 formula for the rest); ms_fp64_everywhere = "
                                       "the FP64 formula for every pair (k_cost_matrix: FP64-issue bound, a second IEEE division per pair)",
                               "models_resident": M_res, "ms": cost_ms, "ms_fp64_everywhere": cost64_ms, "algorithmic_bytes_per_launch": cost_bytes, "GBps": cost_bytes / (cost_ms * 1e-3) / 1e9,
                               "frac_of_hbm_peak": frac("cost_matrix_s4.frac_of_hbm_peak", cost_bytes / (cost_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS),
                               "frac_of_hbm_peak_fp64_everywhere": frac("cost_matrix_s4.frac_of_hbm_peak_fp64_everywhere", cost_bytes / (cost64_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)},
            "symmetric_transfer": {"what": "mh_residual_matrix in MH_RESIDUAL_SYMMETRIC mode on the same batch: forward + backward transfer error, R written "
                                           "(8 B per pair); 57 rounded FP64 operations per pair (2 x 28 + the sum), FP64-issue bound",
                                   "models_resident": M_res, "ms": sym_ms, "ops_per_pair": 57, "ops_per_s": 57.0 * N * M_res / (sym_ms * 1e-3),
                                   "utilisation_vs_2.4GHz_fp64_vector_peak": frac("symmetric_transfer.utilisation", 57.0 * N * M_res / (sym_ms * 1e-3) / (256 * 4 * 16 * 2.4e9)),
                                   "frac_of_hbm_peak": frac("symmetric_transfer.frac_of_hbm_peak", (8.0 * N * M_res) / (sym_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS),
                                   "hypotheses_per_s": M_res / (sym_ms * 1e-3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": frac("roofline.frac", achieved / HBM_PEAK_GBPS), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "k_residual", "algorithmic_bytes_per_launch": alg_bytes,
                         "measured_write_ceiling_GBps": HBM_WRITE_CEILING_GBPS,
                         "frac_of_measured_write_ceiling": frac("roofline.frac_of_measured_write_ceiling", achieved / HBM_WRITE_CEILING_GBPS),
                         # a same-box REFERENCE, not a ceiling (the runtime's fill kernel is not the fastest store stream: a ratio
                         # slightly above 1 is possible and has been measured since the sweep handles 64 models per work item)
                         "memset_of_R_on_this_box_GBps": memset_GBps,
                         "ratio_to_that_memset": (achieved / memset_GBps) if memset_GBps else None,
                         "models_per_launch": M},
            "best_model": head["best_model"], "best_score": head["best_score"], "scores_sha256": head["scores_sha256"],
        }
        if other is not None:
            o_total = float(sum(other["sizes"])) * a.steps
            out["weak_scaling" if a.scaling == "strong" else "strong_scaling"] = {
                "value": o_total / other["dt"], "unit": "hypotheses/s", "ms_per_step": other["dt"] / a.steps * 1e3,
                "models_per_step": int(sum(other["sizes"])), "kernel_ms": {"k_residual": other["res_ms"], "k_dlt4": other["dlt_ms"]},
                "steps": a.steps, "warmup": a.warmup}
        if world == 1 and not a.no_cpu_baseline:
            sustained = {}
            cb, Hs, cs = cpu_baseline(sc, thr2, a.cpu_sample, a.seed, beside=lambda alive: sustained.update(run_sustained(alive)))
            out["cpu_baseline"] = cb
            out["sustained"] = sustained
            # spot-check: the GPU scores the same sample identically (checker, outside the timed region)
            eng.set_models(Hs)
            import numpy as np
            assert np.array_equal(eng.score(thr2), cs), "GPU/oracle score mismatch"
            try:
                out["labeling"] = labeling_extra(mh, eng, a, thr2, lam)
                out["labeling_on_the_intermediate_scene"] = labeling_extra(mh, eng, a, thr2, lam, plane_separation=2.0)
                out["labeling_on_the_r04_scene"] = labeling_extra(mh, eng, a, thr2, lam, legacy=True)
            except Exception as ex:                      # context only: never lose the headline line
                out["labeling"] = {"error": repr(ex)}
            try:
                out["full_loop"] = full_loop_extra(a)
            except Exception as ex:
                out["full_loop"] = {"error": repr(ex)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    eng.close()
    if native_comm is not None:
        torch.cuda.synchronize()
        native_comm[0].mhr_destroy(native_comm[1])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
