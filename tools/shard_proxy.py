#!/usr/bin/env python3
"""One-GPU proxy of BASELINE configs[3]'s strong split: the bench step — adopt the prefetched batch,
prefetch the next one on the second stream, materialise R with fused counts, (all-gather +) arg-max — at the per-rank
shard sizes of N = 1, 2, 4, 8 GPUs (100 000 / 50 000 / 25 000 / 12 500 hypotheses x 50 000 points), with no transport
and with the native RCCL transport on a ONE-RANK communicator (the real ncclAllGather on the engine's streams; a
one-rank gather moves no bytes over xGMI, so this measures the launch/enqueue cost of the exchange, not the wire).
Prints, per size: median step (HIP events at every step boundary), the residual kernel's mean launch time, what the
step costs beyond the kernel, and the efficiency against the full batch = (step(100k) * M / 100k) / step(M).

PSPLIT=a,b,c with MH_LIB=multi-h_amd/libmultih_hip_tuning.so adds a sweep of forced point splits of k_residual."""
import ctypes
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the engine: both bind to the same HIP runtime)

mh = importlib.import_module("multi-h_amd")
N = int(os.environ.get("N", 50000))
SIZES = [int(x) for x in os.environ.get("SIZES", "100000,50000,25000,12500").split(",")]
STEPS, WARM = int(os.environ.get("STEPS", 40)), int(os.environ.get("WARM", 5))
PSPLIT = [int(x) for x in os.environ.get("PSPLIT", "").split(",") if x]
HEADROOM = [int(x) for x in os.environ.get("HEADROOM", "").split(",") if x]
DEPTH = int(os.environ.get("DEPTH", 2))          # batches prepared ahead of the sweep (1: the r03 pipeline, 2: r04)
thr2 = 2.2 ** 2

torch.cuda.set_device(0)
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
eng = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_correspondences(sc.src, sc.dst, sc.aff)


def run(M, label):
    def step(i, last=False):
        eng.adopt_prefetched()
        eng.prefetch_dlt4(1234, (i + DEPTH) * M, M)
        eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
        return eng.select_best(M, fetch=last)

    def timed(profile):
        """STEPS steps between two host waits; HIP events of torch at every step boundary.  profile=True also brackets
        every kernel launch with the engine's own timing events (two more markers per launch on the stream)."""
        for d in range(DEPTH):
            eng.prefetch_dlt4(1234, d * M, M)
        for i in range(WARM):
            step(i)
        torch.cuda.synchronize()
        eng.synchronize()
        eng.profile_reset()
        eng.profile_enable(profile)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(STEPS + 1)]
        marks[0].record()
        for i in range(STEPS):
            best = step(WARM + i, last=(i == STEPS - 1))
            marks[i + 1].record()
        torch.cuda.synchronize()
        eng.synchronize()
        eng.profile_enable(False)
        per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(STEPS))
        for d in range(DEPTH):                       # drain the queue: the next run primes it again
            eng.adopt_prefetched()
        return per, best

    per, best = timed(False)                    # the step as a job runs it: no timing events around the kernels
    per_p, best_p = timed(True)                 # the kernel's own launch time
    assert best == best_p
    n_res, ms_res = eng.profile_get(1)
    return {"label": label, "M": M, "step_ms": per[len(per) // 2], "step_min": per[0], "step_ms_with_kernel_events": per_p[len(per_p) // 2],
            "k_residual_ms": ms_res / max(n_res, 1), "best": best}


def table(rows, title):
    print(f"== {title}")
    full = next((r for r in rows if r["M"] == 100000), rows[0])
    print(f"{'M':>8} {'step ms':>9} {'min':>8} {'k_residual':>11} {'step-kern':>10} {'ideal':>8} {'efficiency':>10}")
    for r in rows:
        ideal = full["step_ms"] * r["M"] / full["M"]
        r["efficiency"] = ideal / r["step_ms"]
        print(f"{r['M']:8d} {r['step_ms']:9.4f} {r['step_min']:8.4f} {r['k_residual_ms']:11.4f} {r['step_ms'] - r['k_residual_ms']:10.4f} "
              f"{ideal:8.4f} {r['efficiency']:10.4f}", flush=True)


out = {"points": N, "steps": STEPS}
rows = [run(M, "no transport") for M in SIZES]
table(rows, "no transport (arg-max only)")
out["no_transport"] = rows

rl = ctypes.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH) if "MH_LIB" not in os.environ else os.path.join(ROOT, "multi-h_amd"),
                              "libmultih_rccl.so"))
rl.mhr_last_error.restype = ctypes.c_char_p
buf = (ctypes.c_ubyte * 128)()
comm = ctypes.c_void_p()
if rl.mhr_unique_id(buf) == 0 and rl.mhr_init(ctypes.byref(comm), 0, 1, buf, 0) == 0:
    eng.set_transport(0, 1, stream_fn=rl.mhr_allgather, ctx=comm)
    rows2 = [run(M, "rccl 1 rank") for M in SIZES]
    table(rows2, "native RCCL transport, one-rank communicator (pad -> ncclAllGather -> arg-max -> publish)")
    out["rccl_one_rank"] = rows2
    for a, b in zip(rows, rows2):
        assert a["best"] == b["best"], (a, b)
    eng.set_transport(0, 1)
else:
    print("RCCL communicator unavailable:", rl.mhr_last_error().decode())

if HEADROOM:
    print("== workgroup slots the resident sweep leaves free for the DLT (mh_set_tuning key 19; -1 = hardware dispatch)")
    for M in SIZES:
      for first in (1, 0):
        eng.set_tuning(20, first)
        for h in HEADROOM:
            eng.set_tuning(19, h)
            r = run(M, f"headroom {h}")
            print(f"M {M:7d} dlt-first {first} headroom {h:4d}: step {r['step_ms']:.4f} ms (min {r['step_min']:.4f})  k_residual {r['k_residual_ms']:.4f} ms", flush=True)
        eng.set_tuning(19, 0)
        eng.set_tuning(20, 1)
if os.environ.get("DLTFORM"):
    print("== the prefetched DLT's form (mh_set_tuning key 25): 1 = LDS-staged, 72 registers (fits beside five sweep waves per SIMD); 2 = columns in registers, 128 registers")
    for M in SIZES:
        for form in (1, 2, 1, 2):
            eng.set_tuning(25, form)
            r = run(M, f"dlt form {form}")
            print(f"M {M:7d} DLT form {form}: step {r['step_ms']:.4f} ms (min {r['step_min']:.4f})  k_residual {r['k_residual_ms']:.4f} ms", flush=True)
    eng.set_tuning(25, 0)
if PSPLIT:
    print("== forced point splits of k_residual (tuning library)")
    for M in SIZES:
        for ps in PSPLIT:
            eng.set_tuning(0, 400 + ps)
            r = run(M, f"psplit {ps}")
            print(f"M {M:7d} psplit {ps:3d}: step {r['step_ms']:.4f} ms  k_residual {r['k_residual_ms']:.4f} ms", flush=True)
        eng.set_tuning(0, 0)
print(json.dumps(out))
eng.close()
if comm:
    torch.cuda.synchronize()
    rl.mhr_destroy(comm)
