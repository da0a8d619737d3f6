#!/usr/bin/env python3
"""What separates two sweeps of the 12 500-hypothesis shard step (40 us of a 965 us step)?  The step is stripped down one
stream operation at a time and timed without markers inside (host clock around STEPS steps, like tools/enqueue_probe.py):
  full        adopt -> prefetch(i+2) -> sweep -> select_best (enqueue only)          [the bench step]
  no-select   adopt -> prefetch(i+2) -> sweep
  no-prefetch sweep -> select_best (enqueue only)      (the same models every step)
  sweeps      sweep only, back to back"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mh = importlib.import_module("multi-h_amd")
M, N, STEPS = int(os.environ.get("M", 12500)), 50000, int(os.environ.get("STEPS", 400))
thr2 = 2.2 ** 2
torch.cuda.set_device(0)
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
eng = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_correspondences(sc.src, sc.dst, sc.aff)


def timed(step, prime=None, drain=None):
    if prime: prime()
    for i in range(20): step(i)
    torch.cuda.synchronize(); eng.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS): step(20 + i)
    torch.cuda.synchronize(); eng.synchronize()
    dt = (time.perf_counter() - t0) / STEPS * 1e3
    if drain: drain()
    return dt


def prime():
    eng.prefetch_dlt4(1, 0, M); eng.prefetch_dlt4(1, M, M)
def drain():
    eng.adopt_prefetched(); eng.adopt_prefetched()
def full(i):
    eng.adopt_prefetched(); eng.prefetch_dlt4(1, (i + 2) * M, M)
    eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False); eng.select_best(M, fetch=False)
def no_select(i):
    eng.adopt_prefetched(); eng.prefetch_dlt4(1, (i + 2) * M, M)
    eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
def no_prefetch(i):
    eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False); eng.select_best(M, fetch=False)
def sweeps(i):
    eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False)

eng.propose_dlt4(1, 0, M)
eng.profile_reset(); eng.profile_enable(True)
for _ in range(20): sweeps(0)
eng.synchronize(); n, ms = eng.profile_get(1); eng.profile_enable(False)
k = ms / n
print(f"M {M}: k_residual alone (events around the launch) {k:.4f} ms")
for name, fn, p, d in (("sweeps", sweeps, None, None), ("no-prefetch", no_prefetch, None, None), ("no-select", no_select, prime, drain), ("full", full, prime, drain),
                       ("sweeps", sweeps, None, None), ("full", full, prime, drain)):
    if p is None: eng.propose_dlt4(1, 0, M)
    dt = timed(fn, p, d)
    print(f"  {name:12s} {dt:.4f} ms per step  (+{(dt - k) * 1e3:6.1f} us beyond the kernel)", flush=True)
eng.close()
