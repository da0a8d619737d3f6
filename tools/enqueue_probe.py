#!/usr/bin/env python3
"""Host enqueue cost of the pipelined step against the GPU time per step (no timing events on the stream): is the host ahead?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mh = importlib.import_module("multi-h_amd")
torch.cuda.set_device(0)
sc = mh.synth.make_scene(50000, 10, seed=1234, with_neighbours=False)
eng = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_correspondences(sc.src, sc.dst, sc.aff)
thr2 = 2.2 ** 2
for M in (12500, 100000):
    eng.prefetch_dlt4(1234, 0, M)
    eng.prefetch_dlt4(1234, M, M)                      # two batches ahead
    for rep in range(3):
        torch.cuda.synchronize(); eng.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            eng.adopt_prefetched(); eng.prefetch_dlt4(1234, (i + 2) * M, M); eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False); eng.select_best(M, fetch=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); eng.synchronize()
        t2 = time.perf_counter()
        print(f"M {M}: host enqueue {(t1 - t0) / 200 * 1e3:.4f} ms per step, GPU done after {(t2 - t0) / 200 * 1e3:.4f} ms per step", flush=True)
    eng.adopt_prefetched(); eng.adopt_prefetched()     # drain the queue before the next size
eng.close()
