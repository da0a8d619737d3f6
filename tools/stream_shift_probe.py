#!/usr/bin/env python3
"""Experiment (r04): does the starvation of a kernel dispatched while the resident sweep runs depend on WHICH hardware queue /
pipe the second stream lands on?  A fresh engine per setting: `shift` dummy streams are created in front of the engine's second
stream (mh_set_tuning key 22), the gate that holds the sweep behind the DLT's dispatch is switched OFF (key 20 = 0), and the
pipelined step is timed at the 8-GPU shard size.  If some shift brings the step down to the gated one's without the gate, the
queues' placement is what starves the DLT."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mh = importlib.import_module("multi-h_amd")
N, M, STEPS, WARM = 50000, int(os.environ.get("M", 12500)), 40, 5
thr2 = 2.2 ** 2
torch.cuda.set_device(0)
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
own = os.environ.get("OWN_STREAM") == "1"
for shift in [int(x) for x in os.environ.get("SHIFTS", "0,1,2,3,4,5,6,7").split(",")]:
    for gate in (0, 1):
        eng = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
        if not own:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_correspondences(sc.src, sc.dst, sc.aff)
        eng.set_tuning(22, shift)
        eng.set_tuning(20, gate)
        eng.prefetch_dlt4(1234, 0, M)
        times = []
        import time
        for rep in range(3):
            for i in range(WARM):
                eng.adopt_prefetched(); eng.prefetch_dlt4(1234, (i + 1) * M, M); eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False); eng.select_best(M, fetch=False)
            eng.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(STEPS):
                eng.adopt_prefetched(); eng.prefetch_dlt4(1234, (i + 1) * M, M); eng.residual_matrix(thr2, fetch_R=False, fetch_counts=False); eng.select_best(M, fetch=False)
            eng.synchronize(); torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / STEPS * 1e3)
        print(f"shift {shift} gate {gate} {'own stream' if own else 'torch stream'}: step {min(times):.4f} ms (three runs: {', '.join(f'{t:.4f}' for t in times)})", flush=True)
        eng.close()
