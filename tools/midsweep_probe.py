#!/usr/bin/env python3
"""Experiment (r04): what happens to a kernel that is dispatched WHILE the resident residual sweep runs?  Engine A sweeps
50k x 100k (7.4 ms) on one stream; 2 ms into it, from other streams: (a) a tiny torch kernel, (b) engine B's k_dlt4 with 64 /
1 024 / 12 500 / 100 000 hypotheses (1 / 16 / 196 / 1 563 workgroups of 4 waves, 72 registers, 78 KB of LDS each).  Reported: how
long after its launch each finishes, against the time the sweep still had to run.  Also with the hardware-dispatched
sweep (key 19 = -1) and with workgroup slots left free (key 19 = 256)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
mh = importlib.import_module("multi-h_amd")
torch.cuda.set_device(0)
sc = mh.synth.make_scene(50000, 10, seed=1234, with_neighbours=False)
A = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
A.set_stream(torch.cuda.current_stream().cuda_stream)
A.set_correspondences(sc.src, sc.dst, sc.aff)
A.propose_dlt4(1, 0, 100000)
B = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)            # its own stream
B.set_correspondences(sc.src, sc.dst, sc.aff)
thr2 = 2.2 ** 2
s2 = torch.cuda.Stream(priority=-1)
x = torch.zeros(1024, device="cuda")
for _ in range(3):
    A.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
    B.propose_dlt4(2, 0, 12500)
torch.cuda.synchronize(); B.synchronize()


def trial(what, headroom):
    A.set_tuning(19, headroom)
    A.residual_matrix(thr2, fetch_R=False, fetch_counts=False); torch.cuda.synchronize()
    end = torch.cuda.Event(); done = torch.cuda.Event()
    t_launch = time.perf_counter()
    A.residual_matrix(thr2, fetch_R=False, fetch_counts=False)
    end.record()
    time.sleep(0.002)
    t0 = time.perf_counter()
    if what == "torch":
        with torch.cuda.stream(s2):
            x.add_(1.0)
            done.record()
        done.synchronize()
    else:
        B.propose_dlt4(2, 0, int(what))
        B.synchronize()
    t1 = time.perf_counter()
    end.synchronize()
    t2 = time.perf_counter()
    return (t0 - t_launch) * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3


for headroom in (0, 256, -1):
    for what in ("torch", "64", "1024", "12500", "100000"):
        r = [trial(what, headroom) for _ in range(5)]
        r.sort(key=lambda t: t[1])
        a, b, c = r[len(r) // 2]
        print(f"sweep form {'hardware dispatch' if headroom < 0 else f'resident, {headroom} slots free':24s} | launched {a:.2f} ms into the sweep: "
              f"{('tiny torch kernel' if what == 'torch' else 'k_dlt4 x ' + what):18s} done after {b:7.3f} ms; the sweep ended after {c:7.3f} ms", flush=True)
A.set_tuning(19, 0)
# reference: the same kernels with nothing beside them
for what in ("torch", "64", "1024", "12500", "100000"):
    torch.cuda.synchronize(); B.synchronize()
    t0 = time.perf_counter()
    if what == "torch":
        with torch.cuda.stream(s2):
            x.add_(1.0)
        s2.synchronize()
    else:
        B.propose_dlt4(2, 0, int(what)); B.synchronize()
    print(f"alone: {what:8s} {(time.perf_counter() - t0) * 1e3:7.3f} ms (launch + run + host wait)")
A.close(); B.close()
