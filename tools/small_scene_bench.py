#!/usr/bin/env python3
"""Process() on small scenes (the sizes of real image pairs: a few hundred to a few thousand correspondences), second call
of a process (the first pays for the HIP runtime), with MULTIH_TIMING=1 for the stages.  Env: CASES=NxKxHYP,...;
TUNE=key:value,... (mh_set_tuning knobs for the engines of the runs, e.g. 10:3); INITS=4,-1 (DLT batch / stable sets)."""
import ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
host = C.CDLL(os.path.join(ROOT, "multi-h_amd", "libmultih_host.so"))
host.mhh_set_device(0)
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    host.mhh_set_engine_tuning(int(kv.split(":")[0]), int(kv.split(":")[1]))
dp = C.POINTER(C.c_double)
cases = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("CASES", "500x2x2000,2000x3x5000,5000x3x10000,20000x6x50000").split(",")]
for N, K, HYP in cases:
    sc = mh.synth.make_scene(N, K, seed=1234, with_neighbours=False)
    src, dst, aff, F, e2 = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff, sc.F, sc.e2))
    for init in [int(v) for v in os.environ.get("INITS", "4,-1").split(",")]:
        for rep in range(3):
            labels = np.full(N, -7, dtype=np.int32); Hout = np.zeros((256, 9))
            it, en, secs = C.c_int(0), C.c_double(0), C.c_double(0)
            t0 = time.time()
            k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), N,
                                     F.ctypes.data_as(dp), e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(2.2),
                                     C.c_double(0.005), C.c_double(0.5), 20, C.c_ulonglong(1234), HYP, 32, 20,
                                     None, 0, labels.ctypes.data_as(C.POINTER(C.c_int)), Hout.ctypes.data_as(dp), 256,
                                     C.byref(it), C.byref(en), C.byref(secs), 0, init)
            wall = time.time() - t0
            sys.stdout.flush()
            print(f"== N={N} planes={K} hypotheses={HYP} init={'DLT batch' if init == 4 else 'reference (stable sets)'} call {rep}: clusters {k}, iterations {it.value}, "
                  f"loop {secs.value * 1e3:.1f} ms, Process() {wall * 1e3:.1f} ms", flush=True)
