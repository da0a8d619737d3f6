#!/usr/bin/env python3
"""Driver for the PMC passes over k_cost32_resident: 100 000 DLT hypotheses x 50 000 points, the int32
cost matrix LAUNCHES times, nothing else of size on the device.  Run directly under rocprofv3 (the program after `--`):
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY ... -- python3 tools/cost32_pmc_driver.py
tools/cost32_pmc_summary.py turns the counter_collection.csv files into per-launch figures."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
N, M, LAUNCHES = int(os.environ.get("N", 50000)), int(os.environ.get("M", 100000)), int(os.environ.get("LAUNCHES", 4))
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
for _ in range(LAUNCHES):
    e.cost_matrix(fetch_C=False, fetch_counts=False)
e.synchronize()
if os.environ.get("ALSO_STORE_ONLY"):            # the same 20 GB written by hipMemset: what the memory takes from stores alone
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ptr, nbytes = e.device_buffer(2)              # MH_BUF_COST_MATRIX
    for _ in range(LAUNCHES):
        hip.hipMemset(ctypes.c_void_p(ptr), 0, ctypes.c_size_t(nbytes))
    hip.hipDeviceSynchronize()
e.close()
