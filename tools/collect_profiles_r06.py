#!/usr/bin/env python3
"""Copies what one tools/evidence_r06.sh session left under gpurun_out/ (scratch) into profiles/ (tracked).
   usage: tools/collect_profiles_r06.py <head-sha of the session>"""
import os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = sys.argv[1] if len(sys.argv) > 1 else "unknown"
G, P, E = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out", "ev6")
DROP = r"^\[Multi-H\]|^Iteration|^Found|^RCCL version|^HIP version|^ROCm version|^Hostname|^Librccl|amdgpu.ids|^Median"


def strip(src, dst, drop=DROP, keep_timing=False):
    if not os.path.exists(src):
        print("missing", src); return
    with open(os.path.join(P, dst), "w") as f:
        for l in open(src, errors="replace"):
            if keep_timing and l.startswith("[Multi-H] iteration"):
                f.write(l); continue
            if not re.search(drop, l):
                f.write(l)
        f.write(f"source: HEAD {sha}\n")


def cp(src, dst):
    if os.path.exists(src): shutil.copy(src, os.path.join(P, dst))
    else: print("missing", src)


s = open(os.path.join(G, "prof_r06", "summary.txt")).read()
assert f"HEAD {sha}" in s.splitlines()[0], s.splitlines()[0]
cp(os.path.join(E, "bench.json"), "r06_bench.json")
cp(os.path.join(G, "prof_r06", "bench_under_trace.json"), "r06_bench_under_trace.json")
cp(os.path.join(G, "prof_r06", "summary.txt"), "r06_rocprof_summary.txt")
cp(os.path.join(G, "prof_r06", re.search(r"kernel stats: (\S+)", s).group(1)), "r06_kernel_stats.csv")
cp(os.path.join(G, "prof_r06", "residual_traffic.json"), "residual_traffic.json")
strip(os.path.join(E, "label_bench.txt"), "r06_label_bench.txt")
strip(os.path.join(E, "label_bench_intermediate_scene.txt"), "r06_label_bench_intermediate_scene.txt")
strip(os.path.join(E, "label_bench_r04_scene.txt"), "r06_label_bench_r04_scene.txt")
strip(os.path.join(E, "loop_timing.txt"), "r06_loop_timing.txt", keep_timing=True)
strip(os.path.join(E, "loop_timing_reference_route.txt"), "r06_loop_timing_reference_route.txt", keep_timing=True)
strip(os.path.join(E, "loop_timing_reference_route_sequential_moves.txt"), "r06_loop_timing_reference_route_sequential_moves.txt", keep_timing=True)
strip(os.path.join(E, "loop_timing_reference_route_20000.txt"), "r06_loop_timing_reference_route_20000.txt", keep_timing=True)
strip(os.path.join(E, "loop_timing_reference_route_20000_sequential_moves.txt"), "r06_loop_timing_reference_route_20000_sequential_moves.txt", keep_timing=True)
strip(os.path.join(E, "batch_probe.txt"), "r06_batch_probe.txt")
strip(os.path.join(E, "batch_trace_probe.txt"), "r06_batch_trace_probe.txt")
strip(os.path.join(E, "barrsmith.txt"), "r06_barrsmith_agreement.txt", drop=DROP + r"|^\{")
strip(os.path.join(E, "at_size_init.txt"), "r06_at_size_alternation.txt")
strip(os.path.join(E, "at_size_dlt.txt"), "r06_at_size_alternation_dlt_route.txt")
strip(os.path.join(E, "stress_parity.txt"), "r06_stress_parity.txt")
strip(os.path.join(E, "tests.log"), "r06_gpu_tests.txt")
if os.path.exists(os.path.join(E, "small_scenes.txt")):
    with open(os.path.join(P, "r06_small_scenes.txt"), "w") as f:
        f.writelines(l[l.index("== N="):] for l in open(os.path.join(E, "small_scenes.txt"), errors="replace") if "== N=" in l)
        f.write(f"source: HEAD {sha}\n")
print("profiles/r06_* written from", E)
