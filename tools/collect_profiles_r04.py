#!/usr/bin/env python3
"""Copies what one tools/evidence_r04.sh session left under gpurun_out/ (scratch) into profiles/ (tracked).
   usage: tools/collect_profiles_r04.py <head-sha of the session>"""
import glob, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = sys.argv[1] if len(sys.argv) > 1 else "unknown"
G, P, E = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out", "ev4")


def cp(src, dst):
    shutil.copy(src, os.path.join(P, dst))


def stats_of(prof):
    s = open(os.path.join(G, prof, "summary.txt")).read()
    assert f"HEAD {sha}" in s.splitlines()[0], (prof, s.splitlines()[0])
    return os.path.join(G, prof, re.search(r"kernel stats: (\S+)", s).group(1))


def strip(src, dst, drop=r"^\[Multi-H\]|^Iteration|^Found|^RCCL version|^HIP version|^ROCm version|^Hostname|^Librccl|amdgpu.ids"):
    with open(os.path.join(P, dst), "w") as f:
        f.writelines(l for l in open(src, errors="replace") if not re.search(drop, l))


cp(os.path.join(E, "bench.json"), "r04_bench.json")
cp(os.path.join(G, "prof_r04", "bench_under_trace.json"), "r04_bench_under_trace.json")
cp(os.path.join(G, "prof_r04", "summary.txt"), "r04_rocprof_summary.txt")
cp(stats_of("prof_r04"), "r04_kernel_stats.csv")
cp(os.path.join(G, "prof_r04", "residual_traffic.json"), "residual_traffic.json")
strip(os.path.join(E, "shard_proxy.txt"), "r04_shard_proxy.txt", drop=r"^RCCL version|^HIP version|^ROCm version|^Hostname|^Librccl|amdgpu.ids|^\{")
with open(os.path.join(P, "r04_shard_proxy.txt"), "a") as f:
    f.write("\n-- A/B: hardware dispatch (headroom -1) vs resident grid, with and without the gate that holds the sweep behind the DLT's dispatch\n")
    f.writelines(l for l in open(os.path.join(E, "shard_proxy_ab.txt"), errors="replace") if l.startswith("M ") or l.startswith("== workgroup"))
    f.write("\n-- A/B: ONE batch prepared ahead (DEPTH=1: the sweep is held until the second stream has reached the DLT's dispatch) instead of two\n")
    keep = False
    for l in open(os.path.join(E, "shard_proxy_depth1.txt"), errors="replace"):
        if l.startswith("== "): keep = True
        if keep and not l.startswith("{") and not re.search(r"^RCCL version|^HIP version|^ROCm version|^Hostname|^Librccl|amdgpu.ids", l): f.write(l)
    if os.path.exists(os.path.join(E, "shard_proxy_dltform.txt")):
        f.write("\n-- A/B: which form of the DLT proposer is prefetched beside the sweep (tools/shard_proxy.py DLTFORM=1)\n")
        f.writelines(open(os.path.join(E, "shard_proxy_dltform.txt"), errors="replace"))
    f.write("\n-- tools/enqueue_probe.py (two batches ahead, no timing events on the stream)\n")
    f.writelines(l for l in open(os.path.join(E, "enqueue_probe.txt"), errors="replace") if l.startswith("M "))
    f.write(f"source: HEAD {sha}\n")
for m in (12500, 100000):
    cp(os.path.join(E, f"timeline_{m}.txt"), f"r04_timeline_{m}.txt")
cp(os.path.join(E, "label_bench.txt"), "r04_label_bench.txt")
cp(os.path.join(G, "prof_r04_label", "summary.txt"), "r04_labeling_summary.txt")
strip(os.path.join(E, "core_components.txt"), "r04_core_components.txt")
strip(os.path.join(E, "barrsmith.txt"), "r04_barrsmith_agreement.txt", drop=r"^\[Multi-H\]|^Iteration|^Found|^\{")
for a, b in (("at_size_init.txt", "r04_at_size_alternation.txt"), ("at_size_dlt.txt", "r04_at_size_dlt_route.txt")):
    strip(os.path.join(E, a), b)
for a, b in (("loop_timing.txt", "r04_loop_timing.txt"), ("loop_timing_reference_init.txt", "r04_loop_timing_reference_init.txt"),
             ("loop_reproposal.txt", "r04_loop_reproposal.txt"), ("score_bench.txt", "r04_score_bench.txt")):
    cp(os.path.join(E, a), b)
with open(os.path.join(P, "r04_small_scenes.txt"), "w") as f:
    f.writelines(l[l.index("== N="):] for l in open(os.path.join(E, "small_scenes.txt"), errors="replace") if "== N=" in l)


for a, b, key in (("dlt_probe.txt", "r04_dlt_probe.txt", "M ="), ("meanshift_probe.txt", "r04_meanshift_session.txt", "N =")):
    if os.path.exists(os.path.join(E, a)):
        with open(os.path.join(P, b), "w") as f:
            f.writelines(l for l in open(os.path.join(E, a), errors="replace") if key in l)
            f.write(f"source: HEAD {sha}\n")


def last(path, pattern):
    hits = [m.group(0).strip() for l in open(path, errors="replace") for m in [re.search(pattern, l)] if m]
    return hits[-1] if hits else "(missing)"


with open(os.path.join(P, "r04_stress.txt"), "w") as f:
    f.write("tools/stress_parity.py SECONDS=150 SEED=4:\n" + last(os.path.join(E, "stress_parity.txt"), r"stress ok.*") + "\n")
    f.write("tools/stress_alternation.py SECONDS=200 SEED=4 (whole Process(), post-filter statistics from the engine):\n"
            + last(os.path.join(E, "stress_process.txt"), r"Process\(\) stress ok.*") + "\n")
    f.write("tools/stress_residual_edges.py SECONDS=90 SEED=4 (with pairs on the cost's truncation threshold):\n"
            + last(os.path.join(E, "stress_residual_edges.txt"), r"residual edge stress ok.*") + "\n")
    f.write("pytest -m gpu: " + last(os.path.join(E, "tests.log"), r"\d+ passed.*") + "\n")
    f.write(f"source: HEAD {sha}\n")
print(open(os.path.join(P, "r04_stress.txt")).read())
print(open(os.path.join(P, "r04_rocprof_summary.txt")).read().splitlines()[0])
