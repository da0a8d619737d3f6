#!/usr/bin/env python3
"""Copies what one tools/evidence_r03.sh session left under gpurun_out/ (scratch) into profiles/ (tracked).
   usage: tools/collect_profiles.py <head-sha of the session>
gpurun merges a session's files into gpurun_out/ without removing older ones, so for the rocprof directories the file the
session's own summary names is taken."""
import glob, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = sys.argv[1] if len(sys.argv) > 1 else "unknown"
G, P, E = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out", "ev")


def cp(src, dst):
    shutil.copy(src, os.path.join(P, dst))


def stats_of(prof):
    s = open(os.path.join(G, prof, "summary.txt")).read()
    assert f"HEAD {sha}" in s.splitlines()[0], (prof, s.splitlines()[0])
    return os.path.join(G, prof, re.search(r"kernel stats: (\S+)", s).group(1))


cp(os.path.join(E, "bench.json"), "r03_bench.json")
cp(os.path.join(G, "prof_r03", "bench_under_trace.json"), "r03_bench_under_trace.json")
cp(os.path.join(G, "prof_r03", "summary.txt"), "r03_rocprof_summary.txt")
cp(stats_of("prof_r03"), "r03_kernel_stats.csv")
cp(os.path.join(G, "prof_r03", "residual_traffic.json"), "residual_traffic.json")
cp(os.path.join(E, "energy.json"), "r03_energy.json")
cp(os.path.join(E, "label_bench.txt"), "r03_label_bench.txt")
cp(os.path.join(G, "prof_r03_label", "summary.txt"), "r03_labeling_summary.txt")
cp(stats_of("prof_r03_label"), "r03_labeling_kernel_stats.csv")
for a, b in (("loop_timing.txt", "r03_loop_timing.txt"), ("loop_timing_reference_init.txt", "r03_loop_timing_reference_init.txt"),
             ("loop_reproposal.txt", "r03_loop_reproposal.txt"), ("score_bench.txt", "r03_score_bench.txt"),
             ("cascade_sweep.txt", "r03_cascade_sweep.txt")):
    cp(os.path.join(E, a), b)
with open(os.path.join(P, "r03_small_scenes.txt"), "w") as f:
    f.writelines(l[l.index("== N="):] for l in open(os.path.join(E, "small_scenes.txt"), errors="replace") if "== N=" in l)


def last(path, pattern):
    hits = [m.group(0).strip() for l in open(path, errors="replace") for m in [re.search(pattern, l)] if m]
    return hits[-1] if hits else "(missing)"


with open(os.path.join(P, "r03_stress.txt"), "w") as f:
    f.write("tools/stress_parity.py SECONDS=150 SEED=3:\n" + last(os.path.join(E, "stress_parity.txt"), r"stress ok.*") + "\n")
    line = last(os.path.join(E, "stress_process.txt"), r"Process\(\) stress ok.*")
    f.write("tools/stress_alternation.py SECONDS=200 SEED=3 (whole Process(), post-filter statistics from the engine):\n" + line + "\n")
    f.write("tools/stress_residual_edges.py SECONDS=60:\n" + last(os.path.join(E, "stress_residual_edges.txt"), r"residual edge stress ok.*") + "\n")
    f.write("pytest -m gpu: " + last(os.path.join(E, "tests.log"), r"\d+ passed.*") + "\n")
    f.write(f"source: HEAD {sha}\n")
print(open(os.path.join(P, "r03_stress.txt")).read())
print(open(os.path.join(P, "r03_rocprof_summary.txt")).read().splitlines()[0])
