#!/usr/bin/env python3
"""Copies the judged summaries of a tools/profile_bench.sh run from gpurun_out/ (scratch) into
profiles/ (tracked) and regenerates profiles/residual_traffic.json, the per-launch HBM traffic
bench.py reports as roofline.traffic.   usage: tools/collect_profiles.py r01"""
import glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, f"{tag}_rocprof_summary.txt"))
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, f"{tag}_bench_under_trace.json"))
stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if stats:        # gpurun merges into gpurun_out/, so older runs may still lie there: newest wins
    shutil.copy(stats[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = os.path.join(ROOT, "gpurun_out", f"bench_{tag}.json")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(dst, f"{tag}_bench.json"))
s = open(os.path.join(src, "summary.txt")).read()


def grab(counter):
    # the product kernel: <PPL 4, MC 16, WRITE_R, !MASK, !NT, FAST, !CALIB, !HSGPR, !SYM, !CONTRACT>
    m = re.search(r"k_residual<4, 16, true, false, false, true, false, false, false, false>\s+%s\s+launches=\s*(\d+)\s+avg=([0-9.e+]+)" % counter, s)
    return float(m.group(2)), int(m.group(1))


w, n = grab("WRITE_SIZE")
f, _ = grab("FETCH_SIZE")
json.dump({"points": 50000, "models": 100000, "kernel": "k_residual",
           "hbm_bytes_per_launch": w * 1024 + 2 * f * 1024, "write_bytes": w * 1024,
           "fetch_bytes_corrected": 2 * f * 1024, "launches_averaged": n,
           "method": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes (tools/profile_bench.sh); "
                     "KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"},
          open(os.path.join(dst, "residual_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, "residual_traffic.json")).read())
