#!/usr/bin/env python3
"""Condenses rocprofv3 CSV outputs (kernel stats + PMC passes) into a short text summary."""
import csv, glob, os, sys, collections
out = sys.argv[1]
KEEP = [k for k in os.environ.get("KERNELS", "k_residual,k_dlt4").split(",") if k]
ROWS = int(os.environ.get("STAT_ROWS", "14"))
def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))
for f in find("trace", "*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < ROWS: print("  ", ",".join(row))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_label"):
    for f in find(sub, "*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")
                name = name[:name.index("(")] if "(" in name else name      # keep the template arguments: variants differ there
                k = (name.replace("void ", "")[:100], row.get("Counter_Name", "?"))
                agg[k][0] += 1
                agg[k][1] += float(row.get("Counter_Value", 0) or 0)
        print("== PMC:", os.path.relpath(f, out))
        for (kn, cn), (n, tot) in sorted(agg.items()):
            if any(k in kn for k in KEEP):
                print(f"   {kn:92s} {cn:24s} launches={n:4d} avg={tot/n:.6g}")
