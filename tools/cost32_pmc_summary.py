#!/usr/bin/env python3
"""Per-launch counter figures of k_cost32_resident from the rocprofv3 --pmc passes of tools/cost32_pmc_driver.py.
   usage: tools/cost32_pmc_summary.py <dir with one sub-directory per pass>"""
import csv, glob, os, sys
root = sys.argv[1]
agg = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_cost32" not in r["Kernel_Name"]:
            continue
        key = r["Counter_Name"]
        d = agg.setdefault(key, {})
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
if not agg:
    sys.exit("no k_cost32 rows found under " + root)
N, M = 50000, 100000
alg = 4.0 * N * M
print(f"k_cost32_resident, {M} hypotheses x {N} points, algorithmic bytes per launch {alg / 1e9:.3f} GB (4 B per pair); per-launch means over "
      f"{len(next(iter(agg.values())))} launches, summed over XCDs / shader engines")
val = {k: sum(v.values()) / len(v) for k, v in agg.items()}
for k in sorted(val):
    print(f"  {k:28s} {val[k]:18.0f}")
g = val.get
if g("WRITE_SIZE"):
    print(f"WRITE_SIZE {g('WRITE_SIZE') * 1024 / 1e9:.3f} GB = {g('WRITE_SIZE') * 1024 / alg:.4f} x algorithmic (rocprofv3 reports KiB)")
if g("TCC_EA0_WRREQ_sum") and g("TCC_EA0_WRREQ_64B_sum") is not None:
    w, w64 = g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_64B_sum")
    print(f"write requests to the fabric: {w:.0f}, of which 64-B {w64:.0f} ({w64 / w:.4f}); the rest are 32-B (partial-line) requests; "
          f"bytes = {(w64 * 64 + (w - w64) * 32) / 1e9:.3f} GB")
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if g(k) is not None:
            print(f"  {k:24s} / SQ_WAVE_CYCLES = {g(k) / wc:.4f}")
if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
    print(f"VALU instructions per wave {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}; per pair {g('SQ_INSTS_VALU') * 64 / (N * M):.2f} lane-instructions"
          + (f"; LDS instructions per pair {g('SQ_INSTS_LDS') * 64 / (N * M):.3f}" if g("SQ_INSTS_LDS") else "")
          + (f"; VMEM write instructions per pair {g('SQ_INSTS_VMEM_WR') * 64 / (N * M):.4f}" if g("SQ_INSTS_VMEM_WR") else ""))
