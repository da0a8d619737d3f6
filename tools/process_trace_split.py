#!/usr/bin/env python3
"""Where a Process() call spends its time, from a rocprofv3 kernel trace of tools/loop_bench.py with REPEAT=1: the SECOND call of the
process (from its upload on): per kernel launches / total / mean / max, the share of the span the device is busy, and the largest gaps
between launches (host stages).
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/loop_bench.py ;  python3 tools/process_trace_split.py DIR/.../*_kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_split_soa" in r["Kernel_Name"]]
seg = rows[idx[1]:] if len(idx) > 1 else rows
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
names, busy = {}, 0
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mh::", "").replace("(anonymous namespace)::", "")[:44]
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = names.setdefault(short(r), [0, 0, 0]); a[0] += 1; a[1] += d; a[2] = max(a[2], d); busy += d
print(f"second Process(): first to last kernel {(t1 - t0) / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms, {len(seg)} launches")
for n, a in sorted(names.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{n:46s} {a[0]:6d} {a[1] / 1e6:9.3f} ms  mean {a[1] / a[0] / 1e3:8.1f} us  max {a[2] / 1e3:8.1f}")
gaps = sorted(((int(seg[j + 1]["Start_Timestamp"]) - int(seg[j]["End_Timestamp"])), j) for j in range(len(seg) - 1))
big = [g for g in gaps if g[0] > 30000]
print(f"gaps above 30 us: {len(big)} summing to {sum(g[0] for g in big) / 1e6:.2f} ms; all positive gaps {sum(max(g[0], 0) for g in gaps) / 1e6:.2f} ms")
for g, j in sorted(big, reverse=True)[:30]:
    print(f"   {g / 1e3:8.1f} us at +{(int(seg[j]['End_Timestamp']) - t0) / 1e6:7.2f} ms after {short(seg[j])} before {short(seg[j + 1])}")
