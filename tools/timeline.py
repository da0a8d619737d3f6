#!/usr/bin/env python3
"""Timeline of a rocprofv3 kernel trace (…_kernel_trace.csv): for a window of dispatches, every kernel's queue, start and
end relative to the window's first start, and the idle gap on its own queue since the previous kernel there.
  tools/timeline.py <kernel_trace.csv> [first_residual_launch_to_show] [number_of_steps] [memory_copy_trace.csv]
A negative first launch counts from the END of the trace (the last phase of tools/shard_proxy.py is the one with the native
RCCL transport); with a memory-copy trace its rows (a one-rank ncclAllGather is a device-to-device copy, not a kernel)
are merged into the timeline."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if len(sys.argv) > 4 and sys.argv[4]:
    for r in csv.DictReader(open(sys.argv[4])):
        rows.append({"Kernel_Name": "COPY " + r.get("Direction", r.get("Name", "?")) + " " + r.get("Bytes", "") + " B", "Queue_Id": "cp",
                     "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"]})
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3


def short(n):
    for k in ("k_residual", "k_dlt4", "k_sel_argmax_gathered", "k_sel_argmax", "k_best_publish", "k_best_fused", "k_pad_scores", "ncclDevKernel", "fillBuffer", "k_model32", "COPY"):
        if k in n:
            return n[:28] if k == "COPY" else k
    return n[:40]


res = [i for i, r in enumerate(rows) if "k_residual" in r["Kernel_Name"]]
if first < 0:
    first = max(0, len(res) + first - steps)
if len(res) <= first + steps:
    first = max(0, len(res) - steps - 1)
lo, hi = res[first], res[first + steps]
t0 = int(rows[lo]["Start_Timestamp"])
last_end = {}
for r in rows[max(0, lo - 4):hi + 1]:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = s - last_end[q] if q in last_end else None
    last_end[q] = e
    print(f"q{q:>3} {short(r['Kernel_Name']):24s} start {s/1e3:10.1f} us  end {e/1e3:10.1f} us  dur {(e-s)/1e3:9.1f} us"
          + (f"  gap on queue {gap/1e3:8.1f} us" if gap is not None else ""))
