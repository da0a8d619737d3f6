#!/bin/bash
# Effective shader clock (GRBM_GUI_ACTIVE / 8 / duration) of the residual, store-only and score kernels.
export TMPDIR=/tmp
OUT=gpurun_out/clock_probe; rm -rf $OUT; mkdir -p $OUT
RV=0,7 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -- python3 tools/kernel_sweep.py > $OUT/sweep.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/clock_probe/p/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/clock_probe/p/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    name, d = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
    if d > 2_000_000:
        agg[name[:70]].append(float(r["Counter_Value"]) / 8 / d)
for k, v in agg.items():
    print(f"{k:70s} launches={len(v):3d}  clock = {sum(v)/len(v):.3f} GHz  (min {min(v):.3f}, max {max(v):.3f})")
PY
