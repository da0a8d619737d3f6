#!/usr/bin/env python3
"""r06: where a batch of concurrent alpha-moves spends its time.  Runs REPS LabelingSteps of the many-label case of
tools/batch_probe.py (540 labels, 20 000 sites) with CTX moves per batch and nothing else, so that a kernel trace of the
process (rocprofv3 --kernel-trace) holds the batched form only; tools/batch_kernel_split.py --summarize <csv> then prints, per
kernel, launches / total / mean time and the share of the expansion's span the device was busy.
Env: NL (20000), EXTRA (400), CTX (16), REPS (2)."""
import csv, importlib, os, sys, time
import numpy as np
if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows = [r for r in rows if "mh::k_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = {}
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mh::", "")
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = names.setdefault(n, [0, 0, 0])
        a[0] += 1; a[1] += d; a[2] = max(a[2], d)
    # the expansions: from each k_ctl_init to the last kernel before the next k_ctl_init / k_reestimate
    spans, busy, gaps = [], [], []
    start = None
    for i, r in enumerate(rows):
        n = r["Kernel_Name"]
        if "k_ctl_init" in n:
            start = i
        elif start is not None and ("k_stats_merge" in n or i == len(rows) - 1):
            seg = rows[start:i + 1]
            spans.append(int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"]))
            busy.append(sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in seg))
            gaps.append(sorted(int(seg[j + 1]["Start_Timestamp"]) - int(seg[j]["End_Timestamp"]) for j in range(len(seg) - 1)))
            start = None
    print(f"{'kernel':28s} {'launches':>8s} {'total ms':>9s} {'mean us':>8s} {'max us':>8s}")
    for n, a in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:28s} {a[0]:8d} {a[1] / 1e6:9.3f} {a[1] / a[0] / 1e3:8.1f} {a[2] / 1e3:8.1f}")
    for s, b, g in zip(spans, busy, gaps):
        big = [x for x in g if x > 20000]
        print(f"expansion: span {s / 1e6:.2f} ms, kernels {b / 1e6:.2f} ms ({b / s:.2f}); {len(g) + 1} launches; gaps: median {g[len(g) // 2] / 1e3:.1f} us, "
              f"{len(big)} above 20 us summing to {sum(big) / 1e6:.2f} ms, all gaps {sum(g) / 1e6:.2f} ms")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
NL, EXTRA, CTX, REPS = int(os.environ.get("NL", 20000)), int(os.environ.get("EXTRA", 400)), int(os.environ.get("CTX", 16)), int(os.environ.get("REPS", 2))
K = 6
sc = mh.synth.make_scene(NL, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2)
e.propose_dlt4(7, 0, EXTRA)
rng = np.random.default_rng(1)
H = np.ascontiguousarray(np.concatenate([sc.H_true, sc.H_true[rng.integers(0, K, EXTRA // 3)] * (1 + rng.normal(0, 3e-3, (EXTRA // 3, 9))), e.get_models()]))
e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
e.set_tuning(37, CTX)
if "TUNE39" in os.environ: e.set_tuning(39, int(os.environ["TUNE39"]))     # sites per wave in a batch's setup / reduction launches
for r in range(REPS):
    e.set_models(H)
    t0 = time.perf_counter()
    lab, en, cyc = e.labeling_step(False, np.full(sc.n, -1, np.int32))
    print(f"step {r}: {(time.perf_counter() - t0) * 1e3:.2f} ms, {cyc} cycles, energy {int(en)}, {e.expand_batch_stats()}", flush=True)
e.close()
