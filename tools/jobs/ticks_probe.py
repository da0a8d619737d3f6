import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mh = importlib.import_module("multi-h_amd")
NL, EXTRA, CTX = int(os.environ.get("NL", 20000)), int(os.environ.get("EXTRA", 400)), int(os.environ.get("CTX", 16))
K = 6
sc = mh.synth.make_scene(NL, K, seed=1234)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff); e.set_epipolar(sc.F, sc.e2); e.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)
e.propose_dlt4(7, 0, EXTRA)
rng = np.random.default_rng(1)
H = np.ascontiguousarray(np.concatenate([sc.H_true, sc.H_true[rng.integers(0, K, EXTRA // 3)] * (1 + rng.normal(0, 3e-3, (EXTRA // 3, 9))), e.get_models()]))
L = H.shape[0] + 1
e.set_tuning(8, 4 * L)
e.set_tuning(37, CTX)
for _ in range(2):
    e.set_models(H)
    t0 = time.perf_counter()
    lab, en, cyc = e.labeling_step(False, np.full(sc.n, -1, np.int32))
    ms = (time.perf_counter() - t0) * 1e3
tr = e.expand_trace(4 * L).copy()
m = tr[:, 1] > 0
tr = tr[m]
start = tr[:, 2].astype(np.int64) & 0xffffffff
end = tr[:, 3].astype(np.int64) & 0xffffffff
order = np.argsort(start)
tr, start, end = tr[order], start[order], end[order]
base = start[0]
start -= base; end -= base
# batches: a move that starts after every earlier move has ended opens a new launch
groups = []
cur = [0]; cur_end = end[0]
for i in range(1, len(start)):
    if start[i] > cur_end + 200:            # 2 us behind the last end: another launch
        groups.append(cur); cur = [i]; cur_end = end[i]
    else:
        cur.append(i); cur_end = max(cur_end, end[i])
groups.append(cur)
print(f"step {ms:.1f} ms, {len(tr)} solver runs in {len(groups)} launches (by their time stamps), {e.expand_batch_stats()}")
span = sum(end[g].max() - start[g].min() for g in groups) / 100
longest = sum((end[g] - start[g]).max() for g in groups) / 100
stagger = [(start[g].max() - start[g].min()) / 100 for g in groups]
print(f"sum over launches: first start to last end {span / 1e3:.2f} ms; longest move {longest / 1e3:.2f} ms; all moves {((end - start).sum()) / 1e5:.2f} ms")
print(f"stagger of the starts inside a launch: mean {np.mean(stagger):.1f} us, max {np.max(stagger):.1f} us")
big = sorted(groups, key=lambda g: -(end[g].max() - start[g].min()))[:8]
for g in big:
    print(f"  launch of {len(g)} solver runs, {(end[g].max() - start[g].min()) / 100:.0f} us: " + ", ".join(f"K={tr[i,0]} P={tr[i,1]} +{(start[i]-start[g].min())/100:.0f}..{(end[i]-start[g].min())/100:.0f} bar={tr[i,5]}" for i in g))
# by core size of the longest move
dur = np.array([(end[g] - start[g]).max() for g in groups]) / 100
kmax = np.array([tr[g, 0].max() for g in groups])
for lo, hi in ((1, 64), (65, 1024), (1025, 4096), (4097, 1 << 30)):
    mm = (kmax >= lo) & (kmax <= hi)
    if mm.any(): print(f"  launches whose largest core has {lo}..{hi} sites: {int(mm.sum())}, longest move mean {dur[mm].mean():.0f} us, sum {dur[mm].sum() / 1e3:.2f} ms")
e.close()
