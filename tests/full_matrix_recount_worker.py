#!/usr/bin/env python3
"""Worker of test_full_size_every_stored_entry_is_recounted_on_the_device (tests/test_gpu_parity.py): BASELINE configs[2],
the residual matrix of 50 000 correspondences x 100 000 DLT hypotheses, every stored entry recounted on the device."""
import importlib, os, sys
import numpy as np
import torch
torch.cuda.init()                                   # before the engine's library loads its own HIP runtime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mh = importlib.import_module("multi-h_amd")
THR2 = 2.2 * 2.2
N, M = 50000, 100000
sc = mh.synth.make_scene(N, 10, seed=1234, with_neighbours=False, legacy_r04=True)
e = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
e.set_correspondences(sc.src, sc.dst, sc.aff)
e.propose_dlt4(1234, 0, M)
e.residual_matrix(THR2, fetch_R=False, fetch_counts=False)            # (allocates R)
ptr, nbytes = e.device_buffer(2)                                       # MH_BUF_RESIDUALS
ld = nbytes // 8 // M
assert ld >= N and ld * M * 8 == nbytes, (ld, nbytes)


class View:
    def __init__(self, p, n, t):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": t, "data": (p, False), "version": 2, "strides": None}


dev = torch.device("cuda:0")
R = torch.as_tensor(View(ptr, M * ld, "<f8"), device=dev).view(M, ld)
R.fill_(float("nan"))                                                  # whatever is not stored stays NaN
torch.cuda.synchronize()
_, cnt = e.residual_matrix(THR2, fetch_R=False)
below = torch.empty(M, dtype=torch.int64, device=dev)
stored = torch.empty(M, dtype=torch.int64, device=dev)
for first in range(0, M, 4096):
    blk = R[first:first + 4096, :N]
    below[first:first + 4096] = (blk < THR2).sum(1)
    stored[first:first + 4096] = (~torch.isnan(blk)).sum(1)
below, stored = below.cpu().numpy(), stored.cpu().numpy()
assert np.array_equal(below, cnt.astype(np.int64)), int((below != cnt).sum())
# a residual is NaN only where the arithmetic makes one (0/0, inf - inf: degenerate models): few rows, and exactly those
assert (stored == N).mean() > 0.99, float((stored == N).mean())
some = np.concatenate([np.arange(0, 192), np.flatnonzero(stored != N)[:64]]).astype(np.int64)
for first in sorted(set(int(q) for q in some)):
    row = e.get_residual_rows(first, 1)[0]
    assert int((~np.isnan(row)).sum()) == int(stored[first]), first
if ld > N:                                                             # the padding of a row is never written
    assert bool(torch.isnan(R[:, N:]).all())
e.close()
print("RECOUNT OK", M, "rows;", int((stored != N).sum()), "rows hold NaN residuals of their own")
