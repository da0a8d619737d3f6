"""Worker of tests/test_gpu_collective_exit.py: one rank of a sharded mh_select_greedy over gloo on a shared GPU.  Prints
one JSON line per rank with what each scenario returned."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mh = importlib.import_module("multi-h_amd")
sh = importlib.import_module("multi-h_amd.sharding")

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
THR2, TOTAL = 2.2 ** 2, 3001
sc = mh.synth.make_scene(3000, 3, seed=3, with_neighbours=False)
eng = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
eng.set_tuning(5, 64)                                   # the ranks share one GPU: a quarter of the chip each
eng.set_correspondences(sc.src, sc.dst, sc.aff)
hook = sh.make_allgather_hook(world, dev)
eng.set_transport(rank, world, host_fn=hook)
first, mine = sh.shard_range(TOTAL, world, rank)
out = {"rank": rank}


def attempt(name, prepare):
    eng.propose_dlt4(77, first, mine)
    prepare()
    try:
        H, counters, counts, _ = eng.select_greedy(THR2, 20, 8, np.ones(sc.n, np.uint8), total_m=TOTAL)
        out[name] = {"ok": True, "counters": counters.tolist(), "counts": counts.tolist()}
    except mh.MultiHError as ex:
        out[name] = {"ok": False, "code": ex.code, "msg": str(ex)}
    dist.barrier()


attempt("clean", lambda: None)
# rank 1 fails in the 2nd scoring round (after one model has been selected)
attempt("fail_round_2", lambda: eng.set_tuning(18, 2) if rank == 1 else None)
# rank 1 holds a model set that is not its shard
attempt("wrong_shard", lambda: eng.propose_dlt4(77, first, mine - 1) if rank == 1 else None)
# rank 0 is in symmetric mode
attempt("symmetric", lambda: eng.set_residual_mode(True) if rank == 0 else None)
eng.set_residual_mode(False)
# r06 (advisor): refitted winners (key 30).  (a) both ranks have the setting on, but rank 1's engine has no epipolar geometry
# (rank-local state): it must go through the collective with its error word set, not return before it.  (b) rank 1 alone has
# the setting on: the records' mode words differ and every rank fails with the same words.  (c) both on and complete: a clean
# refitted selection, the same on both ranks.
eng.set_tuning(30, 1)
if rank == 0:
    eng.set_epipolar(sc.F, sc.e2)
attempt("refit_without_geometry_on_rank_1", lambda: None)
eng.set_epipolar(sc.F, sc.e2)
attempt("refit_on_one_rank", lambda: eng.set_tuning(30, 0) if rank == 0 else None)
eng.set_tuning(30, 1)
attempt("refit_clean", lambda: None)
eng.set_tuning(30, 0)
attempt("clean_again", lambda: None)
for r in range(world):                                  # one rank at a time: the launcher merges the ranks' stdout
    if r == rank:
        print(json.dumps(out), flush=True)
    dist.barrier()
eng.close()
dist.barrier()
dist.destroy_process_group()
