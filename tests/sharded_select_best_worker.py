"""Worker of tests/test_gpu_collective_exit.py::test_select_best_with_an_empty_shard...: one rank of a sharded
mh_select_best over gloo on a shared GPU (the host-synchronised transport).  total_m < world leaves the last rank with an
EMPTY shard.  Prints one JSON line per rank with what each scenario returned."""
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
THR2 = 2.2 ** 2
sc = mh.synth.make_scene(3000, 3, seed=3, with_neighbours=False)
eng = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
eng.set_tuning(5, 64)
eng.set_correspondences(sc.src, sc.dst, sc.aff)
hook = sh.make_allgather_hook(world, dev)
eng.set_transport(rank, world, host_fn=hook)
out = {"rank": rank}


def batch(total, seed=77, score=True, count=None):
    first, mine = sh.shard_range(total, world, rank)
    mine = mine if count is None else count
    if mine > 0:
        eng.propose_dlt4(seed, first, mine)
    else:
        eng.set_models(np.zeros((0, 9)))
    if score:
        eng.residual_matrix(THR2, fetch_R=False, fetch_counts=False)     # a no-op that succeeds on an empty shard


def attempt(name, fn):
    try:
        out[name] = {"ok": True, "result": fn()}
    except mh.MultiHError as ex:
        out[name] = {"ok": False, "code": ex.code, "msg": str(ex)}
    dist.barrier()


def enqueue_then_fetch(total, score_on_empty=True):
    def run():
        batch(total, score=score_on_empty or sh.shard_range(total, world, rank)[1] > 0)
        assert eng.select_best(total, fetch=False) is None
        a = eng.select_best(total)                   # completes the enqueued exchange: NOT a second collective on any rank
        b = eng.select_best(total)                   # and once more
        assert a == b
        return list(a)
    return run


def rescored(total):
    def run():
        batch(total)
        a = eng.select_best(total)
        eng.residual_matrix(THR2, fetch_R=False, fetch_counts=False)     # the same models scored again: a NEW exchange on every rank
        b = eng.select_best(total)
        assert a == b
        return list(a)
    return run


calls0 = hook.stats["calls"]
attempt("empty_shard", enqueue_then_fetch(world - 1))                   # the last rank holds nothing
out["exchanges_empty_shard"] = hook.stats["calls"] - calls0
attempt("empty_shard_unscored", enqueue_then_fetch(world - 1, score_on_empty=False))      # ... and skips the scoring call
calls0 = hook.stats["calls"]
attempt("rescored", rescored(world - 1))
out["exchanges_rescored"] = hook.stats["calls"] - calls0
attempt("full", enqueue_then_fetch(3001))
# rank-local failures go THROUGH the collective: the last rank holds a model set that is not its shard ...
attempt("wrong_shard", lambda: (batch(3001, count=(sh.shard_range(3001, world, rank)[1] - 1) if rank == world - 1 else None),
                                list(eng.select_best(3001)))[1])
# ... rank 0 has not scored its batch
attempt("unscored", lambda: (batch(3001, score=(rank != 0)), list(eng.select_best(3001)))[1])
attempt("full_again", enqueue_then_fetch(3001))
for r in range(world):
    if r == rank:
        print(json.dumps(out), flush=True)
    dist.barrier()
eng.close()
dist.barrier()
dist.destroy_process_group()
