"""World-size-2 (and 3, ragged) test of the multi-GPU plumbing on CPU with the gloo backend:
hypothesis sharding, the all-gather of per-model scores and the replicated selection
(multi-h_amd/sharding.py — the same functions bench.py runs over RCCL).  The scoring kernel is
replaced by the oracle here (tests only); what is under test is the partition + exchange."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    mh = importlib.import_module("multi-h_amd")
    sh = importlib.import_module("multi-h_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = mh.synth.make_scene(600, 3, seed=5, with_neighbours=False)      # replicated on every rank
        if mode == "strong":
            sizes = sh.shard_counts(total, world)
            first, m = sh.shard_range(total, world, rank)
        else:
            sizes = [total] * world
            first, m = sh.batch_first(0, world, rank, total), total
        if m > 0:
            idx = O.sample4(99, first, m, sc.n)
            H, _, _ = O.dlt4(sc.src, sc.dst, idx)
            local = torch.from_numpy(O.score(sc.src, sc.dst, H, 2.2 ** 2))
        else:
            local = torch.zeros(0, dtype=torch.int32)                    # an empty shard still takes part in the gather
        scores = sh.gather_scores(local, world, sizes=sizes)
        best, val = sh.select_best(scores)
        q.put((rank, scores.numpy().copy(), int(best), int(val), sizes))
        dist.barrier()
    finally:
        dist.destroy_process_group()


# (8, 100): configs[3]'s 8-way split, ragged (13 x 4 + 12 x 4); (8, 5): more ranks than hypotheses — three empty shards
@pytest.mark.parametrize("world,total,mode", [(2, 128, "strong"), (3, 100, "strong"), (2, 64, "weak"), (8, 100, "strong"), (8, 5, "strong")])
def test_sharded_scores_equal_single_process(oracle, synth, mh, world, total, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results.sort(key=lambda r: r[0])
    # single-process reference over the same global RNG counters
    sc = mh.synth.make_scene(600, 3, seed=5, with_neighbours=False)       # (the workers' generator)
    n_global = total if mode == "strong" else total * world
    idx = oracle.sample4(99, 0, n_global, sc.n)
    H, _, _ = oracle.dlt4(sc.src, sc.dst, idx)
    ref = oracle.score(sc.src, sc.dst, H, 2.2 ** 2)
    for rank, scores, best, val, sizes in results:
        assert np.array_equal(scores, ref), f"rank {rank}: gathered scores differ from the unsharded batch"
        assert best == int(np.argmax(ref)) and val == int(ref.max())
        assert sum(sizes) == n_global


def test_shard_arithmetic(mh):
    sh = importlib.import_module("multi-h_amd.sharding")
    assert sh.shard_counts(100000, 8) == [12500] * 8
    assert sh.shard_counts(10, 4) == [3, 3, 2, 2]
    assert [sh.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sh.batch_first(2, 8, 3, 100000) == (2 * 8 + 3) * 100000
    assert sh.global_model_index(7, [3, 3, 2, 2]) == (2, 1)
    covered = []
    for r in range(7):
        f, c = sh.shard_range(1000, 7, r)
        covered += list(range(f, f + c))
    assert covered == list(range(1000))


def _hook_worker(rank, world, port, q):
    import ctypes
    sys.path.insert(0, ROOT)
    sh = importlib.import_module("multi-h_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hook = sh.make_allgather_hook(world)
        out = []
        for n in (5, 9):          # int32 scores, then the 72-byte model record
            send = (np.arange(n, dtype=np.int32) + 1000 * rank) if n == 5 else np.full(9, rank + 0.5)
            recv = np.zeros(world * send.size, dtype=send.dtype)
            rc = hook(None, send.ctypes.data_as(ctypes.c_void_p), recv.ctypes.data_as(ctypes.c_void_p), send.nbytes)
            out.append((rc, recv.copy()))
        q.put((rank, out, dict(hook.stats)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_allgather_hook_of_the_host_class(mh):
    """The transport MultiH::SetSharding calls (multi-h_amd/host/MultiH.h): rank-ordered concatenation out, status 0.
    The engine hands it DEVICE pointers; here (no GPU) the hook's host-memory mode carries the same bytes."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hook_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, stats in results:
        (rc1, scores), (rc2, models) = out
        assert rc1 == 0 and rc2 == 0 and stats["calls"] == 2
        assert np.array_equal(scores, np.concatenate([np.arange(5, dtype=np.int32) + 1000 * r for r in range(world)]))
        assert np.array_equal(models, np.concatenate([np.full(9, r + 0.5) for r in range(world)]))
