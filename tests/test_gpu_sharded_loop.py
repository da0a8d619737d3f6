"""BASELINE configs[4] on several ranks: the full loop of the host class with the propose stage
sharded over processes (MultiH::SetSharding) must give the labels and homographies of the
single-process run, bit for bit, for any world size — including a ragged split.  The GPU box has
one device, so the ranks share it and exchange over gloo; the transport hook is the one RCCL runs
through on a real node (multi-h_amd/sharding.py make_allgather_hook)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, env_extra):
    env = dict(os.environ, N="4000", K="3", HYP="3001", ITERS="6", LOOP_BACKEND="gloo", LOOP_DEVICE="0", **env_extra)
    script = os.path.join(ROOT, "tools", "loop_bench.py")
    if world == 1:
        cmd = [sys.executable, script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("iter_hyp", ["0", "700"])
def test_sharded_loop_is_independent_of_world_size(iter_hyp):
    one = _run(1, {"ITER_HYP": iter_hyp})
    assert one["clusters"] >= 2 and one["iterations"] >= 1
    for world in (2, 3):
        got = _run(world, {"ITER_HYP": iter_hyp})
        assert got["ranks_identical"], "ranks disagree on the final labels"
        assert got["exchanges"] > 0, "the sharded propose stage never exchanged scores"
        assert got["digest"] == one["digest"] and got["clusters"] == one["clusters"]
        assert got["energy"] == one["energy"] and got["iterations"] == one["iterations"]


_RCCL_PROBE = r'''
import importlib, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["MH_ROOT"])
import bench
mh = importlib.import_module("multi-h_amd")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sc = mh.synth.make_scene(3000, 3, seed=2, with_neighbours=False)
eng = mh.Engine(0, 2.6, 2.2, 0.005, 0.5, 20)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_correspondences(sc.src, sc.dst, sc.aff)
eng.propose_dlt4(7, 0, 4096)
eng.residual_matrix(2.2 ** 2, fetch_R=False, fetch_counts=False)
ptr, nbytes = eng.device_buffer(0)
counts = torch.as_tensor(bench._DevView(ptr, 4096, "<i4"), device=dev)     # zero-copy view of the engine's buffer
out = torch.full((4096,), -5, dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(out, counts)                                     # RCCL on the engine's own memory
torch.cuda.synchronize()
want = eng.score(2.2 ** 2)
assert np.array_equal(out.cpu().numpy(), want), "RCCL all-gather of the resident score buffer differs"
dist.barrier()
dist.destroy_process_group()
print("RCCL_PROBE_OK", int(want.max()))
'''


def test_rccl_all_gather_reads_the_engine_score_buffer_in_place():
    """bench.py's exchange on a real node: RCCL all-gathers straight out of the engine's resident
    int32 score buffer (zero-copy view, engine on torch's stream).  One rank is all this box has,
    but the collective, the external pointer and the stream ordering are the real ones."""
    env = dict(os.environ, MH_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, "-c", _RCCL_PROBE], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_PROBE_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
