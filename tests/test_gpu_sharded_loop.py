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
