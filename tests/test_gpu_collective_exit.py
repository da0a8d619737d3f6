"""r03 advisor finding (medium): a rank-local failure inside the sharded greedy selection must not strand the other ranks
in the all-gather.  Two ranks share the GPU and exchange over gloo (the host-synchronised transport); one of them fails
— an injected scoring failure in the second round, a resident model set that is not its shard, the symmetric residual
mode — and BOTH must come back with an error in the same round, then run a clean selection again."""
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


def test_a_rank_local_failure_ends_the_selection_on_every_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "sharded_failure_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)          # a stranded rank would run into this
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    recs = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in recs] == [0, 1]
    r0, r1 = recs
    assert r0["clean"]["ok"] and r0["clean"] == r1["clean"] and len(r0["clean"]["counters"]) >= 3
    for name, failing in (("fail_round_2", 1), ("wrong_shard", 1), ("symmetric", 0)):
        bad, other = (r1, r0) if failing == 1 else (r0, r1)
        assert not bad[name]["ok"] and not other[name]["ok"], name
        assert "a rank reported an error" in other[name]["msg"] and other[name]["code"] == -3, other[name]
    assert "injected" in r1["fail_round_2"]["msg"]
    assert "shard" in r1["wrong_shard"]["msg"] and r1["wrong_shard"]["code"] == -2
    assert "forward" in r0["symmetric"]["msg"] and r0["symmetric"]["code"] == -2
    assert r0["clean_again"] == r0["clean"] and r1["clean_again"] == r1["clean"]
