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
    for name, failing in (("fail_round_2", 1), ("wrong_shard", 1)):
        bad, other = (r1, r0) if failing == 1 else (r0, r1)
        assert not bad[name]["ok"] and not other[name]["ok"], name
        assert "a rank reported an error" in other[name]["msg"] and other[name]["code"] == -3, other[name]
    assert "injected" in r1["fail_round_2"]["msg"]
    assert "shard" in r1["wrong_shard"]["msg"] and r1["wrong_shard"]["code"] == -2
    # r05: the selection runs in either residual mode, but the ranks must agree on it — their records carry the mode and
    # every rank fails with the same words when one of them is in the other mode
    for r in (r0, r1):
        assert not r["symmetric"]["ok"] and "same residual mode" in r["symmetric"]["msg"] and r["symmetric"]["code"] == -2, r["symmetric"]
    # r06 (advisor, medium): refitted winners.  The setting travels in the records' mode word; a rank whose engine lacks the
    # geometry the refit needs fails THROUGH the collective
    for r in (r0, r1):
        assert not r["refit_on_one_rank"]["ok"] and "refitted winners" in r["refit_on_one_rank"]["msg"] and r["refit_on_one_rank"]["code"] == -2, r["refit_on_one_rank"]
    assert not r1["refit_without_geometry_on_rank_1"]["ok"] and r1["refit_without_geometry_on_rank_1"]["code"] == -4
    assert "epipolar" in r1["refit_without_geometry_on_rank_1"]["msg"]
    assert not r0["refit_without_geometry_on_rank_1"]["ok"] and "a rank reported an error" in r0["refit_without_geometry_on_rank_1"]["msg"]
    assert r0["refit_clean"]["ok"] and r0["refit_clean"] == r1["refit_clean"] and len(r0["refit_clean"]["counters"]) >= 3
    assert r0["clean_again"] == r0["clean"] and r1["clean_again"] == r1["clean"]


def test_select_best_with_an_empty_shard_and_rank_local_failures():
    """r04 advisor finding (medium): whether an mh_select_best call is a new exchange must come out the same on every
    rank.  With total_m < world the last rank's shard is empty; enqueue-only followed by a fetch (twice) must run ONE
    all-gather on every rank (a second one on the empty rank alone would wait for ever: the timeout below), also when
    that rank skips the scoring call; the same models scored again are a second exchange everywhere.  A shard-size
    mismatch and an unscored batch travel through the collective as error markers: every rank fails, none hangs."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "sharded_select_best_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    recs = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in recs] == [0, 1]
    r0, r1 = recs
    for name in ("empty_shard", "empty_shard_unscored", "rescored", "full", "full_again"):
        assert r0[name]["ok"] and r1[name]["ok"], (name, r0[name], r1[name])
        assert r0[name]["result"] == r1[name]["result"], name
    assert r0["empty_shard"]["result"][0] == 0                       # the only hypothesis of the batch
    assert r0["exchanges_empty_shard"] == r1["exchanges_empty_shard"] == 1
    assert r0["exchanges_rescored"] == r1["exchanges_rescored"] == 2
    assert r0["full"]["result"] == r0["full_again"]["result"]
    assert not r0["wrong_shard"]["ok"] and not r1["wrong_shard"]["ok"]
    assert "shard" in r1["wrong_shard"]["msg"] and r1["wrong_shard"]["code"] == -2
    assert "a rank reported an error" in r0["wrong_shard"]["msg"]
    assert not r0["unscored"]["ok"] and not r1["unscored"]["ok"]
    assert "not been scored" in r0["unscored"]["msg"] and "a rank reported an error" in r1["unscored"]["msg"]
