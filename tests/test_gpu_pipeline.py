"""GPU tests of what r03 added around the bench step: the pipelined propose (second stream), the engine's own best-model
arg-max, the materialised int32 cost matrix (the s = 4 variant of SURVEY 8(d)), and the native RCCL transport
(libmultih_rccl.so) on a one-rank communicator — the whole sharded protocol, RCCL's ncclAllGather on the engine's
stream, without Python in the exchange."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
THR2 = 2.2 * 2.2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(engine, sc):
    engine.set_correspondences(sc.src, sc.dst, sc.aff)
    engine.set_epipolar(sc.F, sc.e2)


def test_prefetched_batches_equal_proposed_batches(engine, synth, oracle):
    """mh_prefetch_dlt4 + mh_adopt_prefetched: the batch is prepared on the engine's second stream while the main stream
    sweeps the previous one; tuples, homographies and the counts of the sweep that follows are those of mh_propose_dlt4,
    bit for bit, over several rounds of the double buffer and with different batch sizes."""
    sc = synth.make_scene(3000, 3, seed=8, with_neighbours=False)
    _load(engine, sc)
    want = []
    for i, m in enumerate((700, 700, 333, 1024)):
        engine.propose_dlt4(21, 1000 * i, m)
        R, cnt = engine.residual_matrix(THR2)
        want.append((engine.get_samples(), engine.get_models(), cnt, R))
    engine.prefetch_dlt4(21, 0, 700)
    for i, m in enumerate((700, 700, 333, 1024)):
        engine.adopt_prefetched()
        if i < 3:
            engine.prefetch_dlt4(21, 1000 * (i + 1), (700, 333, 1024)[i])      # runs beside the sweep below
        R, cnt = engine.residual_matrix(THR2)
        idx, H = engine.get_samples(), engine.get_models()
        assert np.array_equal(idx, want[i][0]) and np.array_equal(H.view(np.uint64), want[i][1].view(np.uint64)), i
        assert np.array_equal(cnt, want[i][2]) and np.array_equal(R.view(np.uint64), want[i][3].view(np.uint64)), i
    assert np.array_equal(want[0][0], oracle.sample4(21, 0, 700, sc.n))
    with pytest.raises(Exception):
        engine.adopt_prefetched()                          # nothing prefetched any more


def test_select_best_is_the_first_maximum(engine, synth):
    sc = synth.make_scene(2000, 3, seed=4, with_neighbours=False)
    _load(engine, sc)
    engine.propose_dlt4(3, 0, 5000)
    H = engine.get_models()
    H[4000:] = H[:1000]                                    # duplicates: ties between a hypothesis and its copy
    engine.set_models(H)
    cnt = engine.score(THR2)
    best, count = engine.select_best()
    assert count == int(cnt.max()) and best == int(np.argmax(cnt))
    assert int(np.flatnonzero(cnt == cnt.max()).size) >= 2, "the case should contain a tie"
    engine.residual_matrix(THR2, fetch_R=False, fetch_counts=False)
    assert engine.select_best() == (best, count)
    assert engine.select_best(fetch=False) is None and engine.select_best() == (best, count)


@pytest.mark.parametrize("n,k,seed", [(129, 2, 1), (1000, 3, 2), (4099, 4, 7)])
def test_cost_matrix_bit_exact(engine, synth, oracle, n, k, seed):
    """mh_cost_matrix against the oracle's dataEnergy (M/MultiH.cpp:473-504) for every (point, model), and its fused
    counts against the score; includes degenerate models (non-finite d2 takes the `beyond` branch like the reference)."""
    sc = synth.make_scene(n, k, seed=seed, with_neighbours=False)
    rng = np.random.default_rng(seed)
    H = np.concatenate([sc.H_true, sc.H_true[rng.integers(0, k, 20)] * (1 + rng.normal(0, 2e-4, (20, 9))),
                        np.array([[1, 0, 0, 0, 1, 0, 0, 0, 0], [1, 0, 0, 0, 1, 0, 1e-3, -1e-3, 0.0]])])
    _load(engine, sc)
    engine.propose_dlt4(seed, 0, 300)                                     # random hypotheses: nearly every pair is far out
    H = np.concatenate([H, engine.get_models(), np.array([[0, 0, 1, 0, 0, 1, 0, 0, 1e90], [np.nan, 0, 0, 0, 1, 0, 0, 0, 1.0]])])
    engine.set_models(H)
    with np.errstate(all="ignore"):
        ref = oracle.data_cost(sc.src, sc.dst, H, 0.5, THR2)           # site-major, label 0 = outlier
        ref_cnt = oracle.score(sc.src, sc.dst, H, THR2)
    # through the FP32 pre-test (k_cost32: the constant for pairs proved far out, the FP64 formula for the rest) and with
    # the FP64 formula for every pair (k_cost_matrix): the same matrix and counts
    for pretest in (1, 0):
        engine.set_tuning(15, pretest)
        try:
            Cm, cnt = engine.cost_matrix()
        finally:
            engine.set_tuning(15, 1)
        assert np.array_equal(Cm, ref[:, 1:].T), pretest
        assert np.array_equal(cnt, ref_cnt), pretest
    assert set(np.unique(Cm)).issubset(set(range(0, 201)) | {9802})
    assert (Cm != 9802).sum() > 0.05 * sc.n


def test_cost_matrix_as_a_resident_grid_equals_the_dispatched_one(engine, synth):
    """r04: k_cost32 walked by a resident grid that hands itself the (model block, point slice) items (mh_set_tuning key 23) —
    the same int32 matrix and counts as one hardware-dispatched workgroup per item, at a size where the resident form runs."""
    sc = synth.make_scene(20011, 4, seed=3, with_neighbours=False)
    _load(engine, sc)
    engine.propose_dlt4(5, 0, 3001)
    try:
        engine.set_tuning(23, 0)
        C0, cnt0 = engine.cost_matrix()
        for v in (8, -1, 3):
            engine.set_tuning(23, v)
            C1, cnt1 = engine.cost_matrix()
            assert np.array_equal(cnt1, cnt0) and np.array_equal(C1, C0), v
    finally:
        engine.set_tuning(23, 8)
    assert (C0 != C0.max()).sum() > 1000


def test_score_as_a_resident_grid_equals_the_dispatched_one_and_the_oracle(engine, synth, oracle):
    """r04: k_score32 walked by a resident grid (mh_set_tuning key 24): the same counts as one hardware-dispatched workgroup
    per item and as the oracle's score loop, at a size where the resident form runs."""
    sc = synth.make_scene(20011, 4, seed=6, with_neighbours=False)
    _load(engine, sc)
    engine.propose_dlt4(8, 0, 12001)
    with np.errstate(all="ignore"):
        want = oracle.score(sc.src, sc.dst, engine.get_models(), THR2)
    try:
        for v in (0, 12, -1, 5):
            engine.set_tuning(24, v)
            assert np.array_equal(engine.score(THR2), want), v
    finally:
        engine.set_tuning(24, 12)
    assert want.max() > 1000


def _rccl(mh):
    lib = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_rccl.so"))
    lib.mhr_last_error.restype = C.c_char_p
    lib.mhr_calls.restype = C.c_longlong
    return lib


def test_native_rccl_transport_on_a_one_rank_communicator(mh, engine, synth, oracle):
    """mh_set_transport with RCCL's ncclAllGather (libmultih_rccl.so, include/multih_rccl.h) as the stream-ordered
    transport.  One rank is all this box has, but the communicator, the collective on the engine's stream, the
    88-byte records and the first round's score all-gather are the real ones: the selection must be the unsharded one
    (which is compared with the oracle's), and the collective must actually have been used."""
    rl = _rccl(mh)
    uid = (C.c_ubyte * 128)()
    assert rl.mhr_unique_id(uid) == 0, rl.mhr_last_error()
    comm = C.c_void_p()
    assert rl.mhr_init(C.byref(comm), 0, 1, uid, 0) == 0, rl.mhr_last_error()
    try:
        # r06: what RCCL itself says about the communicator — the read-back every bench line and harness log carries
        assert rl.mhr_count(comm) == 1 and rl.mhr_rank(comm) == 0, rl.mhr_last_error()
        assert rl.mhr_version() >= 20000, "ncclGetVersion: major * 10000 + minor * 100 + patch"
        assert rl.mhr_count(None) == -1 and b"communicator" in rl.mhr_last_error()
        sc = synth.make_scene(4000, 4, seed=31, with_neighbours=False)
        _load(engine, sc)
        engine.propose_dlt4(77, 0, 3001)
        plain = engine.select_greedy(THR2, 20, 8, np.ones(sc.n, np.uint8))
        engine.set_transport(0, 1, stream_fn=rl.mhr_allgather, ctx=comm)
        before = rl.mhr_calls(comm)
        via = engine.select_greedy(THR2, 20, 8, np.ones(sc.n, np.uint8), total_m=3001)
        used = rl.mhr_calls(comm) - before
        for a, b in zip(plain, via):
            assert np.array_equal(a, b)
        assert len(plain[1]) >= 4
        assert used == len(via[1]) + 1 + 1, "one score all-gather, then one 88-byte record all-gather per round"
        # the bench step's exchange: score all-gather + arg-max over the gathered vector (mh_select_best)
        engine.residual_matrix(THR2, fetch_R=False, fetch_counts=False)
        before = rl.mhr_calls(comm)
        engine.profile_reset()
        engine.profile_enable(True)
        best_via = engine.select_best(3001)
        engine.profile_enable(False)
        assert rl.mhr_calls(comm) - before == 1
        n_x, ms_x = engine.profile_get(7)                  # MH_K_EXCHANGE: the all-gather + the arg-max behind it, timed on the exchange stream
        assert n_x == 1 and 0.0 < ms_x < 50.0
        engine.set_transport(0, 1)
        assert engine.select_best() == best_via
        cnt_all = engine.score(THR2)
        assert best_via == (int(np.argmax(cnt_all)), int(cnt_all.max()))
        engine.set_transport(0, 1, stream_fn=rl.mhr_allgather, ctx=comm)
        H_all, _, _ = oracle.dlt4(sc.src, sc.dst, oracle.sample4(77, 0, 3001, sc.n))
        _, idx_o, cnt_o, mask_o = oracle.select_greedy(sc.src, sc.dst, H_all, THR2, 20, 8)
        assert idx_o.tolist() == via[1].tolist() and cnt_o.tolist() == via[2].tolist() and np.array_equal(mask_o, via[3])
    finally:
        engine.set_transport(0, 1)
        rl.mhr_destroy(comm)


def test_harness_with_rccl_ranks_equals_the_plain_run(mh, synth, tmp_path):
    """multih_harness --ranks 1: the C++ path end to end — fork-free one-rank communicator, MultiH::SetShardingStream,
    every propose batch through the sharded protocol — writes the result file of the plain run."""
    sc = synth.make_scene(3000, 3, seed=12, with_neighbours=False)
    corr = tmp_path / "corr.txt"
    np.savetxt(corr, np.concatenate([sc.src, sc.dst, sc.aff], axis=1), fmt="%.17g")
    epi = tmp_path / "epi.txt"
    np.savetxt(epi, np.concatenate([sc.F, sc.e2])[None], fmt="%.17g")
    harness = os.path.join(os.path.dirname(mh.LIB_PATH), "multih_harness")
    outs = []
    for extra in ([], ["--ranks", "1"]):
        out = tmp_path / f"res{len(outs)}.txt"
        r = subprocess.run([harness, str(corr), str(out), "--epipolar", str(epi), "--hypotheses", "3000"] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        if extra:
            assert "joined the RCCL communicator" in r.stdout
        outs.append(open(out).read())
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == sc.n
    assert len(set(int(l.split()[-1]) for l in outs[0].splitlines())) >= 3


def test_fp32_pretest_score_equals_the_fp64_formula(engine, synth, oracle):
    """mh_score decides most pairs in FP32 behind a rigorous error bound and only the doubtful ones in FP64
    (csrc/score32.hip): the counts must be the oracle's for good models, near-miss models, models whose horizon crosses
    the data and degenerate ones, with and without a point mask — and the pre-test must really carry the load."""
    sc = synth.make_scene(5000, 4, seed=17, with_neighbours=False)
    rng = np.random.default_rng(17)
    engine.set_correspondences(sc.src, sc.dst, sc.aff)
    engine.propose_dlt4(17, 0, 3000)
    H = np.concatenate([engine.get_models(), sc.H_true, sc.H_true[rng.integers(0, 4, 50)] * (1 + rng.normal(0, 3e-4, (50, 9))),
                        np.array([[1, 0, 0, 0, 1, 0, 1e-3, -1e-3, 0.0],            # horizon through the image
                                  [1, 0, 0, 0, 1, 0, 0, 0, 0.0],                   # s == 0 everywhere
                                  [1e150, 0, 0, 0, 1e150, 0, 0, 0, 1e150],         # outside the FP32 range: all pairs in FP64
                                  [1e-300, 0, 0, 0, 1e-300, 0, 0, 0, 1e-300],
                                  [np.nan, 0, 0, 0, 1, 0, 0, 0, 1.0]])])
    engine.set_models(H)
    engine.score_stats(reset=True)
    cnt = engine.score(THR2)
    pairs, pairs64 = engine.score_stats(reset=True)
    with np.errstate(all="ignore"):
        ref = oracle.score(sc.src, sc.dst, H, THR2)
    assert np.array_equal(cnt, ref), f"{int((cnt != ref).sum())} counts differ"
    assert pairs == H.shape[0] * sc.n and 0 < pairs64 < 0.02 * pairs, (pairs, pairs64)
    mask = (rng.random(sc.n) < 0.6).astype(np.uint8)
    with np.errstate(all="ignore"):
        assert np.array_equal(engine.score(THR2, mask), oracle.score(sc.src, sc.dst, H, THR2, mask))
    engine.score_stats(reset=True)
    engine.set_tuning(15, 0)                                  # the FP64 sweep for every pair
    try:
        assert np.array_equal(engine.score(THR2), ref)
        assert engine.score_stats()[0] == 0
    finally:
        engine.set_tuning(15, 1)


def test_fp32_pretest_when_the_fp32_denominator_overflows(engine, oracle):
    """Models whose denominator overflows (or is not representable) in FP32 while the FP64 formula finds inliers: every
    point is mapped next to the origin, where the matches sit.  The pre-test's cheap test works on s (x2 - u), which is
    infinite there — such models must not be eligible for it at all (found by tools/stress_residual_edges.py when the
    cheap test lost its reciprocal: |s| = inf passed "|s| >= tau" with tau = +inf)."""
    rng = np.random.default_rng(3)
    n = 700
    src = rng.uniform(0, 1000, (n, 2))
    dst = rng.uniform(-1.2, 1.2, (n, 2))
    dst[::7] = rng.uniform(-40, 40, (dst[::7].shape[0], 2))                  # some matches far from the origin
    H = np.array([[0, 0, 1, 0, 0, 1, 0, 0, 1e90],                            # u = v = 1e-90
                  [0, 0, 1, 0, 0, 1, 3e35, 3e35, 1e35],                      # finite in FP32, the sum is not
                  [1e-3, 0, 0.5, 0, 1e-3, -0.5, 1e36, 1e36, 1e36],
                  [0, 0, 19.58, 0, 0, 7.9e89, 0.44, -1.2e-136, 1.003e90],    # the stress tool's case, v = 0.79
                  [0, 0, 1, 0, 0, 1, 0, 0, 1e29],                            # still eligible: below 2^100
                  [0, 0, 1, 0, 0, 1, 0, 0, 2e30]])                           # just above
    engine.set_correspondences(src, dst)
    engine.set_models(H)
    engine.score_stats(reset=True)
    cnt = engine.score(THR2)
    pairs, pairs64 = engine.score_stats(reset=True)
    with np.errstate(all="ignore"):
        ref = oracle.score(src, dst, H, THR2)
    assert pairs == H.shape[0] * n, "the pre-test kernel should have run (coordinates are in its range)"
    assert np.array_equal(cnt, ref) and (ref > n // 2).all(), (cnt, ref)


def test_fp32_pretest_with_thresholds_exactly_on_residual_values(engine, synth, oracle):
    """The strict comparison d2 < thr^2 at its sharpest: thresholds set to a pair's own FP64 residual (that pair is NOT an
    inlier) and to the next double above it (now it is).  The FP32 pre-test cannot tell such pairs apart — its bound
    must make it hand them to the FP64 formula."""
    sc = synth.make_scene(2000, 3, seed=23, noise=1.0, with_neighbours=False)
    engine.set_correspondences(sc.src, sc.dst, sc.aff)
    H = np.concatenate([sc.H_true, sc.H_true * (1 + np.random.default_rng(1).normal(0, 1e-4, sc.H_true.shape))])
    engine.set_models(H)
    R, _ = engine.residual_matrix(THR2)
    rng = np.random.default_rng(2)
    tried = 0
    for m in range(H.shape[0]):
        near = np.flatnonzero((R[m] > 0.5) & (R[m] < 50.0))
        for i in rng.choice(near, size=min(6, near.size), replace=False):
            for thr2 in (R[m, i], np.nextafter(R[m, i], np.inf), np.nextafter(R[m, i], -np.inf)):
                cnt = engine.score(float(thr2))
                want = (R < thr2).sum(axis=1)
                assert np.array_equal(cnt, want), (m, int(i), float(thr2))
                tried += 1
    assert tried >= 60


@pytest.mark.parametrize("transport", ["none", "rccl-one-rank"])
def test_select_best_off_the_critical_path(mh, engine, synth, transport):
    """r04: an enqueue-only mh_select_best runs its (all-gather +) arg-max on a third stream behind an event of the
    sweep; the next sweep starts at once into the engine's other counts buffer.  Several pipelined steps without a
    single host wait, then one fetch: every step's winner is the one a synchronous propose + score + arg-max finds, the
    scores the exchange read are the batch's own (no padding kernel), and a scoring call after an enqueue-only call
    lands in a buffer the pending exchange does not read."""
    sc = synth.make_scene(3000, 3, seed=18, with_neighbours=False)
    _load(engine, sc)
    sizes = (901, 901, 640, 1200, 1200, 333)
    want = []
    for i, m in enumerate(sizes):
        engine.propose_dlt4(5, 5000 * i, m)
        cnt = engine.score(THR2)
        want.append((int(np.argmax(cnt)), int(cnt.max())))
    comm = rl = None
    if transport != "none":
        rl = _rccl(mh)
        uid = (C.c_ubyte * 128)()
        assert rl.mhr_unique_id(uid) == 0, rl.mhr_last_error()
        comm = C.c_void_p()
        assert rl.mhr_init(C.byref(comm), 0, 1, uid, 0) == 0, rl.mhr_last_error()
        engine.set_transport(0, 1, stream_fn=rl.mhr_allgather, ctx=comm)
    try:
        # (a) read every step's result: the fetch completes the pending exchange, it does not start a new one
        engine.prefetch_dlt4(5, 0, sizes[0])
        for i, m in enumerate(sizes):
            engine.adopt_prefetched()
            if i + 1 < len(sizes):
                engine.prefetch_dlt4(5, 5000 * (i + 1), sizes[i + 1])
            engine.residual_matrix(THR2, fetch_R=False, fetch_counts=False)
            before = rl.mhr_calls(comm) if comm else 0
            assert engine.select_best(m, fetch=False) is None
            assert engine.select_best(m) == want[i], i
            if comm:
                assert rl.mhr_calls(comm) - before == 1, "the fetch must not run a second collective"
        # (b) no host wait at all between the steps; only the last result is read
        engine.prefetch_dlt4(5, 0, sizes[0])
        for i, m in enumerate(sizes):
            engine.adopt_prefetched()
            if i + 1 < len(sizes):
                engine.prefetch_dlt4(5, 5000 * (i + 1), sizes[i + 1])
            engine.residual_matrix(THR2, fetch_R=False, fetch_counts=False)
            last = engine.select_best(m, fetch=(i + 1 == len(sizes)))
        assert last == want[-1]
        # (c) scoring right behind an enqueue-only call: the sweep's counts must be intact when the exchange reads them
        engine.propose_dlt4(5, 0, sizes[0])
        engine.residual_matrix(THR2, fetch_R=False, fetch_counts=False)
        engine.select_best(sizes[0], fetch=False)
        engine.propose_dlt4(5, 5000, sizes[1])              # another batch, scored into the other buffer at once
        cnt1 = engine.score(THR2)
        assert (int(np.argmax(cnt1)), int(cnt1.max())) == want[1]
        engine.synchronize()                                # completes the pending exchange of batch 0
        assert engine.select_best(sizes[1]) == want[1]
        # a model set that has never been scored has no best model
        engine.propose_dlt4(5, 10000, 77)
        with pytest.raises(mh.MultiHError) as ei:
            engine.select_best(77)
        assert ei.value.code == -4
    finally:
        engine.set_transport(0, 1)
        if comm:
            rl.mhr_destroy(comm)


def test_new_correspondences_drop_a_prefetched_batch(mh, engine, synth):
    """r03 advisor finding: mh_set_correspondences waits for a DLT prefetch in flight on the second stream (it reads the
    point arrays the call overwrites) and drops the batch — it was sampled from the old point set."""
    a = synth.make_scene(3000, 3, seed=1, with_neighbours=False)
    b = synth.make_scene(3000, 3, seed=2, with_neighbours=False)
    _load(engine, a)
    engine.prefetch_dlt4(9, 0, 20000)
    engine.set_correspondences(b.src, b.dst, b.aff)          # same n: only the prefetch state can tell
    with pytest.raises(mh.MultiHError) as ei:
        engine.adopt_prefetched()
    assert ei.value.code == -4
    engine.prefetch_dlt4(9, 0, 512)
    engine.adopt_prefetched()
    got = engine.get_models()
    engine.propose_dlt4(9, 0, 512)
    assert np.array_equal(got.view(np.uint64), engine.get_models().view(np.uint64))


def test_resident_sweep_beside_a_prefetch_equals_the_dispatched_sweep(engine, synth, oracle):
    """r04: while a DLT prefetch is pending the materialising sweep runs as a resident grid that strides over the
    (model block, point slice) items and leaves workgroup slots free for the DLT (k_residual_resident).  Same matrix,
    same counts, bit for bit, as the hardware-dispatched launch — for several headrooms, a ragged model count and a point
    count that is not a multiple of the tile."""
    sc = synth.make_scene(20011, 4, seed=23, with_neighbours=False)
    _load(engine, sc)
    M = 6007
    engine.propose_dlt4(41, 0, M)
    engine.set_tuning(19, -1)                              # reference: one hardware-dispatched workgroup per item
    _, cnt_ref = engine.residual_matrix(THR2, fetch_R=False)
    rows = [0, 1, 15, 16, 3000, M - 1]
    R_ref = np.stack([engine.get_residual_rows(r, 1)[0] for r in rows])
    H = engine.get_models()
    with np.errstate(all="ignore"):
        want = oracle.residual_matrix(sc.src, sc.dst, H[rows])
    ok = ~np.isnan(want)
    assert np.array_equal(R_ref[ok].view(np.uint64), want[ok].view(np.uint64))
    try:
        for headroom in (64, 0, 700, -1):
            engine.set_tuning(19, headroom)
            engine.prefetch_dlt4(41, 0, M)                # pending: the sweep below is held until the DLT has been dispatched
            _, cnt = engine.residual_matrix(THR2, fetch_R=False)
            R = np.stack([engine.get_residual_rows(r, 1)[0] for r in rows])
            assert np.array_equal(cnt, cnt_ref), headroom
            assert np.array_equal(R.view(np.uint64), R_ref.view(np.uint64)), headroom
            engine.adopt_prefetched()
            assert np.array_equal(engine.get_models().view(np.uint64), H.view(np.uint64))
    finally:
        engine.set_tuning(19, 0)


def test_rejected_schedule_variants_are_not_in_the_product_library(mh, engine, synth):
    """r05: the measured-and-rejected schedules of r04 — slice-major item order of the resident sweep
    and of the resident cost-matrix kernel (mh_set_tuning keys 26 / 27), the batched near-pair form of k_cost32 (28), the
    other tilings of the FP32 pre-test score kernel (16) — are compiled into measurement libraries only
    (build.py --tuning; tools/cost32_probe.py, tools/sweep_order_probe.py, tools/score_bench.py load that one).  The
    product library answers MH_ERR_INVALID to any value but 0, and the default schedule is what runs."""
    sc = synth.make_scene(3001, 3, seed=29, with_neighbours=False)
    _load(engine, sc)
    engine.propose_dlt4(43, 0, 1500)
    C_ref, ccnt_ref = engine.cost_matrix()
    for key, value in ((16, 3), (26, 7), (27, 1), (28, 1)):
        with pytest.raises(mh.MultiHError) as ei:
            engine.set_tuning(key, value)
        assert ei.value.code == -2 and "MH_TUNING" in str(ei.value), key
        engine.set_tuning(key, 0)                     # (resetting to the default is always accepted)
    C1, ccnt = engine.cost_matrix()
    assert np.array_equal(ccnt, ccnt_ref) and np.array_equal(C1, C_ref)


def test_two_batches_prefetched_ahead(mh, engine, synth):
    """r04: the prefetch queue holds two batches (the batch after next is prepared too, so the DLT a sweep waits for was
    dispatched a whole sweep earlier).  First in, first out; a third prefetch is refused; tuples, homographies and the counts
    of the sweeps are those of mh_propose_dlt4, bit for bit, through several turns of the three buffers and changing sizes."""
    sc = synth.make_scene(3000, 3, seed=12, with_neighbours=False)
    _load(engine, sc)
    sizes = (700, 512, 1300, 700, 90, 2048, 333)
    want = []
    for i, m in enumerate(sizes):
        engine.propose_dlt4(33, 10000 * i, m)
        _, cnt = engine.residual_matrix(THR2, fetch_R=False)
        want.append((engine.get_samples(), engine.get_models(), cnt))
    engine.prefetch_dlt4(33, 0, sizes[0])
    engine.prefetch_dlt4(33, 10000, sizes[1])
    with pytest.raises(mh.MultiHError) as ei:
        engine.prefetch_dlt4(33, 20000, sizes[2])
    assert ei.value.code == -2
    for i, m in enumerate(sizes):
        engine.adopt_prefetched()
        if i + 2 < len(sizes):
            engine.prefetch_dlt4(33, 10000 * (i + 2), sizes[i + 2])          # beside / behind the sweep below
        _, cnt = engine.residual_matrix(THR2, fetch_R=False)
        engine.select_best(m, fetch=False)
        idx, H = engine.get_samples(), engine.get_models()
        assert np.array_equal(idx, want[i][0]) and np.array_equal(H.view(np.uint64), want[i][1].view(np.uint64)), i
        assert np.array_equal(cnt, want[i][2]), i
    with pytest.raises(mh.MultiHError):
        engine.adopt_prefetched()


def test_process_with_the_approximate_neighbourhood(mh, engine_lib):
    """MultiH::SetNeighbourApprox (r05): Process() with the FLANN-like neighbourhood built inside the class equals Process() with
    the same hit lists handed in through SetNeighbours (mhh_approx_neighbour_hits -> mhh_set_neighbour_hits) — labels, models,
    energy — and differs from the default neighbourhood's run only through the graph."""
    import ctypes as C
    host = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
    dp = C.POINTER(C.c_double)
    sc = mh.synth.make_scene(4000, 3, seed=12, with_neighbours=False)
    src, dst, aff, F, e2 = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff, sc.F, sc.e2))

    def process():
        labels = np.full(sc.n, -7, dtype=np.int32); H = np.zeros((64, 9)); it, en = C.c_int(0), C.c_double(0)
        k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), sc.n, F.ctypes.data_as(dp),
                                 e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(2.2), C.c_double(0.005), C.c_double(0.5), 20,
                                 C.c_ulonglong(5), 5000, 16, 0, None, 0, labels.ctypes.data_as(C.POINTER(C.c_int)), H.ctypes.data_as(dp), 64,
                                 C.byref(it), C.byref(en), None, 0, 4)
        return k, labels, H[:max(k, 0)].copy(), en.value

    seed = 0x464c414e4e
    host.mhh_set_neighbourhood_approx(4, 32, C.c_ulonglong(seed))
    try:
        inside = process()
    finally:
        host.mhh_set_neighbourhood_approx(0, 32, C.c_ulonglong(0))
    rowptr = np.zeros(sc.n + 1, dtype=np.int32); col = np.zeros(sc.n * 32, dtype=np.int32)
    total = host.mhh_approx_neighbour_hits(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), sc.n, 4, 32, C.c_double(1.0 / 0.005), C.c_ulonglong(seed),
                                           rowptr.ctypes.data_as(C.POINTER(C.c_int)), col.ctypes.data_as(C.POINTER(C.c_int)), col.size)
    assert total > 8 * sc.n
    host.mhh_set_neighbour_hits(rowptr.ctypes.data_as(C.POINTER(C.c_int)), col.ctypes.data_as(C.POINTER(C.c_int)), sc.n)
    try:
        handed_in = process()
    finally:
        host.mhh_set_neighbour_hits(None, None, 0)
    assert inside[0] == handed_in[0] >= 3 and np.array_equal(inside[1], handed_in[1]) and np.array_equal(inside[2], handed_in[2])
    assert inside[3] == handed_in[3]
    q = mh.synth.agreement(sc.gt_label, inside[1])
    assert q["planes_recovered"] == 3 and q["ari"] > 0.95, q
