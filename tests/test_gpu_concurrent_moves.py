"""r06: concurrent alpha-moves (csrc/expand.hip: k_batch_check / k_batch_commit, mh_set_tuning keys 37 and 38).

GCoptimization::oneExpansionIteration (GCoptimization.cpp:1278-1289) runs the expansions one label after the other, each on
the labeling the last one left.  The engine solves up to 16 consecutive moves together on the SAME labeling and keeps a move
only when a local test proves that the sequential order would have given the same result (every site the accepted
predecessors changed, and every neighbour of one, keeps its label in every minimum cut of the move); a move that fails heads
the next batch.  Labels, energy and cycle count must therefore be the sequential form's — key 37 = 1 — and the oracle's, for
every batch size, from a cold start and from a given labeling, with few and with hundreds of labels, on graphs where the test
fails often (near-duplicate models) and where it never does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THR, LAM = 2.2, 0.5
THR2 = THR * THR


def _load(engine, sc):
    engine.set_correspondences(sc.src, sc.dst, sc.aff)
    engine.set_epipolar(sc.F, sc.e2)
    engine.set_neighbors_csr(sc.hit_rowptr, sc.hit_col)


def _label_set(sc, rng, copies, spread, dlt=0, engine=None):
    """True planes, `copies` perturbed copies of them (spread 2e-4: near-duplicates whose inlier sets coincide — the moves on
    them touch the same sites, the test between them fails; 3e-3: neighbours in model space) and `dlt` random 4-point fits."""
    H = [sc.H_true]
    if copies:
        H.append(sc.H_true[rng.integers(0, sc.H_true.shape[0], copies)] * (1.0 + rng.normal(0, spread, size=(copies, 9))))
    if dlt:
        engine.propose_dlt4(int(rng.integers(1, 1 << 30)), 0, dlt)
        H.append(engine.get_models())
    return np.ascontiguousarray(np.concatenate(H, axis=0))


def _expand(engine, ctx, min_labels, init=None):
    engine.set_tuning(37, ctx)
    engine.set_tuning(38, min_labels)
    lab, en, cyc = engine.expand(init)
    return lab, en, cyc, engine.expand_batch_stats(), engine.expand_stats()


@pytest.fixture
def restore(engine):
    yield
    engine.set_tuning(37, 16)
    engine.set_tuning(38, 16)
    engine.set_tuning(39, 0)


@pytest.mark.parametrize("n,planes,copies,spread,dlt,seed", [
    (64, 2, 1, 2e-4, 0, 1),            # a handful of sites, three labels
    (1500, 3, 6, 2e-4, 0, 2),          # near-duplicates: most tests fail
    (3000, 4, 10, 3e-3, 20, 3),        # 35 labels, mixed
    (5000, 3, 0, 0.0, 60, 4),          # 64 labels, most of them supported by a few sites
    (4000, 5, 40, 3e-3, 100, 5),       # 146 labels
])
def test_every_batch_size_gives_the_sequential_result_and_the_oracles(engine, synth, oracle, restore, n, planes, copies, spread, dlt, seed):
    sc = synth.make_scene(n, planes, seed=seed)
    _load(engine, sc)
    rng = np.random.default_rng(seed)
    H = _label_set(sc, rng, copies, spread, dlt, engine)
    engine.set_models(H)
    cost = engine.data_cost()
    L = H.shape[0] + 1
    want = oracle.expand(cost, sc.hit_rowptr, sc.hit_col, oracle.potts(LAM))
    init = rng.integers(0, L, size=sc.n).astype(np.int32)              # a labeling nothing is consistent with: every move changes a lot
    want_w = oracle.expand(cost, sc.hit_rowptr, sc.hit_col, oracle.potts(LAM), init_labels=init)
    seq = _expand(engine, 1, 0)
    assert seq[3]["batches"] == 0 and seq[3]["batch_committed"] == 0
    assert np.array_equal(seq[0], want[0]) and seq[1] == want[1] and seq[2] == want[2]
    kept = failed = 0
    for ctx in (2, 3, 8, 16):
        for min_labels in (0, 16):                     # 0: batches from the first cycle on, whatever the label count
            got = _expand(engine, ctx, min_labels)
            assert np.array_equal(got[0], want[0]) and got[1] == want[1] and got[2] == want[2], (ctx, min_labels, got[3])
            b = got[3]
            assert b["moves_per_batch"] == ctx and b["batch_committed"] + b["solo_moves"] + b["host_skipped"] == got[4]["moves"] == got[2] * L, b
            kept += b["batch_committed"]
            failed += b["batch_invalid"]
            got_w = _expand(engine, ctx, min_labels, init)
            assert np.array_equal(got_w[0], want_w[0]) and got_w[1] == want_w[1] and got_w[2] == want_w[2], (ctx, min_labels, "from a given labeling")
            kept += got_w[3]["batch_committed"]
            failed += got_w[3]["batch_invalid"]
    assert kept > 0, "no move was ever kept out of a batch: the path under test did not run"
    if copies and spread < 1e-3:
        assert failed > 0, "near-duplicate models: some move must fail its test against its predecessor's changes"


def test_hundreds_of_labels_batches_pay_off_and_change_nothing(engine, synth, restore):
    """The reference's own route hands the loop several hundred stable-set models (M/MultiH.cpp:604-694).  540 labels on 20 000
    sites: the batched form keeps nearly every move out of a batch (r06: 1 203 of 1 620, 417 never launched, 49-66 batches cut
    short), needs a fifth of the launches, and gives the sequential labels, energy and cycle count."""
    sc = synth.make_scene(20000, 6, seed=1234)
    _load(engine, sc)
    rng = np.random.default_rng(1)
    H = _label_set(sc, rng, 133, 3e-3, 400, engine)
    engine.set_models(H)
    engine.data_cost()
    seq = _expand(engine, 1, 16)
    bat = _expand(engine, 16, 16)
    assert np.array_equal(seq[0], bat[0]) and seq[1] == bat[1] and seq[2] == bat[2]
    b = bat[3]
    moves = bat[4]["moves"]
    assert moves == seq[4]["moves"] == seq[2] * (H.shape[0] + 1)
    assert b["batch_committed"] >= 0.6 * moves and b["solo_moves"] <= 0.02 * moves, b
    assert b["batch_invalid"] < 0.15 * b["batch_committed"], b
    assert bat[4]["launches"] < 0.3 * seq[4]["launches"], (bat[4]["launches"], seq[4]["launches"])
    assert bat[4]["accepted"] == seq[4]["accepted"], "the same moves lower the energy in both forms"


@pytest.mark.parametrize("spw", [16, 32, 64])
def test_sites_per_wave_of_a_batchs_launches_change_nothing(engine, synth, oracle, restore, spw):
    """Key 39: the whole-graph launches of a batch take 16, 32 or 64 sites per wavefront (by the size of the launch when 0): a
    schedule, never a result — 146 labels on 4 000 sites and 35 on 3 000, cold and from a given labeling, against the oracle."""
    for n, planes, copies, dlt, seed in ((4000, 5, 40, 100, 5), (3000, 4, 10, 20, 3)):
        sc = synth.make_scene(n, planes, seed=seed)
        _load(engine, sc)
        rng = np.random.default_rng(seed)
        H = _label_set(sc, rng, copies, 3e-3, dlt, engine)
        engine.set_models(H)
        cost = engine.data_cost()
        init = rng.integers(0, H.shape[0] + 1, size=sc.n).astype(np.int32)
        want = oracle.expand(cost, sc.hit_rowptr, sc.hit_col, oracle.potts(LAM))
        want_w = oracle.expand(cost, sc.hit_rowptr, sc.hit_col, oracle.potts(LAM), init_labels=init)
        engine.set_tuning(39, spw)
        got = _expand(engine, 16, 0)
        got_w = _expand(engine, 16, 0, init)
        assert np.array_equal(got[0], want[0]) and got[1] == want[1] and got[2] == want[2]
        assert np.array_equal(got_w[0], want_w[0]) and got_w[1] == want_w[1] and got_w[2] == want_w[2]
        assert got[3]["batch_committed"] > 0


def test_the_loop_and_process_do_not_depend_on_the_batch_size(mh, engine_lib, synth):
    """Process() by the reference's own initialisation (hundreds of initial models) with 1, 4 and 16 moves per batch: the same
    labels, models, iteration count and energy, bit for bit."""
    import ctypes as C
    import os
    host = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
    sc = synth.make_scene(6000, 4, seed=9, with_neighbours=False)
    dp = C.POINTER(C.c_double)
    src, dst, aff, F, e2 = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff, sc.F, sc.e2))
    out = []
    for ctx in (1, 4, 16):
        host.mhh_set_engine_tuning(-1, 0)
        host.mhh_set_engine_tuning(37, ctx)
        labels = np.full(sc.n, -7, dtype=np.int32)
        Hout = np.zeros((256, 9))
        it, en = C.c_int(-1), C.c_double(-1)
        try:
            k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), sc.n, F.ctypes.data_as(dp),
                                     e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(THR), C.c_double(0.005), C.c_double(LAM), 20,
                                     C.c_ulonglong(9), 0, 0, 0, None, 0, labels.ctypes.data_as(C.POINTER(C.c_int)), Hout.ctypes.data_as(dp), 256,
                                     C.byref(it), C.byref(en), None, 0, -1)
        finally:
            host.mhh_set_engine_tuning(-1, 0)
        assert k >= 2
        out.append((k, labels.tobytes(), Hout[:k].tobytes(), it.value, en.value))
    assert out[0] == out[1] == out[2]
