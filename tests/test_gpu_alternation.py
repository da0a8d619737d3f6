"""The whole merge <-> label alternation (ClusterMergingAndLabeling, M/MultiH.cpp:224-312) through the
host class on the GPU against the ORACLE's independent restatement of the same loop
(oracle/mh_oracle.cpp section 11: mean shift in the reference's summation order, 3-point homographies with the LM
refinement, inlier scoring + collinearity filter, `changed`, LabelingStep with the reference's own GCoptimization
from oracle/_ref where it is built, the stop rule of :295).  Process() starts from SetInitialHomographies (what
EstablishStablePointSets hands over) with F given; the post-filter after the loop is switched off so that the
loop's own output is compared: labels, number of models, GetIterationNumber() and GetEnergy() must be EQUAL, the
homographies equal to 1e-9 relative (they are HAF re-estimates of equal label sets)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
THR, LAM, LOCALITY = 2.2, 0.5, 0.005


def _knn_hits(sc, k):
    """Exact float32 k-NN hit lists with the kernel's association order and tie rule (tests/test_gpu_parity.py
    checks the GPU builder against exactly this), cut at the reference's radius 1/locality like the class default."""
    pv = np.concatenate([sc.src, sc.dst], axis=1).astype(np.float32)
    n = sc.n
    rowptr, col = [0], []
    r2 = np.float32(1.0 / LOCALITY) ** 2
    for a in range(0, n, 1000):
        diff = pv[a:a + 1000, None, :] - pv[None, :, :]
        sq = diff * diff
        d = ((sq[..., 0] + sq[..., 1]) + sq[..., 2]) + sq[..., 3]
        d[np.arange(d.shape[0]), np.arange(a, a + d.shape[0])] = np.inf
        order = np.lexsort((np.broadcast_to(np.arange(n), d.shape), d), axis=1)[:, :k]
        for i in range(d.shape[0]):
            js = order[i][d[i, order[i]] <= r2]
            col.extend(js.tolist())
            rowptr.append(len(col))
    return np.asarray(rowptr, np.int32), np.asarray(col, np.int32)


def _initial_models(sc, seed, duplicates, strays):
    rng = np.random.default_rng(seed)
    H = [sc.H_true * (1.0 + rng.normal(0, 1e-4, size=sc.H_true.shape))]
    for _ in range(duplicates):                    # near-copies: the mean shift merges them -> `changed` iterations
        k = rng.integers(0, sc.H_true.shape[0])
        H.append(sc.H_true[k:k + 1] * (1.0 + rng.normal(0, 2e-4, size=(1, 9))))
    for _ in range(strays):                        # models nothing supports: the collinearity / inlier filter drops them
        H.append((np.eye(3) + rng.normal(0, 0.05, size=(3, 3))).reshape(1, 9))
    return np.ascontiguousarray(np.concatenate(H, axis=0))


@pytest.mark.parametrize("n,planes,seed,duplicates,strays", [(1000, 3, 2, 2, 0), (1000, 2, 7, 0, 2), (5000, 3, 1234, 3, 1),
                                                             (5000, 5, 11, 0, 0), (3000, 4, 5, 4, 2)])
def test_process_loop_equals_the_oracle_alternation(mh, engine_lib, synth, oracle, n, planes, seed, duplicates, strays):
    sc = synth.make_scene(n, planes, seed=seed, with_neighbours=False)
    H0 = _initial_models(sc, seed, duplicates, strays)
    rowptr, col = _knn_hits(sc, 16)
    lab_o, H_o, it_o, en_o, used_ref = oracle.cluster_merging_and_labeling(sc.src, sc.dst, sc.aff, H0, sc.F, sc.e2, LAM, THR,
                                                                            rowptr, col, seed)
    host = C.CDLL(os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so"))
    dp = C.POINTER(C.c_double)
    labels = np.full(n, -7, dtype=np.int32)
    Hout = np.zeros((64, 9))
    it, en = C.c_int(-1), C.c_double(-1)
    src, dst, aff = (np.ascontiguousarray(a) for a in (sc.src, sc.dst, sc.aff))
    F, e2 = np.ascontiguousarray(sc.F), np.ascontiguousarray(sc.e2)
    host.mhh_set_post_filter(0)
    try:
        k = host.mhh_run_process(src.ctypes.data_as(dp), dst.ctypes.data_as(dp), aff.ctypes.data_as(dp), n,
                                 F.ctypes.data_as(dp), e2.ctypes.data_as(dp), C.c_double(2.6), C.c_double(THR),
                                 C.c_double(LOCALITY), C.c_double(LAM), 20, C.c_ulonglong(seed), 0, 0, 0,
                                 H0.ctypes.data_as(dp), H0.shape[0], labels.ctypes.data_as(C.POINTER(C.c_int)),
                                 Hout.ctypes.data_as(dp), 64, C.byref(it), C.byref(en), None, 0, 4)
    finally:
        host.mhh_set_post_filter(1)
    assert k == H_o.shape[0], f"models: GPU {k}, oracle {H_o.shape[0]}"
    assert it.value == it_o, f"GetIterationNumber(): GPU {it.value}, oracle {it_o}"
    assert en.value == en_o, f"GetEnergy(): GPU {en.value}, oracle {en_o}"
    assert np.array_equal(labels, lab_o), f"{int((labels != lab_o).sum())} labels differ"
    if k > 1:
        scale = np.max(np.abs(H_o), axis=1, keepdims=True)
        assert np.max(np.abs(Hout[:k] - H_o) / scale) <= 1e-9
    assert k >= 2 and it_o >= 1
