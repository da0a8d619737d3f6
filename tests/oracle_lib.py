"""ctypes bindings to the CPU oracle (oracle/_build/libmh_oracle.so) and, when
present, to the reference's own alpha-expansion build (oracle/_ref/
libmh_ref_gco.so).  TEST INFRASTRUCTURE: imported only from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke()."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libmh_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmh_ref_gco.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "mh_oracle.cpp")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(ORACLE_SO)
        _lib.mho_labeling_energy.restype = C.c_longlong
    return _lib


def ref():
    """The reference GCO build, or None when oracle/_ref was not built."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = C.CDLL(REF_SO)
    return _ref


def soa(src, dst):
    src, dst = f64(src), f64(dst)
    return f64(src[:, 0]), f64(src[:, 1]), f64(dst[:, 0]), f64(dst[:, 1])


# ---- oracle entry points -------------------------------------------------

def residual_matrix(src, dst, H):
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    R = np.empty((H.shape[0], x1.size), dtype=np.float64)
    lib().mho_residual_matrix(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0], _d(R))
    return R


def residual_matrix_sym(src, dst, H):
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    R = np.empty((H.shape[0], x1.size), dtype=np.float64)
    lib().mho_residual_matrix_sym(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0], _d(R))
    return R


def score(src, dst, H, thr2, mask=None):
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    cnt = np.empty(H.shape[0], dtype=np.int32)
    mp = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        mp = mask.ctypes.data_as(C.POINTER(C.c_ubyte))
    lib().mho_score(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0], C.c_double(thr2), mp, _i(cnt))
    return cnt


def score_sym(src, dst, H, thr2, mask=None):
    """Inlier counts on the symmetric transfer error (mho_score_sym)."""
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    cnt = np.empty(H.shape[0], dtype=np.int32)
    mp = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        mp = mask.ctypes.data_as(C.POINTER(C.c_ubyte))
    lib().mho_score_sym(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0], C.c_double(thr2), mp, _i(cnt))
    return cnt


def score_mt(src, dst, H, thr2):
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    cnt = np.empty(H.shape[0], dtype=np.int32)
    threads = lib().mho_score_mt(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0], C.c_double(thr2), _i(cnt))
    return cnt, int(threads)


def data_cost(src, dst, H, lam, thr2):
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    cost = np.empty((x1.size, H.shape[0] + 1), dtype=np.int32)
    lib().mho_data_cost(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0],
                        C.c_double(lam), C.c_double(thr2), _i(cost))
    return cost


def potts(lam):
    return int(lib().mho_potts(C.c_double(lam)))


def build_sym_graph(n, rowptr, col):
    rowptr, col = i32(rowptr), i32(col)
    nnz = lib().mho_build_sym_graph(n, _i(rowptr), _i(col), None, None, None)
    rp = np.empty(n + 1, dtype=np.int32)
    cl = np.empty(nnz, dtype=np.int32)
    w = np.empty(nnz, dtype=np.int32)
    lib().mho_build_sym_graph(n, _i(rowptr), _i(col), _i(rp), _i(cl), _i(w))
    return rp, cl, w


def expand(cost, rowptr, col, potts_v, init_labels=None, max_cycles=1000):
    cost = i32(cost)
    n, L = cost.shape
    rowptr, col = i32(rowptr), i32(col)
    labels = np.zeros(n, dtype=np.int32) if init_labels is None else i32(init_labels).copy()
    cycles = C.c_int(0)
    energies = np.zeros(max_cycles, dtype=np.int32)
    e = lib().mho_expand(n, L, _i(cost), _i(rowptr), _i(col), int(potts_v), _i(labels), int(max_cycles),
                         C.byref(cycles), _i(energies))
    return labels, int(e), int(cycles.value), energies[:cycles.value].copy()


def labeling_energy(cost, rowptr, col, potts_v, labels):
    cost = i32(cost)
    n, L = cost.shape
    rowptr, col, labels = i32(rowptr), i32(col), i32(labels)
    return int(lib().mho_labeling_energy(n, L, _i(cost), _i(rowptr), _i(col), int(potts_v), _i(labels)))


def jacobi_sym(a):
    a = f64(a)
    n = a.shape[0]
    v = np.empty((n, n))
    d = np.empty(n)
    lib().mho_jacobi_sym(n, _d(a), _d(v), _d(d))
    return d, v


def inlier_moments(src, dst, H, thr2):
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    mo = np.empty((H.shape[0], 6))
    me = np.empty(H.shape[0])
    lib().mho_inlier_moments(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0],
                             C.c_double(thr2), _d(mo), _d(me))
    return mo, me


def haf_reestimate(src, dst, aff, labels, H, F, e2):
    x1, y1, x2, y2 = soa(src, dst)
    aff, labels, F, e2 = f64(aff), i32(labels), f64(F), f64(e2)
    H = f64(H).reshape(-1, 9).copy()
    cnt = np.zeros(H.shape[0], dtype=np.int32)
    lib().mho_haf_reestimate(_d(x1), _d(y1), _d(x2), _d(y2), _d(aff), x1.size, _i(labels),
                             H.shape[0], _d(F), _d(e2), _d(H), _i(cnt))
    return H, cnt


def sample4(seed, m0, M, N):
    idx = np.empty((M, 4), dtype=np.int32)
    lib().mho_sample4(C.c_ulonglong(seed), C.c_longlong(m0), M, N, _i(idx))
    return idx


def rr_schedule():
    s = np.empty((9, 4, 2), dtype=np.int32)
    lib().mho_rr_schedule(_i(s))
    return s


def dlt4(src, dst, idx):
    x1, y1, x2, y2 = soa(src, dst)
    idx = i32(idx).reshape(-1, 4)
    M = idx.shape[0]
    H = np.empty((M, 9))
    wit = np.empty(M)
    sw = np.empty(M, dtype=np.int32)
    lib().mho_dlt4(_d(x1), _d(y1), _d(x2), _d(y2), _i(idx), M, _d(H), _d(wit), _i(sw))
    return H, wit, sw


def poly_roots(coeffs):
    c = f64(coeffs)
    n = c.size - 1
    re, im = np.empty(n), np.empty(n)
    lib().mho_poly_roots(_d(c), n, _d(re), _d(im))
    return re + 1j * im


def refine_points(src, dst, aff, F, e1, e2, in_mask=None, with_reasons=False):
    x1, y1, x2, y2 = soa(src, dst)
    aff, F, e1, e2 = f64(aff), f64(F), f64(e1), f64(e2)
    keep = np.empty(x1.size, dtype=np.uint8)
    out = np.empty((x1.size, 8))
    reason = np.empty(x1.size, dtype=np.uint8)
    mp = None
    if in_mask is not None:
        in_mask = np.ascontiguousarray(in_mask, dtype=np.uint8)
        mp = in_mask.ctypes.data_as(C.POINTER(C.c_ubyte))
    lib().mho_refine_points_ex(_d(x1), _d(y1), _d(x2), _d(y2), _d(aff), x1.size, _d(F), _d(e1), _d(e2), mp,
                               keep.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(out), reason.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return (keep, out, reason) if with_reasons else (keep, out)


def haf_point(src, dst, aff, F, e2, locality):
    x1, y1, x2, y2 = soa(src, dst)
    aff, F, e2 = f64(aff), f64(F), f64(e2)
    H = np.empty((x1.size, 9))
    feat = np.empty((x1.size, 10))
    lib().mho_haf_point(_d(x1), _d(y1), _d(x2), _d(y2), _d(aff), x1.size, _d(F), _d(e2), C.c_double(locality),
                        _d(H), _d(feat))
    return H, feat


def mean_shift(data, bw, seed):
    data = f64(data)
    n, d = data.shape
    modes = np.empty((n, d))
    assign = np.empty(n, dtype=np.int32)
    k = lib().mho_mean_shift(_d(data), n, d, C.c_double(bw), C.c_ulonglong(seed), _d(modes), n, _i(assign))
    return modes[:k].copy(), assign, int(k)


def sample8(seed, m0, M, N):
    idx = np.empty((M, 8), dtype=np.int32)
    lib().mho_sample8(C.c_ulonglong(seed), C.c_longlong(m0), M, N, _i(idx))
    return idx


def fund8(src, dst, idx):
    x1, y1, x2, y2 = soa(src, dst)
    idx = i32(idx).reshape(-1, 8)
    F = np.empty((idx.shape[0], 9))
    lib().mho_fund8(_d(x1), _d(y1), _d(x2), _d(y2), _i(idx), idx.shape[0], _d(F))
    return F


def set_fundamental_metric(metric):
    """0 = Sampson, 1 = the larger squared point-to-epipolar-line distance (what cv::findFundamentalMat thresholds)."""
    lib().mho_set_fundamental_metric(int(metric))


def sampson_score(src, dst, F, thr2):
    x1, y1, x2, y2 = soa(src, dst)
    F = f64(F).reshape(-1, 9)
    cnt = np.empty(F.shape[0], dtype=np.int32)
    lib().mho_sampson_score(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(F), F.shape[0], C.c_double(thr2), _i(cnt))
    return cnt


def sampson(src, dst, F):
    x1, y1, x2, y2 = soa(src, dst)
    F = f64(F).reshape(9)
    d = np.empty(x1.size)
    lib().mho_sampson(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(F), _d(d))
    return d


def fund_refit(src, dst, F, thr2):
    x1, y1, x2, y2 = soa(src, dst)
    F = f64(F).reshape(9)
    out = np.empty(9)
    mask = np.empty(x1.size, dtype=np.uint8)
    cnt = lib().mho_fund_refit(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(F), C.c_double(thr2), _d(out),
                               mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out, mask, int(cnt)


def labeling_step(src, dst, aff, H, lam, thr2, rowptr, col, warm, F, e2, labeling):
    x1, y1, x2, y2 = soa(src, dst)
    aff, F, e2 = f64(aff), f64(F), f64(e2)
    H = f64(H).reshape(-1, 9).copy()
    rowptr, col = i32(rowptr), i32(col)
    lab = i32(labeling).copy()
    cyc = C.c_int(0)
    e = lib().mho_labeling_step(_d(x1), _d(y1), _d(x2), _d(y2), _d(aff), x1.size, _d(H), H.shape[0],
                                C.c_double(lam), C.c_double(thr2), _i(rowptr), _i(col), int(bool(warm)),
                                _d(F), _d(e2), _i(lab), C.byref(cyc))
    return lab, H, int(e), int(cyc.value)


def homography_3pt(p1, p2, F, refine=True):
    p1, p2, F = f64(p1).reshape(-1, 2), f64(p2).reshape(-1, 2), f64(F).reshape(9)
    H = np.zeros(9)
    ok = lib().mho_homography_3pt(_d(p1), _d(p2), p1.shape[0], _d(F), _d(H), int(bool(refine)))
    return H, bool(ok)


def merging_step(src, dst, H, F, thr_h, seed, straightness=0.005):
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    kept = np.zeros_like(H)
    changed, draws = C.c_int(0), C.c_ulonglong(0)
    k = lib().mho_merging_step(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0], _d(f64(F)), C.c_double(thr_h),
                               C.c_double(straightness), C.c_ulonglong(seed), _d(kept), C.byref(changed), C.byref(draws))
    return kept[:k].copy(), bool(changed.value), int(draws.value)


_EXPAND_HOOK = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                           C.POINTER(C.c_int), C.POINTER(C.c_int))


def cluster_merging_and_labeling(src, dst, aff, H0, F, e2, lam, thr_h, rowptr, col, seed, use_reference_gco=True,
                                 straightness=0.005, max_models=256):
    """The oracle's ClusterMergingAndLabeling (M/MultiH.cpp:263-311) from initial models H0.  With
    use_reference_gco and oracle/_ref built, every alpha-expansion inside it is the REFERENCE's own GCoptimization."""
    x1, y1, x2, y2 = soa(src, dst)
    aff, F, e2 = f64(aff), f64(F), f64(e2)
    H0 = f64(H0).reshape(-1, 9)
    H = np.zeros((max(max_models, H0.shape[0]), 9))
    H[:H0.shape[0]] = H0
    rowptr, col = i32(rowptr), i32(col)
    lab = np.empty(x1.size, dtype=np.int32)
    it, en = C.c_int(0), C.c_double(0)
    hook = None
    if use_reference_gco and ref() is not None:
        hook = C.cast(ref().ref_gco_expand_table, _EXPAND_HOOK)
    fn = lib().mho_cluster_merging_and_labeling
    fn.restype = C.c_int
    k = fn(_d(x1), _d(y1), _d(x2), _d(y2), _d(aff), x1.size, _d(H), H0.shape[0], H.shape[0], _d(F), _d(e2),
           C.c_double(lam), C.c_double(thr_h), C.c_double(straightness), _i(rowptr), _i(col), C.c_ulonglong(seed),
           hook if hook is not None else _EXPAND_HOOK(0), _i(lab), C.byref(it), C.byref(en))
    return lab, H[:k].copy(), int(it.value), float(en.value), hook is not None


def select_greedy_refit(src, dst, aff, F, e2, H, thr2, need, max_models, mask=None):
    """mho_select_greedy_refit: every round's winner refitted to its inliers (HAF least squares, one label) before the claim."""
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    aff, F, e2 = f64(aff), f64(F).reshape(9), f64(e2).reshape(2)
    m = np.ones(x1.size, np.uint8) if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).copy()
    Hs = np.zeros((max_models, 9))
    idx = np.zeros(max_models, np.int64)
    cnt = np.zeros(max_models, np.int32)
    k = lib().mho_select_greedy_refit(_d(x1), _d(y1), _d(x2), _d(y2), _d(aff), x1.size, _d(F), _d(e2), _d(H), H.shape[0],
                                      C.c_double(thr2), int(need), int(max_models), m.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(Hs),
                                      idx.ctypes.data_as(C.POINTER(C.c_longlong)), _i(cnt))
    return Hs[:k].copy(), idx[:k].copy(), cnt[:k].copy(), m


def select_greedy(src, dst, H, thr2, need, max_models, mask=None, symmetric=False):
    """The oracle's sequential best-first selection (oracle/mh_oracle.cpp section 12).  Returns
    (H_selected, hypothesis indices, counts, mask_out).  symmetric: scores and claims on the symmetric transfer error."""
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    m = np.ones(x1.size, np.uint8) if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).copy()
    Hs = np.zeros((max_models, 9))
    idx = np.zeros(max_models, np.int64)
    cnt = np.zeros(max_models, np.int32)
    fn = lib().mho_select_greedy_sym if symmetric else lib().mho_select_greedy
    k = fn(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0], C.c_double(thr2), int(need),
           int(max_models), m.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(Hs),
           idx.ctypes.data_as(C.POINTER(C.c_longlong)), _i(cnt))
    return Hs[:k].copy(), idx[:k].copy(), cnt[:k].copy(), m


def compatibility_check(src, dst, labels, H, F, sqr_thr, min_inliers, seed):
    """HomographyCompatibilityCheck (M/MultiH.cpp:100-222) as the oracle restates it, on the oracle's own 3-point solver.
    Returns (labels_out, H_kept, medians)."""
    s, d = f64(src).reshape(-1, 2), f64(dst).reshape(-1, 2)
    H = f64(H).reshape(-1, 9).copy()
    lab = i32(labels).copy()
    med = np.full(H.shape[0], np.nan)
    k = lib().mho_compatibility_check(_d(s), _d(d), s.shape[0], _i(lab), _d(H), H.shape[0], _d(f64(F)), C.c_double(sqr_thr),
                                      int(min_inliers), C.c_ulonglong(seed), _d(med))
    return lab, H[:k].copy(), med


def establish_stable_point_sets(src, dst, aff, F, e2, locality, thr_h, ms_seed, max_models=None):
    x1, y1, x2, y2 = soa(src, dst)
    cap = x1.size if max_models is None else max_models
    H = np.zeros((cap, 9))
    k = lib().mho_establish_stable_point_sets(_d(x1), _d(y1), _d(x2), _d(y2), _d(f64(aff)), x1.size, _d(f64(F)), _d(f64(e2)),
                                              C.c_double(locality), C.c_double(thr_h), C.c_ulonglong(ms_seed), _d(H), cap)
    return H[:k].copy()


def process(src, dst, aff, F, e2, thr_h, locality, lam, min_inliers, seed, rowptr, col, *, init_H=None, init_mode=None,
            hypotheses=0, max_propose=32, post_filter=True, use_reference_gco=True, straightness=0.005, max_models=None):
    """The oracle's Process() from a known F (oracle/mh_oracle.cpp section 12; M/MultiH.cpp:42-98).  init_mode 0 = init_H,
    1 = stable point sets, 2 = DLT proposals + greedy selection.  Returns a dict: labels, H, iterations, energy,
    removed_by_filter, degenerate_tail, used_reference_gco."""
    x1, y1, x2, y2 = soa(src, dst)
    if init_mode is None:
        init_mode = 0 if init_H is not None else 2
    H0 = f64(init_H).reshape(-1, 9) if init_H is not None else np.zeros((0, 9))
    cap = max_models or (x1.size if init_mode == 1 else max(256, H0.shape[0]))
    H = np.zeros((cap, 9))
    rowptr, col = i32(rowptr), i32(col)
    lab = np.empty(x1.size, dtype=np.int32)
    it, en, rem, deg = C.c_int(0), C.c_double(0), C.c_int(0), C.c_int(0)
    hook = None
    if use_reference_gco and ref() is not None:
        hook = C.cast(ref().ref_gco_expand_table, _EXPAND_HOOK)
    fn = lib().mho_process
    fn.restype = C.c_int
    k = fn(_d(x1), _d(y1), _d(x2), _d(y2), _d(f64(aff)), x1.size, _d(f64(F)), _d(f64(e2)), C.c_double(thr_h),
           C.c_double(locality), C.c_double(lam), int(min_inliers), C.c_double(straightness), C.c_ulonglong(seed),
           int(init_mode), _d(H0) if H0.size else None, H0.shape[0], int(hypotheses), int(max_propose), _i(rowptr), _i(col),
           hook if hook is not None else _EXPAND_HOOK(0), int(bool(post_filter)), _d(H), cap, _i(lab), C.byref(it),
           C.byref(en), C.byref(rem), C.byref(deg))
    assert k >= 0, "the oracle's model capacity was too small"
    return {"labels": lab, "H": H[:k].copy(), "iterations": int(it.value), "energy": float(en.value),
            "removed_by_filter": int(rem.value), "degenerate_tail": bool(deg.value), "used_reference_gco": hook is not None}


# ---- reference GCO (oracle/_ref) -----------------------------------------

def ref_expand_table(cost, rowptr, col, potts_v, init_labels=None):
    r = ref()
    cost = i32(cost)
    n, L = cost.shape
    rowptr, col = i32(rowptr), i32(col)
    out = np.empty(n, dtype=np.int32)
    init = None if init_labels is None else _i(i32(init_labels))
    e = r.ref_gco_expand_table(n, L, _i(cost), _i(rowptr), _i(col), int(potts_v), init, _i(out))
    return out, int(e)


def ref_expand_formula(src, dst, H, lam, thr2, rowptr, col, init_labels=None):
    r = ref()
    x1, y1, x2, y2 = soa(src, dst)
    H = f64(H).reshape(-1, 9)
    rowptr, col = i32(rowptr), i32(col)
    out = np.empty(x1.size, dtype=np.int32)
    init = None if init_labels is None else _i(i32(init_labels))
    e = r.ref_gco_expand_formula(_d(x1), _d(y1), _d(x2), _d(y2), x1.size, _d(H), H.shape[0],
                                 C.c_double(lam), C.c_double(thr2), _i(rowptr), _i(col), init, _i(out))
    return out, int(e)


def merge_candidates(H, F, thr_h, seed):
    """mho_merge_candidates: (features [Nh,6], modes [k,6], candidates [nc,9], the mode of each candidate, draws)."""
    H = f64(H).reshape(-1, 9)
    nh = H.shape[0]
    feat, modes, cand = np.zeros((nh, 6)), np.zeros((nh, 6)), np.zeros((nh, 9))
    cand_mode = np.zeros(nh, dtype=np.int32)
    k, draws = C.c_int(0), C.c_ulonglong(0)
    nc = lib().mho_merge_candidates(_d(H), nh, _d(f64(F).reshape(9)), C.c_double(thr_h), C.c_ulonglong(seed), _d(feat), _d(modes),
                                    C.byref(k), _d(cand), _i(cand_mode), C.byref(draws))
    return feat, modes[:k.value].copy(), cand[:nc].copy(), cand_mode[:nc].copy(), int(draws.value)


def epipoles(F):
    e1, e2 = np.zeros(2), np.zeros(2)
    lib().mho_epipoles(_d(f64(F).reshape(9)), _d(e1), _d(e2))
    return e1, e2


def front_half(src, dst, aff, seed, hypotheses, thr_f, with_reasons=False):
    """mho_front_half: (kept or -1, F, e1, e2, keep mask, refined [n,8]); with_reasons: + the stage each row left at
    (0 kept, 1 not in the RANSAC mask, 2 OptimalTriangulation, 3 distanceError > 1)."""
    x1, y1, x2, y2 = soa(src, dst)
    aff = f64(aff)
    F, e1, e2 = np.zeros(9), np.zeros(2), np.zeros(2)
    keep = np.zeros(x1.size, dtype=np.uint8)
    refined = np.zeros((x1.size, 8))
    reason = np.zeros(x1.size, dtype=np.uint8)
    k = lib().mho_front_half_ex(_d(x1), _d(y1), _d(x2), _d(y2), _d(aff), x1.size, C.c_ulonglong(seed), int(hypotheses), C.c_double(thr_f),
                                _d(F), _d(e1), _d(e2), keep.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(refined),
                                reason.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return (int(k), F, e1, e2, keep, refined, reason) if with_reasons else (int(k), F, e1, e2, keep, refined)
