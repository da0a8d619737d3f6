"""ctypes binding of include/multih_hip.h (libmultih_hip.so) for tests, bench
and multi-GPU plumbing.  One method per C entry point, numpy in / numpy out; no
computation happens in Python and nothing here falls back to the CPU: if the
library is missing or no gfx950 device is visible, construction raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MH_LIB: another build of the same ABI, e.g. the measurement library of `build.py --tuning` (tools/ only)
LIB_PATH = os.environ.get("MH_LIB") or os.path.join(HERE, "libmultih_hip.so")

MH_OK = 0
ERR_NAMES = {-1: "MH_ERR_NO_DEVICE", -2: "MH_ERR_INVALID", -3: "MH_ERR_HIP", -4: "MH_ERR_NOT_SET",
             -5: "MH_ERR_OVERFLOW"}
BUF_COUNTS, BUF_MODELS, BUF_RESIDUALS, BUF_LABELS, BUF_COST = 0, 1, 2, 3, 4
K_DLT4, K_RESIDUAL, K_SCORE, K_DATACOST, K_EXPAND, K_REESTIMATE, K_COSTMATRIX = 0, 1, 2, 3, 4, 5, 6

# every symbol include/multih_hip.h declares (tests check the export table against this)
SYMBOLS = [
    "mh_abi_version", "mh_last_error", "mh_device_count", "mh_create", "mh_destroy", "mh_set_params",
    "mh_set_stream", "mh_synchronize", "mh_set_correspondences", "mh_set_epipolar",
    "mh_set_neighbors_csr", "mh_build_neighbors_knn", "mh_build_neighbors_knn_radius", "mh_build_neighbors_radius", "mh_get_sym_graph", "mh_set_fundamental_metric", "mh_propose_fund8",
    "mh_get_fund_hypotheses", "mh_score_sampson", "mh_refit_fundamental", "mh_estimate_fundamental", "mh_epipoles", "mh_refine_correspondences", "mh_get_refine_reasons",
    "mh_local_homographies", "mh_mean_shift", "mh_propose_dlt4",
    "mh_set_models", "mh_get_models", "mh_get_model_count", "mh_get_samples", "mh_set_residual_mode", "mh_score",
    "mh_residual_matrix", "mh_cost_matrix", "mh_get_residual_rows", "mh_set_transport", "mh_select_greedy", "mh_get_score_stats", "mh_prefetch_dlt4", "mh_adopt_prefetched", "mh_select_best", "mh_get_copy_stats", "mh_inliers_of_model", "mh_inliers_of_homography", "mh_compat_trial_stats", "mh_compat_trial_stats_fit", "mh_inlier_moments", "mh_data_cost", "mh_expand",
    "mh_get_expand_stats", "mh_get_expand_batch_stats", "mh_get_expand_trace", "mh_get_core_components", "mh_reestimate", "mh_labeling_step", "mh_device_buffer", "mh_profile_enable", "mh_profile_reset",
    "mh_profile_get", "mh_set_tuning",
]


class MultiHError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen the engine; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} is not built; run `python multi-h_amd/build.py` "
                                "(or __graft_entry__.build())")
    lib = C.CDLL(p)
    lib.mh_last_error.restype = C.c_char_p
    lib.mh_destroy.restype = None
    if path is None:
        _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Engine:
    """Thin owner of one `mh_engine*`."""

    def __init__(self, device: int = 0, thr_fund_mat: float = 3.0, thr_hom: float = 2.5,
                 locality: float = 0.002, lam: float = 0.5, min_inliers: int = 0):
        self.lib = load_library()
        self._h = C.c_void_p()
        self._check(self.lib.mh_create(C.byref(self._h), int(device)))
        self.n = 0
        self.set_params(thr_fund_mat, thr_hom, locality, lam, min_inliers)

    # -- plumbing -----------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != MH_OK:
            raise MultiHError(rc, (self.lib.mh_last_error() or b"").decode())

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.mh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- parameters / inputs ------------------------------------------------
    def set_params(self, thr_fund_mat, thr_hom, locality, lam, min_inliers=0):
        self._check(self.lib.mh_set_params(self._h, C.c_double(thr_fund_mat), C.c_double(thr_hom),
                                           C.c_double(locality), C.c_double(lam), int(min_inliers)))
        self.thr_hom, self.lam = float(thr_hom), float(lam)

    def set_stream(self, stream_ptr: int | None, external: bool = True):
        """Launch on the caller's hipStream_t (0 / None = the HIP default stream); external=False
        returns to the engine's own stream."""
        self._check(self.lib.mh_set_stream(self._h, C.c_void_p(stream_ptr or 0), int(bool(external))))

    def synchronize(self):
        self._check(self.lib.mh_synchronize(self._h))

    def set_correspondences(self, src, dst, aff=None):
        src, dst = _f64(src), _f64(dst)
        n = src.shape[0]
        ap = None
        if aff is not None:
            aff = _f64(aff).reshape(n, 4)
            ap = _p(aff, C.c_double)
        self._check(self.lib.mh_set_correspondences(self._h, _p(src, C.c_double), _p(dst, C.c_double), ap, n))
        self.n = n

    def set_epipolar(self, F, e2):
        F, e2 = _f64(F).reshape(9), _f64(e2).reshape(2)
        self._check(self.lib.mh_set_epipolar(self._h, _p(F, C.c_double), _p(e2, C.c_double)))

    def set_neighbors_csr(self, rowptr, col):
        rowptr, col = _i32(rowptr), _i32(col)
        cp = _p(col, C.c_int) if col.size else None
        self._check(self.lib.mh_set_neighbors_csr(self._h, _p(rowptr, C.c_int), cp, rowptr.size - 1))

    def build_neighbors_knn(self, k: int, radius: float = 0.0):
        """k nearest hits per query; radius > 0 keeps only those within it (the reference's 1/locality)."""
        if radius > 0.0:
            self._check(self.lib.mh_build_neighbors_knn_radius(self._h, int(k), C.c_double(radius)))
        else:
            self._check(self.lib.mh_build_neighbors_knn(self._h, int(k)))

    def build_neighbors_radius(self, radius: float, max_hits: int = 0) -> int:
        hits = C.c_longlong(0)
        self._check(self.lib.mh_build_neighbors_radius(self._h, C.c_double(radius), C.c_longlong(max_hits), C.byref(hits)))
        return int(hits.value)

    def get_sym_graph(self):
        nnz = C.c_int(0)
        self._check(self.lib.mh_get_sym_graph(self._h, None, None, None, C.byref(nnz)))
        rp = np.empty(self.n + 1, dtype=np.int32)
        col = np.empty(nnz.value, dtype=np.int32)
        w = np.empty(nnz.value, dtype=np.int32)
        self._check(self.lib.mh_get_sym_graph(self._h, _p(rp, C.c_int), _p(col, C.c_int), _p(w, C.c_int), None))
        return rp, col, w

    # -- epipolar front half ---------------------------------------------------
    def propose_fund8(self, seed: int, first: int, m: int):
        self._check(self.lib.mh_propose_fund8(self._h, C.c_ulonglong(seed), C.c_longlong(first), int(m)))
        self._fm = int(m)

    def get_fund_hypotheses(self):
        F = np.empty((self._fm, 9), dtype=np.float64)
        idx = np.empty((self._fm, 8), dtype=np.int32)
        self._check(self.lib.mh_get_fund_hypotheses(self._h, _p(F, C.c_double), _p(idx, C.c_int)))
        return F, idx

    def score_sampson(self, thr2: float):
        cnt = np.empty(self._fm, dtype=np.int32)
        self._check(self.lib.mh_score_sampson(self._h, C.c_double(thr2), _p(cnt, C.c_int)))
        return cnt

    def refit_fundamental(self, F, thr2: float, iterations: int = 1):
        F = _f64(F).reshape(9)
        out = np.empty(9, dtype=np.float64)
        mask = np.empty(self.n, dtype=np.uint8)
        inl = C.c_int(0)
        self._check(self.lib.mh_refit_fundamental(self._h, _p(F, C.c_double), C.c_double(thr2), int(iterations),
                                                  _p(out, C.c_double), _p(mask, C.c_ubyte), C.byref(inl)))
        return out, mask, inl.value

    def set_fundamental_metric(self, metric: int) -> None:
        """0 = Sampson (default), 1 = the larger squared point-to-epipolar-line distance (cv::findFundamentalMat's)."""
        self._check(self.lib.mh_set_fundamental_metric(self._h, int(metric)))

    def estimate_fundamental(self, seed: int, hypotheses: int, thr: float):
        F = np.empty(9, dtype=np.float64)
        e2 = np.empty(2, dtype=np.float64)
        mask = np.empty(self.n, dtype=np.uint8)
        inl = C.c_int(0)
        self._check(self.lib.mh_estimate_fundamental(self._h, C.c_ulonglong(seed), int(hypotheses), C.c_double(thr),
                                                     _p(F, C.c_double), _p(e2, C.c_double), _p(mask, C.c_ubyte),
                                                     C.byref(inl)))
        return F, e2, mask, inl.value

    def epipoles(self, F):
        F = _f64(F).reshape(9)
        e1, e2 = np.empty(2), np.empty(2)
        self._check(self.lib.mh_epipoles(self._h, _p(F, C.c_double), _p(e1, C.c_double), _p(e2, C.c_double)))
        return e1, e2

    def refine_correspondences(self, F, e1, e2, in_mask=None):
        F, e1, e2 = _f64(F).reshape(9), _f64(e1).reshape(2), _f64(e2).reshape(2)
        keep = np.empty(self.n, dtype=np.uint8)
        out = np.empty((self.n, 8), dtype=np.float64)
        mp = None
        if in_mask is not None:
            in_mask = np.ascontiguousarray(in_mask, dtype=np.uint8)
            mp = _p(in_mask, C.c_ubyte)
        self._check(self.lib.mh_refine_correspondences(self._h, _p(F, C.c_double), _p(e1, C.c_double), _p(e2, C.c_double),
                                                       mp, _p(keep, C.c_ubyte), _p(out, C.c_double)))
        return keep, out

    def refine_reasons(self):
        """Per row of the last refine_correspondences call: 0 kept, 1 not in the mask, 2 triangulation, 3 affine test."""
        r = np.empty(self.n, dtype=np.uint8)
        self._check(self.lib.mh_get_refine_reasons(self._h, _p(r, C.c_ubyte), int(self.n)))
        return r

    # -- reference-style initialisation ------------------------------------------
    def local_homographies(self, locality: float):
        H = np.empty((self.n, 9), dtype=np.float64)
        feat = np.empty((self.n, 10), dtype=np.float64)
        self._check(self.lib.mh_local_homographies(self._h, C.c_double(locality), _p(H, C.c_double), _p(feat, C.c_double)))
        return H, feat

    def mean_shift(self, data, band_width: float, seed: int):
        data = _f64(data)
        n, d = data.shape
        modes = np.empty((n, d), dtype=np.float64)
        assign = np.empty(n, dtype=np.int32)
        k = C.c_int(0)
        self._check(self.lib.mh_mean_shift(self._h, _p(data, C.c_double), n, d, C.c_double(band_width),
                                           C.c_ulonglong(seed), _p(modes, C.c_double), n, _p(assign, C.c_int), C.byref(k)))
        return modes[:k.value].copy(), assign, k.value

    # -- propose ------------------------------------------------------------
    def propose_dlt4(self, seed: int, first: int, m: int):
        self._check(self.lib.mh_propose_dlt4(self._h, C.c_ulonglong(seed), C.c_longlong(first), int(m)))

    def set_models(self, H):
        H = _f64(H).reshape(-1, 9)
        self._check(self.lib.mh_set_models(self._h, _p(H, C.c_double), H.shape[0]))

    @property
    def model_count(self) -> int:
        m = C.c_int(0)
        self._check(self.lib.mh_get_model_count(self._h, C.byref(m)))
        return m.value

    def get_models(self):
        H = np.empty((self.model_count, 9), dtype=np.float64)
        self._check(self.lib.mh_get_models(self._h, _p(H, C.c_double)))
        return H

    def get_samples(self):
        idx = np.empty((self.model_count, 4), dtype=np.int32)
        self._check(self.lib.mh_get_samples(self._h, _p(idx, C.c_int)))
        return idx

    # -- score --------------------------------------------------------------
    def set_residual_mode(self, symmetric: bool):
        self._check(self.lib.mh_set_residual_mode(self._h, 1 if symmetric else 0))

    def score(self, thr2: float, mask=None, fetch: bool = True):
        cnt = np.empty(self.model_count, dtype=np.int32) if fetch else None
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            mp = _p(mask, C.c_ubyte)
        self._check(self.lib.mh_score(self._h, C.c_double(thr2), mp, _p(cnt, C.c_int) if fetch else None))
        return cnt

    def residual_matrix(self, thr2: float, fetch_R: bool = True, fetch_counts: bool = True):
        m = self.model_count
        R = np.empty((m, self.n), dtype=np.float64) if fetch_R else None
        cnt = np.empty(m, dtype=np.int32) if fetch_counts else None
        self._check(self.lib.mh_residual_matrix(self._h, C.c_double(thr2),
                                                _p(R, C.c_double) if fetch_R else None,
                                                _p(cnt, C.c_int) if fetch_counts else None))
        return R, cnt

    def cost_matrix(self, fetch_C: bool = True, fetch_counts: bool = True):
        """mh_cost_matrix: int32 data cost of every model against every point (model-major), fused inlier counts."""
        m = self.model_count
        Cm = np.empty((m, self.n), dtype=np.int32) if fetch_C else None
        cnt = np.empty(m, dtype=np.int32) if fetch_counts else None
        self._check(self.lib.mh_cost_matrix(self._h, _p(Cm, C.c_int) if fetch_C else None, _p(cnt, C.c_int) if fetch_counts else None))
        return Cm, cnt

    def get_residual_rows(self, first: int, count: int):
        rows = np.empty((count, self.n), dtype=np.float64)
        self._check(self.lib.mh_get_residual_rows(self._h, int(first), int(count), _p(rows, C.c_double)))
        return rows

    def select_greedy(self, thr2: float, need: int, max_models: int, mask=None, total_m: int = 0):
        """Greedy selection over the resident batch on the device (mh_select_greedy, one rank).
        Returns (H [k,9], counters [k], counts [k], mask_out or None)."""
        H = np.zeros((int(max_models), 9))
        counters = np.zeros(int(max_models), dtype=np.int64)
        counts = np.zeros(int(max_models), dtype=np.int32)
        k = C.c_int(0)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).copy()
        self._check(self.lib.mh_select_greedy(self._h, C.c_double(thr2), int(need), int(max_models),
                                              None if m is None else m.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                              _p(H, C.c_double), counters.ctypes.data_as(C.POINTER(C.c_longlong)),
                                              _p(counts, C.c_int), C.byref(k), C.c_longlong(int(total_m))))
        return H[:k.value].copy(), counters[:k.value].copy(), counts[:k.value].copy(), m

    def set_transport(self, rank: int, world: int, stream_fn=None, host_fn=None, ctx=None):
        """mh_set_transport: stream_fn = a C function pointer that enqueues the all-gather on the engine's stream
        (mhr_allgather of libmultih_rccl.so), host_fn = a host-synchronised ctypes hook (sharding.make_allgather_hook)."""
        self._transport_keepalive = (stream_fn, host_fn, ctx)
        self._check(self.lib.mh_set_transport(self._h, int(rank), int(world), C.cast(stream_fn, C.c_void_p) if stream_fn else None,
                                              C.cast(host_fn, C.c_void_p) if host_fn else None,
                                              ctx if isinstance(ctx, C.c_void_p) else C.c_void_p(ctx or 0)))

    def prefetch_dlt4(self, seed: int, first: int, m: int):
        self._check(self.lib.mh_prefetch_dlt4(self._h, C.c_ulonglong(seed), C.c_longlong(first), int(m)))

    def adopt_prefetched(self):
        self._check(self.lib.mh_adopt_prefetched(self._h))

    def select_best(self, total_m: int = 0, fetch: bool = True):
        """(index in the whole batch, count) of the best-supported model; fetch=False only enqueues."""
        if not fetch:
            self._check(self.lib.mh_select_best(self._h, C.c_longlong(int(total_m)), None, None))
            return None
        idx, cnt = C.c_longlong(0), C.c_int(0)
        self._check(self.lib.mh_select_best(self._h, C.c_longlong(int(total_m)), C.byref(idx), C.byref(cnt)))
        return int(idx.value), int(cnt.value)

    def score_stats(self, reset=False):
        """(pairs scored through the FP32 pre-test, pairs among them that needed the FP64 formula)"""
        a, b = C.c_longlong(0), C.c_longlong(0)
        self._check(self.lib.mh_get_score_stats(self._h, C.byref(a), C.byref(b), int(bool(reset))))
        return int(a.value), int(b.value)

    def copy_stats(self, reset=False):
        a, b = C.c_longlong(0), C.c_longlong(0)
        self._check(self.lib.mh_get_copy_stats(self._h, C.byref(a), C.byref(b), int(bool(reset))))
        return int(a.value), int(b.value)

    def inliers_of_model(self, idx: int, thr2: float, label_value: int, labels):
        labels = _i32(labels).copy()
        self._check(self.lib.mh_inliers_of_model(self._h, int(idx), C.c_double(thr2), int(label_value),
                                                 _p(labels, C.c_int)))
        return labels

    def inliers_of_homography(self, H, thr2: float, label_value: int, labels):
        labels = _i32(labels).copy()
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
        self._check(self.lib.mh_inliers_of_homography(self._h, _p(H, C.c_double), C.c_double(thr2), int(label_value),
                                                      _p(labels, C.c_int)))
        return labels

    def inlier_moments(self, thr2: float):
        m = self.model_count
        mo = np.empty((m, 6), dtype=np.float64)
        me = np.empty(m, dtype=np.float64)
        self._check(self.lib.mh_inlier_moments(self._h, C.c_double(thr2), _p(mo, C.c_double), _p(me, C.c_double)))
        return mo, me

    # -- label --------------------------------------------------------------
    def data_cost(self, fetch: bool = True):
        cost = np.empty((self.n, self.model_count + 1), dtype=np.int32) if fetch else None
        self._check(self.lib.mh_data_cost(self._h, _p(cost, C.c_int) if fetch else None))
        return cost

    def expand(self, init_labels=None):
        labels = np.empty(self.n, dtype=np.int32)
        ip = None
        if init_labels is not None:
            init_labels = _i32(init_labels)
            ip = _p(init_labels, C.c_int)
        energy, cycles = C.c_int(0), C.c_int(0)
        self._check(self.lib.mh_expand(self._h, ip, _p(labels, C.c_int), C.byref(energy), C.byref(cycles)))
        return labels, energy.value, cycles.value

    def compat_trial_stats(self, pts_xyxy, cluster_begin, tri, H, ok):
        """mh_compat_trial_stats: per (cluster, trial) the five middle order statistics and the three largest squared
        transfer errors.  tri [clusters, trials, 3], H [clusters, trials, 9], ok [clusters, trials]."""
        pts = _f64(pts_xyxy).reshape(-1, 4)
        begin = np.ascontiguousarray(cluster_begin, dtype=np.int32)
        tri = np.ascontiguousarray(tri, dtype=np.int32)
        clusters, trials = tri.shape[0], tri.shape[1]
        H = _f64(H).reshape(clusters, trials, 9)
        ok = np.ascontiguousarray(ok, dtype=np.uint8).reshape(clusters, trials)
        out = np.empty((clusters, trials, 8), dtype=np.float64)
        self._check(self.lib.mh_compat_trial_stats(self._h, _p(pts, C.c_double), _p(begin, C.c_int), clusters, _p(tri, C.c_int),
                                                   _p(H, C.c_double), _p(ok, C.c_ubyte), trials, _p(out, C.c_double)))
        return out

    def compat_trial_stats_fit(self, pts_xyxy, cluster_begin, tri, F):
        """mh_compat_trial_stats_fit: the trials' 3-point homographies fitted on the device.  Returns (stats [clusters, trials, 8],
        H [clusters, trials, 9], ok [clusters, trials])."""
        pts = _f64(pts_xyxy).reshape(-1, 4)
        begin = np.ascontiguousarray(cluster_begin, dtype=np.int32)
        tri = np.ascontiguousarray(tri, dtype=np.int32)
        clusters, trials = tri.shape[0], tri.shape[1]
        F = _f64(F).reshape(9)
        out = np.empty((clusters, trials, 8), dtype=np.float64)
        H = np.empty((clusters, trials, 9), dtype=np.float64)
        ok = np.empty((clusters, trials), dtype=np.uint8)
        self._check(self.lib.mh_compat_trial_stats_fit(self._h, _p(pts, C.c_double), _p(begin, C.c_int), clusters, _p(tri, C.c_int),
                                                       _p(F, C.c_double), trials, _p(out, C.c_double), _p(H, C.c_double), _p(ok, C.c_ubyte)))
        return out, H, ok

    def expand_stats(self):
        st = (C.c_longlong * 24)()
        self._check(self.lib.mh_get_expand_stats(self._h, st))
        return dict(zip(("cycles", "moves", "accepted", "push_phases", "relax_intervals", "host_syncs",
                         "reduce_launches", "flow_moves", "launches", "moves_run", "moves_solved",
                         "core_sites", "core_max", "barriers", "relabels", "solve_us", "barrier_us", "relax_us", "push_us", "tail_us",
                         "barrier_timeout_retries", "solver_workgroups", "xcd_local_moves", "longest_barrier_wait_us"),
                        list(st)))

    def expand_batch_stats(self):
        st = (C.c_longlong * 8)()
        self._check(self.lib.mh_get_expand_batch_stats(self._h, st))
        return dict(zip(("batches", "batch_committed", "batch_invalid", "host_skipped", "solo_moves", "reserved", "moves_per_batch",
                         "batch_min_labels"), list(st)))

    def expand_trace(self, moves: int):
        out = np.zeros((int(moves), 8), dtype=np.int32)
        self._check(self.lib.mh_get_expand_trace(self._h, _p(out, C.c_int), int(moves)))
        return out

    def core_components(self, moves: int):
        out = np.zeros((int(moves), 16), dtype=np.int32)
        self._check(self.lib.mh_get_core_components(self._h, _p(out, C.c_int), int(moves)))
        return out

    def reestimate(self, labels):
        labels = _i32(labels)
        H = np.empty((self.model_count, 9), dtype=np.float64)
        self._check(self.lib.mh_reestimate(self._h, _p(labels, C.c_int), _p(H, C.c_double)))
        return H

    def labeling_step(self, warm: bool, labeling):
        lab = _i32(labeling).copy()
        energy, cycles = C.c_double(0), C.c_int(0)
        self._check(self.lib.mh_labeling_step(self._h, int(bool(warm)), _p(lab, C.c_int), C.byref(energy),
                                              C.byref(cycles)))
        return lab, energy.value, cycles.value

    # -- device access / profiling -----------------------------------------
    def device_buffer(self, which: int):
        ptr, nbytes = C.c_void_p(), C.c_ulonglong(0)
        self._check(self.lib.mh_device_buffer(self._h, int(which), C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def profile_enable(self, on: bool = True):
        self._check(self.lib.mh_profile_enable(self._h, int(bool(on))))

    def profile_reset(self):
        self._check(self.lib.mh_profile_reset(self._h))

    def profile_get(self, kernel: int):
        n, ms = C.c_int(0), C.c_double(0)
        self._check(self.lib.mh_profile_get(self._h, int(kernel), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def set_tuning(self, key: int, value: int):
        self._check(self.lib.mh_set_tuning(self._h, int(key), int(value)))


def device_count() -> int:
    return int(load_library().mh_device_count())
