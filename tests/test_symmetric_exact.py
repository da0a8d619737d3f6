"""The symmetric transfer error against an INDEPENDENT statement of its definition in exact rational arithmetic
(until r04 the mode was checked against the builder's own restatement only).

Definition (Hartley & Zisserman, symmetric transfer error):  d2 = ||p2 - H p1||^2 + ||p1 - H^-1 p2||^2,  with the TRUE
inverse of H (fractions.Fraction on the doubles' exact values: no rounding anywhere).  Three implementations are held
against it on a 200 x 50 block: the oracle's C restatement (CPU, this file's first test) and the engine's residual matrix
(GPU, second test; bit-equal to the oracle's, tests/test_gpu_parity.py::test_symmetric_transfer_mode).

Tolerance.  A bound in ulps of d2 itself ("2 ulp") is not what a one-pass FP64 evaluation can meet: the forward image
u = N / s carries the rounding of N and s relative to the SUM OF MAGNITUDES of their terms, so (x2 - u) is off by about
10^-13 px on coordinates of 10^3 however small the residual, and a hypothesis whose horizon passes near a point (|s| small
against its terms) is ill-conditioned by that ratio.  The test therefore carries a first-order running error bound through
the formula in exact arithmetic — 3 roundings on each of N1, N2, s (relative to the magnitudes of their terms), the two
divisions, the subtractions and squares; for the backward part also the rounding of the adjugate's entries, each a
difference of two rounded products — doubles it for the second-order terms, and asserts |d2 - exact| <= bound pair by
pair (pairs whose denominator is uncertain by more than 0.1 % are counted and skipped: fewer than 2 % here).  Every INLIER
DECISION the exact value makes outside its bound must be the implementation's."""
from fractions import Fraction

import numpy as np
import pytest


U = Fraction(1, 2 ** 53)


def _transfer(c, dc, x, y, X, Y):
    """Exact squared transfer distance of (x, y) -> (X, Y) under the 3x3 map c, and a first-order bound on the error of
    its FP64 evaluation in the reference's operation order when the coefficients themselves are uncertain by dc.
    Returns (d2, bound) or None when the denominator is zero / too uncertain to bound."""
    ax, ay = abs(x), abs(y)
    n1, n2, s = c[0] * x + c[1] * y + c[2], c[3] * x + c[4] * y + c[5], c[6] * x + c[7] * y + c[8]
    if s == 0:
        return None
    mag = lambda i: abs(c[i] * x) + abs(c[i + 1] * y) + abs(c[i + 2])
    unc = lambda i: dc[i] * ax + dc[i + 1] * ay + dc[i + 2]
    e1, e2, es = 3 * U * mag(0) + unc(0), 3 * U * mag(3) + unc(3), 3 * U * mag(6) + unc(6)
    if es * 1000 > abs(s):
        return None
    u, v = n1 / s, n2 / s
    eu = (e1 + abs(u) * es) / abs(s) + U * abs(u)
    ev = (e2 + abs(v) * es) / abs(s) + U * abs(v)
    dx, dy = X - u, Y - v
    edx, edy = eu + U * (abs(X) + abs(u)), ev + U * (abs(Y) + abs(v))
    d2 = dx * dx + dy * dy
    return d2, 2 * (2 * abs(dx) * edx + edx * edx + 2 * abs(dy) * edy + edy * edy + 3 * U * d2)


def _exact_block(src, dst, H):
    """out[m, n] = (exact symmetric transfer error through the TRUE inverse, bound on an FP64 evaluation's error) or None."""
    out = np.empty((H.shape[0], src.shape[0]), dtype=object)
    zero = [Fraction(0)] * 9
    for m in range(H.shape[0]):
        h = [Fraction(float(v)) for v in H[m]]
        a, b, c, d, e, f, g, hh, i = h
        det = a * (e * i - f * hh) - b * (d * i - f * g) + c * (d * hh - e * g)
        prods = [(e * i, f * hh), (c * hh, b * i), (b * f, c * e), (f * g, d * i), (a * i, c * g), (c * d, a * f),
                 (d * hh, e * g), (b * g, a * hh), (a * e, b * d)]
        adj = [p - q for p, q in prods]
        dadj = [2 * U * (abs(p) + abs(q)) for p, q in prods]            # two rounded products and their rounded difference
        for n in range(src.shape[0]):
            x1, y1, x2, y2 = (Fraction(float(v)) for v in (src[n, 0], src[n, 1], dst[n, 0], dst[n, 1]))
            if det == 0:
                out[m, n] = None
                continue
            # the definition, through the true inverse adj / det (one projective map with adj: the scale cancels)
            inv = [v / det for v in adj]
            fw = _transfer(h, zero, x1, y1, x2, y2)
            bw_def = _transfer(inv, zero, x2, y2, x1, y1)
            bw = _transfer(adj, dadj, x2, y2, x1, y1)
            if fw is None or bw is None or bw_def is None:
                out[m, n] = None
                continue
            assert bw_def[0] == bw[0]                                    # adj and adj / det are the same map, exactly
            total = fw[0] + bw_def[0]
            out[m, n] = (total, fw[1] + bw[1] + 2 * U * total)
    return out


def _block(synth, oracle):
    sc = synth.make_scene(200, 3, seed=41, with_neighbours=False)
    idx = oracle.sample4(9, 0, 46, sc.n)
    Hd, _, _ = oracle.dlt4(sc.src, sc.dst, idx)
    H = np.concatenate([sc.H_true, sc.H_true[:1] * 3.7, Hd])[:50]        # truth, a rescaled copy (scale must cancel), DLT hypotheses
    return sc, np.ascontiguousarray(H)


def _check(R, exact, thr2, what):
    worst, n_pairs, decisions = 0.0, 0, 0
    for m in range(R.shape[0]):
        for n in range(R.shape[1]):
            if exact[m, n] is None or not np.isfinite(R[m, n]):
                continue
            ex, tol = exact[m, n]
            err = abs(Fraction(float(R[m, n])) - ex)
            assert err <= tol, f"{what}: model {m}, point {n}: {float(R[m, n])!r} vs exact {float(ex)!r} (|error| {float(err):.3e} > bound {float(tol):.3e})"
            worst = max(worst, float(err / tol))
            n_pairs += 1
            if abs(ex - Fraction(thr2)) > tol:                            # outside the bound the decision is the exact one
                assert (float(R[m, n]) < thr2) == (ex < Fraction(thr2))
                decisions += 1
    assert n_pairs > 0.98 * R.size and decisions > 0.99 * n_pairs, (n_pairs, decisions, R.size)
    return worst


def test_oracle_symmetric_residual_against_exact_rationals(synth, oracle):
    sc, H = _block(synth, oracle)
    exact = _exact_block(sc.src, sc.dst, H)
    with np.errstate(all="ignore"):
        R = oracle.residual_matrix_sym(sc.src, sc.dst, H)
    worst = _check(R, exact, 2.2 ** 2, "oracle")
    print(f"worst |d2 - exact| / bound = {worst:.3f}")
    assert worst > 1e-3, "a bound a thousand times above every error checks nothing"
    # the rescaled copy of a model scores exactly like the model wherever both are finite (H and 3.7 H are one map; the
    # roundings differ, the decisions must not outside the band) — and the definition is symmetric in the two images
    k = sc.H_true.shape[0]
    assert np.allclose(R[0], R[k], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_gpu_symmetric_residual_against_exact_rationals(engine, synth, oracle):
    sc, H = _block(synth, oracle)
    exact = _exact_block(sc.src, sc.dst, H)
    engine.set_correspondences(sc.src, sc.dst, sc.aff)
    engine.set_models(H)
    engine.set_residual_mode(True)
    try:
        R, cnt = engine.residual_matrix(2.2 ** 2)
    finally:
        engine.set_residual_mode(False)
    _check(R, exact, 2.2 ** 2, "engine")
    assert np.array_equal(cnt, (R < 2.2 ** 2).sum(axis=1))
