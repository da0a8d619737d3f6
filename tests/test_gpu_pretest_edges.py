"""Directed tests on the edges of the FP32 pre-test of the score kernels (csrc/score32.hip:32-48).
The cheap sufficient test declares a pair "provably not an inlier" when, in FP32,
    sigma = |s^| >= tau = 64 E_s      and      W^ = max(|Wx^|, |Wy^|) >= max(k1 sigma, 25.2 A)
with Wx^ = fl(x2~ s^ - nx^).  The pass-one arithmetic is re-enacted here in exact rational arithmetic rounded to FP32 once
per fused operation, the model constants as k_model32 forms them, and destination coordinates are then placed so that W^
lands ON each edge and one / two FP32 steps either side of it — for models whose horizon is far (k1 sigma decides), for
points next to a model's horizon (25.2 A decides, and sigma on tau itself), with coordinates just under the 2^20
eligibility limit where 25.4 u Cmax takes over from 1.12 thr, and with thr^2 on the limits 2^-40 / 2^40.  Whatever path a
pair takes, the counts must be the FP64 formula's (mho_score)."""
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U = 2.0 ** -24


def rn32(q: Fraction) -> np.float32:
    """q rounded to the nearest FP32 (ties to even); normal range only."""
    if q == 0:
        return np.float32(0.0)
    sign = -1.0 if q < 0 else 1.0
    a = abs(q)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    scaled = a / Fraction(2) ** (e - 23)                    # in [2^23, 2^24)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return np.float32(sign * n * 2.0 ** (e - 23))


def fma32(a, b, c) -> np.float32:
    return rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def model_constants(h, X, Y, Cmax):
    """tau and 25.2 A of k_model32 (FP64 arithmetic, each operation rounded once, then to FP32)."""
    a_s = abs(h[6]) * X + abs(h[7]) * Y + abs(h[8])
    a_n = max(abs(h[0]) * X + abs(h[1]) * Y + abs(h[2]), abs(h[3]) * X + abs(h[4]) * Y + abs(h[5]))
    es, en = 5.0 * U * a_s, 5.0 * U * a_n
    up = 1.0 + 2.0 ** -22
    return np.float32(64.0 * es * up), np.float32(25.2 * 1.01 * (Cmax * es + en) * up)


def k1_of(thr2, Cmax):
    return np.float32(np.float32(max(1.12 * np.sqrt(abs(thr2)), 25.4 * U * Cmax) * (1.0 + 1e-6)) + np.float32(1e-30))


def pass_one(h32, fx, fy):
    s = fma32(h32[6], fx, fma32(h32[7], fy, h32[8]))
    nx = fma32(h32[0], fx, fma32(h32[1], fy, h32[2]))
    ny = fma32(h32[3], fx, fma32(h32[4], fy, h32[5]))
    return s, nx, ny


def place_on_edge(s, nx, edge, steps=(-2, -1, 0, 1, 2)):
    """FP32 destination coordinates gx for which |fl(gx s - nx)| crosses `edge`: the first gx (going outwards from the
    model's image of the point) that reaches it, and its FP32 neighbours."""
    out = []
    for sign in (1.0, -1.0):
        g = np.float32((sign * float(edge) + float(nx)) / float(s))
        toward = np.float32(np.inf) if (sign > 0) == (float(s) > 0) else np.float32(-np.inf)
        away = np.float32(-toward)
        for _ in range(200):                                   # walk back until below the edge ...
            if abs(fma32(g, s, -nx)) < edge:
                break
            g = np.nextafter(g, away)
        for _ in range(400):                                   # ... then forward to the first value that reaches it
            if abs(fma32(g, s, -nx)) >= edge:
                break
            g = np.nextafter(g, toward)
        assert abs(fma32(g, s, -nx)) >= edge > abs(fma32(np.nextafter(g, away), s, -nx))
        for k in steps:
            v = g
            for _ in range(abs(k)):
                v = np.nextafter(v, toward if k > 0 else away)
            out.append(v)
    return out


def _check(engine, oracle, src, dst, H, thr2):
    engine.set_correspondences(src, dst)
    engine.set_models(H)
    engine.score_stats(reset=True)
    got = engine.score(thr2)
    pairs, in_fp64 = engine.score_stats(reset=True)
    with np.errstate(all="ignore"):
        want = oracle.score(src, dst, H, thr2)
    assert np.array_equal(got, want)
    engine.set_tuning(15, 0)
    try:
        assert np.array_equal(engine.score(thr2), want)
    finally:
        engine.set_tuning(15, 1)
    return pairs, in_fp64


@pytest.mark.parametrize("thr,cmax", [(2.2, 2.0 ** 19), (1.0, 2.0 ** 20 * (1 - 2.0 ** -30)), (2.2, 3000.0)])
def test_cheap_test_edges(engine, oracle, thr, cmax):
    rng = np.random.default_rng(int(thr * 10) + int(np.log2(cmax)))
    thr2 = thr * thr
    X = Y = 1000.0
    k1 = k1_of(thr2, cmax)
    if cmax > 2.0 ** 19.5:
        assert float(k1) > 1.12 * thr * 1.3, "here 25.4 u Cmax is what sets k1"
    # (With Cmax that large 25.2 A >= 5 x 25.4 u Cmax sigma: the per-model constant is the binding edge for every pair, and
    # the k1 sigma edge only binds for moderate Cmax — the third scene.)
    models = []
    for _ in range(6):                                          # horizon far from the image: sigma ~ 1
        models.append(np.array([1, 0, 0, 0, 1, 0, 0, 0, 1.0]) + rng.normal(0, 1, 9) * np.array([.05, .05, 20, .05, .05, 20, 2e-5, 2e-5, .01]))
    for _ in range(6):                                          # horizon through the image
        a = rng.uniform(0, 2 * np.pi)
        c = rng.uniform(300, 700)
        models.append(np.array([1, 0.1, 5, -0.1, 1, -3, np.cos(a) * 1e-3, np.sin(a) * 1e-3, -c * 1e-3 * (np.cos(a) + np.sin(a))]))
    H = np.ascontiguousarray(np.array(models))
    src, dst = [[X, Y], [0.0, 0.0]], [[cmax, cmax], [0.0, -cmax]]          # sentinels: they fix X, Y and Cmax
    edges_hit = {"k1": 0, "a25": 0, "tau": 0}
    for mi, h in enumerate(H):
        h32 = h.astype(np.float32)
        tau, a25 = model_constants(h, X, Y, cmax)
        for _ in range(6):
            if mi < 6:
                x, y = rng.uniform(0, 1000, 2)
            else:                                               # a point next to the horizon: sigma between tau and 25.2 A / k1
                y = rng.uniform(0, 1000)
                lo, hi = float(tau) * 1.0000005, max(float(a25) / float(k1) * 0.5, float(tau) * 4)
                s_t = rng.choice([lo, float(tau), float(tau) * (1 - 1e-6), rng.uniform(lo, hi)])
                x = (s_t - h[7] * y - h[8]) / h[6]
                if not (0 <= x <= 1000):
                    continue
            fx, fy = np.float32(x), np.float32(y)
            s, nx, ny = pass_one(h32, fx, fy)
            sigma = abs(s)
            if sigma == 0:
                continue
            edge = max(rn32(Fraction(float(k1)) * Fraction(float(sigma))), a25)
            which = "a25" if a25 >= rn32(Fraction(float(k1)) * Fraction(float(sigma))) else "k1"
            gy0 = np.float32(float(ny) / float(s))             # the model's own image: Wy^ ~ 0
            for gx in place_on_edge(s, nx, edge):
                if abs(float(gx)) >= cmax or abs(float(gy0)) >= cmax:
                    continue
                src.append([float(fx), float(fy)])
                dst.append([float(gx), float(gy0)])
                edges_hit[which] += 1
                if abs(float(sigma) - float(tau)) <= 4e-6 * float(tau):
                    edges_hit["tau"] += 1
            # the same with the roles of x and y exchanged (W^ is a maximum of two)
            gx0 = np.float32(float(nx) / float(s))
            for gy in place_on_edge(s, ny, edge, steps=(-1, 0, 1)):
                if abs(float(gy)) < cmax and abs(float(gx0)) < cmax:
                    src.append([float(fx), float(fy)])
                    dst.append([float(gx0), float(gy)])
    src, dst = np.array(src), np.array(dst)
    assert edges_hit["k1" if cmax < 1e4 else "a25"] >= 50, edges_hit
    pairs, _ = _check(engine, oracle, src, dst, H, thr2)
    assert pairs == src.shape[0] * H.shape[0], "the FP32 pre-test must have been the path that ran"
    # the same points under thresholds that put the edges elsewhere, and shuffled (a pair's wave neighbours change)
    perm = rng.permutation(src.shape[0])
    for t2 in (thr2 * 1.0000001, thr2 * 0.97, 4.0 * thr2):
        _check(engine, oracle, src[perm], dst[perm], H, t2)


@pytest.mark.parametrize("log2_thr2", [-40, 40, -41, 41])
def test_threshold_on_the_eligibility_limits(engine, oracle, synth, log2_thr2):
    """thr^2 = 2^-40 and 2^40 are the last thresholds the pre-test takes (beyond them the FP64 sweep runs): scenes scaled so
    that the threshold separates inliers from outliers there, coordinates up to just under 2^20."""
    thr2 = 2.0 ** log2_thr2
    sc = synth.make_scene(3000, 3, seed=5, with_neighbours=False)
    scale = np.sqrt(thr2) / 2.2
    lim = 2.0 ** 20 * (1 - 2.0 ** -30)
    src, dst = sc.src * scale, sc.dst * scale
    if np.abs(np.concatenate([src, dst])).max() >= lim:
        f = lim / np.abs(np.concatenate([src, dst])).max()
        src, dst = src * f, dst * f
    src[0], dst[0] = [lim, 0.0], [-lim, lim]
    S = np.diag([scale, scale, 1.0])
    H = np.array([(S @ h.reshape(3, 3) @ np.linalg.inv(S)).reshape(9) for h in sc.H_true])
    H = np.concatenate([H, H * (1 + np.random.default_rng(1).normal(0, 1e-4, H.shape))])
    pairs, _ = _check(engine, oracle, src, dst, H, thr2)
    assert (pairs > 0) == (abs(log2_thr2) <= 40)
