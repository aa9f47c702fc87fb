"""The no-op bound of the composite kernels (csrc/gs_composite.hip stage_record, gs_config.alpha_cull) restated in
NumPy and checked against brute force: for random conics, boxes and tiles the bound is never below the largest
log2(alpha) any pixel of the tile reaches (so no contributing entry can be dropped), and it is tight (equal to the
maximum over the continuous rectangle)."""
import numpy as np

K = np.float32(-0.72134752044448170368)       # -1/2 log2 e


def bound_log2_alpha(mu, inv, box, tile_xy, l2s):
    """Upper bound of log2(alpha) over tile /\\ box; -inf when no pixel of the tile lies in the box.
    mu (2,), inv (4,) column-major conic, box (xmin, ymin, xmax, ymax) 1-based inclusive, tile_xy = first pixel."""
    A, B, C = K * inv[0], K * (inv[1] + inv[2]), K * inv[3]
    rx0, rx1 = max(tile_xy[0], box[0]) - mu[0], min(tile_xy[0] + 15, box[2]) - mu[0]
    ry0, ry1 = max(tile_xy[1], box[1]) - mu[1], min(tile_xy[1] + 15, box[3]) - mu[1]
    if rx0 > rx1 or ry0 > ry1:
        return -np.inf
    if not (A < 0 and C < 0 and 4 * A * C - B * B > 0):
        return np.inf                                                   # not provably concave: keep
    cx, cy = float(np.clip(0.0, rx0, rx1)), float(np.clip(0.0, ry0, ry1))
    dy1 = float(np.clip(-0.5 * B * cx / C, ry0, ry1))
    dx2 = float(np.clip(-0.5 * B * cy / A, rx0, rx1))
    f1 = A * cx * cx + dy1 * (B * cx + C * dy1)
    f2 = C * cy * cy + dx2 * (B * cy + A * dx2)
    fm = max(f1, f2) if (cx != 0 and cy != 0) else (f1 if cx != 0 else f2)
    return fm + l2s


def brute(mu, inv, box, tile_xy, l2s, sub=1):
    best = -np.inf
    xs = np.arange(tile_xy[0], tile_xy[0] + 15 + 1e-9, 1.0 / sub)
    ys = np.arange(tile_xy[1], tile_xy[1] + 15 + 1e-9, 1.0 / sub)
    for x in xs:
        if x < box[0] or x > box[2]:
            continue
        for y in ys:
            if y < box[1] or y > box[3]:
                continue
            dx, dy = x - mu[0], y - mu[1]
            p = K * (inv[0] * dx * dx + (inv[1] + inv[2]) * dx * dy + inv[3] * dy * dy)
            best = max(best, float(p) + l2s)
    return best


def continuous_max(mu, inv, box, tile_xy, l2s):
    from scipy.optimize import minimize
    lo = (max(tile_xy[0], box[0]) - mu[0], max(tile_xy[1], box[1]) - mu[1])
    hi = (min(tile_xy[0] + 15, box[2]) - mu[0], min(tile_xy[1] + 15, box[3]) - mu[1])
    a, b, c = float(K) * inv[0], float(K) * (inv[1] + inv[2]), float(K) * inv[3]
    f = lambda d: -(a * d[0] * d[0] + b * d[0] * d[1] + c * d[1] * d[1])
    g = lambda d: -np.array([2 * a * d[0] + b * d[1], b * d[0] + 2 * c * d[1]])
    best = np.inf
    for x0 in ((lo[0], lo[1]), (hi[0], hi[1]), (0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]))):
        r = minimize(f, np.array(x0), jac=g, bounds=list(zip(lo, hi)), method="L-BFGS-B", options=dict(ftol=1e-15, gtol=1e-12))
        best = min(best, r.fun)
    return -best + l2s


def _random_case(rng):
    s1, s2 = np.exp(rng.uniform(-0.5, 3.5, 2))
    th = rng.uniform(0, np.pi)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    cov = R @ np.diag([s1 * s1, s2 * s2]) @ R.T + 0.3
    ic = np.linalg.inv(cov)
    inv = np.array([ic[0, 0], ic[1, 0], ic[0, 1], ic[1, 1]])
    mu = rng.uniform(-20, 84, 2)
    r = np.ceil(3 * max(s1, s2))
    box = (np.floor(mu[0] - r), np.floor(mu[1] - r), np.ceil(mu[0] + r), np.ceil(mu[1] + r))
    tile = (16 * rng.integers(0, 4) + 1, 16 * rng.integers(0, 4) + 1)
    return mu, inv, box, tile, float(np.log2(rng.uniform(0.05, 1.0)))


def test_bound_is_conservative_and_tight():
    rng = np.random.default_rng(5)
    seen_inside = seen_edge = seen_none = 0
    for _ in range(2500):
        mu, inv, box, tile, l2s = _random_case(rng)
        b = bound_log2_alpha(mu, inv, box, tile, l2s)
        px = brute(mu, inv, box, tile, l2s)
        if px == -np.inf:
            seen_none += b == -np.inf
            continue
        assert b >= px - 1e-9 * max(1.0, abs(px)), (b, px)               # never below what a pixel reaches
        cont = continuous_max(mu, inv, box, tile, l2s)                   # bounded maximisation over the continuous rectangle
        assert abs(b - cont) <= 1e-6 * max(1.0, abs(cont)), (b, cont)    # the bound IS that maximum
        seen_inside += b == l2s
        seen_edge += b < l2s
    assert seen_inside > 30 and seen_edge > 200 and seen_none > 10


def test_non_concave_conic_is_kept():
    assert bound_log2_alpha((8.0, 8.0), np.array([1.0, 2.0, 2.0, 1.0]), (1, 1, 16, 16), (17, 1), 0.0) == -np.inf   # box misses the tile
    assert bound_log2_alpha((8.0, 8.0), np.array([1.0, 2.0, 2.0, 1.0]), (1, 1, 40, 40), (17, 1), 0.0) == np.inf    # indefinite: kept
    assert bound_log2_alpha((8.0, 8.0), np.array([np.nan, 0, 0, 1.0]), (1, 1, 40, 40), (17, 1), 0.0) == np.inf     # NaN: kept
