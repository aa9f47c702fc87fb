"""Deterministic synthetic scenes for parity tests and the benchmark.

Not part of the reference (it only loads trained PLY files, splat.jl:106-119); this is the
workload generator BASELINE.json's configs are stated on.  PCG64 via numpy default_rng, seed =
1234 + config index.
"""
from __future__ import annotations

import numpy as np

from .camera import Camera, default_camera

# BASELINE.json configs: (n_gaussians, W, H, sh_degree)
CONFIGS = {
    "C1": (10_000, 256, 256, 0),
    "C2": (100_000, 800, 800, 3),
    "C3": (1_000_000, 1920, 1080, 3),
    "C4": (1_000_000, 1920, 1080, 3),     # x 8 views, one per GPU
    "C5": (5_000_000, 3840, 2160, 3),
}


def scene_camera(W: int, view: int = 0) -> Camera:
    """Reference defaultCamera with fx=fy scaled by W/1920 (footprints resolution independent);
    view k rotates the eye about +y by k*45 degrees (8-view batches)."""
    cam = default_camera(id=view)
    cam.fx = cam.fy = float(np.float32(3200.0 * W / 1920.0))
    if view:
        a = np.deg2rad(45.0 * view)
        e = cam.eye.astype(np.float64)
        cam.eye = np.array([np.cos(a) * e[0] + np.sin(a) * e[2], e[1], -np.sin(a) * e[0] + np.cos(a) * e[2]], np.float32)
    return cam


def make_scene(n: int, W: int, H: int, sh_degree: int, seed: int = 1234, clustered: bool = False, raw_quaternions: bool = False):
    """Returns dict(means[n,3], scales[n,3], quats[n,4], opacities[n], shs[n,K,3]) float32.

    Quaternions are drawn N(0,1)^4 and normalised HERE: the reference kernel does not
    normalise (projection.jl:126) and applies R four times (J*R*Sigma*(J*R)'), so raw N(0,1)
    draws would inflate every footprint by |q|^4; trained scenes carry near-unit quaternions.
    raw_quaternions = True leaves them as drawn, exactly as SURVEY 8d words the scene.

    clustered = True: a HEAVY-TAILED scene in the shape of a trained one (what splat.jl:106-119 loads) instead of the spatially
    uniform BASELINE scene: 60 % of the gaussians sit in three blobs that cover 5 % of the frame and are faint (opacity logits
    U(-5, -2): their tiles walk thousands of entries before they saturate), and 0.1 % are huge (scale logits U(-0.6, 0.2): footprints
    of hundreds of pixels, in the list of hundreds of tiles each).  The rest is the uniform scene.  Tile lists beyond the reference's
    UInt16 limit of 65 535 entries (forward.jl:137,141) at 1080p-class sizes.
    """
    rng = np.random.default_rng(seed)
    fx = 3200.0 * W / 1920.0
    Wv, Hv = W * 30.0 / fx, H * 30.0 / fx
    means = np.empty((n, 3), np.float32)
    means[:, 0] = rng.uniform(-0.5 * Wv, 0.5 * Wv, n)
    means[:, 1] = rng.uniform(-0.5 * Hv, 0.5 * Hv, n)
    means[:, 2] = rng.uniform(-4.0, 4.0, n)
    scales = rng.uniform(-4.5, -2.5, (n, 3)).astype(np.float32)
    q = rng.standard_normal((n, 4))
    quats = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opacities = rng.uniform(-2.0, 4.0, n).astype(np.float32)
    K = (sh_degree + 1) ** 2
    shs = (rng.standard_normal((n, K, 3)) * 0.1).astype(np.float32)
    shs[:, 0, :] = (rng.standard_normal((n, 3)) * 0.3).astype(np.float32)
    if raw_quaternions:
        quats = q.astype(np.float32)
    if clustered:
        r2 = np.random.default_rng(seed + 77)                                   # (a stream of its own: the uniform scene above is untouched)
        kind = r2.random(n)
        blob = kind < 0.6
        huge = kind > 0.999
        centres = np.array([[-0.25, 0.10], [0.20, -0.22], [0.30, 0.28]]) * np.array([Wv, Hv])
        rb = np.sqrt(0.05 * Wv * Hv / (3.0 * np.pi))                            # three discs of this radius cover 5 % of the frame
        which = r2.integers(0, 3, n)
        ang, rad = r2.uniform(0.0, 2.0 * np.pi, n), rb * np.sqrt(r2.random(n))
        means[blob, 0] = (centres[which, 0] + rad * np.cos(ang))[blob]
        means[blob, 1] = (centres[which, 1] + rad * np.sin(ang))[blob]
        opacities[blob] = r2.uniform(-5.0, -2.0, n).astype(np.float32)[blob]
        scales[huge] = r2.uniform(-0.6, 0.2, (n, 3)).astype(np.float32)[huge]
    return dict(means=means, scales=scales, quats=quats, opacities=opacities, shs=shs)


def make_scene_2d(n: int, W: int, H: int, seed: int = 1234, scale_lo: float = 0.0, scale_hi: float = 2.5):
    """2-D image-fitting model in the shape of initData(Val(SPLAT2D)) (splat.jl:74-87): means U[0,1)^2 (fractions of
    the image), rotations pi/2*(U-0.5), opacities and colors U[0,1); the reference draws the log-scales U[0,1) too
    (1-2.7 px standard deviations) -- a wider range is used by default so footprints span several tiles.
    Returns dict(means[n,2], scales[n,2], rots[n], opacities[n], colors[n,3]) float32."""
    rng = np.random.default_rng(seed)
    return dict(means=rng.random((n, 2), dtype=np.float32),
                scales=rng.uniform(scale_lo, scale_hi, (n, 2)).astype(np.float32),
                rots=(np.float32(np.pi / 2) * (rng.random(n, dtype=np.float32) - np.float32(0.5))).astype(np.float32),
                opacities=rng.random(n, dtype=np.float32), colors=rng.random((n, 3), dtype=np.float32))


def make_dC(W: int, H: int, seed: int = 1234) -> np.ndarray:
    """Upstream image gradient, [3, H, W] float32 ~ N(0,1)."""
    return np.random.default_rng(seed + 7919).standard_normal((3, H, W)).astype(np.float32)
