"""Closed-form known-answer tests that pin the oracle without the (unrunnable) reference."""
import ctypes as C
import math

import numpy as np

from gaussiansplat_amd import camera as gcam


def _rect(O, bb, gx=8, gy=8):
    rc = (C.c_int32 * 4)()
    a = np.asarray(bb, np.float32)
    ok = O.lib().gso_tile_rect(a.ctypes.data_as(C.POINTER(C.c_float)), 16, gx, gy, rc)
    return tuple(rc) if ok else None


def test_tile_rect_formula(oracle):
    """binning.jl:14-17: tile = div(v, 16) + 1 on 1-based pixel coordinates, both ends inclusive."""
    O = oracle
    assert _rect(O, [1, 1, 15, 15]) == (1, 1, 1, 1)
    assert _rect(O, [1, 1, 16, 16]) == (1, 2, 1, 2)              # pixel 16 -> tile 2 (reference off-by-one)
    assert _rect(O, [16, 17, 20, 31]) == (2, 2, 2, 2)            # xmin 16 -> starts at tile 2
    assert _rect(O, [1, 1, 128, 128]) == (1, 8, 1, 8)            # 128/16+1 = 9 clipped to the grid
    assert _rect(O, [1, 1, -5, 40]) == (1, 1, 1, 3)              # div truncates toward zero: -5 -> tile 1 (literal)
    assert _rect(O, [1, 1, -16, 40]) is None                     # off-screen left: max tile 0 < min tile 1
    assert _rect(O, [100, 1, 40, 40]) is None                    # min > max
    assert _rect(O, [1, 1, float("nan"), 4]) is None             # spec: non-finite boxes are dropped
    assert _rect(O, [1, 1, float("inf"), 4]) is None
    assert _rect(O, [1, 1, 3.0e9, 3.0e9]) == (1, 8, 1, 8)        # saturating Int32 conversion


def test_depth_key_is_isless_order(oracle):
    O = oracle
    zs = np.array([-np.inf, -3.5, -0.0, 0.0, 1e-30, 2.0, np.inf, np.nan], np.float32)
    asc = [O.lib().gso_depth_key(C.c_float(z), O.ORDER_DEPTH_ASC) for z in zs]
    assert asc == sorted(asc) and len(set(asc)) == len(asc)       # strictly increasing, NaN last, -0 < +0
    desc = [O.lib().gso_depth_key(C.c_float(z), O.ORDER_DEPTH_DESC) for z in zs[:-1]]
    assert desc == sorted(desc, reverse=True)
    assert O.lib().gso_depth_key(C.c_float(float("nan")), O.ORDER_DEPTH_DESC) == 0xFFFFFFFF


def test_camera_matrices_default_camera(oracle):
    """camera.jl:88-111 on defaultCamera: 4th row of T is zero (m[4,4]=0), P has p43 = 1."""
    cam = gcam.default_camera()
    T = gcam.compute_transform(cam).reshape(4, 4, order="F").astype(np.float64)
    P = gcam.compute_projection(cam, 512, 512).reshape(4, 4, order="F").astype(np.float64)
    assert np.all(T[3] == 0)
    w = -cam.eye.astype(np.float64); w /= np.linalg.norm(w)
    u = np.cross([0, 1, 0], w); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    assert np.allclose(T[:3, :3], np.stack([u, v, w]), atol=1e-6)
    assert np.allclose(T[:3, 3], -np.stack([u, v, w]) @ cam.eye.astype(np.float64), atol=1e-5)
    assert np.isclose(P[0, 0], 2 * 3200 / 512) and np.isclose(P[1, 1], 2 * 3200 / 512)
    assert np.isclose(P[2, 2], 100.1 / 99.9, rtol=1e-6) and np.isclose(P[2, 3], -2 * 10 / 99.9, rtol=1e-6) and P[3, 2] == 1
    assert np.count_nonzero(P) == 5


def test_single_isotropic_gaussian_on_axis(oracle):
    """One gaussian at lookAt, identity rotation, isotropic scale: everything is closed form."""
    O = oracle
    W = H = 64
    cam = gcam.default_camera()
    ocam = O.make_camera(cam.eye, cam.lookAt, cam.up, cam.fx, cam.fy, cam.near, cam.far, W, H)
    s, o = math.log(0.05), 0.3
    means = np.zeros((1, 3), np.float32); scales = np.full((1, 3), s, np.float32)
    quats = np.array([[1, 0, 0, 0]], np.float32); opac = np.array([o], np.float32)
    shs = np.array([[[0.4, -0.2, 0.1]]], np.float32)           # degree 0
    r = O.render(means, scales, quats, opac, shs, 0, ocam, order=O.ORDER_INDEX)
    pre = r["pre"]
    tz = float(np.linalg.norm(cam.eye.astype(np.float64)))
    assert np.allclose(pre["ts"][0], [0, 0, tz, 0], atol=2e-5)             # ts[4] == 0: T's 4th row is zero
    assert np.allclose(pre["mu"][0], [W / 2 + 0.5, H / 2 + 0.5], atol=1e-4)
    a = (3200.0 * math.exp(s) / tz) ** 2
    assert np.allclose(pre["cov2d"][0], [a + 0.3, 0.3, 0.3, a + 0.3], rtol=1e-5)   # +0.3 on ALL FOUR entries
    det = (a + 0.3) ** 2 - 0.09
    assert np.allclose(pre["invcov"][0], np.array([a + 0.3, -0.3, -0.3, a + 0.3]) / det, rtol=1e-5)
    lam = (a + 0.3) + math.sqrt(max(0.1, 0.09))                            # boundingbox.jl:21-22 (0.1 floor)
    rad = math.ceil(3.0 * math.sqrt(lam))
    mu = W / 2 + 0.5
    assert list(pre["bbs"][0]) == [max(1, math.floor(mu - rad)), max(1, math.floor(mu - rad)), min(W, math.ceil(mu + rad)), min(H, math.ceil(mu + rad))]
    assert np.isclose(pre["sig"][0], 1 / (1 + math.exp(-o)), rtol=1e-6)
    assert np.allclose(pre["rgb"][0], 0.28209479177387814 * shs[0, 0] + 0.5, rtol=1e-6)
    # centre pixel (i, j) = (W/2, H/2): delta = (-0.5, -0.5)
    M = np.array([[a + 0.3, -0.3], [-0.3, a + 0.3]]) / det
    d = np.array([-0.5, -0.5])
    alpha = pre["sig"][0] * math.exp(-0.5 * d @ M @ d)
    i = j = W // 2
    assert np.allclose(r["image"][:, j - 1, i - 1], pre["rgb"][0] * alpha, rtol=1e-5)
    assert np.isclose(r["trans"][j - 1, i - 1], 1 - alpha, rtol=1e-5)
    # outside the pixel box nothing is drawn (splat.jl:240)
    xmin, ymin, xmax, ymax = (int(v) for v in pre["bbs"][0])
    mask = np.ones((H, W), bool); mask[ymin - 1:ymax, xmin - 1:xmax] = False
    assert np.all(r["image"][:, mask] == 0) and np.all(r["trans"][mask] == 1)


def test_lattice_scene_one_gaussian_per_tile(oracle):
    """src/test.jl:1-11 idea: gaussians on a tile lattice; every tile list must contain its own gaussian."""
    O = oracle
    W = H = 128
    cam = gcam.default_camera()
    cam.eye = np.array([0, 0, 30], np.float32)                  # look straight down -z so the lattice is axis aligned
    ocam = O.make_camera(cam.eye, cam.lookAt, cam.up, cam.fx, cam.fy, cam.near, cam.far, W, H)
    g = 8
    # pixel centre of tile (tx,ty) = 16*t + 8.5 ; mu = fx * x / 30 * (sign from u axis) + 0.5 + W/2
    T = np.array(ocam.T[:], np.float64).reshape(4, 4, order="F")
    px = (np.arange(g) * 16 + 8.5)
    xs = (px - 0.5 - W / 2) * 30 / 3200
    means = np.array([[T[0, 0] * x, T[1, 1] * y, 0.0] for y in xs for x in xs], np.float32)   # undo axis signs
    n = g * g
    r = O.render(means, np.full((n, 3), math.log(0.005), np.float32), np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1)),
                 np.zeros(n, np.float32), np.zeros((n, 1, 3), np.float32), 0, ocam, order=O.ORDER_INDEX)
    assert np.allclose(r["pre"]["mu"].reshape(g, g, 2)[..., 0], px[None, :], atol=1e-3)
    assert np.allclose(r["pre"]["mu"].reshape(g, g, 2)[..., 1], px[:, None], atol=1e-3)
    for t in range(n):
        lst = r["ids"][r["ranges"][t, 0]:r["ranges"][t, 1]]
        assert t in lst
