"""Shared helpers of the parity tests: scene/camera set-up identical for the oracle and the HIP path."""
import numpy as np

from gaussiansplat_amd import camera as gcam
from gaussiansplat_amd import synthetic


def scene_and_cameras(n, W, H, deg, seed, view=0):
    from oracle import oracle as O
    sc = synthetic.make_scene(n, W, H, deg, seed=seed)
    cam = synthetic.scene_camera(W, view=view)
    T = gcam.compute_transform(cam)
    P = gcam.compute_projection(cam, W, H)
    ocam = O.camera_from_arrays(T, P, np.float32(cam.fx), np.float32(cam.fy), np.float32(cam.near), np.float32(cam.far),
                                cam.eye, cam.lookAt, W, H)
    return sc, cam, T, P, ocam


def hip_context(sc, cam, T, P, W, H, deg, **kw):
    from gaussiansplat_amd import backend as B
    ctx = B.Context(**kw)
    n = sc["means"].shape[0]
    ctx.set_model_host(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"].reshape(n, 3 * (deg + 1) ** 2), deg)
    ctx.set_camera(T, P, float(np.float32(cam.fx)), float(np.float32(cam.fy)), float(np.float32(cam.near)),
                   float(np.float32(cam.far)), cam.eye, cam.lookAt, W, H)
    return ctx


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.linalg.norm(a - b)
    return d / max(np.linalg.norm(b), 1e-30)
