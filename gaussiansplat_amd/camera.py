"""Host-side mirror of the reference camera (src/camera.jl).

`Camera`, `default_camera`, `compute_transform`, `compute_projection` and `get_camera`
keep the reference's names, argument meaning and quirks:

* camera.jl:24-47   defaultCamera: eye (1,3,30), lookAt 0, up +y, fx=fy=3200, near .1, far 100
* camera.jl:88-100  computeTransform: rows u,v,w with the whole 4th row ZERO (m[4,4]=0), composed
                    with the inverse eye translation -> ts[4] == 0 for every gaussian
* camera.jl:102-111 computeProjection: p11=2fx/w, p22=2fy/h, p33=(f+n)/(f-n), p34=-2fn/(f-n), p43=1
* camera.jl:119-151 getCamera: cameras.json entry -> eye/lookAt from position + rotation

Matrices are returned flattened column-major (what Julia's `.linear |> CuArray` hands to
the kernels and what the C ABI `gs_set_camera` takes).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

_f32 = np.float32


@dataclass
class Camera:
    fx: float
    fy: float
    far: float
    near: float
    eye: np.ndarray
    lookAt: np.ndarray
    up: np.ndarray
    scale: np.ndarray = field(default_factory=lambda: np.ones(3, _f32))
    aspectRatio: float = 1.0
    id: int = 0
    data: object = None


def default_camera(id: int = 0) -> Camera:
    return Camera(fx=3200.0, fy=3200.0, far=100.0, near=0.1, eye=np.array([1.0, 3.0, 30.0], _f32),
                  lookAt=np.zeros(3, _f32), up=np.array([0.0, 1.0, 0.0], _f32), id=id)


def _unit(v: np.ndarray) -> np.ndarray:
    # LinearAlgebra.normalize on Vector{Float32}: f32 squares summed in Float64, times inv(norm)
    nrm = _f32(np.sqrt(np.sum((v * v).astype(np.float64))))
    return (v * (_f32(1.0) / nrm)).astype(_f32)


def compute_transform(camera: Camera) -> np.ndarray:
    eye = np.asarray(camera.eye, _f32)
    w = _unit(np.asarray(camera.lookAt, _f32) - eye)
    u = _unit(np.cross(np.asarray(camera.up, _f32), w).astype(_f32))
    v = np.cross(w, u).astype(_f32)
    T = np.zeros((4, 4), _f32)
    for r, axis in enumerate((u, v, w)):
        T[r, :3] = axis
        # m * inv(translate(eye)): ((a1*(-e1) + a2*(-e2)) + a3*(-e3)) + 0*1, fp32 left to right
        acc = axis[0] * -eye[0]
        acc = _f32(acc + axis[1] * -eye[1])
        acc = _f32(acc + axis[2] * -eye[2])
        T[r, 3] = acc
    return T.flatten(order="F")


def compute_projection(camera: Camera, w: int, h: int) -> np.ndarray:
    fx, fy, far, near = (_f32(x) for x in (camera.fx, camera.fy, camera.far, camera.near))
    P = np.zeros((4, 4), _f32)
    P[0, 0] = _f32(2.0) * fx / _f32(w)
    P[1, 1] = _f32(2.0) * fy / _f32(h)
    P[2, 2] = (far + near) / (far - near)
    P[2, 3] = _f32(-2.0) * (far * near) / (far - near)
    P[3, 2] = 1.0
    return P.flatten(order="F")


def get_camera(path: str, idx: int) -> Camera:
    """cameras.json loader (camera.jl:113-151); `idx` is 1-based like the reference."""
    with open(path) as fh:
        cams = json.load(fh)
    c = cams[idx - 1]
    position = np.asarray(c["position"], _f32)
    rotation = np.asarray(c["rotation"], _f32).T        # cat(rows..., dims=2): the json rows become columns
    eye = (-(rotation.T) @ position).astype(_f32)
    lookAt = (-(rotation.T) @ np.array([0.0, 0.0, 1.0], _f32)).astype(_f32)
    return Camera(fx=float(c["fx"]), fy=float(c["fy"]), far=100.0, near=0.010, eye=eye, lookAt=lookAt,
                  up=np.array([0.0, 1.0, 0.0], _f32), id=c["id"], data=c["img_name"])
