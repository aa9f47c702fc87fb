"""ctypes front-end of the C oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by gaussiansplat_amd/.  PARITY UNPINNED by the reference (see
gs_oracle.h).  Arrays use the [n, comp] row-major convention (== the reference's
column-major [comp, n]); images are [3, H, W] (== reference cimage[W, H, 3]).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORDER_INDEX, ORDER_DEPTH_DESC, ORDER_DEPTH_ASC = 0, 1, 2


class Camera(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("T", C.c_float * 16), ("P", C.c_float * 16),
                ("fx", C.c_float), ("fy", C.c_float), ("near_", C.c_float), ("far_", C.c_float),
                ("eye", C.c_float * 3), ("lookAt", C.c_float * 3)]


def build(force: bool = False) -> None:
    """Compile the oracle with the committed Makefile (gcc; seconds)."""
    out = os.path.join(_HERE, "_build", "libgs_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("gs_oracle.c", "gs_oracle.h", "Makefile")]
    if force or not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)


_libs: dict = {}


def lib(omp: bool = False):
    key = "omp" if omp else "st"
    if key not in _libs:
        build()
        path = os.path.join(_HERE, "_build", "libgs_oracle_omp.so" if omp else "libgs_oracle.so")
        path = os.environ.get("GS_ORACLE_LIB", path)        # `make -C oracle asan-test`: the ASan/UBSan build of the same source
        L = C.CDLL(path)
        fp, u32p, u64p, dp = (C.POINTER(t) for t in (C.c_float, C.c_uint32, C.c_uint64, C.c_double))
        L.gso_expf.restype = C.c_float; L.gso_expf.argtypes = [C.c_float]
        L.gso_camera_matrices.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(Camera)]
        L.gso_preprocess.argtypes = [C.c_int64, C.c_int] + [fp] * 5 + [C.POINTER(Camera)] + [fp] * 9
        L.gso_depth_order.argtypes = [C.c_int64, fp, C.c_int, u32p]
        L.gso_depth_key.restype = C.c_uint32; L.gso_depth_key.argtypes = [C.c_float, C.c_int]
        L.gso_tile_rect.restype = C.c_int; L.gso_tile_rect.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.gso_bin.restype = C.c_int64
        L.gso_bin.argtypes = [C.c_int64, fp, fp, u32p, C.c_int, C.c_int, C.c_int, C.c_int, u32p, u32p, u64p, C.c_int64]
        L.gso_bin_dense_literal.restype = C.c_int64
        L.gso_bin_dense_literal.argtypes = [C.c_int64, fp, C.c_int, C.c_int, C.c_int, u32p, C.c_int64]
        L.gso_composite_forward.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, u32p, u32p] + [fp] * 6 + [C.c_float, fp, fp]
        L.gso_backward.argtypes = ([C.c_int64, C.c_int] + [fp] * 5 + [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, u32p, u32p, fp,
                                   C.c_float, fp] + [dp] * 6)
        L.gso_sincosf.argtypes = [C.c_float, fp, fp]
        L.gso_preprocess2d.argtypes = [C.c_int64] + [fp] * 5 + [C.c_int, C.c_int] + [fp] * 6
        L.gso_backward2d.argtypes = ([C.c_int64] + [fp] * 5 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u32p, u32p, C.c_float, fp]
                                     + [dp] * 6)
        L.gso_num_threads.restype = C.c_int
        _libs[key] = L
    return _libs[key]


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def expf(x: float) -> float:
    return float(lib().gso_expf(C.c_float(x)))


def make_camera(eye, lookAt, up, fx, fy, near, far, W, H) -> Camera:
    cam = Camera()
    e, l, u = (_c32(v) for v in (eye, lookAt, up))
    lib().gso_camera_matrices(_fp(e), _fp(l), _fp(u), fx, fy, near, far, W, H, C.byref(cam))
    return cam


def camera_from_arrays(T, P, fx, fy, near, far, eye, lookAt, W, H) -> Camera:
    cam = Camera()
    cam.W, cam.H = int(W), int(H)
    cam.T[:] = [float(v) for v in np.asarray(T, np.float32).reshape(-1)]
    cam.P[:] = [float(v) for v in np.asarray(P, np.float32).reshape(-1)]
    cam.fx, cam.fy, cam.near_, cam.far_ = float(fx), float(fy), float(near), float(far)
    cam.eye[:] = [float(v) for v in eye]
    cam.lookAt[:] = [float(v) for v in lookAt]
    return cam


def preprocess(means, scales, quats, opacities, shs, sh_degree, cam: Camera, omp=False) -> dict:
    means, scales, quats, shs = (_c32(a) for a in (means, scales, quats, shs))
    opac = _c32(opacities).reshape(-1)
    n = means.shape[0]
    o = dict(ts=np.empty((n, 4), np.float32), tps=np.empty((n, 4), np.float32), mu=np.empty((n, 2), np.float32),
             cov3d=np.empty((n, 9), np.float32), cov2d=np.empty((n, 4), np.float32), invcov=np.empty((n, 4), np.float32),
             bbs=np.empty((n, 4), np.float32), rgb=np.empty((n, 3), np.float32), sig=np.empty((n,), np.float32))
    lib(omp).gso_preprocess(n, sh_degree, _fp(means), _fp(scales), _fp(quats), _fp(opac), _fp(shs), C.byref(cam),
                            *(_fp(o[k]) for k in ("ts", "tps", "mu", "cov3d", "cov2d", "invcov", "bbs", "rgb", "sig")))
    return o


def depth_order(tps, order) -> np.ndarray:
    tps = _c32(tps)
    perm = np.empty(tps.shape[0], np.uint32)
    lib().gso_depth_order(tps.shape[0], _fp(tps), order, _p(perm, C.c_uint32))
    return perm


def bin_lists(bbs, tps, order, tile, gx, gy):
    """-> (ranges[nt,2] uint32, ids uint32[I], keys uint64[I])."""
    bbs, tps = _c32(bbs), _c32(tps)
    n = bbs.shape[0]
    perm = depth_order(tps, order) if order != ORDER_INDEX else None
    L = lib()
    total = L.gso_bin(n, _fp(bbs), _fp(tps), _p(perm, C.c_uint32), order, tile, gx, gy, None, None, None, 0)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    ids = np.zeros(max(total, 1), np.uint32)
    keys = np.zeros(max(total, 1), np.uint64)
    L.gso_bin(n, _fp(bbs), _fp(tps), _p(perm, C.c_uint32), order, tile, gx, gy,
              _p(ranges, C.c_uint32), _p(ids, C.c_uint32), _p(keys, C.c_uint64), total)
    return ranges, ids[:total], keys[:total]


def bin_dense_literal(bbs, tile, gx, gy):
    """Literal hits/scan/compact path -> hitIdxs[maxBin, gy, gx] (1-based ids, 0 empty), maxHits."""
    bbs = _c32(bbs)
    n = bbs.shape[0]
    L = lib()
    max_hits = L.gso_bin_dense_literal(n, _fp(bbs), tile, gx, gy, None, 0)
    max_bin = min(65535, 1 << max(0, int(max_hits - 1).bit_length())) if max_hits > 0 else 1   # nextpow(2, maxHits)
    hit = np.zeros((max_bin, gy, gx), np.uint32)
    L.gso_bin_dense_literal(n, _fp(bbs), tile, gx, gy, _p(hit, C.c_uint32), max_bin)
    return hit, int(max_hits)


def composite_forward(pre, ranges, ids, cam: Camera, tile, gx, gy, t_min=0.0, omp=False):
    image = np.zeros((3, cam.H, cam.W), np.float32)
    trans = np.ones((cam.H, cam.W), np.float32)
    ranges = np.ascontiguousarray(ranges, np.uint32)
    ids = np.ascontiguousarray(ids, np.uint32) if len(ids) else np.zeros(1, np.uint32)
    lib(omp).gso_composite_forward(C.byref(cam), tile, gx, gy, _p(ranges, C.c_uint32), _p(ids, C.c_uint32),
                                   _fp(pre["mu"]), _fp(pre["invcov"]), _fp(pre["bbs"]), _fp(pre["sig"]), _fp(pre["rgb"]),
                                   _fp(pre["tps"]), t_min, _fp(image), _fp(trans))
    return image, trans


def render(means, scales, quats, opacities, shs, sh_degree, cam: Camera, order=ORDER_DEPTH_DESC, tile=16,
           t_min=0.0, omp=False):
    """preprocess -> compactIdxs -> forward, the call sequence of examples/main.jl:32-34."""
    gx, gy = (cam.W + tile - 1) // tile, (cam.H + tile - 1) // tile
    pre = preprocess(means, scales, quats, opacities, shs, sh_degree, cam, omp=omp)
    ranges, ids, keys = bin_lists(pre["bbs"], pre["tps"], order, tile, gx, gy)
    image, trans = composite_forward(pre, ranges, ids, cam, tile, gx, gy, t_min, omp=omp)
    return dict(pre=pre, ranges=ranges, ids=ids, keys=keys, image=image, trans=trans)


def backward(means, scales, quats, opacities, shs, sh_degree, cam: Camera, ranges, ids, dC, tile=16, t_min=0.0,
             omp=False):
    """fp64 adjoint; returns dict of float64 gradient arrays shaped like the parameters."""
    means, scales, quats, shs = (_c32(a) for a in (means, scales, quats, shs))
    opac = _c32(opacities).reshape(-1)
    n = means.shape[0]
    gx, gy = (cam.W + tile - 1) // tile, (cam.H + tile - 1) // tile
    dC = _c32(dC)
    ranges = np.ascontiguousarray(ranges, np.uint32)
    ids = np.ascontiguousarray(ids, np.uint32) if len(ids) else np.zeros(1, np.uint32)
    g = dict(means=np.zeros((n, 3)), scales=np.zeros((n, 3)), quats=np.zeros((n, 4)), opacities=np.zeros(n),
             shs=np.zeros(shs.shape), g2d=np.zeros((n, 10)))
    dp = C.POINTER(C.c_double)
    lib(omp).gso_backward(n, sh_degree, _fp(means), _fp(scales), _fp(quats), _fp(opac), _fp(shs), C.byref(cam), tile, gx, gy,
                          _p(ranges, C.c_uint32), _p(ids, C.c_uint32), None, t_min, _fp(dC),
                          *(g[k].ctypes.data_as(dp) for k in ("means", "scales", "quats", "opacities", "shs", "g2d")))
    return g


# ---------------------------------------------------------------- 2-D renderer (GAUSSIAN_2D)

def sincosf(x: float):
    sn, cs = C.c_float(), C.c_float()
    lib().gso_sincosf(C.c_float(x), C.cast(C.byref(sn), C.POINTER(C.c_float)), C.cast(C.byref(cs), C.POINTER(C.c_float)))
    return sn.value, cs.value


def image_camera(W: int, H: int) -> Camera:
    """Camera carrying only the image size (the 2-D renderer has no projection)."""
    cam = Camera()
    cam.W, cam.H = int(W), int(H)
    cam.near_, cam.far_ = -1.0, 1.0
    return cam


def preprocess2d(means, scales, rots, opacities, colors, W, H, omp=False) -> dict:
    means, scales, colors = (_c32(a) for a in (means, scales, colors))
    rots = _c32(rots).reshape(-1); opac = _c32(opacities).reshape(-1)
    n = means.shape[0]
    o = dict(mu=np.empty((n, 2), np.float32), cov2d=np.empty((n, 4), np.float32), invcov=np.empty((n, 4), np.float32),
             bbs=np.empty((n, 4), np.float32), rgb=np.empty((n, 3), np.float32), sig=np.empty((n,), np.float32))
    lib(omp).gso_preprocess2d(n, _fp(means), _fp(scales), _fp(rots), _fp(opac), _fp(colors), int(W), int(H),
                              *(_fp(o[k]) for k in ("mu", "cov2d", "invcov", "bbs", "rgb", "sig")))
    o["tps"] = np.zeros((n, 4), np.float32)
    return o


def render2d(means, scales, rots, opacities, colors, W, H, tile=16, t_min=0.0, omp=False):
    """preprocess -> compactIdxs -> forward for the 2-D renderer; lists in gaussian-index order (no depth)."""
    gx, gy = (W + tile - 1) // tile, (H + tile - 1) // tile
    pre = preprocess2d(means, scales, rots, opacities, colors, W, H, omp=omp)
    ranges, ids, keys = bin_lists(pre["bbs"], pre["tps"], ORDER_INDEX, tile, gx, gy)
    cam = image_camera(W, H)
    image = np.zeros((3, H, W), np.float32)
    trans = np.ones((H, W), np.float32)
    idsc = np.ascontiguousarray(ids, np.uint32) if len(ids) else np.zeros(1, np.uint32)
    lib(omp).gso_composite_forward(C.byref(cam), tile, gx, gy, _p(np.ascontiguousarray(ranges, np.uint32), C.c_uint32), _p(idsc, C.c_uint32),
                                   _fp(pre["mu"]), _fp(pre["invcov"]), _fp(pre["bbs"]), _fp(pre["sig"]), _fp(pre["rgb"]),
                                   None, t_min, _fp(image), _fp(trans))
    return dict(pre=pre, ranges=ranges, ids=ids, keys=keys, image=image, trans=trans)


def backward2d(means, scales, rots, opacities, colors, W, H, ranges, ids, dC, tile=16, t_min=0.0, omp=False):
    means, scales, colors = (_c32(a) for a in (means, scales, colors))
    rots = _c32(rots).reshape(-1); opac = _c32(opacities).reshape(-1)
    n = means.shape[0]
    gx, gy = (W + tile - 1) // tile, (H + tile - 1) // tile
    dC = _c32(dC)
    ranges = np.ascontiguousarray(ranges, np.uint32)
    ids = np.ascontiguousarray(ids, np.uint32) if len(ids) else np.zeros(1, np.uint32)
    g = dict(means=np.zeros((n, 2)), scales=np.zeros((n, 2)), rots=np.zeros(n), opacities=np.zeros(n), colors=np.zeros((n, 3)),
             g2d=np.zeros((n, 10)))
    dp = C.POINTER(C.c_double)
    lib(omp).gso_backward2d(n, _fp(means), _fp(scales), _fp(rots), _fp(opac), _fp(colors), int(W), int(H), tile, gx, gy,
                            _p(ranges, C.c_uint32), _p(ids, C.c_uint32), C.c_float(t_min), _fp(dC),
                            *(g[k].ctypes.data_as(dp) for k in ("means", "scales", "rots", "opacities", "colors", "g2d")))
    return g
