"""Independent NumPy-float32 restatement of the arhik/GaussianSplat forward hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product (gaussiansplat_amd/).  PARITY
UNPINNED by the reference (no tests, no golden vectors, not runnable here).  This file
is the *second* restatement: it was written from the Julia sources, not from
gs_oracle.c, and the two must agree bit-for-bit on every fp32 / integer output
(tests/test_oracle_cross.py).  Citations are /root/reference/src/<file>:<line>.

Numeric contract: every fp32 operation individually rounded in the written order
(NumPy float32 ufuncs do exactly that), Float64-literal promotions reproduced, exp =
``expf_spec`` (no fma anywhere), Julia NaN-propagating max/min.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
f64 = np.float64

ORDER_INDEX, ORDER_DEPTH_DESC, ORDER_DEPTH_ASC = 0, 1, 2
EARLY_BATCH = 64


def expf_spec(x):
    """The spec's exp (stands in for CUDA.exp / exp, splat.jl:176, projection.jl:133)."""
    x = np.asarray(x, dtype=f32)
    with np.errstate(all="ignore"):
        xc = np.where(np.isnan(x), f32(0), np.clip(x, f32(-87.33654), f32(88.72283)))
        n = np.rint(xc * f32(1.44269504))
        r = xc - n * f32(0.693359375)
        r = r - n * f32(-2.12194440e-4)
        z = r * r
        y = np.full_like(r, f32(1.9875691500e-4))
        for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
            y = y * r + f32(c)
        y = y * z
        y = y + r
        y = y + f32(1.0)
        ni = n.astype(np.int32)
        n1 = (np.sign(ni) * (np.abs(ni) // 2)).astype(np.int32)  # C truncating division
        n2 = ni - n1
        s1 = ((n1 + 127).astype(np.uint32) << np.uint32(23)).view(f32)
        s2 = ((n2 + 127).astype(np.uint32) << np.uint32(23)).view(f32)
        out = (y * s1) * s2
        out = np.where(x > f32(88.72283), f32(np.inf), out)
        out = np.where(x < f32(-87.33654), f32(0), out)
        out = np.where(np.isnan(x), x, out)
    return out.astype(f32)


def sincosf_spec(x):
    """The spec's sin/cos (stands in for CUDA.sin / CUDA.cos, cov2d.jl:6-8): k = rint(x*2/pi), three-term
    Cody-Waite reduction by pi/2, Cephes polynomials on [-pi/4, pi/4], quadrant fix-up; fp32 mul/add only."""
    x = np.asarray(x, dtype=f32)
    with np.errstate(all="ignore"):
        fin = np.isfinite(x)
        xs = np.where(fin, x, f32(0))
        kf = np.rint(xs * f32(0.636619772))
        r = xs - kf * f32(1.5703125)
        r = r - kf * f32(4.837512969970703125e-4)
        r = r - kf * f32(7.54978995489188216e-8)
        z = r * r
        ps = np.full_like(r, f32(-1.9515295891e-4))
        ps = ps * z + f32(8.3321608736e-3)
        ps = ps * z + f32(-1.6666654611e-1)
        ps = ps * z
        ps = ps * r
        ps = ps + r
        pc = np.full_like(r, f32(2.443315711809948e-5))
        pc = pc * z + f32(-1.388731625493765e-3)
        pc = pc * z + f32(4.166664568298827e-2)
        pc = pc * z
        pc = pc * z
        pc = pc - f32(0.5) * z
        pc = pc + f32(1.0)
        q = kf.astype(np.int64) & 3
        sn = np.select([q == 0, q == 1, q == 2], [ps, pc, -ps], -pc)
        cs = np.select([q == 0, q == 1, q == 2], [pc, -ps, -pc], ps)
        sn = np.where(fin, sn, f32(np.nan)); cs = np.where(fin, cs, f32(np.nan))
    return sn.astype(f32), cs.astype(f32)


def jl_max(a, b):
    """Julia max: NaN if either argument is NaN."""
    with np.errstate(all="ignore"):
        return np.where(np.isnan(a) | np.isnan(b), np.nan, np.maximum(a, b))


def jl_min(a, b):
    with np.errstate(all="ignore"):
        return np.where(np.isnan(a) | np.isnan(b), np.nan, np.minimum(a, b))


# ----------------------------------------------------------------------------- camera

def default_camera_params():
    """camera.jl:24-47."""
    return dict(eye=np.array([1.0, 3.0, 30.0], f32), lookAt=np.zeros(3, f32), up=np.array([0, 1, 0], f32),
                fx=f32(3200.0), fy=f32(3200.0), near=f32(0.1), far=f32(100.0))


def _normalize(a):
    # LinearAlgebra.norm on Vector{Float32}: squares in f32, accumulated in Float64
    s = f64(a[0] * a[0]) + f64(a[1] * a[1]) + f64(a[2] * a[2])
    nrm = f32(np.sqrt(s))
    return a * (f32(1.0) / nrm)


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], f32)


def camera_matrices(eye, lookAt, up, fx, fy, near, far, W, H):
    """computeTransform (camera.jl:88-100) and computeProjection (camera.jl:102-111).
    Returns column-major flattened T, P (as Julia's `.linear |> CuArray` hands them over)."""
    eye, lookAt, up = (np.asarray(v, f32) for v in (eye, lookAt, up))
    w = _normalize(lookAt - eye)
    u = _normalize(_cross(up, w))
    v = _cross(w, u)
    m = np.zeros((4, 4), f32)
    m[0, :3], m[1, :3], m[2, :3] = u, v, w          # m[4,4] = 0 (:96): the whole 4th row is zero
    ti = np.eye(4, dtype=f32)
    ti[:3, 3] = -eye                                 # inv(translate(eye)), camera.jl:65-77
    T = np.zeros((4, 4), f32)
    for i in range(4):
        for j in range(4):
            s = m[i, 0] * ti[0, j]
            for k in range(1, 4):
                s = f32(s + m[i, k] * ti[k, j])
            T[i, j] = s
    P = np.zeros((4, 4), f32)
    fx, fy, near, far = f32(fx), f32(fy), f32(near), f32(far)
    P[0, 0] = f32(2.0) * fx / f32(W)
    P[1, 1] = f32(2.0) * fy / f32(H)
    P[2, 2] = (far + near) / (far - near)
    P[2, 3] = f32(-2.0) * (far * near) / (far - near)
    P[3, 2] = f32(1.0)
    return T.flatten(order="F"), P.flatten(order="F")


# ----------------------------------------------------------------------------- preprocess

_C0 = f32(0.28209479177387814)
_C1 = f32(0.48860251190291990)
_C2 = [f32(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                        -1.0925484305920792, 0.5462742152960396)]
_C3 = [f32(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                        -0.4570457994644658, 1.445305721320277, -0.5900435899266435)]


def sh_basis(deg, x, y, z):
    b = [np.full_like(x, _C0)]
    if deg >= 1:
        b += [-y * _C1, z * _C1, -x * _C1]            # splat.jl:190
    if deg >= 2:                                       # extension: standard 3DGS polynomials
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [_C2[0] * xy, _C2[1] * yz, _C2[2] * ((f32(2) * zz - xx) - yy), _C2[3] * xz, _C2[4] * (xx - yy)]
    if deg >= 3:
        b += [(_C3[0] * y) * (f32(3) * xx - yy), (_C3[1] * xy) * z, (_C3[2] * y) * ((f32(4) * zz - xx) - yy),
              (_C3[3] * z) * ((f32(2) * zz - f32(3) * xx) - f32(3) * yy), (_C3[4] * x) * ((f32(4) * zz - xx) - yy),
              (_C3[5] * z) * (xx - yy), (_C3[6] * x) * (xx - f32(3) * yy)]
    return b


def _matmul_lr(A, B):
    """Row-by-column products summed left to right (StaticArrays unrolled `*`).
    A: dict[(i,k)] -> array, B: dict[(k,j)] -> array; returns dict[(i,j)]."""
    rows = sorted({i for i, _ in A}); inner = sorted({k for _, k in A}); cols = sorted({j for _, j in B})
    out = {}
    for i in rows:
        for j in cols:
            s = A[(i, inner[0])] * B[(inner[0], j)]
            for k in inner[1:]:
                s = s + A[(i, k)] * B[(k, j)]
            out[(i, j)] = s
    return out


def preprocess(means, scales, quats, opacities, shs, sh_degree, T, P, fx, fy, eye, lookAt, W, H):
    """projection.jl:39-155 + cov2d.jl:30-45 + boundingbox.jl:4-36 + splat.jl:175-193.
    Inputs are [n, comp] float32 arrays (row g = the reference's column g)."""
    with np.errstate(all="ignore"):
        means, scales, quats = (np.ascontiguousarray(a, f32) for a in (means, scales, quats))
        opac = np.asarray(opacities, f32).reshape(-1)
        n = means.shape[0]
        K = (sh_degree + 1) ** 2
        shs = np.asarray(shs, f32).reshape(n, K, 3)              # shs[g, k, c] == reference shs[c + 3k, g]
        T = np.asarray(T, f32); P = np.asarray(P, f32)
        fx, fy = f32(fx), f32(fy)
        one = np.ones(n, f32)
        mv = [means[:, 0], means[:, 1], means[:, 2], one]
        ts = []
        for i in range(4):                                        # projection.jl:59
            s = T[i] * mv[0]
            for k in range(1, 4):
                s = s + T[i + 4 * k] * mv[k]
            ts.append(s)
        tps = []
        for i in range(4):                                        # :77
            s = P[i] * ts[0]
            for k in range(1, 4):
                s = s + P[i + 4 * k] * ts[k]
            tps.append(s)
        cx, cy = f64(W) / 2.0, f64(H) / 2.0                       # forward.jl:58-59
        mux = ((((f32(W) * tps[0]) / tps[3] + f32(1)) / f32(2)).astype(f64) + cx).astype(f32)   # :88
        muy = ((((f32(H) * tps[1]) / tps[3] + f32(1)) / f32(2)).astype(f64) + cy).astype(f32)   # :89
        tx, ty, tz = ts[0], ts[1], ts[2]
        zero = np.zeros(n, f32)
        J = {(0, 0): fx / tz, (1, 0): zero, (0, 1): zero, (1, 1): fy / tz,
             (0, 2): (-fx) * tx / (tz * tz), (1, 2): (-fy) * ty / (tz * tz)}                   # :113-118
        w, x, y, z = quats[:, 0], quats[:, 1], quats[:, 2], quats[:, 3]
        two = f32(2)
        R = {(0, 0): f32(1) - two * (y * y + z * z), (1, 0): two * (x * y + w * z), (2, 0): two * (x * z - w * y),
             (0, 1): two * (x * y - w * z), (1, 1): f32(1) - two * (x * x - z * z), (2, 1): two * (y * z + w * x),
             (0, 2): two * (x * z + w * y), (1, 2): two * (y * z - w * x), (2, 2): f32(1) - two * (x * x + y * y)}
        S = {(i, j): zero for i in range(3) for j in range(3)}
        for d in range(3):
            S[(d, d)] = expf_spec(scales[:, d])                   # :133-135
        Wm = _matmul_lr(R, S)                                     # :136
        WmT = {(j, i): v for (i, j), v in Wm.items()}
        C3 = _matmul_lr(Wm, WmT)                                  # :137
        JR = _matmul_lr(J, R)                                     # :144
        JCR = _matmul_lr(JR, C3)                                  # :145
        JRT = {(j, i): v for (i, j), v in JR.items()}
        c2 = _matmul_lr(JCR, JRT)                                 # :146
        c2 = {k: (v.astype(f64) + 0.3).astype(f32) for k, v in c2.items()}   # :148-152
        a0, a1, a2, a3 = c2[(0, 0)], c2[(1, 0)], c2[(0, 1)], c2[(1, 1)]
        det = a0 * a3 - a2 * a1                                   # StaticArrays det 2x2
        idet = f32(1) / det
        inv = np.stack([a3 * idet, -(a1 * idet), -(a2 * idet), a0 * idet], axis=1)   # cov2d.jl:38
        halfad = (a0 + a3) / f32(2)                               # boundingbox.jl:20
        disc = (halfad * halfad - det).astype(f64)
        sq = np.sqrt(jl_max(f64(0.1), disc))
        e1 = halfad.astype(f64) - sq
        e2 = halfad.astype(f64) + sq
        r = np.ceil(3.0 * np.sqrt(jl_max(e1, e2)))                # :23
        bxmin = jl_max(1.0, np.floor(-r + mux.astype(f64))).astype(f32)
        bxmax = jl_min(f64(W), np.ceil(r + mux.astype(f64))).astype(f32)
        bymin = jl_max(1.0, np.floor(-r + muy.astype(f64))).astype(f32)
        bymax = jl_min(f64(H), np.ceil(r + muy.astype(f64))).astype(f32)
        eye = np.asarray(eye, f32); lookAt = np.asarray(lookAt, f32)
        le = lookAt - eye
        d0, d1, d2 = tps[0] - le[0], tps[1] - le[1], tps[2] - le[2]
        nrm = np.sqrt((d0 * d0 + d1 * d1) + d2 * d2)
        ninv = f32(1) / nrm
        basis = sh_basis(sh_degree, ninv * d0, ninv * d1, ninv * d2)
        rgb = np.zeros((n, 3), f32)
        for c in range(3):
            s = shs[:, 0, c] * basis[0]
            for k in range(1, K):
                s = s + shs[:, k, c] * basis[k]
            rgb[:, c] = (s.astype(f64) + 0.5).astype(f32)         # splat.jl:192
        ez = expf_spec(opac)
        sig = ez / (f32(1) + ez)                                  # splat.jl:175-178
        cov3d = np.stack([C3[(i, j)] for j in range(3) for i in range(3)], axis=1)
        return dict(ts=np.stack(ts, 1), tps=np.stack(tps, 1), mu=np.stack([mux, muy], 1), cov3d=cov3d,
                    cov2d=np.stack([a0, a1, a2, a3], 1), invcov=inv.astype(f32),
                    bbs=np.stack([bxmin, bymin, bxmax, bymax], 1), rgb=rgb, sig=sig.astype(f32))


# ----------------------------------------------------------------------------- 2-D renderer

def preprocess2d(means, scales, rots, opacities, colors, W, H):
    """preprocess(::GaussianRenderer2D), forward.jl:9-33: computeCov2d_kernel (cov2d.jl:3-28), computeInvCov2d
    (cov2d.jl:30-45), computeBB (boundingbox.jl:4-36) on the pixel position (w*mx, h*my) of splat.jl:337-339.
    alpha uses the raw opacity (splat.jl:341), the colour is `colors` itself."""
    with np.errstate(all="ignore"):
        means = np.ascontiguousarray(means, f32); scales = np.ascontiguousarray(scales, f32)
        theta = np.asarray(rots, f32).reshape(-1)
        sn, cs = sincosf_spec(theta)                                          # cov2d.jl:6-8
        zero = np.zeros_like(sn)
        R = {(0, 0): cs, (0, 1): -sn, (1, 0): sn, (1, 1): cs}                 # :9-12
        S = {(0, 0): expf_spec(scales[:, 0]), (0, 1): zero, (1, 0): zero, (1, 1): expf_spec(scales[:, 1])}   # :14-17
        Wm = {(i, j): R[(i, 0)] * S[(0, j)] + R[(i, 1)] * S[(1, j)] for i in range(2) for j in range(2)}     # :18
        Jm = {(i, j): Wm[(i, 0)] * Wm[(j, 0)] + Wm[(i, 1)] * Wm[(j, 1)] for i in range(2) for j in range(2)}  # :19
        a0 = (Jm[(0, 0)].astype(f64) + 0.3).astype(f32)                       # :25
        a1, a2 = Jm[(1, 0)], Jm[(0, 1)]
        a3 = (Jm[(1, 1)].astype(f64) + 0.3).astype(f32)                       # :26
        det = a0 * a3 - a2 * a1
        idet = f32(1) / det
        inv = np.stack([a3 * idet, -(a1 * idet), -(a2 * idet), a0 * idet], axis=1)       # cov2d.jl:38
        mux = f32(W) * means[:, 0]; muy = f32(H) * means[:, 1]                # splat.jl:337-339
        halfad = (a0 + a3) / f32(2)                                           # boundingbox.jl:20
        disc = (halfad * halfad - det).astype(f64)
        sq = np.sqrt(jl_max(f64(0.1), disc))
        e1 = halfad.astype(f64) - sq
        e2 = halfad.astype(f64) + sq
        r = np.ceil(3.0 * np.sqrt(jl_max(e1, e2)))
        bxmin = jl_max(1.0, np.floor(-r + mux.astype(f64))).astype(f32)
        bxmax = jl_min(f64(W), np.ceil(r + mux.astype(f64))).astype(f32)
        bymin = jl_max(1.0, np.floor(-r + muy.astype(f64))).astype(f32)
        bymax = jl_min(f64(H), np.ceil(r + muy.astype(f64))).astype(f32)
        n = means.shape[0]
        tps = np.zeros((n, 4), f32)                                           # no clip z: never skipped (near <= 0 <= far)
        return dict(mu=np.stack([mux, muy], 1), cov2d=np.stack([a0, a1, a2, a3], 1), invcov=inv.astype(f32),
                    bbs=np.stack([bxmin, bymin, bxmax, bymax], 1), rgb=np.ascontiguousarray(colors, f32).reshape(n, 3),
                    sig=np.fmin(np.fmax(np.asarray(opacities, f32).reshape(-1), f32(0)), f32(0.99999994)), tps=tps)   # C fmaxf/fminf: NaN -> 0


# ----------------------------------------------------------------------------- order / binning

def depth_keys(clipz, order):
    clipz = np.asarray(clipz, f32)
    if order == ORDER_INDEX:
        return np.zeros(clipz.shape, np.uint32)
    v = -clipz if order == ORDER_DEPTH_DESC else clipz           # forward.jl:103
    u = v.view(np.uint32)
    key = np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000))
    return np.where(np.isnan(v), np.uint32(0xFFFFFFFF), key).astype(np.uint32)


def depth_order(clipz, order):
    n = len(clipz)
    if order == ORDER_INDEX:
        return np.arange(n, dtype=np.uint32)
    return np.argsort(depth_keys(clipz, order), kind="stable").astype(np.uint32)


def tile_rects(bbs, tile, gx, gy):
    """binning.jl:3-35 -> (valid, x0, x1, y0, y1), 1-based inclusive, clipped to the grid."""
    bbs = np.asarray(bbs, f32)
    with np.errstate(all="ignore"):
        fin = np.isfinite(bbs).all(axis=1)
        bs = f32(tile)

        def jdiv(v):
            vv = np.where(fin, v, f32(0))
            q = np.rint((vv - np.fmod(vv, bs)) / bs)              # Julia div(::Float32, ::Float32)
            return np.clip(q, -1e9, 1e9).astype(np.int64)
        x0 = jdiv(np.floor(bbs[:, 0])) + 1
        y0 = jdiv(np.floor(bbs[:, 1])) + 1
        x1 = jdiv(np.ceil(bbs[:, 2])) + 1
        y1 = jdiv(np.ceil(bbs[:, 3])) + 1
        ok = fin & (x0 <= x1) & (y0 <= y1)
        x0c, y0c = np.maximum(x0, 1), np.maximum(y0, 1)
        x1c, y1c = np.minimum(x1, gx), np.minimum(y1, gy)
        ok &= (x0c <= x1c) & (y0c <= y1c)
    return ok, x0c, x1c, y0c, y1c


def bin_lists(bbs, clipz, order, tile, gx, gy):
    """Per-tile lists; returns (ranges[nt,2] uint32, ids uint32, keys uint64)."""
    ok, x0, x1, y0, y1 = tile_rects(bbs, tile, gx, gy)
    perm = depth_order(clipz, order)
    dk = depth_keys(clipz, order)
    per_tile = [[] for _ in range(gx * gy)]
    for g in perm:
        g = int(g)
        if not ok[g]:
            continue
        for ty in range(int(y0[g]), int(y1[g]) + 1):
            for tx in range(int(x0[g]), int(x1[g]) + 1):
                per_tile[(ty - 1) * gx + (tx - 1)].append(g)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    ids, keys, pos = [], [], 0
    for t, lst in enumerate(per_tile):
        ranges[t] = (pos, pos + len(lst))
        pos += len(lst)
        ids += lst
        keys += [(t << 32) | (g if order == ORDER_INDEX else int(dk[g])) for g in lst]
    return ranges, np.asarray(ids, np.uint32), np.asarray(keys, np.uint64)


# ----------------------------------------------------------------------------- composite

def composite_forward(pre, ranges, ids, near, far, W, H, tile, gx, gy, t_min=0.0):
    """splatDraw, splat.jl:195-269, vectorised over the 16x16 pixels of a tile."""
    image = np.zeros((3, H, W), f32)          # image[c, j-1, i-1]  == reference cimage[i, j, c]
    trans = np.ones((H, W), f32)
    mu, inv, bbs, sig, rgb, tps = (pre[k] for k in ("mu", "invcov", "bbs", "sig", "rgb", "tps"))
    near, far, t_min = f32(near), f32(far), f32(t_min)
    with np.errstate(all="ignore"):
        for t in range(gx * gy):
            bx, by = t % gx + 1, t // gx + 1
            i = ((bx - 1) * tile + np.arange(1, tile + 1)).astype(np.int64)
            j = ((by - 1) * tile + np.arange(1, tile + 1)).astype(np.int64)
            i = i[i <= W]; j = j[j <= H]
            if len(i) == 0 or len(j) == 0:
                continue
            fi = np.broadcast_to(i.astype(f32)[None, :], (len(j), len(i)))
            fj = np.broadcast_to(j.astype(f32)[:, None], (len(j), len(i)))
            C = np.zeros((3, len(j), len(i)), f32)
            Tr = np.ones((len(j), len(i)), f32)
            dead = np.zeros_like(Tr, bool)
            for k in range(int(ranges[t, 0]), int(ranges[t, 1])):
                b = int(ids[k])
                if t_min > 0 and (k - int(ranges[t, 0])) % EARLY_BATCH == 0:     # early-out extension: checked per batch
                    dead |= (Tr < t_min)
                live = ~dead
                cz = tps[b, 2]
                if (cz < near) or (cz > far):                                    # splat.jl:227
                    continue
                hit = (bbs[b, 0] <= fi) & (fi <= bbs[b, 2]) & (bbs[b, 1] <= fj) & (fj <= bbs[b, 3])  # :240
                m = hit & live
                if not m.any():
                    continue
                dX = fi - mu[b, 0]
                dY = fj - mu[b, 1]
                v1 = inv[b, 0] * dX + inv[b, 2] * dY
                v2 = inv[b, 1] * dX + inv[b, 3] * dY
                dist = f32(0.5) * (v1 * dX + v2 * dY)                            # :246
                alpha = sig[b] * expf_spec(-dist)                                # :247
                for c in range(3):
                    C[c] = np.where(m, C[c] + (rgb[b, c] * alpha) * Tr, C[c])    # :255-257
                Tr = np.where(m, Tr * (f32(1) - alpha), Tr)                      # :259
            image[:, j[0] - 1:j[-1], i[0] - 1:i[-1]] = C
            trans[j[0] - 1:j[-1], i[0] - 1:i[-1]] = Tr
    return image, trans
