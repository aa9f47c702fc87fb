"""fp64 torch restatement of the reference forward, used ONLY to check the adjoint by autograd.

Third, independent statement of the forward math (after oracle/gs_oracle.c and
oracle/gs_oracle_np.py): differentiable in means/scales/quats/opacities/shs; the discrete
decisions (pixel boxes, tile lists, near/far skip) are taken from the fp32 oracle.
Citations: /root/reference/src/projection.jl:39-155, cov2d.jl:30-45, splat.jl:175-269.
"""
import numpy as np
import torch

C0, C1 = 0.28209479177387814, 0.48860251190291990
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_basis(deg, x, y, z):
    b = [torch.full_like(x, C0)]
    if deg >= 1:
        b += [-y * C1, z * C1, -x * C1]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg >= 3:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, 1)          # [n, K]


def per_gaussian(means, scales, quats, opac, shs, deg, T, P, fx, fy, eye, lookAt, W, H):
    dt = torch.float64
    T = torch.as_tensor(np.asarray(T, np.float64).reshape(4, 4, order="F"), dtype=dt)
    P = torch.as_tensor(np.asarray(P, np.float64).reshape(4, 4, order="F"), dtype=dt)
    n = means.shape[0]
    mh = torch.cat([means, torch.ones(n, 1, dtype=dt)], 1)
    t = mh @ T.T
    p = t @ P.T
    mux = (W * p[:, 0] / p[:, 3] + 1) / 2 + W / 2
    muy = (H * p[:, 1] / p[:, 3] + 1) / 2 + H / 2
    tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
    z0 = torch.zeros_like(tx)
    J = torch.stack([torch.stack([fx / tz, z0, -fx * tx / tz ** 2], 1), torch.stack([z0, fy / tz, -fy * ty / tz ** 2], 1)], 1)
    w, x, y, z = quats[:, 0], quats[:, 1], quats[:, 2], quats[:, 3]
    R = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x - z * z), 2 * (y * z - w * x)], 1),      # projection.jl:8 quirk
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    Wm = R * torch.exp(scales)[:, None, :]
    Sg = Wm @ Wm.transpose(1, 2)
    A = J @ R
    cov = A @ Sg @ A.transpose(1, 2) + 0.3
    M = torch.linalg.inv(cov)
    sig = torch.sigmoid(opac.reshape(-1))
    le = torch.as_tensor(np.asarray(lookAt, np.float64) - np.asarray(eye, np.float64))
    v = p[:, :3] - le
    d = v / v.norm(dim=1, keepdim=True)
    B = sh_basis(deg, d[:, 0], d[:, 1], d[:, 2])
    rgb = torch.einsum("nk,nkc->nc", B, shs.reshape(n, -1, 3)) + 0.5
    return mux, muy, M, sig, rgb


def render(params, deg, T, P, fx, fy, eye, lookAt, near, far, W, H, ranges, ids, bbs32, clipz32, t_min=0.0):
    means, scales, quats, opac, shs = params
    mux, muy, M, sig, rgb = per_gaussian(means, scales, quats, opac, shs, deg, T, P, fx, fy, eye, lookAt, W, H)
    return composite(mux, muy, M, sig, rgb, near, far, W, H, ranges, ids, bbs32, clipz32, t_min)


def per_gaussian2d(means, scales, rots, opac, colors, W, H):
    """2-D renderer (cov2d.jl:3-45, splat.jl:337-345): Sigma = R S^2 R' + 0.3 I, mu = (W mx, H my), raw opacity."""
    c, s = torch.cos(rots), torch.sin(rots)
    e = torch.exp(scales)
    Wm = torch.stack([torch.stack([c * e[:, 0], -s * e[:, 1]], 1), torch.stack([s * e[:, 0], c * e[:, 1]], 1)], 1)   # [n, 2, 2]
    cov = Wm @ Wm.transpose(1, 2) + 0.3 * torch.eye(2, dtype=torch.float64)
    M = torch.linalg.inv(cov)
    return W * means[:, 0], H * means[:, 1], M, torch.clamp(opac, 0.0, 0.99999994), colors


def render2d(params, W, H, ranges, ids, bbs32, t_min=0.0):
    means, scales, rots, opac, colors = params
    mux, muy, M, sig, rgb = per_gaussian2d(means, scales, rots, opac, colors, W, H)
    return composite(mux, muy, M, sig, rgb, -1.0, 1.0, W, H, ranges, ids, bbs32, np.zeros(len(bbs32)), t_min)


def composite(mux, muy, M, sig, rgb, near, far, W, H, ranges, ids, bbs32, clipz32, t_min=0.0):
    gx = (W + 15) // 16
    img = torch.zeros(3, H, W, dtype=torch.float64)
    trans = torch.ones(H, W, dtype=torch.float64)
    for t in range(ranges.shape[0]):
        s0, s1 = int(ranges[t, 0]), int(ranges[t, 1])
        bx, by = t % gx, t // gx
        i = torch.arange(bx * 16 + 1, min(bx * 16 + 16, W) + 1, dtype=torch.float64)
        j = torch.arange(by * 16 + 1, min(by * 16 + 16, H) + 1, dtype=torch.float64)
        if len(i) == 0 or len(j) == 0:
            continue
        fi, fj = i[None, :].expand(len(j), len(i)), j[:, None].expand(len(j), len(i))
        C = torch.zeros(3, len(j), len(i), dtype=torch.float64)
        Tr = torch.ones(len(j), len(i), dtype=torch.float64)
        dead = torch.zeros(len(j), len(i), dtype=torch.bool)
        for k in range(s0, s1):
            b = int(ids[k])
            if t_min > 0 and (k - s0) % 64 == 0:
                dead = dead | (Tr.detach() < t_min)
            if clipz32[b] < near or clipz32[b] > far:
                continue
            bb = bbs32[b]
            hit = (fi >= float(bb[0])) & (fi <= float(bb[2])) & (fj >= float(bb[1])) & (fj <= float(bb[3]))
            hit = hit & ~dead
            if not bool(hit.any()):
                continue
            dX, dY = fi - mux[b], fj - muy[b]
            v1 = M[b, 0, 0] * dX + M[b, 0, 1] * dY
            v2 = M[b, 1, 0] * dX + M[b, 1, 1] * dY
            alpha = torch.where(hit, sig[b] * torch.exp(-0.5 * (v1 * dX + v2 * dY)), torch.zeros_like(dX))
            C = C + rgb[b][:, None, None] * (alpha * Tr)[None]
            Tr = Tr * (1 - alpha)
        img[:, int(j[0]) - 1:int(j[-1]), int(i[0]) - 1:int(i[-1])] = C
        trans[int(j[0]) - 1:int(j[-1]), int(i[0]) - 1:int(i[-1])] = Tr
    return img, trans
