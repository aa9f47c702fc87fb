"""Pin the oracle's derived adjoint (the reference has no valid 3-D backward to compare with):
fp64 torch autograd of an independent forward restatement, plus central finite differences."""
import numpy as np
import pytest
import torch

import torch_ref
from common import rel_l2, scene_and_cameras
from gaussiansplat_amd import synthetic


def _setup(oracle, n, W, H, deg, seed, order, t_min):
    O = oracle
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    if n >= 900:
        sc["scales"] += 1.0            # long lists (> 64 per tile): the batch early-out rule triggers
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=order, t_min=t_min)
    dC = synthetic.make_dC(W, H, seed)
    return sc, cam, T, P, ocam, ref, dC


@pytest.mark.parametrize("n,W,H,deg,seed,order,t_min", [
    (160, 64, 48, 0, 21, 1, 0.0), (140, 48, 40, 1, 22, 0, 0.0), (120, 56, 40, 2, 23, 2, 0.0), (120, 48, 48, 3, 24, 1, 0.0),
    (160, 64, 48, 3, 25, 1, 1e-3), (900, 32, 32, 1, 26, 1, 0.3)])
def test_oracle_forward_and_adjoint_vs_torch_autograd(oracle, n, W, H, deg, seed, order, t_min):
    O = oracle
    sc, cam, T, P, ocam, ref, dC = _setup(O, n, W, H, deg, seed, order, t_min)
    params = [torch.tensor(np.asarray(sc[k], np.float64), requires_grad=True) for k in ("means", "scales", "quats", "opacities", "shs")]
    img, trans = torch_ref.render(params, deg, T, P, float(cam.fx), float(cam.fy), cam.eye, cam.lookAt, float(np.float32(cam.near)),
                                  float(np.float32(cam.far)), W, H, ref["ranges"], ref["ids"], ref["pre"]["bbs"], ref["pre"]["tps"][:, 2],
                                  t_min=t_min)
    # forward: the fp32 oracle against the fp64 restatement
    assert np.abs(img.detach().numpy() - ref["image"]).max() < 2e-5
    assert np.abs(trans.detach().numpy() - ref["trans"]).max() < 2e-5
    (img * torch.tensor(dC, dtype=torch.float64)).sum().backward()
    g = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=t_min)
    for p, name in zip(params, ("means", "scales", "quats", "opacities", "shs")):
        want = p.grad.numpy().reshape(-1)
        got = g[name].reshape(-1)
        assert rel_l2(got, want) < 1e-5, (name, rel_l2(got, want))      # fp32 inputs -> fp64 math on both sides


def test_oracle_adjoint_vs_finite_differences(oracle):
    """Directional derivative of L = sum(image * dC) by central differences of the fp64 restatement
    (lists and boxes frozen) against the oracle gradient."""
    O = oracle
    n, W, H, deg = 60, 48, 32, 2
    sc, cam, T, P, ocam, ref, dC = _setup(O, n, W, H, deg, 31, 1, 0.0)
    g = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC)
    rng = np.random.default_rng(5)
    names = ("means", "scales", "quats", "opacities", "shs")
    base = [np.asarray(sc[k], np.float64) for k in names]
    dirs = [rng.standard_normal(b.shape) for b in base]

    def loss(eps):
        ps = [torch.tensor(b + eps * d) for b, d in zip(base, dirs)]
        img, _ = torch_ref.render(ps, deg, T, P, float(cam.fx), float(cam.fy), cam.eye, cam.lookAt, float(np.float32(cam.near)),
                                  float(np.float32(cam.far)), W, H, ref["ranges"], ref["ids"], ref["pre"]["bbs"], ref["pre"]["tps"][:, 2])
        return float((img * torch.tensor(dC, dtype=torch.float64)).sum())
    h = 1e-6
    fd = (loss(h) - loss(-h)) / (2 * h)
    an = sum(float((g[k].reshape(d.shape) * d).sum()) for k, d in zip(names, dirs))
    assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (fd, an)


def test_gradients_accumulate_and_linear_in_dC(oracle):
    """Contract kept from the reference (splat.jl:137-173): gradients accumulate; adjoint is linear in dC."""
    O = oracle
    sc, cam, T, P, ocam, ref, dC = _setup(O, 100, 48, 48, 1, 41, 1, 0.0)
    args = (sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], 1, ocam, ref["ranges"], ref["ids"])
    g1 = O.backward(*args, dC)
    g2 = O.backward(*args, 2.0 * dC)
    for k in ("means", "scales", "quats", "opacities", "shs"):
        assert np.allclose(g2[k], 2.0 * g1[k], rtol=1e-12, atol=1e-14)
