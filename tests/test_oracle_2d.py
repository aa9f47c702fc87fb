"""2-D image-fitting renderer (GAUSSIAN_2D): the C oracle against the independent NumPy
restatement (bit-exact), closed-form known answers, and the derived adjoint against fp64 torch autograd and
finite differences.  PARITY UNPINNED by the reference (no tests; its 2-D pieces are inconsistent)."""
import numpy as np
import pytest

from gaussiansplat_amd import synthetic
from oracle import gs_oracle_np as ONP


def test_sincos_spec_c_vs_numpy_and_accuracy(oracle):
    O = oracle
    x = np.concatenate([np.linspace(-8, 8, 4001), np.array([0.0, -0.0, np.pi / 4, -np.pi / 4, 100.0, -1000.5, 1e4])]).astype(np.float32)
    sn, cs = ONP.sincosf_spec(x)
    for v, s, c in zip(x[::37], sn[::37], cs[::37]):
        sc, cc = O.sincosf(float(v))
        assert np.float32(sc).tobytes() == np.float32(s).tobytes() and np.float32(cc).tobytes() == np.float32(c).tobytes()
    small = np.abs(x) <= 8
    assert np.abs(sn[small] - np.sin(x[small].astype(np.float64))).max() < 2.5e-7
    assert np.abs(cs[small] - np.cos(x[small].astype(np.float64))).max() < 2.5e-7
    assert np.isnan(ONP.sincosf_spec(np.array([np.inf], np.float32))[0][0]) and np.isnan(O.sincosf(float("nan"))[1])


@pytest.mark.parametrize("n,W,H,seed", [(600, 96, 64, 3), (301, 70, 50, 4)])
def test_c_vs_numpy_bit_exact_2d(oracle, n, W, H, seed):
    O = oracle
    sc = synthetic.make_scene_2d(n, W, H, seed)
    sc["opacities"][:6] = np.array([-0.3, 0.0, 1.0, 1.7, np.nan, 0.99999994], np.float32)      # the clamp of the spec
    a = O.preprocess2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H)
    b = ONP.preprocess2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H)
    for k in ("mu", "cov2d", "invcov", "bbs", "rgb", "sig"):
        assert a[k].tobytes() == np.ascontiguousarray(b[k], np.float32).tobytes(), k
    r = O.render2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges_np, ids_np, _ = ONP.bin_lists(b["bbs"], np.zeros(n, np.float32), ONP.ORDER_INDEX, 16, gx, gy)
    assert np.array_equal(ranges_np, r["ranges"]) and np.array_equal(ids_np, r["ids"])
    img, tr = ONP.composite_forward(b, ranges_np, ids_np, -1.0, 1.0, W, H, 16, gx, gy)
    assert img.tobytes() == r["image"].tobytes() and tr.tobytes() == r["trans"].tobytes()
    assert float(r["trans"].min()) < 0.9            # the scene actually covers pixels
    assert tuple(a["sig"][:6]) == (0.0, 0.0, np.float32(0.99999994), np.float32(0.99999994), 0.0, np.float32(0.99999994))
    g = O.backward2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H, r["ranges"], r["ids"],
                     synthetic.make_dC(W, H, seed))
    assert np.all(g["opacities"][:5] == 0.0) and np.isfinite(g["opacities"]).all() and np.abs(g["opacities"][6:]).max() > 0


def test_known_answer_axis_aligned(oracle):
    """theta = 0: Sigma = diag(e^2s1 + 0.3, e^2s2 + 0.3); on-centre pixel alpha = opacity; box = ceil(3 sqrt(lmax))."""
    O = oracle
    W, H = 64, 48
    s1, s2 = np.float32(np.log(3.0)), np.float32(np.log(1.5))
    pre = O.preprocess2d(np.array([[0.5, 0.5]], np.float32), np.array([[s1, s2]], np.float32), np.zeros(1, np.float32),
                         np.array([0.8], np.float32), np.array([[0.2, 0.4, 0.6]], np.float32), W, H)
    np.testing.assert_allclose(pre["cov2d"][0], [9.3, 0.0, 0.0, 2.55], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pre["invcov"][0], [1 / 9.3, 0.0, 0.0, 1 / 2.55], rtol=3e-6, atol=1e-7)
    assert tuple(pre["mu"][0]) == (32.0, 24.0)
    r = np.ceil(3 * np.sqrt(9.3))
    assert tuple(pre["bbs"][0]) == (32 - r, 24 - r, 32 + r, 24 + r)
    out = O.render2d(np.array([[0.5, 0.5]], np.float32), np.array([[s1, s2]], np.float32), np.zeros(1, np.float32),
                     np.array([0.8], np.float32), np.array([[0.2, 0.4, 0.6]], np.float32), W, H)
    np.testing.assert_allclose(out["image"][:, 23, 31], np.array([0.2, 0.4, 0.6]) * 0.8, rtol=1e-6)   # pixel (32, 24), 1-based
    np.testing.assert_allclose(out["trans"][23, 31], 0.2, rtol=1e-6)
    a = 0.8 * np.exp(-0.5 * (3 * 3 / 9.3))                                                            # pixel (35, 24)
    np.testing.assert_allclose(out["image"][0, 23, 34], 0.2 * a, rtol=2e-6)
    assert out["image"][0, 23, 31 + int(r) + 1] == 0.0                                                # outside the box


@pytest.mark.parametrize("t_min", [0.0, 1e-3])
def test_adjoint_2d_vs_torch_autograd(oracle, t_min):
    import torch
    import torch_ref as TR
    O = oracle
    n, W, H = 160, 48, 40
    sc = synthetic.make_scene_2d(n, W, H, 11, scale_hi=2.0)
    r = O.render2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H, t_min=t_min)
    dC = synthetic.make_dC(W, H, 5)
    g = O.backward2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H, r["ranges"], r["ids"], dC, t_min=t_min)
    params = [torch.tensor(np.asarray(sc[k], np.float64), requires_grad=True) for k in ("means", "scales", "rots", "opacities", "colors")]
    img, tr = TR.render2d(params, W, H, r["ranges"], r["ids"], r["pre"]["bbs"], t_min=t_min)
    assert np.abs(img.detach().numpy() - r["image"]).max() < 2e-5
    (img * torch.tensor(dC.astype(np.float64))).sum().backward()
    for k, p in zip(("means", "scales", "rots", "opacities", "colors"), params):
        got, want = g[k].reshape(-1), p.grad.numpy().reshape(-1)
        assert np.linalg.norm(got - want) <= 1e-6 * max(np.linalg.norm(want), 1e-30), k


def test_adjoint_2d_finite_differences_and_accumulation(oracle):
    O = oracle
    n, W, H = 40, 32, 32
    sc = synthetic.make_scene_2d(n, W, H, 21, scale_hi=1.8)
    r = O.render2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H)
    dC = synthetic.make_dC(W, H, 6)
    g = O.backward2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H, r["ranges"], r["ids"], dC)

    def loss(name, idx, h):
        p = {k: v.copy() for k, v in sc.items()}
        p[name].reshape(-1)[idx] += np.float32(h)
        # fixed lists/boxes are not available through render2d; perturbations are small enough not to move a box edge
        q = O.render2d(p["means"], p["scales"], p["rots"], p["opacities"], p["colors"], W, H)
        return float((q["image"].astype(np.float64) * dC).sum())

    rng = np.random.default_rng(0)
    for name in ("rots", "opacities", "colors", "scales"):
        flat = g[name].reshape(-1)
        for idx in rng.choice(flat.size, 4, replace=False):
            h = 2e-3
            fd = (loss(name, idx, h) - loss(name, idx, -h)) / (2 * np.float64(np.float32(h)))
            assert abs(fd - flat[idx]) <= 3e-2 * max(abs(flat[idx]), 0.05), (name, idx, fd, flat[idx])
    g2 = O.backward2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H, r["ranges"], r["ids"], 2 * dC)
    for k in ("means", "scales", "rots", "opacities", "colors"):
        np.testing.assert_allclose(g2[k], 2 * g[k], rtol=1e-12, atol=1e-12)
