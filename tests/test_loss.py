"""src/loss.jl restatement (oracle/loss_oracle_np.py): known answers + fp64 torch autograd of the gradient."""
import numpy as np
import torch

from oracle import loss_oracle_np as LO


def torch_loss(img, gt, k, lam=0.1, fft=False):
    """fp64 restatement for autograd.  fft=True: the zero-padded 'same' convolution through rfft2 (the window is point
    symmetric, so convolution == correlation) -- the direct grouped conv2d needs ~50 s per 1920x1080x3 backward on a few cores."""
    k4 = torch.as_tensor(k, dtype=torch.float64)[None, None].repeat(img.shape[0], 1, 1, 1)
    conv = lambda t: torch.nn.functional.conv2d(t[None], k4, padding=k.shape[0] // 2, groups=img.shape[0])[0]
    if fft:
        p = k.shape[0] // 2
        Hh, Ww = img.shape[-2] + 2 * p, img.shape[-1] + 2 * p
        kf = torch.fft.rfft2(torch.as_tensor(k, dtype=torch.float64), s=(Hh, Ww))
        conv = lambda t: torch.fft.irfft2(torch.fft.rfft2(t, s=(Hh, Ww)) * kf, s=(Hh, Ww))[..., p:p + img.shape[-2], p:p + img.shape[-1]]
    mux, muy = conv(img), conv(gt)
    s2x, s2y, sxy = conv(img * img) - mux * mux, conv(gt * gt) - muy * muy, conv(img * gt) - mux * muy
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    ssim = ((2 * mux * muy + c1) / (mux ** 2 + muy ** 2 + c1) * (2 * sxy + c2) / (s2x + s2y + c2)).mean()
    return (1 - lam) * (img - gt).abs().sum() / (2.0 * img.numel()) + lam * (1 - ssim) / 2.0


def test_kernel_window_known_answers():
    k = LO.kernel_window(11, 1.5)
    assert k.shape == (11, 11) and k.dtype == np.float32
    assert abs(float(k.sum()) - 1.0) < 1e-6
    assert np.allclose(k, k.T) and np.allclose(k, k[::-1, ::-1])           # symmetric: flipkernel is immaterial
    assert k[5, 5] == k.max()                                               # centre ceil(11/2) = 6 (1-based)
    assert np.isclose(k[5, 6] / k[5, 5], np.exp(-1.0), rtol=1e-6)           # exp(-r), not exp(-r^2/2s^2)
    assert np.isclose(k[4, 4] / k[5, 5], np.exp(-np.sqrt(2.0)), rtol=1e-6)
    assert np.allclose(LO.kernel_window(11, 3.0), k, rtol=1e-6)             # sigma cancels in the normalisation


def test_loss_identical_images_and_against_torch():
    rng = np.random.default_rng(0)
    img = rng.random((3, 40, 56), dtype=np.float32)
    gt = rng.random((3, 40, 56), dtype=np.float32)
    k = LO.kernel_window()
    assert abs(LO.loss(img, img, k)) < 1e-6                                 # ssim = 1, L1 = 0
    assert abs(float(LO.ssim_score(img, img, k)) - 1.0) < 1e-6
    want = float(torch_loss(torch.tensor(img, dtype=torch.float64), torch.tensor(gt, dtype=torch.float64), k))
    assert abs(LO.loss(img, gt, k) - want) < 2e-6
    # interior pixel of a constant image: mu = value, variance 0 -> ssim map = 1 there
    c = np.full((1, 32, 32), 0.3, np.float32)
    assert abs(float(LO.ssim_score(c, c, k)) - 1.0) < 1e-6


def test_fft_form_of_the_torch_reference_equals_the_direct_one():
    rng = np.random.default_rng(5)
    img = rng.random((3, 37, 50)); gt = rng.random((3, 37, 50))
    k = LO.kernel_window()
    g = []
    for fft in (False, True):
        x = torch.tensor(img, dtype=torch.float64, requires_grad=True)
        l = torch_loss(x, torch.tensor(gt, dtype=torch.float64), k, fft=fft)
        l.backward()
        g.append((float(l), x.grad.numpy().copy()))
    assert abs(g[0][0] - g[1][0]) < 1e-13 and np.abs(g[0][1] - g[1][1]).max() < 1e-13
