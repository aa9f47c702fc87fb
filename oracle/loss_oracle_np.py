"""NumPy restatement of the reference loss (src/loss.jl) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: loss.jl is not runnable as shipped (it needs NNlib/Flux `conv`, absent from
the reference's Manifest, and the reference has no tests).  This file follows the written
arithmetic (citations /root/reference/src/loss.jl:<line>); NNlib's grouped `conv`
(DenseConvDims, padding = windowSize/2 zeros, stride 1, flipkernel=false, groups = channels;
not vendored) is restated as a zero-padded per-channel 2-D convolution -- the window is
point-symmetric, so flipping is immaterial.  The gradient is pinned by fp64 torch autograd
(tests/test_loss.py).

Images are [C, H, W] (== the reference's [W, H, C, 1] column-major).
"""
from __future__ import annotations

import numpy as np

C1 = np.float32(0.01) ** 2          # loss.jl:37
C2 = np.float32(0.03) ** 2          # loss.jl:38


def kernel_window(window_size: int = 11, sigma: float = 1.5) -> np.ndarray:
    """loss.jl:5-12: exp(-r)/sqrt(2 sigma^2) with r the EUCLIDEAN distance from (ceil(w/2), ceil(w/2)),
    normalised to sum 1 (so sigma cancels; it is not a Gaussian).  Float64, then Float32 (loss.jl:22)."""
    c = np.ceil(window_size / 2.0)
    idx = np.arange(1, window_size + 1, dtype=np.float64)
    r = np.sqrt((c - idx[:, None]) ** 2 + (c - idx[None, :]) ** 2)
    k = np.exp(-r) / np.sqrt(2.0 * sigma ** 2)
    return (k / k.sum()).astype(np.float32)


def _conv(x: np.ndarray, k: np.ndarray) -> np.ndarray:
    """per-channel zero-padded 'same' convolution (loss.jl:25-33 cdims), float32 accumulation."""
    Cn, H, W = x.shape
    w = k.shape[0]
    p = w // 2
    xp = np.zeros((Cn, H + 2 * p, W + 2 * p), np.float32)
    xp[:, p:p + H, p:p + W] = x
    out = np.zeros_like(x, dtype=np.float32)
    for a in range(w):
        for b in range(w):
            out += k[a, b] * xp[:, a:a + H, b:b + W]
    return out


def ssim_score(x: np.ndarray, y: np.ndarray, k: np.ndarray) -> np.float32:
    """loss.jl:41-58."""
    x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
    mux, muy = _conv(x, k), _conv(y, k)
    mux2, muy2, muxy = mux * mux, muy * muy, mux * muy
    s2x = _conv(x * x, k) - mux2
    s2y = _conv(y * y, k) - muy2
    sxy = _conv(x * y, k) - muxy
    lp = (np.float32(2) * muxy + C1) / (mux2 + muy2 + C1)
    cp = (np.float32(2) * sxy + C2) / (s2x + s2y + C2)
    return np.float32(np.mean(lp * cp, dtype=np.float64))


def loss(img: np.ndarray, gt: np.ndarray, k: np.ndarray | None = None, lam: float = 0.1) -> float:
    """loss.jl:62-72: (1-lam) * sum|img-gt| / (2 length) + lam * (1 - ssim) / 2   (lam is Float64)."""
    k = kernel_window() if k is None else k
    img = np.asarray(img, np.float32); gt = np.asarray(gt, np.float32)
    l1 = float(np.sum(np.abs(img - gt), dtype=np.float64))
    d = 1.0 - float(ssim_score(img, gt, k))
    return (1.0 - lam) * l1 / (2.0 * img.size) + lam * d / 2.0
