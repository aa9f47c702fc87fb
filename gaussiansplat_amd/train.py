"""Host-side mirror of the reference's loss and training step (src/loss.jl, src/train.jl).

`kernelWindow`, `getLossFunction` and `train` keep the reference's names and argument meaning.
The reference's loop is stale (it needs an AD package that is not in its Manifest and its
backward/SGD lines are commented out, train.jl:39-46); what it intends is implemented:

    preprocess -> compactIdxs -> forward -> loss + dL/dimage -> backward -> param .-= lr*grad -> resetGrads

with the loss, its image gradient and the SGD update running on the GPU (csrc/gs_loss.hip) so the
step has no host round trip.
"""
from __future__ import annotations

import numpy as np

from . import renderer as R


def kernelWindow(windowSize: int = 11, σ: float = 1.5) -> np.ndarray:
    """loss.jl:5-12 (host copy for inspection; the kernels rebuild the same window)."""
    c = np.ceil(windowSize / 2.0)
    idx = np.arange(1, windowSize + 1, dtype=np.float64)
    k = np.exp(-np.sqrt((c - idx[:, None]) ** 2 + (c - idx[None, :]) ** 2)) / np.sqrt(2.0 * σ ** 2)
    return (k / k.sum()).astype(np.float32)


class LossFunction:
    """What getLossFunction returns: callable (img, gt) -> loss like the reference's closure
    (loss.jl:60-72), plus .value_and_grad for the training step."""

    def __init__(self, renderer, imSize, windowSize: int, nChannels: int, λ: float = 0.1):
        if windowSize != 11:
            raise NotImplementedError("only the reference's windowSize = 11 (loss.jl:14) is built")
        self.r, self.λ = renderer, float(λ)
        self.W, self.H, self.C = int(imSize[0]), int(imSize[1]), int(nChannels)
        self._dC = None

    def _ptrs(self, img, gt):
        import torch
        dev = self.r.imageData.device
        img = img if isinstance(img, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(img, np.float32))
        gt = gt if isinstance(gt, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(gt, np.float32))
        img = img.to(dev, torch.float32).contiguous(); gt = gt.to(dev, torch.float32).contiguous()
        assert tuple(img.shape) == (self.C, self.H, self.W) == tuple(gt.shape)
        return img, gt

    def value_and_grad(self, img, gt, want_loss: bool = True):
        import torch
        img, gt = self._ptrs(img, gt)
        if self._dC is None:
            self._dC = torch.empty_like(img)
        self.r._begin()
        val = self.r.ctx.loss_device(img.data_ptr(), gt.data_ptr(), self._dC.data_ptr(), self.W, self.H, self.C, self.λ, want_loss)
        self.r._end()
        self._keep = (img, gt)
        return val, self._dC

    def __call__(self, img, gt) -> float:
        return self.value_and_grad(img, gt)[0]


def getLossFunction(imSize, windowSize: int, nChannels: int, renderer=None, λ: float = 0.1) -> LossFunction:
    """loss.jl:60-72.  `renderer` supplies the GPU context (the reference's closure captures a CuArray kernel)."""
    if renderer is None:
        raise ValueError("getLossFunction needs the renderer whose GPU context runs the loss kernels")
    return LossFunction(renderer, imSize, windowSize, nChannels, λ)


def trainStep(renderer, gtimg, lr: float, lossFunc: LossFunction, camera=None, want_loss: bool = True, fused_sgd: bool = False):
    """One iteration of train.jl:33-56 as intended (see module docstring).
    fused_sgd (3-D renderer): backward and the parameter update in one pass (gs_backward_sgd) -- the same parameters bit for
    bit (deterministic mode), but renderer.splatGrads is not filled."""
    tps = R.preprocess(renderer, camera)
    R.compactIdxs(renderer)
    R.forward(renderer, tps)
    loss, ΔC = lossFunc.value_and_grad(renderer.imageData, gtimg, want_loss)
    if fused_sgd:
        renderer._dC_keepalive = ΔC
        renderer._begin()
        renderer.ctx.backward_sgd(ΔC.data_ptr(), float(lr))
        renderer._end()
        return loss
    R.backward(renderer, ΔC)
    renderer._begin()
    renderer.ctx.sgd_step(float(lr), renderer._grads)        # param .-= lr * Δparam (train.jl:42-46)
    renderer._end()
    R.resetGrads(renderer)                                   # train.jl:55
    return loss


def train(renderer, gtimg, lr: float, lossFunc: LossFunction, iterations: int = 100, camera=None, log_every: int = 0):
    """train.jl:16-59 without the GUI; the reference loops `while score < 0.99` on a score it never updates."""
    losses = []
    for it in range(iterations):
        l = trainStep(renderer, gtimg, lr, lossFunc, camera, want_loss=True)
        losses.append(l)
        if log_every and it % log_every == 0:
            print(f"loss : {l}")                             # loss.jl:69
    return losses
