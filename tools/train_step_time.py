#!/usr/bin/env python3
"""Time of one training iteration of src/train.jl as intended (preprocess, lists, forward, L1 + DSSIM loss with its image
gradient, backward, SGD step) at C3 -- the loss / optimiser rows of SURVEY 8(f).2 next to the fwd+bwd headline.
    python3 tools/train_step_time.py [C3]     (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gaussiansplat_amd import renderer as R, synthetic, train as TR  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n, W, H, deg = synthetic.CONFIGS[cfg]
gx, gy = (W + 15) // 16, (H + 15) // 16
scene = synthetic.make_scene(n, W, H, deg, seed=1236)
r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, device=0, t_min=1e-5)
cam = synthetic.scene_camera(W, view=0)
gt = torch.rand((3, H, W), device="cuda")
lossFunc = TR.getLossFunction((W, H, 3), 11, 3, renderer=r)
for want, fused in ((False, False), (True, False), (False, True)):
    for _ in range(5):
        TR.trainStep(r, gt, 1e-4, lossFunc, cam, want_loss=want, fused_sgd=fused)
    torch.cuda.synchronize()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K):
        TR.trainStep(r, gt, 1e-4, lossFunc, cam, want_loss=want, fused_sgd=fused)
    torch.cuda.synchronize()
    print({"config": cfg, "want_loss_value_on_host": want, "fused_backward_sgd": fused,
           "ms_per_iteration": round((time.perf_counter() - t0) / K * 1e3, 3)})
# the pieces
def timeit(f, k=50):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print({"loss_and_grad_ms": round(timeit(lambda: lossFunc.value_and_grad(r.imageData, gt, False)), 3)})
r._begin(); g = r._grads
print({"sgd_step_ms": round(timeit(lambda: r.ctx.sgd_step(1e-4, g)), 3)}); r._end()
