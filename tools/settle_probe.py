#!/usr/bin/env python3
"""Why do the first ~20 frames of a fresh renderer run 4 % slower than the rest?  (GPU box)
Fresh renderer each time, then: A 5 warm-up frames; B 50 ms of unrelated GPU work (torch) then 5 frames; C 25 warm-up frames;
D 5 frames, 50 ms host sleep, 5 frames -- then 20 timed frames each.  B == C and A slow: the chip's clocks (idle while the model is
uploaded) need tens of ms of load to come up; A == B: something in the renderer settles over its first frames."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplat_amd import renderer as R, synthetic, distributed as D

n, W, H, deg = synthetic.CONFIGS["C3"]
scene = synthetic.make_scene(n, W, H, deg, seed=1234 + 2)
gx, gy = (W + 15) // 16, (H + 15) // 16
cams = [synthetic.scene_camera(W, view=v) for v in (0, 4)]
dCs = [torch.as_tensor(synthetic.make_dC(W, H, 1236 + v)).cuda() for v in (0, 4)]
big = torch.empty(64 * 1024 * 1024, device="cuda")


def frames(hv, k0, k):
    for i in range(k0, k0 + k):
        D.multi_view_step(hv, [cams[i % 2]], [dCs[i % 2]], sync="allreduce", overlap=False, pipeline=False)


def run(mode):
    r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, device=0)
    hv = D.HipViewRenderer(r)
    torch.cuda.synchronize()
    if mode == "B":
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            big.mul_(1.0001); torch.cuda.synchronize()
    k = 0
    if mode == "C":
        frames(hv, 0, 25); k = 25
    elif mode == "D":
        frames(hv, 0, 5); torch.cuda.synchronize(); time.sleep(0.05); frames(hv, 5, 5); k = 10
    else:
        frames(hv, 0, 5); k = 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames(hv, k, 20)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


for rep in range(2):
    for mode in "ABCD":
        print("mode %s: %.4f ms/frame" % (mode, run(mode)), flush=True)
