#!/usr/bin/env python3
"""Why is the composite forward slower inside a frame than in back-to-back launches?  (GPU box)
A: eight launches back to back (tools/abtest.py's figure); B: single launches, each after a 1 GB fill that evicts L2 and the
infinity cache; C: single launches after the fill AND a re-run of the frame's own preprocess + binning (what precedes it in a frame)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras
from gaussiansplat_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n, W, H, deg = synthetic.CONFIGS[cfg]
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
dC = synthetic.make_dC(W, H, 1)
ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
for _ in range(2):
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")       # 1 GB
res = {"A_back_to_back": [], "B_after_1GB_fill": [], "A2_back_to_back": []}
for rnd in range(6):
    res["A_back_to_back"].append(ctx.time_composite(0, 30, 8))
    t = []
    for k in range(6):
        big.fill_(float(k)); torch.cuda.synchronize()
        t.append(ctx.time_composite(0, 30, -1))
    res["B_after_1GB_fill"].append(float(np.median(t)))
    res["A2_back_to_back"].append(ctx.time_composite(0, 30, 8))
for w in ("bwd",):
    a = [ctx.time_composite(1, 30, 5) for _ in range(4)]
    b = []
    for k in range(6):
        big.fill_(float(k)); torch.cuda.synchronize()
        b.append(ctx.time_composite(1, 30, -1))
    print("bwd back to back median %.4f   after 1 GB fill median %.4f" % (float(np.median(a)), float(np.median(b))))
for k, v in res.items():
    print("fwd %-18s median %.4f  min %.4f  max %.4f" % (k, float(np.median(v)), min(v), max(v)))
