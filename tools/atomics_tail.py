#!/usr/bin/env python3
"""Do the two same-address counter atomics per tile (walked / evaluated totals) lengthen the composite kernels?  (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras
from gaussiansplat_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n, W, H, deg = synthetic.CONFIGS[cfg]
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
dC = synthetic.make_dC(W, H, 1)
ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
ctx.preprocess(); ctx.bin(); ctx.forward_host()
g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
res = {}
for rnd in range(6):
    for v in (30, 10030):
        res.setdefault(("fwd", v), []).append(ctx.time_composite(0, v, 8))
        res.setdefault(("bwd", v), []).append(ctx.time_composite(1, v, 5))
for (w, v), ts in sorted(res.items()):
    print(f"{w} variant {v:5d}  median {np.median(ts):.4f}  min {min(ts):.4f}  max {max(ts):.4f} ms")
