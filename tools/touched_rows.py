#!/usr/bin/env python3
"""How many gaussians does a view's backward touch at all?  (GPU box)  python3 tools/touched_rows.py [C3]
Rows of the per-gaussian 2-D gradient buffer that the composite backward left exactly zero: the per-gaussian chain reads 272 B and
writes 236 B for each of them to produce zeros."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras
from gaussiansplat_amd import backend as B, synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n, W, H, deg = synthetic.CONFIGS[cfg]
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
ctx = hip_context(sc, cam, T, P, W, H, deg)
ctx.preprocess(); ctx.bin(); ctx.forward_host()
g = ctx.grads_alloc(); ctx.backward(synthetic.make_dC(W, H, 1), g)
g2 = ctx.get_array(B.ARR_GRAD2D)
touched = np.any(g2 != 0, axis=1)
rect = ctx.get_array(B.ARR_TILE_RECT)
print(json.dumps({"config": cfg, "gaussians": n, "with_a_tile": int((rect[:, 0] > 0).sum()), "touched_by_the_backward": int(touched.sum()),
                  "share_touched": float(touched.mean())}))
