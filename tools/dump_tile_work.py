#!/usr/bin/env python3
"""Per-tile evaluated entries of a config's views (GPU box) -> gpurun_out/tile_work_<cfg>.npz (to design launch orders offline)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras
from gaussiansplat_amd import synthetic
out = {}
for cfg in sys.argv[1:] or ["C3"]:
    n, W, H, deg = synthetic.CONFIGS[cfg]
    for view in (0, 3, 4):
        sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg), view=view)
        ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
        ctx.preprocess(); ctx.bin(); ctx.forward_host()
        clk = ctx.tile_clock(0, 10)
        out[f"{cfg}_v{view}_eval"] = (clk[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        out[f"{cfg}_v{view}_walked"] = (clk[:, 3] >> np.uint64(32)).astype(np.uint32)
        out[f"{cfg}_grid"] = np.array([(W + 15) // 16, (H + 15) // 16])
        ctx.close()
np.savez_compressed("gpurun_out/tile_work.npz", **out)
print({k: v.shape for k, v in out.items()})
