#!/usr/bin/env python3
"""Time the per-gaussian backward kernels (SH kernel + geometry chain) of a frame on their own (GPU box):

    [GSPLAT_HIP_LIB=.../lib_TAG/libgsplat_hip.so] python3 tools/ab_params.py [C3] [reps]      (variant: build --tag TAG -DGS_SHBWD_THREADS=128)

One frame is rendered, its composite adjoint run once, then GS_BWD_PARAMS_ONLY | GS_BWD_PARAMS_SH and | GS_BWD_PARAMS_GEOM are
repeated back to back and timed with stream events (overwrite mode, as in the first backward after a reset)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplat_amd import renderer as R, synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, W, H, deg = synthetic.CONFIGS[cfg]
scene = synthetic.make_scene(n, W, H, deg, seed=1234 + list(synthetic.CONFIGS).index(cfg))
cam = synthetic.scene_camera(W)
dC = torch.as_tensor(synthetic.make_dC(W, H, 1)).cuda()
r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), None, scene)
tps = R.preprocess(r, cam); R.compactIdxs(r); R.forward(r, tps)
R.backward(r, dC, phase="composite")
out = {"config": cfg, "lib": os.environ.get("GSPLAT_HIP_LIB")}
for phase in ("params_sh", "params_geom", "params"):
    for _ in range(3):
        r.ctx.backward(dC.data_ptr(), r._grads, overwrite=True, phase=phase)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r.ctx.backward(dC.data_ptr(), r._grads, overwrite=True, phase=phase)
    e1.record(); torch.cuda.synchronize()
    out[phase + "_us"] = e0.elapsed_time(e1) / reps * 1e3
print(json.dumps(out))
