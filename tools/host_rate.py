#!/usr/bin/env python3
"""Is a small frame bound by the GPU or by the caller?  (GPU box)   python3 tools/host_rate.py [C1|C2] [frames]
Four ABI calls per frame (gs_preprocess, gs_bin, gs_forward, gs_backward_ex) through ctypes, no synchronisation inside the loop:
host_ms = the loop's own time per frame (the GPU queue never blocks it), frame_ms = with the final synchronise."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from gaussiansplat_amd import backend as B, camera as gcam, synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n, W, H, deg = synthetic.CONFIGS[cfg]
seed = 1234 + ["C1", "C2", "C3", "C4", "C5"].index(cfg)
sc = synthetic.make_scene(n, W, H, deg, seed=seed)
out = {}
for bp in (0, 3):
    ctx = B.Context(t_min=1e-5, bin_path=bp)
    ctx.set_model_host(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"].reshape(n, -1), deg)
    cams = [synthetic.scene_camera(W, view=v) for v in (0, 4)]
    dC = torch.as_tensor(synthetic.make_dC(W, H, 1)).cuda()
    g = ctx.grads_alloc()
    def frame(k):
        cam = cams[k & 1]
        ctx.set_view_slot(k & 1)
        ctx.set_camera(gcam.compute_transform(cam), gcam.compute_projection(cam, W, H), float(cam.fx), float(cam.fy), float(cam.near), float(cam.far), cam.eye, cam.lookAt, W, H)
        ctx.preprocess(); ctx.bin(); ctx.forward_device(); ctx.backward(dC.data_ptr(), g, overwrite=True)
    def frame_same_cam(k):
        ctx.preprocess(); ctx.bin(); ctx.forward_device(); ctx.backward(dC.data_ptr(), g, overwrite=True)
    for name, f in (("two_cameras", frame), ("one_camera_4_calls", frame_same_cam)):
        for k in range(50):
            f(k)
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            f(k)
        t1 = time.perf_counter()
        ctx.synchronize()
        t2 = time.perf_counter()
        out["bin_path_%d_%s" % (bp, name)] = dict(host_ms=round((t1 - t0) / K * 1e3, 4), frame_ms=round((t2 - t0) / K * 1e3, 4))
    ctx.close()
import json
print(json.dumps(dict(config=cfg, frames=K, **out)))
