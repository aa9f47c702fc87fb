#!/usr/bin/env python3
"""A trained-scene-like workload: many small gaussians (sigma 0.2-1.2 px: ~3 tiles each, no tile saturates).
    python3 tools/small_splats.py [n=3000000]      (GPU box)  -> ms per fwd+bwd frame, stage times, Msplats/s
The synthetic C-configs of BASELINE.json have footprints of ~90 px; this is the other end of the distribution."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras  # noqa: E402
from gaussiansplat_amd import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
W, H, deg = 1920, 1080, 3
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 4242)
sc = dict(sc)
sc["scales"] = (sc["scales"] - np.float32(4.0)).astype(np.float32)          # exp(-4) of the C-config footprints
dC = synthetic.make_dC(W, H, 1)
ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, profile_stages=True)
g = ctx.grads_alloc()
import torch  # noqa: E402
dCd = torch.as_tensor(dC).cuda()
def frame():
    ctx.preprocess(); ctx.bin(); ctx.forward_device(); ctx.backward(dCd.data_ptr(), g, overwrite=True)
for _ in range(5):
    frame()
ctx.synchronize()
ctx.stage_stats(reset=True)
t0 = time.perf_counter()
K = 50
for _ in range(K):
    frame()
ctx.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
st = ctx.stage_stats()
print({"n": n, "instances": ctx.num_instances, "tiles_per_gaussian": ctx.num_instances / n, "ms_per_frame": round(ms, 3),
       "Msplats_per_s": round(n / ms / 1e3, 1), "work": ctx.work_counters_ex()})
print({k: round(v[0] / max(v[1], 1), 4) for k, v in st.items()})
