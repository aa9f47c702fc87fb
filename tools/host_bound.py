#!/usr/bin/env python3
"""Is a small frame bound by the host (enqueue time) or by the GPU?  (GPU box)

    python3 tools/host_bound.py [C1] [steps]

Times K fwd+bwd steps twice: the host time to ENQUEUE them (no synchronisation inside) and the wall time including the final
synchronise.  enqueue ~ wall: the host is the bottleneck (fewer, fatter launches or a hipGraph help); enqueue << wall: the GPU is."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplat_amd import renderer as R, synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, W, H, deg = synthetic.CONFIGS[cfg]
scene = synthetic.make_scene(n, W, H, deg, seed=1234 + list(synthetic.CONFIGS).index(cfg))
cams = [synthetic.scene_camera(W, view=v) for v in (0, 4)]
dC = torch.as_tensor(synthetic.make_dC(W, H, 1)).cuda()
r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), None, scene)


def step(k):
    R.resetGrads(r)
    tps = R.preprocess(r, cams[k & 1]); R.compactIdxs(r); R.forward(r, tps); R.backward(r, dC)


for k in range(20):
    step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# the C ABI alone (no torch / renderer mirror around it)
ctx = r.ctx
g = r._grads
t3 = time.perf_counter()
for k in range(K):
    ctx.preprocess(); ctx.bin(); ctx.forward_device(r.imageData.data_ptr(), r.transmittance.data_ptr()); ctx.backward(dC.data_ptr(), g, overwrite=True)
t4 = time.perf_counter()
torch.cuda.synchronize()
t5 = time.perf_counter()
print(json.dumps({"config": cfg, "steps": K, "mirror_enqueue_us_per_step": (t1 - t0) / K * 1e6, "mirror_wall_us_per_step": (t2 - t0) / K * 1e6,
                  "abi_enqueue_us_per_step": (t4 - t3) / K * 1e6, "abi_wall_us_per_step": (t5 - t3) / K * 1e6}))
