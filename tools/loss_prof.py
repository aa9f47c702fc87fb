#!/usr/bin/env python3
"""The loss kernels (gs_loss.hip: L1 + DSSIM of src/loss.jl with its image gradient) at the size they are quoted for, for rocprofv3:

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/loss_prof.py [W H reps]

Prints the hipEvent time per call; the per-kernel split comes from the profiler."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplat_amd import backend as B  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ctx = B.Context()
gt = torch.rand((3, H, W), device="cuda")
img = (gt + 0.1 * torch.randn_like(gt)).clamp(0, 1)
dC = torch.empty_like(img)
for _ in range(3):
    ctx.loss_device(img.data_ptr(), gt.data_ptr(), dC.data_ptr(), W, H, 3, 0.1, want_loss=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ctx.set_stream(torch.cuda.current_stream().cuda_stream or 1)
e0.record()
for _ in range(reps):
    ctx.loss_device(img.data_ptr(), gt.data_ptr(), dC.data_ptr(), W, H, 3, 0.1, want_loss=False)
e1.record(); torch.cuda.synchronize()
val = ctx.loss_device(img.data_ptr(), gt.data_ptr(), dC.data_ptr(), W, H, 3, 0.1, want_loss=True)
print(json.dumps({"W": W, "H": H, "us_per_call": e0.elapsed_time(e1) / reps * 1e3, "loss": val,
                  "algorithmic_bytes": 4 * 3 * W * H * 3, "note": "reads img + gt, writes dC (12 B per pixel-channel); the five window statistics stay in LDS/registers"}))
