#!/usr/bin/env python3
"""The first 48 frames of a fresh renderer, for the "settle" transient of bench.py (GPU box).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/settle -o settle -- python3 tools/settle_trace.py [frames] [mode]
    python3 tools/settle_analyse.py gpurun_out/settle/*/settle_kernel_trace.csv profiles/r04_settle_frames.json

The frames are exactly bench.py's C3 steps (two cameras alternating, one view slot each, gradients into the flat buffer, no host
synchronisation between frames).  mode: "default"; "noslots" (no view slots: no launch-order history, no list caps);
"nocap" (list_cap = 1).  Without rocprof it prints the wall time of every group of four frames and, with per-frame
synchronisation in a second renderer, the per-stage hipEvent times of every frame.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplat_amd import distributed as D, renderer as R, synthetic  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 48
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
n, W, H, deg = synthetic.CONFIGS["C3"]
scene = synthetic.make_scene(n, W, H, deg, seed=1234 + 2)
gx, gy = (W + 15) // 16, (H + 15) // 16
cams = [synthetic.scene_camera(W, view=v) for v in (0, 4)]
if mode == "noslots":
    for c in cams:
        c.id = None
dCs = [torch.as_tensor(synthetic.make_dC(W, H, 1236 + v)).cuda() for v in (0, 4)]
kw = dict(list_cap=1) if mode == "nocap" else {}


def run(profile_stages, sync_each):
    r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, device=0, profile_stages=profile_stages, **kw)
    hv = D.HipViewRenderer(r)
    torch.cuda.synchronize()
    wall, stages = [], []
    t0 = time.perf_counter()
    for k in range(frames):
        D.multi_view_step(hv, [cams[k % 2]], [dCs[k % 2]], sync="allreduce", overlap=False, pipeline=False)
        if sync_each:
            torch.cuda.synchronize()
            stages.append({s: round(v, 4) for s, v in r.ctx.stage_times().items() if v > 0})
            t1 = time.perf_counter(); wall.append((t1 - t0) * 1e3); t0 = t1
        elif k % 4 == 3:
            torch.cuda.synchronize()
            t1 = time.perf_counter(); wall.append((t1 - t0) / 4 * 1e3); t0 = t1
    torch.cuda.synchronize()
    del hv, r
    return wall, stages


w, _ = run(0, False)
print(json.dumps({"mode": mode, "ms_per_frame_by_group_of_4": [round(x, 4) for x in w]}), flush=True)
if os.environ.get("GS_SETTLE_STAGES") == "1":
    w2, st = run(1, True)
    print(json.dumps({"mode": mode, "per_frame_sync_ms": [round(x, 4) for x in w2], "stage_ms_by_frame": st}), flush=True)
