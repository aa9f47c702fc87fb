#!/usr/bin/env python3
"""Is the settle transient of a fresh renderer the chip's shader clock?  (GPU box)

    python3 tools/clock_ramp.py [out.json]

rocprofv3 (profiles/r04b_settle_frames.json) shows that over a fresh renderer's first ~24 frames ONLY the two composite kernels get
faster (backward 803 -> 699 us, forward 397 -> 328 us); the HBM-bound kernels do not move.  The composite kernels are bound by VALU
issue, i.e. by the shader clock.  This probe reads that clock INSIDE the backward composite kernel: the debug instantiation stamps
every tile with shader cycles (s_memtime, inside + outside the per-entry loops) and with 100 MHz real-time ticks (s_memrealtime);
cycles / time over all tiles = the clock the waves actually ran at.  Sampled after 1, 2, 4, 8, 16, 32 and 48 frames of a fresh
renderer (no synchronisation between the frames before a sample), next to the frame time of the group of frames before the sample,
and repeated after one second of idle.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplat_amd import distributed as D, renderer as R, synthetic  # noqa: E402

n, W, H, deg = synthetic.CONFIGS["C3"]
scene = synthetic.make_scene(n, W, H, deg, seed=1234 + 2)
gx, gy = (W + 15) // 16, (H + 15) // 16
cams = [synthetic.scene_camera(W, view=v) for v in (0, 4)]
dCs = [torch.as_tensor(synthetic.make_dC(W, H, 1236 + v)).cuda() for v in (0, 4)]


def in_kernel_mhz(ctx, which):
    clk = ctx.tile_clock(which, 30)
    ran = clk[:, 1] > 0
    cyc = (clk[ran, 4] + clk[ran, 5]).astype(np.float64)
    ticks = (clk[ran, 1] - clk[ran, 0]).astype(np.float64)
    return float(cyc.sum() / ticks.sum() * 100.0), float((clk[ran, 1].max() - clk[ran, 0].min()) * 0.01)


def run(label):
    r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, device=0)
    hv = D.HipViewRenderer(r)
    torch.cuda.synchronize()
    marks = [1, 2, 4, 8, 16, 32, 48]
    out, k, t0 = [], 0, time.perf_counter()
    for m in marks:
        nfr = m - k
        while k < m:
            D.multi_view_step(hv, [cams[k % 2]], [dCs[k % 2]], sync="allreduce", overlap=False, pipeline=False)
            k += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / nfr * 1e3
        mb, span_b = in_kernel_mhz(r.ctx, 1)
        mf, span_f = in_kernel_mhz(r.ctx, 0)
        out.append({"after_frames": m, "ms_per_frame_since_last_sample": round(ms, 4), "backward_in_kernel_MHz": round(mb, 1), "backward_span_us": round(span_b, 1),
                    "forward_in_kernel_MHz": round(mf, 1), "forward_span_us": round(span_f, 1), "idle_probe_MHz": round(r.ctx.clock_mhz(), 1)})
        t0 = time.perf_counter()
    print(label, json.dumps(out), flush=True)
    del hv, r
    return out


res = {"fresh_process": run("fresh"), "second_renderer_same_process": run("second")}
time.sleep(1.0)
res["after_1s_idle"] = run("after_idle")
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        json.dump(res, fh, indent=1)
