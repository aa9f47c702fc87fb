#!/usr/bin/env python3
"""SIMD throughput of the composite kernels by resident waves, measured directly (GPU box).

    python3 tools/occupancy_curve.py [C3] [out.json]

tools/tile_tail.py infers "entries per us by resident waves" from tile lifetimes inside the production launch, assuming a tile
progresses uniformly over its lifetime; round 3's forced-occupancy sweep ran over all tiles.  The two tables disagree (a lone wave
at 0.19 vs 0.60 of a SIMD's full rate) and the tail model depends on which is right.  Here a WINDOW of the frame's longest-first
launch order is launched on its own (gs_debug_set_window): M = 1024, 2048 .. 5120 workgroups start at once on 1024 SIMDs, i.e.
M / 1024 waves per SIMD for (nearly) the tiles' whole lifetime, once over the heaviest tiles of the order (its start) and once
over light ones (behind the first 5120).  Per window: launch time, evaluated entries, per-tile microseconds per evaluated entry
(median), entries per us per SIMD.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras  # noqa: E402
from gaussiansplat_amd import synthetic  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", f"occupancy_curve_{cfg}.json")
    n, W, H, deg = synthetic.CONFIGS[cfg]
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
    dC = synthetic.make_dC(W, H, 1)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
    for _ in range(30):                                                  # the chip's clock ramp
        ctx.time_composite(1, 30, 2)
    res = {"config": cfg, "clock_mhz": ctx.clock_mhz(), "windows": []}
    for which, name in ((0, "fwd"), (1, "bwd")):
        for start, label in ((0, "heaviest"), (5120, "light")):
            for M in (1024, 2048, 3072, 4096, 5120):
                if start + M > 8160 - 600 and start:                     # (the tail of the order holds holes)
                    continue
                ctx.set_debug_window(start, M)
                ms = min(ctx.time_composite(which, 30, 4) for _ in range(3))
                clk = ctx.tile_clock(which, 30)
                ran = clk[:, 1] > 0
                dur = (clk[ran, 1] - clk[ran, 0]).astype(np.float64) * 0.01      # us
                ev = (clk[ran, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64)
                wk = (clk[ran, 3] >> np.uint64(32)).astype(np.float64)
                span = (clk[ran, 1].max() - clk[ran, 0].min()) * 0.01
                hw = clk[ran, 2]
                simd = ((hw & np.uint64(0xFFF0)) | ((hw >> np.uint64(32)) << np.uint64(16))).astype(np.int64)
                per_simd = np.bincount(np.unique(simd, return_inverse=True)[1])
                row = {"kernel": name, "tiles": label, "start": start, "workgroups": M, "tiles_run": int(ran.sum()), "simds_used": int(len(per_simd)), "waves_per_simd_max": int(per_simd.max()), "launch_ms": ms,
                       "clocked_span_us": float(span), "evaluated": float(ev.sum()), "walked": float(wk.sum()),
                       "us_per_evaluated_entry_median": float(np.median(dur / np.maximum(ev, 1))),
                       "tile_us_median": float(np.median(dur)), "tile_us_max": float(dur.max()),
                       "entries_per_us_per_simd_launch": float(ev.sum() / (ms * 1e3) / 1024),
                       "entries_per_us_per_wave_median": float(np.median(ev / np.maximum(dur, 1e-9)))}
                res["windows"].append(row)
                print(json.dumps(row), flush=True)
    ctx.set_debug_window(0, 0)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
