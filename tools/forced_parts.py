#!/usr/bin/env python3
"""Measured cost of running EVERY tile of a config as 1, 2 or 4 waves (gs_config.tile_parts), isolated composite kernels (GPU box).

    python3 tools/forced_parts.py [C3]

The ratio time(2 parts) / time(1 part) at C3 (8160 tiles: 16320 half-tile waves on 5120 slots, the SIMDs issue-bound) is the measured
instruction overhead of a split tile that tools/tail_sim.py needs.
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras
from gaussiansplat_amd import synthetic

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n, W, H, deg = synthetic.CONFIGS[cfg]
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
dC = synthetic.make_dC(W, H, 1)
ctxs = {}
for parts in (1, 2, 4):
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, tile_parts=parts, list_cap=1)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
    assert ctx.tile_parts_of_frame() == parts
    ctxs[parts] = ctx
res = {p: {"fwd": [], "bwd": []} for p in ctxs}
for rnd in range(5):
    for p, ctx in ctxs.items():
        res[p]["fwd"].append(ctx.time_composite(0, 30, 6))
        res[p]["bwd"].append(ctx.time_composite(1, 30, 4))
out = {"config": cfg}
for p in ctxs:
    out[f"parts{p}"] = {k: {"min_ms": min(v), "median_ms": sorted(v)[len(v) // 2]} for k, v in res[p].items()}
for p in (2, 4):
    out[f"ratio_parts{p}_over_1"] = {k: out[f"parts{p}"][k]["min_ms"] / out["parts1"][k]["min_ms"] for k in ("fwd", "bwd")}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/forced_parts_{cfg}.json", "w"), indent=1)
