#!/usr/bin/env python3
"""A/B timing of composite kernel variants on the C3 workload (GPU box)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras
from gaussiansplat_amd import synthetic, backend as B
import torch

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
# variant tens digit = scheduling (gs_composite.hip: apply_sched_variant): 1 one wave per tile in tile order, 3 (or 0) the frame's
# longest-first launch order (production); + 1000 the other alpha_cull setting; + 10000 with the two work-counter atomics per tile.
# Other kernel BODIES are compared as whole libraries: python -m gaussiansplat_amd.build --tag NAME -D... and tools/ab_libs.sh.
variants_f = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["10", "30"])]
variants_b = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["10", "30"])]
n, W, H, deg = synthetic.CONFIGS[cfg]
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
dC = synthetic.make_dC(W, H, 1)
K3 = 3 * (deg + 1) ** 2
rounds = int(os.environ.get("AB_ROUNDS", "6"))
tmins = [float(x) for x in os.environ.get("AB_TMIN", "0,1e-5").split(",")]
for t_min in tmins:
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=t_min)
    ctx.preprocess(); ctx.bin(); img0, tr0 = ctx.forward_host()
    g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
    res = {}
    for rnd in range(rounds):                      # interleaved rounds: clock ramps and noise hit every variant alike
        for v in variants_f:
            res.setdefault(("fwd", v), []).append(ctx.time_composite(0, v, 8))
        for v in variants_b:
            res.setdefault(("bwd", v), []).append(ctx.time_composite(1, v, 5))
    for (w, v), ts in res.items():
        ts = sorted(ts)
        print(f"{w} t_min={t_min:g} v{v:<5d} min {ts[0]:7.3f}  median {ts[len(ts) // 2]:7.3f}  max {ts[-1]:7.3f} ms")
    ctx.close()
