#!/usr/bin/env python3
"""Would a static schedule balanced by MEASURED tile time beat the dynamic longest-first launch?  (GPU box)

    python3 tools/static_balance_probe.py [C3]

The static schedule of gs_config.sched_rounds deals tiles to waves by their evaluated-entry count; measured (profiles/r05c_*): the
cost per evaluated entry varies 1.7 x between tiles (fewer live strips per entry on dense tiles), the waves' sums of TIME are
unequal and the launch is 33 % slower than the dynamic one.  This probe rebuilds the launch order from the shader cycles each tile
took in the previous launch of the same kernel (gs_debug_rebuild_order, work_mode 1), for R = 1 (dynamic, longest first by time)
and R = 2, 3 (static), a few iterations each, and times the isolated kernels.
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
ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, sched_rounds=1)
ctx.preprocess(); ctx.bin(); ctx.forward_host()
g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
for _ in range(30):
    ctx.time_composite(1, 30, 2)
out = {"config": cfg, "clock_mhz": ctx.clock_mhz()}
for which, name, reps in ((1, "bwd", 4), (0, "fwd", 6)):
    rows = []
    def t(label):
        ms = min(ctx.time_composite(which, 30, reps) for _ in range(3))
        rows.append({"order": label, "ms": ms}); print(name, label, "%.4f ms" % ms, flush=True)
    t("dynamic, by evaluated entries (production)")
    ctx.rebuild_order(0, 1, 1); t("dynamic, by evaluated entries (rebuilt; leaves cycles)")
    for mode, what in ((2, "instruction-cost model"), (1, "measured cycles")):
        ctx.rebuild_order(0, 1, mode); t(f"dynamic, by evaluated entries (leaves {what})")
        for R in (1, 2):
            for it in range(3):
                ctx.rebuild_order(which, R, mode)
                t(f"R={R} by {what}, iteration {it}")
    out[name] = rows
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/static_balance_probe_{cfg}.json", "w"), indent=1)
