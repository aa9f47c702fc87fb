#!/usr/bin/env python3
"""What a small frame's composite launches are made of (GPU box)   python3 tools/small_frame_clocks.py [C1|C2] [tile_parts=0]
One record per workgroup of the production launch (tile_parts 0: a tile has 2-8 waves), or per tile with one wave per tile (tile_parts 1): when each tile's wave started and ended relative to the
launch's first start, how long it ran, the shader cycles inside / outside its per-entry loops, and the walked / evaluated entries per tile --
with every wave resident at once the launch lasts as long as its slowest wave."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras  # noqa: E402
from gaussiansplat_amd import synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
n, W, H, deg = synthetic.CONFIGS[cfg]
seed = 1234 + ["C1", "C2", "C3", "C4", "C5"].index(cfg)
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
import torch  # noqa: E402
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 0                   # gs_config.tile_parts: 0 = what production does, 1 = one wave per tile
ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, tile_parts=parts)
dC = torch.as_tensor(synthetic.make_dC(W, H, 1)).cuda()
g = ctx.grads_alloc()
ctx.set_view_slot(0)
for _ in range(6):
    ctx.preprocess(); ctx.bin(); ctx.forward_device(); ctx.backward(dC.data_ptr(), g, overwrite=True)
ctx.synchronize()
mhz = ctx.clock_mhz()
out = {"config": cfg, "tile_parts": ctx.tile_parts_of_frame(), "shader_mhz": mhz}
for which, name in ((0, "forward"), (1, "backward")):
    clk = ctx.tile_clock(which, 30 if parts == 1 else -30)
    ran = clk[:, 1] > 0
    c = clk[ran]
    st, en = c[:, 0].astype(np.int64), c[:, 1].astype(np.int64)
    t0 = st.min()
    dur = (en - st) * 0.01
    inloop = c[:, 4].astype(np.float64) / mhz
    outloop = c[:, 5].astype(np.float64) / mhz
    walked = (c[:, 3] >> np.uint64(32)).astype(np.int64); ev = (c[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    q = lambda a: [round(float(x), 2) for x in np.percentile(a, [0, 10, 50, 90, 100])]
    out[name] = dict(workgroups=int(ran.sum()), span_us=round(float((en.max() - t0) * 0.01), 2), start_us_p0_10_50_90_100=q((st - t0) * 0.01),
                     end_us=q((en - t0) * 0.01), duration_us=q(dur), in_loops_us=q(inloop), outside_loops_us=q(outloop),
                     walked=q(walked), evaluated=q(ev))
print(json.dumps(out))
