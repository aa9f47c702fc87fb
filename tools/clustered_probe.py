#!/usr/bin/env python3
"""A heavy-tailed scene (synthetic.make_scene(clustered=True)) beside the uniform one of the same size: frame time, evaluated entries,
per-tile spread, tile-tail summary (GPU box).   python3 tools/clustered_probe.py [C3]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gaussiansplat_amd import synthetic, renderer as R, backend as B
from tile_tail import analyse

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n, W, H, deg = synthetic.CONFIGS[cfg]
gx, gy = (W + 15) // 16, (H + 15) // 16
seed = 1234 + list(synthetic.CONFIGS).index(cfg)
cam = synthetic.scene_camera(W)
dC = torch.as_tensor(synthetic.make_dC(W, H, seed)).cuda()
out = {"config": cfg}
for name, clustered in (("uniform", False), ("clustered", True)):
    sc = synthetic.make_scene(n, W, H, deg, seed=seed, clustered=clustered)
    kw = {k: int(v) for k, v in (("tile_parts", os.environ.get("PROBE_TILE_PARTS", "")),) if v}
    r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), sc, device=0, t_min=1e-5, profile_stages=1, **kw)
    def frame():
        tps = R.preprocess(r, cam); R.compactIdxs(r, (16, 16), (gx, gy)); R.forward(r, tps, (16, 16), (gx, gy)); R.backward(r, dC)
    for _ in range(30):
        frame()
    torch.cuda.synchronize(); r.ctx.stage_stats(reset=True)
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        frame()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    st = {k: round(s / c, 4) for k, (s, c) in r.ctx.stage_stats().items() if c}
    wc = r.ctx.work_counters_ex()
    clk = r.ctx.tile_clock(1, 30)
    ev = (clk[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    a = analyse(clk)
    try:                                                    # the launch as production runs it: heavy tiles split (records per workgroup)
        clkb = r.ctx.tile_clock(1, -30)
        ab = analyse(clkb)
        split_units = int((clkb[:2304, 1] > 0).sum())
    except Exception as e:
        ab, split_units = {"error": str(e)}, -1
    rg = r.ctx.get_array(B.ARR_TILE_RANGES).reshape(-1, 2)
    L = (rg[:, 1] - rg[:, 0]).astype(np.int64)
    row = {"ms_per_frame": ms, "stage_ms": st, "instances": r.ctx.num_instances, "work": wc,
           "list_len_max_median": [int(L.max()), float(np.median(L))], "evaluated_per_tile_max_median": [int(ev.max()), float(np.median(ev))],
           "split_units": split_units, "bwd_tile_tail_as_run": {k: ab.get(k) for k in ("span_us", "mean_over_peak", "tile_duration_us_percentiles", "simd_time_share_by_resident_waves_0_to_8")},
           "bwd_tile_tail": {k: a[k] for k in ("span_us", "mean_over_peak", "tile_duration_us_percentiles", "simd_time_share_by_resident_waves_0_to_8")}}
    out[name] = row
    print(name, json.dumps(row), flush=True)
    del r
u, c = out["uniform"], out["clustered"]
out["clustered_over_uniform_same_evaluated"] = c["ms_per_frame"] / (u["ms_per_frame"] * c["work"]["evaluated_fwd"] / u["work"]["evaluated_fwd"])
print("clustered / (uniform scaled to the same evaluated entries):", out["clustered_over_uniform_same_evaluated"])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/clustered_probe_{cfg}.json", "w"), indent=1)
