#!/usr/bin/env python3
"""Per-tile clocks of the composite kernels (GPU box): where the kernel time goes besides VALU issue.

    python3 tools/tile_tail.py [C3] [out.json]

For every scheduling variant (tens digit: 1 = one wave per tile in launch order; 3 = plain launch over the longest-first
permutation that keeps tile % 8 (the default of the backward); 0 = persistent waves on per-XCD ticket queues, heaviest first;
2 = the same queues in arbitrary order; units digit 1 = the reduce-scatter-tree backward body) and both kernels it launches
once with gs_debug_tile_clock and reports
  * kernel span (first start .. last end, 100 MHz s_memrealtime ticks -> microseconds),
  * the concurrency profile: time-weighted mean of waves in flight, and the share of the span spent below 50 % / 25 % of
    the peak concurrency (the tail),
  * when the last 10 % / 1 % of the tiles finish relative to the span,
  * per-SIMD balance of the evaluated entries (max / mean over the SIMDs that ran anything),
  * histogram of evaluated entries per tile and the cost per evaluated entry (ticks) by decile.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras  # noqa: E402
from gaussiansplat_amd import synthetic  # noqa: E402

TICK_US = 0.01


def analyse(clk):
    start, end = clk[:, 0].astype(np.int64), clk[:, 1].astype(np.int64)
    ran = end > 0
    start, end = start[ran], end[ran]
    hw = clk[ran, 2]
    walked = (clk[ran, 3] >> np.uint64(32)).astype(np.int64)
    evaluated = (clk[ran, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    t0, t1 = int(start.min()), int(end.max())
    span = t1 - t0
    ev = np.concatenate([np.stack([start - t0, np.ones_like(start)], 1), np.stack([end - t0, -np.ones_like(end)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    conc = np.cumsum(ev[:, 1])
    dt = np.diff(np.concatenate([ev[:, 0], [span]]))
    peak = int(conc.max())
    mean_conc = float((conc * dt).sum() / max(span, 1))
    below50 = float(dt[conc < 0.5 * peak].sum() / max(span, 1))
    below25 = float(dt[conc < 0.25 * peak].sum() / max(span, 1))
    fin = np.sort(end - t0)
    simd = ((hw & np.uint64(0xFFF0)) | ((hw >> np.uint64(32)) << np.uint64(16))).astype(np.int64)      # xcc | se/sh/cu/simd bits
    keys, inv = np.unique(simd, return_inverse=True)
    per_simd = np.bincount(inv, weights=evaluated.astype(np.float64))
    per_simd_end = np.zeros(len(keys)); np.maximum.at(per_simd_end, inv, (end - t0).astype(np.float64))
    dur = (end - start).astype(np.float64)
    order = np.argsort(evaluated)
    dec = np.array_split(order, 10)
    return {
        "tiles": int(ran.sum()), "span_us": span * TICK_US, "peak_waves_in_flight": peak, "mean_waves_in_flight": mean_conc,
        "mean_over_peak": mean_conc / max(peak, 1), "share_of_span_below_50pct_of_peak": below50, "share_of_span_below_25pct_of_peak": below25,
        "t90_over_span": float(fin[int(0.9 * len(fin))] / max(span, 1)), "t99_over_span": float(fin[int(0.99 * len(fin))] / max(span, 1)),
        "simds_seen": int(len(keys)), "evaluated_per_simd_max_over_mean": float(per_simd.max() / per_simd.mean()),
        "simd_finish_spread_us": [float(np.percentile(per_simd_end, q)) * TICK_US for q in (5, 50, 95, 100)],
        "evaluated_total": int(evaluated.sum()), "walked_total": int(walked.sum()),
        "evaluated_per_tile_percentiles": [int(np.percentile(evaluated, q)) for q in (0, 10, 50, 90, 99, 100)],
        "tile_duration_us_percentiles": [float(np.percentile(dur, q)) * TICK_US for q in (0, 10, 50, 90, 99, 100)],
        "ticks_per_evaluated_entry_by_decile": [float(dur[d].sum() / max(evaluated[d].sum(), 1)) for d in dec],
    }


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", f"tile_tail_{cfg}.json")
    n, W, H, deg = synthetic.CONFIGS[cfg]
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
    dC = synthetic.make_dC(W, H, 1)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
    res = {"config": cfg, "instances": ctx.num_instances, "work": ctx.work_counters_ex()}
    variants = {"fwd": [10, 30, 0], "bwd": [10, 30, 0, 20, 11, 31]}
    for which, name in ((0, "fwd"), (1, "bwd")):
        for v in variants[name]:
            a = analyse(ctx.tile_clock(which, v))
            a["mean_ms_of_8_launches"] = ctx.time_composite(which, v, 8)
            res[f"{name}_v{v}"] = a
            print(name, v, json.dumps({k: a[k] for k in ("span_us", "mean_ms_of_8_launches", "peak_waves_in_flight", "mean_over_peak",
                                                         "share_of_span_below_50pct_of_peak", "t90_over_span",
                                                         "evaluated_per_simd_max_over_mean")}), flush=True)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
