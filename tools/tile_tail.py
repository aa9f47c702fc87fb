#!/usr/bin/env python3
"""Per-tile clocks of the composite kernels (GPU box): where the kernel time goes besides VALU issue.

    python3 tools/tile_tail.py [C3] [out.json]

For every scheduling variant (tens digit: 1 = one wave per tile in launch (tile) order; 3 = plain launch over the frame's
longest-first order (gs_config.schedule 3: what the backward and the next forward of the view slot use)) and both kernels it launches once with gs_debug_tile_clock and reports
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
    # SIMD-time by the number of waves resident on the SIMD (the kernels are built for five): a SIMD issues for one wave at
    # roughly half the rate it reaches with several, and not at all with none
    occ_hist = np.zeros(9)
    for k in range(len(keys)):
        m = inv == k
        e2 = np.concatenate([np.stack([start[m] - t0, np.ones(m.sum(), np.int64)], 1), np.stack([end[m] - t0, -np.ones(m.sum(), np.int64)], 1)])
        e2 = e2[np.lexsort((e2[:, 1], e2[:, 0]))]
        c2 = np.cumsum(e2[:, 1])
        d2 = np.diff(np.concatenate([e2[:, 0], [span]]))
        occ_hist[0] += e2[0, 0]                                          # before its first wave
        np.add.at(occ_hist, np.clip(c2, 0, 8), d2)
    occ_hist = occ_hist / max(occ_hist.sum(), 1)
    # SIMD throughput by occupancy: every tile is taken to progress uniformly over its lifetime (evaluated / duration entries per
    # tick); a SIMD's rate in an interval is the sum over its resident tiles; averaged (time-weighted) by the number resident.
    rate_num, rate_den = np.zeros(9), np.zeros(9)
    tile_rate = evaluated / np.maximum(end - start, 1)
    for k in range(len(keys)):
        m = np.nonzero(inv == k)[0]
        ev2 = sorted([(int(start[i]), 1, tile_rate[i]) for i in m] + [(int(end[i]), -1, -tile_rate[i]) for i in m])
        c, r, prev = 0, 0.0, None
        for t, dc, dr in ev2:
            if prev is not None and t > prev and c > 0:
                rate_num[min(c, 8)] += r * (t - prev); rate_den[min(c, 8)] += (t - prev)
            c += dc; r += dr; prev = t
    simd_rate = [float(rate_num[i] / rate_den[i]) if rate_den[i] > 0 else 0.0 for i in range(9)]
    order = np.argsort(evaluated)
    dec = np.array_split(order, 10)
    # in-kernel shader-cycle stamps: per-entry loops vs everything else (staging a batch, waiting for its gathers, prologue/epilogue)
    loop_cyc = clk[ran, 4].astype(np.float64) if clk.shape[1] > 4 else np.zeros(len(dur))
    stage_cyc = clk[ran, 5].astype(np.float64) if clk.shape[1] > 5 else np.zeros(len(dur))
    # frozen-pixel work inside live strips: per evaluated entry the kernel executes `exec` strip slots (forward:
    # always 4; backward: the strips the entry's box can reach that still hold a live pixel); if the tile's live pixels were
    # packed 64 to a slot it would need `ideal`; `alive` = strips holding any live pixel, `pix` = live pixels.
    frozen = None
    if clk.shape[1] > 7:
        ex = (clk[ran, 6] >> np.uint64(32)).astype(np.float64); ideal = (clk[ran, 6] & np.uint64(0xFFFFFFFF)).astype(np.float64)
        alive = (clk[ran, 7] >> np.uint64(32)).astype(np.float64); pix = (clk[ran, 7] & np.uint64(0xFFFFFFFF)).astype(np.float64)
        evs = max(float(evaluated.sum()), 1.0)
        frozen = {"strip_slots_executed_per_entry": float(ex.sum() / evs), "strips_with_a_live_pixel_per_entry": float(alive.sum() / evs),
                  "slots_if_live_pixels_packed_per_entry": float(ideal.sum() / evs), "live_pixels_per_entry": float(pix.sum() / evs),
                  "packed_over_executed": float(ideal.sum() / max(ex.sum(), 1.0)), "live_pixel_share_of_executed_lanes": float(pix.sum() / max(64.0 * ex.sum(), 1.0))}
        if clk.shape[1] > 13 and clk[ran, 8:14].sum() > 0:              # forward: evaluated entries by the slots K they would need
            for w, name in enumerate(("any_pixel_anywhere", "whole_rows", "inside_columns")):
                h = [float((clk[ran, 8 + 2 * w] >> np.uint64(32)).sum()), float((clk[ran, 8 + 2 * w] & np.uint64(0xFFFFFFFF)).sum()),
                     float((clk[ran, 9 + 2 * w] >> np.uint64(32)).sum()), float((clk[ran, 9 + 2 * w] & np.uint64(0xFFFFFFFF)).sum())]
                tot = max(sum(h), 1.0)
                frozen["entries_share_by_slots_1_to_4_" + name] = [round(x / tot, 4) for x in h]
                frozen["mean_slots_" + name] = sum((k + 1) * x for k, x in enumerate(h)) / tot
        if clk.shape[1] > 14 and clk[ran, 14].sum() > 0:
            frozen["evaluated_if_culled_against_live_rectangle_over_evaluated"] = float(clk[ran, 14].astype(np.float64).sum() / evs)
    # least squares: tile duration ~ a * walked + b * evaluated + c  (what a launch order should sort by)
    A = np.stack([walked.astype(np.float64), evaluated.astype(np.float64), np.ones(len(dur))], 1)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    resid = dur - A @ coef
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
        "simd_time_share_by_resident_waves_0_to_8": [round(float(x), 4) for x in occ_hist],
        "simd_entries_per_us_by_resident_waves_0_to_8": [round(x * 100.0, 4) for x in simd_rate],
        "loop_cycles_per_evaluated_entry_by_decile": [float(loop_cyc[d].sum() / max(evaluated[d].sum(), 1)) for d in dec],
        "stage_cycles_per_walked_entry_by_decile": [float(stage_cyc[d].sum() / max(walked[d].sum(), 1)) for d in dec],
        "stage_share_of_stamped_cycles_by_decile": [float(stage_cyc[d].sum() / max((stage_cyc[d] + loop_cyc[d]).sum(), 1)) for d in dec],
        "stage_share_of_stamped_cycles": float(stage_cyc.sum() / max((stage_cyc + loop_cyc).sum(), 1)),
        "walked_per_evaluated_by_decile": [float(walked[d].sum() / max(evaluated[d].sum(), 1)) for d in dec],
        "duration_fit_ticks": {"per_walked": float(coef[0]), "per_evaluated": float(coef[1]), "const": float(coef[2]),
                               "rms_residual_over_mean": float(np.sqrt((resid ** 2).mean()) / dur.mean())},
        "frozen_pixels": frozen,
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
    for _ in range(30):                                                       # the chip's clock ramp: profile the settled clock
        ctx.time_composite(1, 30, 2)
    res = {"config": cfg, "instances": ctx.num_instances, "work": ctx.work_counters_ex(), "clock_mhz": ctx.clock_mhz()}
    variants = {"fwd": [10, 30], "bwd": [10, 30]}
    for which, name in ((0, "fwd"), (1, "bwd")):
        for v in variants[name]:
            a = analyse(ctx.tile_clock(which, v))
            a["mean_ms_of_8_launches"] = ctx.time_composite(which, v, 8)
            res[f"{name}_v{v}"] = a
            print(name, v, json.dumps({k: a[k] for k in ("span_us", "mean_ms_of_8_launches", "peak_waves_in_flight", "mean_over_peak",
                                                         "share_of_span_below_50pct_of_peak", "t90_over_span",
                                                         "evaluated_per_simd_max_over_mean", "simd_finish_spread_us",
                                                         "simd_time_share_by_resident_waves_0_to_8", "simd_entries_per_us_by_resident_waves_0_to_8", "stage_share_of_stamped_cycles",
                                                         "duration_fit_ticks", "frozen_pixels")}), flush=True)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
