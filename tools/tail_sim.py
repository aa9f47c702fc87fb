#!/usr/bin/env python3
"""List-scheduling model of the composite kernels' ragged end.  CPU only.

    python3 tools/tail_sim.py [out.json]

8160 tiles with C3's measured work distribution (evaluated entries per tile, percentiles of profiles/r04u_tile_tail_C3.json) on
1024 SIMDs x 5 wave slots; workgroups are dealt in launch order to the slot that frees first; a SIMD's throughput depends on
its resident waves and is shared equally among them.  Two rate tables:

  forced     0.60 / 0.74 / 0.78 / 0.93 / 1.0   the forced-occupancy sweep over ALL tiles (round 3); predicted LPT = 0.92 of ideal,
                                               which the tile clocks contradict (measured mean_over_peak 0.80, 61 % of SIMD time at 5 waves)
  in_kernel  0.19 / 0.39 / 0.61 / 0.74 / 1.0   per-SIMD entries / us by resident waves inside the production launch
                                               (r04u_tile_tail_C3.json bwd_v30: 1.11 / 2.29 / 3.57 / 4.37 / 5.88): a wave's own speed hardly
                                               depends on its neighbours -- it is bound by the latency of its per-entry chain, and
                                               five of them just fill the SIMD's issue slots.  Reproduces the measured 0.80 / 0.61.

A part of a split tile walks the whole list but composites only its strips.  Instruction cost of a part relative to the whole tile,
from the disassembly of the backward (tools/loop_cost.py: 150 model cycles per entry outside the strips + 76 per live strip, 3.02
live strips per entry; staging, 11 % of a tile's cycles, is repeated by every part):

    2 parts: (150 + 76 * 1.51) / 380 = 0.70 of the loop -> 0.73 of the tile each (two halves: 1.46 x)
    4 parts: (150 + 76 * 0.755) / 380 = 0.55            -> 0.60 each           (four quarters: 2.4 x)

The forward is worse off: staging is 44 % of its cycles (0.44 + 0.56 * (54+...)): a half costs ~0.80, a quarter ~0.66.
"""
import heapq
import json
import sys

import numpy as np

RATES = {
    "forced": {0: 0.0, 1: 0.60, 2: 0.74, 3: 0.78, 4: 0.93, 5: 1.0},
    "in_kernel": {0: 0.0, 1: 0.19, 2: 0.39, 3: 0.61, 4: 0.74, 5: 1.0},
    # round 5, measured directly: windows of 1024 .. 5120 tiles of the launch order launched alone (tools/occupancy_curve.py,
    # profiles/r05a_occupancy_curve_C3.json; backward, heaviest tiles: 1.75 / 3.22 / 4.23 / 4.72 / 5.03 entries per us per SIMD by the
    # launch time, 0.31 / 0.59 / 0.77 / 0.95 / 1.0 by the per-tile medians; forward 0.39 / 0.73 / 0.88 / 0.94 / 1.0)
    "measured": {0: 0.0, 1: 0.33, 2: 0.62, 3: 0.80, 4: 0.94, 5: 1.0},
}
PCT = ([0, 10, 50, 90, 99, 100], [235, 356, 423, 494, 556, 631])          # evaluated entries per tile, C3


def tile_work(n=8160, seed=1):
    rng = np.random.default_rng(seed)
    return np.interp(rng.uniform(0, 100, n), PCT[0], PCT[1])


def simulate(units, rate, nsimd=1024, slots=5):
    """units: work in dispatch order -> (makespan in units of work / full SIMD rate, share of SIMD time by resident waves)."""
    res = [[] for _ in range(nsimd)]
    tlast = np.zeros(nsimd)
    ver = [0] * nsimd
    n = len(units)
    nxt = 0
    for _ in range(slots):
        for i in range(nsimd):
            if nxt < n:
                res[i].append(units[nxt]); nxt += 1
    heap = []

    def push(i):
        k = len(res[i])
        if k:
            heapq.heappush(heap, (tlast[i] + min(res[i]) / (rate[k] / k), i, ver[i]))

    for i in range(nsimd):
        push(i)
    busy = np.zeros(slots + 1)
    t = 0.0
    while heap:
        tt, i, v = heapq.heappop(heap)
        if v != ver[i]:
            continue
        k = len(res[i]); per = rate[k] / k
        dt = tt - tlast[i]
        busy[k] += dt
        res[i] = [r - dt * per for r in res[i]]
        res[i].pop(int(np.argmin(res[i])))
        tlast[i] = tt; t = tt
        if nxt < n:
            res[i].append(units[nxt]); nxt += 1
        ver[i] += 1
        push(i)
    busy[0] = t * nsimd - busy[1:].sum()
    return t, busy / (t * nsimd)


def split(ws, lo, hi, parts, cost):
    """tiles ws (sorted, heaviest first): those with rank share in [lo, hi) run as `parts` units of cost * work each; LPT order of the units."""
    n = len(ws)
    a, b = int(lo * n), int(hi * n)
    units = np.concatenate([ws[:a], np.repeat(ws[a:b] * cost, parts), ws[b:]])
    return np.sort(units)[::-1]


def main():
    out = {}
    w = tile_work()
    ws = np.sort(w)[::-1]
    ideal = w.sum() / 1024
    # cost of one part relative to the whole tile: measured with EVERY C3 tile split (tools/forced_parts.py,
    # profiles/r05a_forced_parts_C3.json: backward 1.405 x / 2.15 x, forward 1.23 x / 1.58 x for 2 / 4 parts)
    for kernel, half, quarter in (("backward", 0.70, 0.54), ("forward", 0.615, 0.395)):
        for name, rate in RATES.items():
            rows = {}

            def run(label, units):
                t, b = simulate(list(units), rate)
                rows[label] = {"eff_vs_unsplit_ideal": round(ideal / t, 3), "five_wave_share": round(float(b[5]), 2), "idle_share": round(float(b[0]), 2)}
                print(f"{kernel:8s} {name:9s} {label:44s} eff {ideal / t:.3f}  5-wave share {b[5]:.2f}  idle {b[0]:.2f}")

            run("LPT whole tiles", ws)
            run("tile order", w)
            for frac in (0.1, 0.2, 0.3):
                run(f"heaviest {frac:.0%} in 2", split(ws, 0.0, frac, 2, half))
            for frac in (0.1, 0.2, 0.3):
                run(f"lightest {frac:.0%} in 2", split(ws, 1.0 - frac, 1.0, 2, half))
            for frac in (0.05, 0.1, 0.2):
                run(f"lightest {frac:.0%} in 4", split(ws, 1.0 - frac, 1.0, 4, quarter))
            run("all in 2", split(ws, 0.0, 1.0, 2, half))
            # a heavy and a light tile back to back in one wave slot: 4080 pairs of (rank k, rank n-1-k), each one unit
            pairs = ws[:4080] + ws[::-1][:4080]
            run("heavy + light paired in one slot", np.sort(pairs)[::-1])
            out[f"{kernel}_{name}"] = rows
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
