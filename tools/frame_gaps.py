#!/usr/bin/env python3
"""Idle gaps between the kernels of a frame, from a rocprofv3 --kernel-trace CSV (GPU box):

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ...
    python3 tools/frame_gaps.py gpurun_out/trace [out.json]

A frame starts at gs_preprocess_kernel and ends with gs_geom_bwd_kernel.  Reports, over the frames of the timed region (the
last ones), the mean frame span, the sum of the kernel durations, and for every kernel its mean duration and the mean idle
gap in front of it (start minus the end of the previous kernel on the device)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    frames, cur = [], None
    for s, e, k in rows:
        if "gs_preprocess_kernel" in k:
            cur = []
        if cur is not None:
            cur.append((s, e, k))
            if "gs_geom_bwd_kernel" in k:
                frames.append(cur); cur = None
    # frames of the default (early-out) path only, the last 60 %
    frames = [f for f in frames if any("composite_fwd_kernel<true" in k for _, _, k in f)]
    frames = frames[int(0.4 * len(frames)):]
    span = sum(f[-1][1] - f[0][0] for f in frames) / max(len(frames), 1)
    busy = sum(sum(e - s for s, e, _ in f) for f in frames) / max(len(frames), 1)
    # whole trace: device time covered by at least one kernel vs the sum of the kernel durations (their difference = time two
    # kernels of different streams ran at once), over the second half of the run
    half = rows[len(rows) // 2:]
    cover, cur_s, cur_e = 0, None, None
    for s0, e0, _ in half:
        if cur_e is None or s0 > cur_e:
            if cur_e is not None:
                cover += cur_e - cur_s
            cur_s, cur_e = s0, e0
        else:
            cur_e = max(cur_e, e0)
    if cur_e is not None:
        cover += cur_e - cur_s
    total = sum(e0 - s0 for s0, e0, _ in half)
    wall = half[-1][1] - half[0][0] if half else 0
    overlap = {"second_half_wall_us": wall / 1e3, "covered_us": cover / 1e3, "kernel_sum_us": total / 1e3, "overlapped_us": (total - cover) / 1e3,
               "idle_us": (wall - cover) / 1e3, "preprocess_launches": sum("gs_preprocess_kernel" in k for _, _, k in half)}
    per = defaultdict(lambda: [0.0, 0.0, 0])
    for f in frames:
        prev_end = None
        for s, e, k in f:
            name = k.split("(")[0].replace("void ", "")
            p = per[name]
            p[0] += e - s
            if prev_end is not None:
                p[1] += max(0, s - prev_end)
            p[2] += 1
            prev_end = max(prev_end or 0, e)
    nf = max(len(frames), 1)
    res = {"overlap": overlap, "frames": len(frames), "frame_span_us": span / 1e3, "kernel_sum_us": busy / 1e3, "idle_us": (span - busy) / 1e3,
           "kernels": {k: {"per_frame_us": v[0] / nf / 1e3, "gap_before_us": v[1] / nf / 1e3, "launches_per_frame": v[2] / nf} for k, v in per.items()}}
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["per_frame_us"]):
        print(f"{v['per_frame_us']:9.1f} us  gap {v['gap_before_us']:6.1f} us  x{v['launches_per_frame']:.0f}  {k[:100]}")
    if out:
        with open(out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
