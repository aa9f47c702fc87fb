#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of tools/settle_trace.py -> per-frame kernel durations (which kernels are longer in the first frames?).

    python3 tools/settle_analyse.py <kernel_trace.csv> <out.json>

A frame starts at each gs_preprocess_kernel dispatch.  Per frame: its span (first kernel start .. last kernel end), the sum of its
kernel durations, the gap to the previous frame, and the duration of every kernel by (short) name.  Summary: mean over the frames
1-4, 5-12, 13-24, 25-48 of every kernel's duration and of the span -- the settle transient, kernel by kernel.
"""
import csv
import json
import re
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
rows = []
with open(src) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)", name)
    s = m.group(1) if m else name
    if "composite" in s or "rs_" in s or "ds_local" in s or "gs_sh_bwd" in s or "gs_geom" in s or "preprocess" in s or "l2_write" in s:
        t = re.search(r"<([^>]*)>", name)
        if "composite" in s and t:
            s += "<" + t.group(1).split(",")[0] + ">"
    return s


frames = []
for st, en, name in rows:
    s = short(name)
    if s.startswith("gs_preprocess_kernel"):
        frames.append([])
    if frames:
        frames[-1].append((st, en, s))
per = []
prev_end = None
for f in frames:
    d = defaultdict(float)
    for st, en, s in f:
        d[s] += (en - st) / 1e3
    span = (max(e for _, e, _ in f) - f[0][0]) / 1e3
    per.append({"span_us": span, "kernel_sum_us": sum(d.values()), "gap_before_us": (f[0][0] - prev_end) / 1e3 if prev_end else 0.0, "kernels_us": dict(d)})
    prev_end = max(e for _, e, _ in f)
# the process renders two renderers when GS_SETTLE_STAGES is set; keep the first run only (frames until a long gap)
groups = {"1-4": (0, 4), "5-12": (4, 12), "13-24": (12, 24), "25-48": (24, 48)}
names = sorted({k for p in per for k in p["kernels_us"]})
summary = {}
for g, (a, b) in groups.items():
    sel = per[a:b]
    if not sel:
        continue
    summary[g] = {"frames": len(sel), "span_us": sum(p["span_us"] for p in sel) / len(sel), "kernel_sum_us": sum(p["kernel_sum_us"] for p in sel) / len(sel),
                  "gap_before_us": sum(p["gap_before_us"] for p in sel) / len(sel),
                  "kernels_us": {k: round(sum(p["kernels_us"].get(k, 0.0) for p in sel) / len(sel), 2) for k in names}}
last, first = summary.get("25-48") or summary[list(summary)[-1]], summary["1-4"]
delta = {k: round(first["kernels_us"][k] - last["kernels_us"][k], 2) for k in names}
res = {"source": src, "frames": len(per), "summary_by_frame_group": summary,
       "first4_minus_last_group_us_by_kernel": dict(sorted(delta.items(), key=lambda kv: -abs(kv[1]))),
       "span_us_by_frame": [round(p["span_us"], 1) for p in per], "kernel_sum_us_by_frame": [round(p["kernel_sum_us"], 1) for p in per]}
with open(out, "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps({"frames": len(per), "span_us_by_group": {g: round(v["span_us"], 1) for g, v in summary.items()},
                  "kernel_sum_us_by_group": {g: round(v["kernel_sum_us"], 1) for g, v in summary.items()},
                  "largest_first4_minus_last": dict(list(res["first4_minus_last_group_us_by_kernel"].items())[:8])}))
