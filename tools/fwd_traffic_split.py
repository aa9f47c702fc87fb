#!/usr/bin/env python3
"""Isolated composite FORWARD launches of the C3 frame for a --pmc pass (tools/fwd_traffic_split.sh): `reps` launches of one
scheduling variant, nothing else in the process besides one frame of set-up.   python3 tools/fwd_traffic_split.py VARIANT [reps]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras
from gaussiansplat_amd import synthetic, backend as B

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 30
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, W, H, deg = synthetic.CONFIGS["C3"]
sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + 2)
ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
ctx.preprocess(); ctx.bin(); ctx.forward_host()
ms = ctx.time_composite(0, variant, reps)
out = {"variant": variant, "reps": reps, "ms": ms}
if os.environ.get("FWD_SPLIT_HOST_STATS"):
    # what the launch needs at least: every gaussian with a walked entry once (64-byte row), the walked ids, the image
    rg = ctx.get_array(B.ARR_TILE_RANGES).reshape(-1, 2).astype(np.int64)
    ids = ctx.get_array(B.ARR_SORTED_IDS)
    clk = ctx.tile_clock(0, 10)
    walked = (clk[:, 3] >> np.uint64(32)).astype(np.int64)
    touched = np.zeros(n, bool)
    ahead_rows = 0
    for t in range(len(rg)):
        s0, s1 = rg[t]
        w = min(walked[t], s1 - s0)
        touched[ids[s0:s0 + w]] = True
        ahead_rows += min(64, (s1 - s0) - w)                      # the batch gathered ahead of the early-out decision and never used
    out.update(walked=int(walked.sum()), unique_rows=int(touched.sum()), unique_row_bytes=int(touched.sum()) * 64, walked_id_bytes=int(walked.sum()) * 4,
               rows_gathered_ahead_and_unused=int(ahead_rows), bytes_gathered_ahead_and_unused=int(ahead_rows) * 64 + 8160 * 128 * 4,
               algorithmic_bytes=40 * int(walked.sum()) + 16 * W * H)
print(json.dumps(out))
ctx.close()
