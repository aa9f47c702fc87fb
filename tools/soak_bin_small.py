#!/usr/bin/env python3
"""Soak (GPU box): random small frames, gs_bin's small path (bin_path 0) against the two-level path (bin_path 3): ranges, ids, sortIdxs equal.
    python3 tools/soak_bin_small.py [cases=80] [seed=1]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gaussiansplat_amd import backend as B, camera as gcam, synthetic  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
done = small = 0
while done < cases:
    W, H = int(rng.integers(1, 33)) * 16 - int(rng.integers(0, 16)), int(rng.integers(1, 33)) * 16 - int(rng.integers(0, 16))
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    n = int(rng.integers(1, min(16384, (4 << 20) // nt) + 1))
    grow = float(rng.uniform(-1.0, 3.5)); order = int(rng.integers(0, 3)); view = int(rng.integers(0, 8))
    sc = synthetic.make_scene(n, max(W, 16), max(H, 16), 0, seed=int(rng.integers(1, 1 << 30)), clustered=bool(rng.integers(0, 2)) and n > 100)
    sc["scales"] = (sc["scales"] + np.float32(grow)).astype(np.float32)
    if rng.integers(0, 3) == 0:                      # ties and outliers
        k = int(rng.integers(1, max(2, n // 3)))
        sc["means"][:k, 2] = sc["means"][0, 2]
        sc["means"][-1] += np.float32(1e5)
    cam = synthetic.scene_camera(max(W, 16), view=view)
    out = []
    for bp in (0, 3):
        ctx = B.Context(order=order, t_min=0.0, bin_path=bp)
        ctx.set_model_host(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"].reshape(n, -1), 0)
        ctx.set_camera(gcam.compute_transform(cam), gcam.compute_projection(cam, W, H), float(cam.fx), float(cam.fy), float(cam.near), float(cam.far), cam.eye, cam.lookAt, W, H)
        for _ in range(2):
            ctx.preprocess(); ctx.bin()
        path = ctx.bin_path_of_frame()
        out.append((path, ctx.num_instances, ctx.get_array(B.ARR_TILE_RANGES), ctx.get_array(B.ARR_SORTED_IDS), ctx.get_array(B.ARR_SORT_IDXS)))
        ctx.close()
    a, b = out
    assert a[0] == 3 and b[0] == 0, (a[0], b[0], n, W, H)
    assert a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]), (n, W, H, grow, order, view)
    done += 1
    if done % 10 == 0:
        print("ok", done, "last:", dict(n=n, W=W, H=H, grow=round(grow, 2), order=order, instances=a[1]), flush=True)
print("soak passed:", done, "cases")
