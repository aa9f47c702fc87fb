#!/usr/bin/env python3
"""Writes tests/golden/*.npz: small seeded scenes with the outputs of the CPU oracle (oracle/gs_oracle.c).

    python tests/golden/make_golden.py            (from the repo root; needs gcc only)

These are REGRESSION fixtures of the build's own oracle, not vectors of the reference: the reference (Julia +
CUDA.jl) ships no tests or fixtures and cannot run in this pipeline, so parity stays "unpinned by the reference"
(DESIGN.md section 2).  What they pin: the numeric spec (every fp32 / integer output bit for bit) against accidental
change in later rounds, for the C oracle, its NumPy twin (tests/test_golden.py, CPU) and the HIP path (GPU).
Inputs are stored next to the outputs, so the files are self-contained."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import scene_and_cameras           # noqa: E402
from gaussiansplat_amd import synthetic        # noqa: E402
from oracle import oracle as O                 # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_3d(name, n, W, H, deg, seed, order, t_min):
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    r = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=order, t_min=t_min)
    dC = synthetic.make_dC(W, H, seed)
    g = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, r["ranges"], r["ids"], dC, t_min=t_min)
    np.savez_compressed(os.path.join(HERE, name), kind="3d", n=n, W=W, H=H, deg=deg, seed=seed, order=order, t_min=np.float32(t_min),
                        means=sc["means"], scales=sc["scales"], quats=sc["quats"], opacities=sc["opacities"], shs=sc["shs"],
                        T=np.asarray(T, np.float32), P=np.asarray(P, np.float32), fx=np.float32(cam.fx), fy=np.float32(cam.fy),
                        near=np.float32(cam.near), far=np.float32(cam.far), eye=np.asarray(cam.eye, np.float32),
                        lookAt=np.asarray(cam.lookAt, np.float32), dC=dC,
                        **{"pre_" + k: v for k, v in r["pre"].items()}, ranges=r["ranges"], ids=r["ids"], keys=r["keys"],
                        image=r["image"], trans=r["trans"], **{"g_" + k: v for k, v in g.items()})


def golden_2d(name, n, W, H, seed, t_min):
    sc = synthetic.make_scene_2d(n, W, H, seed, scale_hi=2.2)
    r = O.render2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H, t_min=t_min)
    dC = synthetic.make_dC(W, H, seed)
    g = O.backward2d(sc["means"], sc["scales"], sc["rots"], sc["opacities"], sc["colors"], W, H, r["ranges"], r["ids"], dC, t_min=t_min)
    np.savez_compressed(os.path.join(HERE, name), kind="2d", n=n, W=W, H=H, seed=seed, t_min=np.float32(t_min),
                        means=sc["means"], scales=sc["scales"], rots=sc["rots"], opacities=sc["opacities"], colors=sc["colors"], dC=dC,
                        **{"pre_" + k: v for k, v in r["pre"].items()}, ranges=r["ranges"], ids=r["ids"], keys=r["keys"],
                        image=r["image"], trans=r["trans"], **{"g_" + k: v for k, v in g.items()})


if __name__ == "__main__":
    golden_3d("g3d_sh1_depth.npz", 96, 72, 40, 1, 501, O.ORDER_DEPTH_DESC, 0.0)        # ragged image (H % 16 != 0), reference SH degree
    golden_3d("g3d_sh3_index_early.npz", 80, 48, 48, 3, 502, O.ORDER_INDEX, 1e-3)      # literal list order + early-out rule
    golden_2d("g2d.npz", 64, 56, 40, 503, 0.0)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
