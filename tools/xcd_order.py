#!/usr/bin/env python3
"""Experiment (GS_EXPERIMENTS build): launch orders that give every XCD spatially compact groups of tiles, so that the tiles listing
a gaussian share one L2, against the production scheme (XCD = tile mod 8: neighbours in x land on different XCDs).

    GSPLAT_HIP_LIB=gaussiansplat_amd/lib_exp/libgsplat_hip.so python3 tools/xcd_order.py [C3] [single:<name>]

Block b of a plain launch runs on XCD b mod 8, so order[8 j + c] is the j-th tile of XCD c.  Orders: `mod8` the production classes
(tile mod 8), each sorted by work; `super8` / `super4` / `super2`: square groups of 8x8 / 4x4 / 2x2 tiles dealt to the XCDs by greedy
longest-first on the groups' work, tiles of an XCD sorted by work.  `single:<name>` launches only that order (for a --pmc pass)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import hip_context, scene_and_cameras  # noqa: E402
from gaussiansplat_amd import synthetic  # noqa: E402


def interleave(classes, ntiles):
    """order[8 j + c] = j-th tile of class c; a class that runs out borrows from the tail of the longest one"""
    lists = [list(c) for c in classes]
    out = []
    j = 0
    while len(out) < ntiles:
        for c in range(8):
            if len(out) >= ntiles:
                break
            if lists[c]:
                out.append(lists[c].pop(0))
            else:
                k = max(range(8), key=lambda i: len(lists[i]))
                if lists[k]:
                    out.append(lists[k].pop())
        j += 1
    return np.asarray(out, np.uint32)


def make_orders(work, gx, gy):
    ntiles = gx * gy
    t = np.arange(ntiles)
    orders = {}
    cls = [t[(t & 7) == c] for c in range(8)]
    orders["mod8"] = interleave([c[np.argsort(-work[c], kind="stable")] for c in cls], ntiles)
    for s in (8, 4, 2):
        bx, by = (t % gx) // s, (t // gx) // s
        grp = by * ((gx + s - 1) // s) + bx
        ng = int(grp.max()) + 1
        gw = np.bincount(grp, weights=work, minlength=ng)
        load = np.zeros(8)
        g2x = np.zeros(ng, np.int64)
        for g in np.argsort(-gw):                                  # greedy longest-first: balanced work per XCD
            c = int(np.argmin(load)); g2x[g] = c; load[c] += gw[g]
        cl = g2x[grp]
        cls = [t[cl == c] for c in range(8)]
        orders[f"super{s}"] = interleave([c[np.argsort(-work[c], kind="stable")] for c in cls], ntiles)
        print(f"super{s}: groups {ng}, XCD work max/mean {load.max() / load.mean():.3f}, tiles per XCD {[len(c) for c in cls]}", flush=True)
    # what an order kernel can do in one pass: groups of 8x8 tiles ranked by work and dealt to the XCDs in snake order (rank k -> XCD
    # k & 7, reversed in every second round), tiles of an XCD in 32 work classes, inside a class in (group ordinal, local) order
    s = 8
    bx, by = (t % gx) // s, (t // gx) // s
    grp = by * ((gx + s - 1) // s) + bx
    ng = int(grp.max()) + 1
    gw = np.bincount(grp, weights=work, minlength=ng)
    rank = np.empty(ng, np.int64); rank[np.argsort(-gw, kind="stable")] = np.arange(ng)
    gc = np.where((rank >> 3) & 1, 7 - (rank & 7), rank & 7)
    cl = gc[grp]
    local = (rank[grp] >> 3) * 64 + ((t // gx) & 7) * 8 + ((t % gx) & 7)
    wc = np.minimum(31, np.maximum(0, 31 - (work * (32.0 / max(work.max(), 1))).astype(np.int64)))
    cls = []
    for c in range(8):
        m = t[cl == c]
        cls.append(m[np.lexsort((local[m], wc[m]))])
    orders["snake8b"] = interleave(cls, ntiles)
    loads = np.bincount(cl, weights=work, minlength=8)
    print(f"snake8b: XCD work max/mean {loads.max() / loads.mean():.3f}, tiles per XCD {[len(c) for c in cls]}", flush=True)
    cls = [t[cl == c][np.argsort(-work[t[cl == c]], kind="stable")] for c in range(8)]
    orders["snake8"] = interleave(cls, ntiles)
    return orders


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("single:") else "C3"
    single = next((a.split(":", 1)[1] for a in sys.argv[1:] if a.startswith("single:")), None)
    n, W, H, deg = synthetic.CONFIGS[cfg]
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234 + list(synthetic.CONFIGS).index(cfg))
    dC = synthetic.make_dC(W, H, 1)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    os.environ.pop("GS_DEBUG_ORDER_FILE", None)
    clk = ctx.tile_clock(0, 10)
    work = (clk[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64)
    orders = make_orders(work, gx, gy)
    d = tempfile.mkdtemp()
    files = {}
    for k, o in orders.items():
        assert sorted(o.tolist()) == list(range(gx * gy)), k
        files[k] = os.path.join(d, k + ".u32"); o.tofile(files[k])
    if single:
        os.environ["GS_DEBUG_ORDER_FILE"] = files[single]
        print(single, "fwd", ctx.time_composite(0, 30, 10), "bwd", ctx.time_composite(1, 30, 10))
        return
    res = {}
    for rnd in range(5):
        for k in ["prod"] + list(orders):
            if k == "prod":
                os.environ.pop("GS_DEBUG_ORDER_FILE", None)
            else:
                os.environ["GS_DEBUG_ORDER_FILE"] = files[k]
            res.setdefault((k, "fwd"), []).append(ctx.time_composite(0, 30, 8))
            res.setdefault((k, "bwd"), []).append(ctx.time_composite(1, 30, 5))
    for (k, w), ts in res.items():
        ts = sorted(ts)
        print(f"{k:8s} {w}  min {ts[0]:.4f}  median {ts[len(ts) // 2]:.4f}  max {ts[-1]:.4f} ms")
    ctx.close()


if __name__ == "__main__":
    main()
