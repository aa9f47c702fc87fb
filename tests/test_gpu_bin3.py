"""Two-level tile binning (gaussiansplat_amd/csrc/gs_bin3.hip, the default path of gs_bin) against the oracle's lists
(oracle gso_bin = the reference's hitBinning / scan! / compactHits result, src/forward.jl:118-161, src/compact.jl:3-21)
on the cases its structure makes special:

  * footprints that cover many super-tiles (8 x 8 tiles) -- the level-1 staging buffer overflows and the surplus takes the
    direct-write path; super-tile lists far longer than one level-2 segment;
  * grids whose last super-tile row / column is ragged (gx, gy not multiples of 8) and grids smaller than one super-tile;
  * fewer list positions than one level-1 workgroup, a single gaussian, gaussians without any tile;
  * a 4K-class grid (super-tiles of 16 x 16 tiles by default there; 8 x 8 forced: 510 super-tiles, 512 positions per level-1
    workgroup), an 8K grid, and super-tiles of 16 x 16 tiles forced on the small grids.
Lists are compared bit for bit, for the depth and the index order, and with the generate-in-pass radix path (bin_path 2).
"""
import numpy as np
import pytest

from common import hip_context, scene_and_cameras

pytestmark = pytest.mark.gpu


def _check_lists(O, B, sc, cam, T, P, ocam, W, H, deg, order, bin_path=3, **kw):      # 3: two-level whatever the size (0 hands small frames to gs_bin_small.hip)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pre = O.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam)
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=order, t_min=0.0, bin_path=bin_path, **kw)
    ranges, ids, okeys = O.bin_lists(pre["bbs"], pre["tps"], order, 16, gx, gy)
    for frame in range(2):          # frame 0: the host reads the totals before the lists; frame 1: lists enqueued speculatively
        ctx.preprocess(); ctx.bin()
        assert ctx.num_instances == len(ids)
        assert np.array_equal(ctx.get_array(B.ARR_TILE_RANGES), ranges), frame
        assert np.array_equal(ctx.get_array(B.ARR_SORTED_IDS), ids), frame
        assert np.array_equal(ctx.get_array(B.ARR_SORTED_KEYS), okeys), frame
    n_inst = ctx.num_instances
    ctx.close()
    return n_inst


@pytest.mark.parametrize("super16", [False, True])
@pytest.mark.parametrize("order", [1, 0])
@pytest.mark.parametrize("n,W,H,grow", [
    (3000, 640, 400, 3.2),       # footprints of hundreds of pixels: most gaussians span many super-tiles (staging overflow, long lists)
    (1500, 1000, 600, 4.0),      # 63 x 38 tiles: ragged last super-tile row and column
    (700, 100, 90, 2.0),         # 7 x 6 tiles: the whole grid is inside one super-tile
    (1, 333, 222, 3.0),          # one gaussian
    (65, 2040, 72, 2.5),         # 128 x 5 tiles: one row of super-tiles
])
def test_big_footprints_and_ragged_grids(oracle, n, W, H, grow, order, super16):
    """super16: super-tiles of 16 x 16 tiles (what 4K-class grids take by themselves) forced on these small grids"""
    from gaussiansplat_amd import backend as B
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 1, 4000 + n)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(grow)).astype(np.float32)        # log-scales: exp(grow) times larger
    ni = _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 1, order, debug_flags=B.GS_DEBUG_SUPER16 if super16 else 0)
    if n >= 1500:
        assert ni > 20 * n                                                                     # the footprints really are large


def test_gaussians_without_tiles_and_duplicates(oracle):
    """Half of the gaussians behind the camera / off screen (empty rectangles), the rest exact duplicates (equal depth keys:
    ties must stay in gaussian-index order through both levels)."""
    from gaussiansplat_amd import backend as B
    n, W, H = 4099, 512, 384
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 99)
    sc = dict(sc)
    m = sc["means"].copy()
    m[::2] += np.float32(1e6)                                      # far off screen
    m[1::2] = m[1]                                                 # all the others: the same gaussian
    sc["means"] = m
    for k in ("scales", "quats", "opacities"):
        a = sc[k].copy(); a[1::2] = a[1]; sc[k] = a
    sc["scales"] = (sc["scales"] + np.float32(2.0)).astype(np.float32)
    _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1)
    _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, debug_flags=B.GS_DEBUG_SUPER16)


def test_4k_grid_small_scene(oracle):
    """3840 x 2160: 240 x 135 tiles.  By default super-tiles of 16 x 16 tiles (15 x 9 = 135 of them: 8 x 8 would be 30 x 17 =
    510, and level 1 costs per (chunk, super-tile)); GS_DEBUG_SUPER8 forces the 510 (512 positions per level-1 workgroup)."""
    from gaussiansplat_amd import backend as B
    n, W, H = 20_000, 3840, 2160
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 77)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(1.0)).astype(np.float32)
    a = _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1)
    b = _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, bin_path=2)
    c = _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, debug_flags=B.GS_DEBUG_SUPER8)
    assert a == b == c


def test_8k_grid_and_wide_cursors(oracle):
    """7680 x 4320: 480 x 270 tiles = 60 x 34 super-tiles (2040: 256 positions per level-1 workgroup, > 64 KB of LDS; the
    backward's tile order kernel is beyond its LDS and the launch order is used).  GS_DEBUG_WIDE_CURSORS (gs_config.debug_flags)
    forces the 64-bit cursors that lists beyond 4 GB take."""
    from gaussiansplat_amd import backend as B, synthetic
    n, W, H = 3_000, 7680, 4320
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 78)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(2.0)).astype(np.float32)
    a = _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1)
    b = _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, debug_flags=B.GS_DEBUG_WIDE_CURSORS)
    assert a == _check_lists(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, debug_flags=B.GS_DEBUG_WIDE_CURSORS | B.GS_DEBUG_SUPER8)   # 2040 super-tiles of 8 x 8
    n2, W2, H2 = 5_000, 640, 480
    sc2, cam2, T2, P2, ocam2 = scene_and_cameras(n2, W2, H2, 0, 79)
    sc2 = dict(sc2); sc2["scales"] = (sc2["scales"] + np.float32(1.5)).astype(np.float32)
    _check_lists(oracle, B, sc2, cam2, T2, P2, ocam2, W2, H2, 0, 1, debug_flags=B.GS_DEBUG_WIDE_CURSORS)
    assert a == b
    # forward + backward run at this size (plain tile order in the backward)
    ctx = hip_context(sc, cam, T, P, W, H, 0, t_min=1e-5)
    ctx.preprocess(); ctx.bin()
    img, tr = ctx.forward_host()
    g = ctx.grads_alloc(); ctx.backward(synthetic.make_dC(W, H, 2), g); ctx.synchronize()
    gr = ctx.grads_read(g, 0)
    assert np.isfinite(img).all() and all(np.isfinite(v).all() for v in gr.values()) and float(tr.min()) < 1.0
    ctx.close()


def test_render_through_two_level_lists_matches_radix_lists(oracle):
    """forward + backward on the two-level lists == on the radix lists (same lists => bit-identical image and T)."""
    import torch
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 30_000, 800, 608, 1
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 5)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(1.5)).astype(np.float32)
    dC = synthetic.make_dC(W, H, 3)
    out = []
    for bp in (3, 2):
        ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, bin_path=bp, deterministic=True)
        ctx.preprocess(); ctx.bin()
        img, tr = ctx.forward_host()
        g = ctx.grads_alloc(); ctx.backward(dC, g); ctx.synchronize()
        out.append((img.copy(), tr.copy(), {k: v.copy() for k, v in ctx.grads_read(g, deg).items()}))
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    for k in out[0][2]:
        assert np.array_equal(out[0][2][k], out[1][2][k]), k        # deterministic mode: bitwise


def test_speculative_lists_survive_growth_and_shrinkage(oracle):
    """From a ctx's second frame on gs_bin enqueues the tile lists against the capacities of the buffers it has, before the
    host has seen the frame's instance counts; gs_forward reads them after it has enqueued the composite.  A frame whose
    lists outgrow a buffer must list nothing (no out-of-bounds store), be detected, and be redone with larger buffers --
    lists, image and gradients equal to the oracle's for a scene that grows 30-fold, shrinks again and grows a little."""
    from gaussiansplat_amd import backend as B, synthetic
    O = oracle
    n, W, H, deg = 6000, 480, 352, 1
    gx, gy = (W + 15) // 16, (H + 15) // 16
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 123)
    dC = synthetic.make_dC(W, H, 9)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, deterministic=True, bin_path=3)
    base = sc["scales"].copy()
    counts = []
    for shift in (0.0, 1.8, -0.5, 0.1, 0.12):
        s2 = dict(sc); s2["scales"] = (base + np.float32(shift)).astype(np.float32)
        ctx.set_model_host(s2["means"], s2["scales"], s2["quats"], s2["opacities"], s2["shs"].reshape(n, -1), deg)
        ctx.preprocess(); ctx.bin()
        img, tr = ctx.forward_host()
        g = ctx.grads_alloc(); ctx.backward(dC, g)
        grads = ctx.grads_read(g, deg)
        ref = O.render(s2["means"], s2["scales"], s2["quats"], s2["opacities"], s2["shs"], deg, ocam, order=1, t_min=1e-5)
        assert ctx.num_instances == len(ref["ids"])
        assert np.array_equal(ctx.get_array(B.ARR_TILE_RANGES), ref["ranges"]), shift
        assert np.array_equal(ctx.get_array(B.ARR_SORTED_IDS), ref["ids"]), shift
        assert np.all(np.abs(img - ref["image"]) <= 1e-4 + 1e-4 * np.abs(ref["image"])), shift
        gref = O.backward(s2["means"], s2["scales"], s2["quats"], s2["opacities"], s2["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=1e-5)
        for k in ("means", "scales", "quats", "opacities", "shs"):
            a, b = grads[k].astype(np.float64).reshape(-1), gref[k].reshape(-1)
            assert np.linalg.norm(a - b) <= 1e-3 * max(np.linalg.norm(b), 1e-30), (shift, k)
        counts.append(ctx.num_instances)
    assert counts[1] > 8 * counts[0] and counts[2] < counts[0]             # the growth really outran the 12.5 % slack of the buffers
    ctx.close()
