"""gs_bin for mid-size frames (gaussiansplat_amd/csrc/gs_bin_mid.hip: TWO launches -- unordered candidate lists per super-tile of 8 x 8 tiles
plus a 2-D difference array of the rectangles, then one workgroup per tile that tests its super-tile's candidates, ranks its hits by
(depth key, id) in LDS and finds its list's start from the difference array; what bin_path 0 takes between the small-frame path and
262 144 gaussians x 8192 tiles, BASELINE C2) against the oracle's lists (gso_bin = hitBinning / scan! / compactHits,
src/forward.jl:118-161, src/compact.jl:3-21; depth order = CUDA.sortperm, forward.jl:103), bit for bit: tile ranges, sorted ids,
sorted keys -- and sortIdxs, which this path computes only when asked.

Cases the structure makes special: BASELINE C2; the first size beyond the small path; a 1080p grid (8160 tiles, 135 super-tiles, ragged
last rows); ragged and one-row grids; few hits per tile (ranked by counting) and hundreds to thousands (the bitonic network); exact
duplicates (ties in index order); index order and the 2-D renderer; and the three ways a frame can fail to fit -- a tile with more than
4096 hits, a super-tile with more candidates than its region, lists beyond the ids buffer -- after which the host bins the SAME frame
again with the general path (same lists) and keeps to it.
"""
import numpy as np
import pytest

from common import hip_context, scene_and_cameras

pytestmark = pytest.mark.gpu


def _lists(B, ctx):
    return (ctx.get_array(B.ARR_TILE_RANGES), ctx.get_array(B.ARR_SORTED_IDS), ctx.get_array(B.ARR_SORTED_KEYS), ctx.get_array(B.ARR_SORT_IDXS))


def _check(O, B, sc, cam, T, P, ocam, W, H, deg, order, frames=2, expect=(4, 4), **kw):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pre = O.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, omp=True)
    ranges, ids, okeys = O.bin_lists(pre["bbs"], pre["tps"], order, 16, gx, gy)
    perm = O.depth_order(pre["tps"], order) if order != 0 else np.arange(sc["means"].shape[0], dtype=np.uint32)
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=order, t_min=0.0, **kw)
    seen = []
    for frame in range(frames):
        ctx.preprocess(); ctx.bin()
        assert ctx.num_instances == len(ids)                       # (settles the frame: a frame that did not fit has been binned again by now)
        seen.append(ctx.bin_path_of_frame())
        got = _lists(B, ctx)
        assert np.array_equal(got[0], ranges), frame
        assert np.array_equal(got[1], ids), frame
        assert np.array_equal(got[2], okeys), frame
        assert np.array_equal(got[3], perm), frame
    ctx.close()
    assert tuple(seen) == tuple(expect[:frames]), seen
    return len(ids)


@pytest.mark.parametrize("order", [1, 2, 0])
@pytest.mark.parametrize("n,W,H,grow", [
    (100_000, 800, 800, 0.0),     # BASELINE C2
    (16_385, 256, 256, 0.0),      # one gaussian beyond the small path
    (30_000, 1920, 1080, 0.5),    # 8160 tiles, 135 super-tiles, ragged last super-tile row
    (20_000, 700, 370, 1.5),      # 44 x 24 tiles: ragged grid; hundreds of hits per tile (counting and bitonic ranks side by side)
    (40_000, 2000, 16, 1.0),      # 125 x 1 tiles
])
def test_lists_match_the_oracle(oracle, n, W, H, grow, order):
    from gaussiansplat_amd import backend as B
    if order != 1 and n > 40_000:
        pytest.skip("the large case once")
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 9000 + n)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(grow)).astype(np.float32)
    _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, order)


def test_the_largest_model(oracle):
    from gaussiansplat_amd import backend as B
    n, W, H = 262_144, 640, 480
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 77)
    sc = dict(sc); sc["scales"] = (sc["scales"] - np.float32(0.7)).astype(np.float32)
    ni = _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, frames=1)
    assert ni > n
    sc2, cam2, T2, P2, ocam2 = scene_and_cameras(n + 1, W, H, 0, 78)
    sc2 = dict(sc2); sc2["scales"] = (sc2["scales"] - np.float32(0.7)).astype(np.float32)
    _check(oracle, B, sc2, cam2, T2, P2, ocam2, W, H, 0, 1, frames=1, expect=(0,))        # one more: the general path


def test_duplicates_and_gaussians_without_tiles(oracle):
    from gaussiansplat_amd import backend as B
    n, W, H = 30_001, 640, 400
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 99)
    sc = dict(sc)
    m = sc["means"].copy()
    m[::3] += np.float32(1e6)                                      # off screen
    m[1::30] = m[1]                                                # a thousand exact duplicates: one tile neighbourhood holds them all, ties by index
    m[5] = np.nan; m[8, 2] = np.inf
    sc["means"] = m
    for k in ("scales", "quats", "opacities"):
        a = sc[k].copy(); a[1::30] = a[1]; sc[k] = a
    for order in (1, 2):
        _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, order)


def test_a_tile_with_more_hits_than_its_workgroup_ranks(oracle):
    """Footprints of hundreds of pixels: tiles with more than 4096 hits.  The frame is reported, binned again by the general path -- same
    lists -- and the following frames go there directly."""
    from gaussiansplat_amd import backend as B
    n, W, H = 24_000, 640, 400
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 5)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(3.2)).astype(np.float32)
    ni = _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, frames=3, expect=(0, 0, 0))
    assert ni > 4096 * 40 * 25 // 4


def test_a_super_tile_with_more_candidates_than_its_region(oracle):
    """The whole model inside one super-tile: its candidate region (eight times an even share) overflows; binned again by the general
    path."""
    from gaussiansplat_amd import backend as B
    n, W, H = 60_000, 800, 800
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 6)
    sc = dict(sc)
    m = sc["means"].copy(); m[:, 0] *= np.float32(0.08); m[:, 1] *= np.float32(0.08); sc["means"] = m
    sc["scales"] = (sc["scales"] - np.float32(1.5)).astype(np.float32)
    _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, frames=2, expect=(0, 0))


def test_lists_beyond_the_first_guess_of_the_ids_buffer(oracle):
    """A fresh ctx sizes the ids buffer for 64 instances per gaussian; a frame with more (and no tile over 4096) is binned again."""
    from gaussiansplat_amd import backend as B
    n, W, H = 20_000, 1600, 1200
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 7)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(2.4)).astype(np.float32)
    ni = _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, frames=2, expect=(0, 0))
    assert ni > 64 * n


@pytest.mark.parametrize("n,W,H", [(30_000, 512, 512), (20_000, 300, 200)])
def test_2d_renderer_takes_it_too(oracle, n, W, H):
    from gaussiansplat_amd import backend as B, synthetic
    from test_gpu_2d import _ctx
    sc = synthetic.make_scene_2d(n, W, H, seed=31, scale_hi=1.5)
    out = []
    for bp in (0, 3):
        ctx = _ctx(sc, W, H, bin_path=bp)
        ctx.preprocess(); ctx.bin()
        ni = ctx.num_instances
        assert ctx.bin_path_of_frame() == (4 if bp == 0 else 0)
        out.append((ni, ctx.get_array(B.ARR_TILE_RANGES), ctx.get_array(B.ARR_SORTED_IDS)))
        ctx.close()
    assert out[0][0] == out[1][0] and out[0][0] > n
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_frames_with_the_early_out_and_slot_history(oracle):
    """C2 frames with the early-out, view slots and the backward: image, transmittance and deterministic gradients equal those of the
    two-level path bit for bit over a sequence of frames (the last tile's workgroup carries the previous forward's walked counts to the
    host; tile parts and list segments of small grids read the slot's history)."""
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 100_000, 800, 800, 1
    dC = synthetic.make_dC(W, H, 3)
    res = []
    for bp in (0, 3):
        sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1235)
        ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, bin_path=bp, deterministic=True, tile_parts=1)
        g = ctx.grads_alloc()
        frames = []
        for k in range(4):
            ctx.set_view_slot(k % 2)
            ctx.preprocess(); ctx.bin()
            img, tr = ctx.forward_host()
            ctx.backward(dC, g, overwrite=True)
            assert ctx.bin_path_of_frame() == (4 if bp == 0 else 0)
            frames.append((img, tr, ctx.grads_read(g, deg), ctx.work_counters_ex()))
        res.append(frames)
        ctx.close()
    for a, b in zip(*res):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[3] == b[3]
        for k in a[2]:
            assert np.array_equal(a[2][k], b[2][k]), k
