"""gs_bin for small frames (gaussiansplat_amd/csrc/gs_bin_small.hip: ONE launch, one workgroup per tile -- every tile tests every
gaussian, ranks its own hits by (depth key, id) in LDS, and finds its list's start from a closed form over the rectangles; what
bin_path 0 takes up to 16 384 gaussians x 1024 tiles, BASELINE C1) against the oracle's lists (gso_bin = hitBinning / scan! /
compactHits, src/forward.jl:118-161, src/compact.jl:3-21; depth order = CUDA.sortperm, forward.jl:103) and against the general
paths, bit for bit: tile ranges, sorted ids, sorted keys -- and sortIdxs, which this path computes only when asked.

Cases the structure makes special: the path's limits (16 384 gaussians; 1024 tiles; the gaussian x tile budget), fewer gaussians than
one workgroup, one gaussian, a ragged image, footprints covering every tile (a tile's hits = the whole model: the bitonic network at
its largest; few hits are ranked by counting), gaussians without tiles, exact duplicates and a wall of gaussians at ONE depth (ties
in index order), far outliers and non-finite depths, index order, the 2-D renderer, and frame sequences whose history the next
frame reads (walked counts summed by the last tile's workgroup).
"""
import numpy as np
import pytest

from common import hip_context, scene_and_cameras

pytestmark = pytest.mark.gpu


def _lists(B, ctx):
    return (ctx.get_array(B.ARR_SORT_IDXS), ctx.get_array(B.ARR_TILE_RANGES), ctx.get_array(B.ARR_SORTED_IDS), ctx.get_array(B.ARR_SORTED_KEYS))


def _check(O, B, sc, cam, T, P, ocam, W, H, deg, order, frames=2, expect_path=3, **kw):
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pre = O.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam)
    ranges, ids, okeys = O.bin_lists(pre["bbs"], pre["tps"], order, 16, gx, gy)
    perm = O.depth_order(pre["tps"], order) if order != 0 else np.arange(sc["means"].shape[0], dtype=np.uint32)
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=order, t_min=0.0, **kw)
    for frame in range(frames):
        ctx.preprocess(); ctx.bin()
        assert ctx.bin_path_of_frame() == expect_path, frame
        assert ctx.num_instances == len(ids)
        got = _lists(B, ctx)
        assert np.array_equal(got[0], perm), frame
        assert np.array_equal(got[1], ranges), frame
        assert np.array_equal(got[2], ids), frame
        assert np.array_equal(got[3], okeys), frame
    ctx.close()
    return len(ids)


@pytest.mark.parametrize("order", [1, 2, 0])
@pytest.mark.parametrize("n,W,H,grow", [
    (10_000, 256, 256, 0.0),     # BASELINE C1
    (16_384, 256, 240, 0.0),     # the path's largest model (16 items per thread of the order kernel)
    (4096, 512, 512, 0.5),       # 1024 tiles: the largest grid, at the pair budget
    (1023, 70, 37, 1.0),         # fewer gaussians than one workgroup; a ragged image (5 x 3 tiles)
    (1, 333, 222, 3.0),          # one gaussian
    (700, 160, 96, 4.5),         # footprints covering every tile of the grid
    (3000, 1000, 16, 1.0),       # 63 x 1 tiles
])
def test_lists_match_the_oracle(oracle, n, W, H, grow, order):
    from gaussiansplat_amd import backend as B
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 1, 7000 + n)
    sc = dict(sc); sc["scales"] = (sc["scales"] + np.float32(grow)).astype(np.float32)
    ni = _check(oracle, B, sc, cam, T, P, ocam, W, H, 1, order)
    if grow >= 4.0:
        assert ni > 0.3 * n * ((W + 15) // 16) * ((H + 15) // 16)


def test_beyond_the_limits_the_general_path_runs(oracle):
    from gaussiansplat_amd import backend as B
    for n, W, H in [(16_385, 256, 256), (2000, 528, 512), (8192, 512, 272)]:     # one gaussian too many; 1056 tiles; 8192 x 544 pairs > 4 M
        sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 11)
        _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, 1, frames=1, expect_path=0)
    sc, cam, T, P, ocam = scene_and_cameras(3000, 256, 256, 0, 12)
    _check(oracle, B, sc, cam, T, P, ocam, 256, 256, 0, 1, frames=1, expect_path=0, bin_path=3)          # asked for: two-level


def test_without_tiles_duplicates_and_non_finite_depths(oracle):
    """A third of the gaussians far off screen (empty rectangles), a third exact duplicates (equal keys: ties stay in gaussian-index
    order), and a few with NaN / Inf positions (non-finite depth keys: outside the bucket range, clamped into the end buckets)."""
    from gaussiansplat_amd import backend as B
    n, W, H = 6001, 256, 192
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 99)
    sc = dict(sc)
    m = sc["means"].copy()
    m[::3] += np.float32(1e6)
    m[1::3] = m[1]
    m[5] = np.nan; m[8, 2] = np.inf; m[11, 2] = -np.inf
    sc["means"] = m
    for k in ("scales", "quats", "opacities"):
        a = sc[k].copy(); a[1::3] = a[1]; sc[k] = a
    sc["scales"] = (sc["scales"] + np.float32(1.5)).astype(np.float32)
    for order in (1, 2):
        _check(oracle, B, sc, cam, T, P, ocam, W, H, 0, order, frames=3)


def test_a_wall_at_one_depth(oracle):
    """Every gaussian at the same point: one depth key for the whole model (ties resolve by gaussian index), and the centre tiles list
    all 5000 of them (the LDS sort of a tile at 8192 padded pairs)."""
    from gaussiansplat_amd import backend as B
    n, W, H = 5000, 256, 256
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, 0, 5)
    sc = dict(sc)
    m = sc["means"].copy(); m[:, 2] = np.float32(0.25); m[:, 0] *= np.float32(0.0); m[:, 1] *= np.float32(0.0); sc["means"] = m     # one point: one key
    pre = oracle.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], 0, ocam)
    assert len(np.unique(pre["tps"][:, 2])) == 1
    ranges, ids, okeys = oracle.bin_lists(pre["bbs"], pre["tps"], 1, 16, 16, 16)
    ctx = hip_context(sc, cam, T, P, W, H, 0, order=1, t_min=0.0)
    seen = []
    for frame in range(3):
        ctx.preprocess(); ctx.bin()
        seen.append(ctx.bin_path_of_frame())
        got = _lists(B, ctx)
        assert np.array_equal(got[0], np.arange(n, dtype=np.uint32)), frame
        assert np.array_equal(got[1], ranges) and np.array_equal(got[2], ids) and np.array_equal(got[3], okeys), frame
    assert seen == [3, 3, 3], seen
    ctx.close()


@pytest.mark.parametrize("n,W,H", [(5000, 256, 256), (900, 100, 60)])
def test_2d_renderer_takes_it_too(oracle, n, W, H):
    """GaussianRenderer2D (forward.jl:9-33): no depth, lists in index order -- the order kernel sorts nothing and the list kernel reads
    the model's own rectangles."""
    from gaussiansplat_amd import backend as B, synthetic
    from test_gpu_2d import _ctx
    sc = synthetic.make_scene_2d(n, W, H, seed=31)
    out = []
    for bp in (0, 3):
        ctx = _ctx(sc, W, H, bin_path=bp)
        ctx.preprocess(); ctx.bin()
        assert ctx.bin_path_of_frame() == (3 if bp == 0 else 0)
        out.append((ctx.num_instances, ctx.get_array(B.ARR_TILE_RANGES), ctx.get_array(B.ARR_SORTED_IDS)))
        ctx.close()
    assert out[0][0] == out[1][0] and out[0][0] > n
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_frames_with_the_early_out_and_slot_history(oracle):
    """C1-like frames with the early-out, view slots and the backward: image, transmittance and deterministic gradients equal those of
    the two-level path bit for bit over a sequence of frames (the order kernel carries the previous forward's walked counts to the
    host; tile parts and list segments of small grids read the slot's history)."""
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 10_000, 256, 256, 0
    dC = synthetic.make_dC(W, H, 3)
    res = []
    for bp in (0, 3):
        sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1234)
        ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, bin_path=bp, deterministic=True, tile_parts=1)
        g = ctx.grads_alloc()
        frames = []
        for k in range(4):
            ctx.set_view_slot(k % 2)
            ctx.preprocess(); ctx.bin()
            img, tr = ctx.forward_host()
            ctx.backward(dC, g, overwrite=True)
            assert ctx.bin_path_of_frame() == (3 if bp == 0 else 0)
            frames.append((img, tr, ctx.grads_read(g, deg), ctx.work_counters_ex()))
        res.append(frames)
        ctx.close()
    for a, b in zip(*res):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[3] == b[3]
        for k in a[2]:
            assert np.array_equal(a[2][k], b[2][k]), k
