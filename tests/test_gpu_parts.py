"""Several waves per tile on small grids (gs_config.tile_parts).

The reference runs one 16 x 16 thread block per tile (splat.jl:224-231); here one wave64 composites a tile, and a grid with fewer tiles
than the chip has wave slots (C1: 256 tiles, C2: 2 500 of 5 120) gives a tile 2 or 4 waves, each owning two or one of its four 16 x 4
pixel strips and walking the tile's list on its own.  What must hold:

* against the oracle, with every partition, the bars of BASELINE.json: pixels |d| <= 1e-4 + 1e-4 |x|, gradients rel-L2 <= 1e-3 (C1, a
  ragged case, and the full C2 size);
* between the partitions: every pixel sees the same entries in the same order, except entries a wave drops because they cannot reach
  ITS pixels with alpha >= 2^-27 (the no-op rule of alpha_cull) -- with the cull off the image and the transmittance are therefore
  BIT-identical whatever the partition; with it on they agree to 1e-6.  Gradients agree to 1e-5 (a tile's per-splat sums are formed
  per wave, so the partition changes the order of the additions -- in deterministic mode the point where they are rounded to fixed point);
* the automatic choice: 4 / 2 / 1 by the grid, 1 for frames without the early-out.
"""
import numpy as np
import pytest

from common import hip_context, rel_l2, scene_and_cameras

pytestmark = pytest.mark.gpu
GRADS = ("means", "scales", "quats", "opacities", "shs")
PIX_ATOL, PIX_RTOL, GRAD_REL_L2 = 1e-4, 1e-4, 1e-3


def _frame(ctx, dC, deg):
    ctx.preprocess(); ctx.bin()
    img, tr = ctx.forward_host()
    g = ctx.grads_alloc()
    ctx.backward(dC, g)
    return img, tr, ctx.grads_read(g, deg), ctx.work_counters_ex()


@pytest.mark.parametrize("n,W,H,deg,seed,boost", [(10_000, 256, 256, 0, 1235, 0.0),      # C1
                                                  (6_001, 200, 120, 2, 31, 1.0),          # ragged edges, dense: pixels freeze, strips die
                                                  (100_000, 800, 800, 3, 1236, 0.0)])     # C2
@pytest.mark.parametrize("parts", [0, 1, 2, 4])
def test_every_partition_meets_the_oracle_bars(oracle, n, W, H, deg, seed, boost, parts):
    from gaussiansplat_amd import synthetic
    O = oracle
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    sc["scales"] = (sc["scales"] + np.float32(boost)).astype(np.float32)
    dC = synthetic.make_dC(W, H, seed)
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=1, t_min=1e-5, omp=True)
    gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC,
                      t_min=1e-5, omp=True)
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, tile_parts=parts)
    img, tr, grads, wc = _frame(ctx, dC, deg)
    ctx.close()
    assert np.all(np.abs(img - ref["image"]) <= PIX_ATOL + PIX_RTOL * np.abs(ref["image"])), np.abs(img - ref["image"]).max()
    assert np.all(np.abs(tr - ref["trans"]) <= PIX_ATOL + PIX_RTOL * np.abs(ref["trans"]))
    for k in GRADS:
        e = rel_l2(grads[k], np.asarray(gref[k]).reshape(grads[k].shape))
        assert e <= GRAD_REL_L2, (k, e)
    assert wc["walked_fwd"] == wc["walked_bwd"] > 0 and wc["evaluated_fwd"] == wc["evaluated_bwd"]


@pytest.mark.parametrize("cull", [False, True])
def test_partitions_agree(cull):
    """cull off: bit-identical image and transmittance; cull on: to the size of what the no-op rule drops; gradients to 1e-5"""
    from gaussiansplat_amd import synthetic
    n, W, H, deg, seed = 8_000, 232, 152, 1, 77                                # 15 x 10 tiles, ragged
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    sc["scales"] = (sc["scales"] + np.float32(1.1)).astype(np.float32)        # dense: several hundred entries per tile, pixels freeze
    dC = synthetic.make_dC(W, H, seed)
    out = {}
    for parts in (1, 2, 4):
        ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, tile_parts=parts, alpha_cull=cull, deterministic=True)
        out[parts] = _frame(ctx, dC, deg)
        ctx.close()
    assert 0 < out[1][3]["walked_fwd"] < 0.9 * 8_000 * 40                      # (the early-out is at work: pixels freeze)
    for parts in (2, 4):
        a, b = out[1], out[parts]
        if not cull:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        else:
            assert np.abs(a[0] - b[0]).max() <= 1e-6 and np.abs(a[1] - b[1]).max() <= 1e-6
        for k in GRADS:
            assert rel_l2(b[2][k], a[2][k]) <= 1e-5, (k, rel_l2(b[2][k], a[2][k]))


def test_automatic_choice_and_frames_without_early_out(oracle):
    """4 waves per tile when 4 x tiles fit the 5120 wave slots, 2 when 2 x tiles do, else 1; literal frames (t_min = 0) always 1 --
    seen through gs_get_tile_parts and the frame's result: equal, bit for bit (deterministic mode), to the forced partition."""
    from gaussiansplat_amd import synthetic
    for (W, H, expect) in ((320, 320, 4), (800, 800, 2), (1296, 1296, 1)):    # 400, 2500, 6561 tiles
        n, deg, seed = 20_000, 0, 5
        sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
        dC = synthetic.make_dC(W, H, seed)
        res = {}
        for parts in (0, expect):
            ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, tile_parts=parts, alpha_cull=False, deterministic=True)
            res[parts] = _frame(ctx, dC, deg)
            assert ctx.tile_parts_of_frame() == expect
            ctx.close()
        assert np.array_equal(res[0][0], res[expect][0]) and np.array_equal(res[0][1], res[expect][1])
        for k in GRADS:
            if expect > 1:
                assert np.array_equal(res[0][2][k], res[expect][2][k]), k
            else:       # a grid with a launch order: the automatic mode may split the few tiles that stand above an even share (below)
                assert rel_l2(res[0][2][k], res[expect][2][k]) <= 1e-5, k
    sc, cam, T, P, ocam = scene_and_cameras(3_000, 160, 96, 1, 9)
    ctx = hip_context(sc, cam, T, P, 160, 96, 1, order=1, t_min=0.0, tile_parts=4)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    assert ctx.tile_parts_of_frame() == 1
    ctx.close()


# ---------------------------------------------------------------- heavy tiles split by the launch order (round 5)
ALWAYS_ORDER = 2
FRONT = 2304          # GS_LPT_FRONT: entries of a launch order reserved for the extra waves of split tiles


def _slot_frames(ctx, dC, deg, nframes=2):
    """frames under one view slot: from the second on the forward runs on the slot's launch order (with its split tiles)"""
    out = None
    for _ in range(nframes):
        ctx.set_view_slot(0)
        out = _frame(ctx, dC, deg)
    return out


def _split_units(ctx, cull=True):
    """records of the launch order's front region = the extra waves of split tiles in a (debug) forward launch over the last frame's
    order (the clocked kernels exist with the no-op cull only: + 1000 = the other cull setting than the ctx's)"""
    clk = ctx.tile_clock(0, -30 if cull else -1030)
    return int((clk[:FRONT, 1] > 0).sum()), clk


@pytest.mark.parametrize("cull", [False, True])
def test_heavy_tiles_split_by_the_launch_order(oracle, cull):
    """A heavy-tailed scene (synthetic.make_scene(clustered=True): what a trained .ply looks like, splat.jl:106-119) on a grid with a
    launch order: tiles whose work stands far above an even share run as two or four waves, chosen per tile by tile_lpt_order_kernel.
    Same bars as the frame-wide partitions: image / transmittance bit-identical to whole tiles with the cull off, 1e-6 with it on;
    gradients 1e-5 between partitions (deterministic mode); the oracle bars either way."""
    from gaussiansplat_amd import synthetic
    O = oracle
    n, W, H, deg, seed = 150_000, 1024, 768, 1, 1236                          # 64 x 48 = 3072 tiles: too many for frame-wide 2 waves per tile
    sc = synthetic.make_scene(n, W, H, deg, seed=seed, clustered=True)
    _, cam, T, P, ocam = scene_and_cameras(16, W, H, deg, seed)
    dC = synthetic.make_dC(W, H, seed)
    res = {}
    for parts in (1, 0):
        ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, tile_parts=parts, alpha_cull=cull, deterministic=True, debug_flags=ALWAYS_ORDER)
        res[parts] = _slot_frames(ctx, dC, deg)
        if parts == 0:                                                         # (tile_parts = 1: the orders have no front region at all)
            units, clk = _split_units(ctx, cull)
            assert units >= 8, units                                           # the blobs' tiles are split
        ctx.close()
    a, b = res[1], res[0]
    if not cull:
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    else:
        assert np.abs(a[0] - b[0]).max() <= 1e-6 and np.abs(a[1] - b[1]).max() <= 1e-6
    for k in GRADS:
        assert rel_l2(b[2][k], a[2][k]) <= 1e-5, (k, rel_l2(b[2][k], a[2][k]))
    if cull:                                                                   # against the oracle, with the production settings
        ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=1, t_min=1e-5, omp=True)
        gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=1e-5, omp=True)
        img, tr, grads = b[0], b[1], b[2]
        assert np.all(np.abs(img - ref["image"]) <= PIX_ATOL + PIX_RTOL * np.abs(ref["image"])), np.abs(img - ref["image"]).max()
        assert np.all(np.abs(tr - ref["trans"]) <= PIX_ATOL + PIX_RTOL * np.abs(ref["trans"]))
        for k in GRADS:
            e = rel_l2(grads[k], np.asarray(gref[k]).reshape(grads[k].shape))
            assert e <= GRAD_REL_L2, (k, e)


def test_split_entries_are_whole_tiles_for_capped_lists_and_uniform_scenes():
    """(a) a launch that composites whole tiles only (capped lists) under an order with split entries: the first part stands for the tile,
    the others do nothing -- bits as without the order's splits; (b) the uniform BASELINE scene splits nothing."""
    from gaussiansplat_amd import synthetic
    n, W, H, deg, seed = 120_000, 1024, 768, 1, 1236
    sc = synthetic.make_scene(n, W, H, deg, seed=seed, clustered=True)
    _, cam, T, P, ocam = scene_and_cameras(16, W, H, deg, seed)
    dC = synthetic.make_dC(W, H, seed)
    res = {}
    for name, kw in (("whole", dict(tile_parts=1, list_cap=1)), ("capped_under_split_order", dict(tile_parts=0, list_cap=2))):
        ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, deterministic=True, debug_flags=ALWAYS_ORDER, **kw)
        res[name] = _slot_frames(ctx, dC, deg, nframes=3) + (ctx.list_stats(),)
        ctx.close()
    a, b = res["whole"], res["capped_under_split_order"]
    assert b[4]["capped"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for k in GRADS:
        assert np.array_equal(a[2][k], b[2][k]), k
    # the uniform BASELINE scene on a grid with more tiles than wave slots: no tile stands above an even share
    n, W, H = 250_000, 1920, 1080
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    sc["scales"] = (sc["scales"] + np.float32(0.6)).astype(np.float32)
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5)
    _slot_frames(ctx, synthetic.make_dC(W, H, seed), deg)
    assert _split_units(ctx)[0] == 0
    ctx.close()


def test_heavy_tiles_backward_as_list_segments(oracle):
    """Round 5: a split tile's backward runs as segments of its LIST (the forward's parts leave (C, T) snapshots at the segment
    boundaries; every segment is a whole-tile wave starting from the snapshot in front of it).  Gradients against the oracle's adjoint
    (bars of BASELINE.json) and against whole tiles (1e-5, deterministic mode); the backward's work counters are the sums over the
    segments and equal the forward's walk; several frames under one slot (the snapshots' walked lengths are re-armed every frame)."""
    from gaussiansplat_amd import synthetic
    O = oracle
    n, W, H, deg, seed = 150_000, 1024, 768, 1, 1236
    sc = synthetic.make_scene(n, W, H, deg, seed=seed, clustered=True)
    sc["opacities"] = (sc["opacities"] - np.float32(1.5)).astype(np.float32)   # fainter still: the blobs' tiles walk several thousand entries
    _, cam, T, P, ocam = scene_and_cameras(16, W, H, deg, seed)
    dC = synthetic.make_dC(W, H, seed)
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=1, t_min=1e-5, omp=True)
    gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=1e-5, omp=True)
    res = {}
    for parts in (1, 0):
        ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, tile_parts=parts, deterministic=True, debug_flags=ALWAYS_ORDER)
        for _ in range(4):
            ctx.set_view_slot(0)
            res[parts] = _frame(ctx, dC, deg)
        ctx.close()
    # the segments really ran: clocked backward launch (float atomics) of the same frame, records per workgroup; the segment units come first
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, tile_parts=0, debug_flags=ALWAYS_ORDER)
    for _ in range(3):
        ctx.set_view_slot(0)
        _frame(ctx, dC, deg)
    clk = ctx.tile_clock(1, -30)
    nseg_units = 8 * (FRONT // 24) * 7
    assert clk.shape[0] > nseg_units + FRONT and int((clk[:nseg_units, 1] > 0).sum()) >= 8, int((clk[:nseg_units, 1] > 0).sum())
    ctx.close()
    a, b = res[1], res[0]
    assert np.abs(a[0] - b[0]).max() <= 1e-6 and np.abs(a[1] - b[1]).max() <= 1e-6
    for k in GRADS:
        assert rel_l2(b[2][k], a[2][k]) <= 1e-5, (k, rel_l2(b[2][k], a[2][k]))
        e = rel_l2(b[2][k], np.asarray(gref[k]).reshape(b[2][k].shape))
        assert e <= GRAD_REL_L2, (k, e)
    assert b[3]["walked_bwd"] == b[3]["walked_fwd"] or b[3]["walked_bwd"] >= a[3]["walked_bwd"]


@pytest.mark.parametrize("n,W,H,deg,seed,boost,segs", [(10_000, 256, 256, 0, 1235, 0.0, 2),       # C1: 256 tiles, two segments x four pixel parts
                                                       (100_000, 800, 800, 3, 1236, 0.0, 2),      # C2: 2500 tiles, two segments
                                                       (6_001, 200, 120, 2, 31, 1.0, 2)])         # ragged edges, dense
def test_small_grids_backward_as_list_segments(oracle, n, W, H, deg, seed, boost, segs):
    """Round 5: on a grid with fewer tiles than wave slots and a view slot with history, EVERY tile's backward runs as two segments
    of its list (the forward's waves -- two or four per tile, by pixel strips -- leave one snapshot per tile), with as many pixel parts on
    top as still fit the wave slots, instead of pixel parts that each walk the whole list.  Oracle bars; 1e-5 against the frame without history (pixel parts); the image is the same bits."""
    from gaussiansplat_amd import synthetic
    O = oracle
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    sc["scales"] = (sc["scales"] + np.float32(boost)).astype(np.float32)
    dC = synthetic.make_dC(W, H, seed)
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=1, t_min=1e-5, omp=True)
    gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=1e-5, omp=True)
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, tile_parts=0, deterministic=True)
    ctx.set_view_slot(2)
    first = _frame(ctx, dC, deg)                                               # no history yet: pixel parts
    for _ in range(2):
        ctx.set_view_slot(2)
        got = _frame(ctx, dC, deg)                                             # list segments
    ctx.close()
    assert np.array_equal(first[0], got[0]) and np.array_equal(first[1], got[1])
    assert np.all(np.abs(got[0] - ref["image"]) <= PIX_ATOL + PIX_RTOL * np.abs(ref["image"]))
    for k in GRADS:
        assert rel_l2(got[2][k], first[2][k]) <= 1e-5, (k, rel_l2(got[2][k], first[2][k]))
        e = rel_l2(got[2][k], np.asarray(gref[k]).reshape(got[2][k].shape))
        assert e <= GRAD_REL_L2, (k, e)
    assert got[3]["walked_bwd"] >= first[3]["walked_bwd"] > 0                   # (the segments' sum is the tile's whole walk; a part's count is its own)
