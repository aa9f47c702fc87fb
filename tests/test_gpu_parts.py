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
            assert np.array_equal(res[0][2][k], res[expect][2][k]), k
    sc, cam, T, P, ocam = scene_and_cameras(3_000, 160, 96, 1, 9)
    ctx = hip_context(sc, cam, T, P, 160, 96, 1, order=1, t_min=0.0, tile_parts=4)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    assert ctx.tile_parts_of_frame() == 1
    ctx.close()
