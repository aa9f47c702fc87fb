"""Static schedule of the composite launches (gs_config.sched_rounds): several tiles per wave.

The reference launches one thread block per tile (splat.jl:224-231, forward.jl:176-197) and leaves the order to the hardware.  Here,
on a grid with more tiles than the chip holds waves, a wave composites R = ceil(tiles / wave slots) tiles one after the other, dealt
from the view slot's work history so that all waves carry the same total (gs_composite.hip: tile_lpt_order_kernel).  Speed only --
what must hold:

* a tile's pixels see the same list in the same order whatever wave composites it: image and transmittance BIT-identical to the
  one-tile-per-wave launch, gradients bit-identical in deterministic mode, for the automatic R and forced ones, with and without the
  no-op cull, with capped lists extended inside the kernel;
* against the oracle, the bars of BASELINE.json (pixels |d| <= 1e-4 + 1e-4 |x|, gradients rel-L2 <= 1e-3), also with float atomics;
* every tile is composited exactly once (walked / evaluated totals equal those of the one-tile launch), also when an XCD's list is
  shorter than the others' (ragged grids) and when rounds do not divide the list.
"""
import numpy as np
import pytest

from common import hip_context, rel_l2, scene_and_cameras

pytestmark = pytest.mark.gpu
GRADS = ("means", "scales", "quats", "opacities", "shs")
PIX_ATOL, PIX_RTOL, GRAD_REL_L2 = 1e-4, 1e-4, 1e-3
ALWAYS_ORDER, TINY_CAPS = 2, 4


def _frames(ctx, dC, deg, nframes=2, slot=0):
    """nframes frames under one view slot: from the second on the forward runs on the slot's launch order"""
    out = None
    for _ in range(nframes):
        ctx.set_view_slot(slot)
        ctx.preprocess(); ctx.bin()
        img, tr = ctx.forward_host()
        g = ctx.grads_alloc()
        ctx.backward(dC, g)
        out = (img, tr, ctx.grads_read(g, deg), ctx.work_counters_ex(), ctx.sched_rounds_of_frame())
    return out


@pytest.mark.parametrize("cull", [True, False])
def test_bit_identical_to_one_tile_per_wave_on_a_full_hd_grid(cull):
    from gaussiansplat_amd import synthetic
    n, W, H, deg, seed = 120_000, 1920, 1080, 1, 41                            # 8160 tiles > 5120 wave slots: automatic R = 2
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    sc["scales"] = (sc["scales"] + np.float32(0.9)).astype(np.float32)        # dense enough for pixels to freeze
    dC = synthetic.make_dC(W, H, seed)
    res = {}
    for rounds, expect in ((1, 1), (0, 2), (3, 3), (16, 16)):
        ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, sched_rounds=rounds, alpha_cull=cull, deterministic=True)
        res[rounds] = _frames(ctx, dC, deg)
        assert res[rounds][4] == expect, (rounds, res[rounds][4])
        ctx.close()
    a = res[1]
    assert a[3]["walked_fwd"] > 0
    for rounds in (0, 3, 16):
        b = res[rounds]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), rounds
        for k in GRADS:
            assert np.array_equal(a[2][k], b[2][k]), (rounds, k)
        assert a[3] == b[3], (rounds, a[3], b[3])                              # every tile exactly once


@pytest.mark.parametrize("n,W,H,deg,seed,boost", [(6_001, 200, 120, 2, 31, 1.0),          # 13 x 8 tiles, ragged edges, dense
                                                  (40_000, 640, 400, 3, 1236, 0.3)])      # 40 x 25 tiles
@pytest.mark.parametrize("rounds", [2, 3, 7])
def test_forced_rounds_meet_the_oracle_bars(oracle, n, W, H, deg, seed, boost, rounds):
    from gaussiansplat_amd import synthetic
    O = oracle
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    sc["scales"] = (sc["scales"] + np.float32(boost)).astype(np.float32)
    dC = synthetic.make_dC(W, H, seed)
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=1, t_min=1e-5, omp=True)
    gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC,
                      t_min=1e-5, omp=True)
    ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, sched_rounds=rounds, debug_flags=ALWAYS_ORDER)   # float atomics
    img, tr, grads, wc, used = _frames(ctx, dC, deg)
    ctx.close()
    assert used == rounds
    assert np.all(np.abs(img - ref["image"]) <= PIX_ATOL + PIX_RTOL * np.abs(ref["image"])), np.abs(img - ref["image"]).max()
    assert np.all(np.abs(tr - ref["trans"]) <= PIX_ATOL + PIX_RTOL * np.abs(ref["trans"]))
    for k in GRADS:
        e = rel_l2(grads[k], np.asarray(gref[k]).reshape(grads[k].shape))
        assert e <= GRAD_REL_L2, (k, e)
    assert wc["walked_fwd"] == wc["walked_bwd"] > 0 and wc["evaluated_fwd"] == wc["evaluated_bwd"]


def test_rounds_with_capped_lists_extended_inside_the_kernel():
    """capped lists with the minimum cap on every tile (every busy tile appends segments of its list itself) under a static
    schedule: bit-identical to full lists, one tile per wave"""
    from gaussiansplat_amd import synthetic
    n, W, H, deg, seed = 30_000, 480, 320, 1, 5
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    sc["scales"] = (sc["scales"] + np.float32(1.2)).astype(np.float32)
    dC = synthetic.make_dC(W, H, seed)
    res = {}
    for name, kw in (("plain", dict(sched_rounds=1, list_cap=1, debug_flags=ALWAYS_ORDER)),
                     ("rounds_caps", dict(sched_rounds=3, list_cap=2, debug_flags=ALWAYS_ORDER | TINY_CAPS))):
        ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, deterministic=True, tile_parts=1, **kw)
        res[name] = _frames(ctx, dC, deg, nframes=3) + (ctx.list_stats(),)
        ctx.close()
    a, b = res["plain"], res["rounds_caps"]
    assert b[4] == 3 and b[5]["capped"] and b[5]["extended_segments"] > 0
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for k in GRADS:
        assert np.array_equal(a[2][k], b[2][k]), k


def test_no_static_schedule_without_early_out_or_on_small_grids():
    sc, cam, T, P, ocam = scene_and_cameras(3_000, 160, 96, 1, 9)
    for kw, expect in ((dict(t_min=0.0, sched_rounds=4, debug_flags=ALWAYS_ORDER), 1),    # literal frames: one tile per wave
                       (dict(t_min=1e-5, sched_rounds=4), 1),                                # 60 tiles, no forced launch order
                       (dict(t_min=1e-5, sched_rounds=4, debug_flags=ALWAYS_ORDER), 4)):
        ctx = hip_context(sc, cam, T, P, 160, 96, 1, order=1, **kw)
        ctx.set_view_slot(1); ctx.preprocess(); ctx.bin(); ctx.forward_host()
        assert ctx.sched_rounds_of_frame() == expect, (kw, ctx.sched_rounds_of_frame())
        ctx.close()
