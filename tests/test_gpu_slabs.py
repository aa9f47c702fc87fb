"""Binning in depth slabs (gs_config.slab_mode) must be invisible: same entries in the same order with the same 64-entry
batch boundaries as the classic single list, so image and transmittance are BIT-identical, deterministic-mode gradients
are BIT-identical, float-atomic gradients agree to atomic-order noise, and the oracle bars hold."""
import os

import numpy as np
import pytest

from common import hip_context, rel_l2, scene_and_cameras

pytestmark = pytest.mark.gpu
GRADS = ("means", "scales", "quats", "opacities", "shs")


@pytest.fixture(autouse=True)
def _one_wave_per_tile(monkeypatch):
    """Frames binned in depth slabs are composited by one wave per tile (gs_config.tile_parts); the plain frames they are compared with
    BIT FOR BIT here must be too: on these small grids the default would give a tile two or four waves, each testing its entries against
    its own pixels (differences below 2^-27 per entry, tests/test_gpu_parts.py)."""
    monkeypatch.setenv("GSPLAT_TILE_PARTS", "1")


def _frame(ctx, dC, deg):
    ctx.preprocess(); ctx.bin()
    rounds = ctx.num_rounds
    img, tr = ctx.forward_host()
    g = ctx.grads_alloc()
    ctx.backward(dC, g)
    grads = ctx.grads_read(g, deg)
    return rounds, img, tr, grads, ctx.work_counters_ex()


@pytest.mark.parametrize("fractions,rounds", [((0.3,), 2), ((0.15, 0.5), 3), ((0.1, 0.2, 0.4), 4), ((0.999,), 2), ((0.0005, 0.6), 3)])
@pytest.mark.parametrize("t_min", [1e-3, 1e-5])
def test_forced_slabs_bit_identical_to_classic(oracle, fractions, rounds, t_min):
    from gaussiansplat_amd import synthetic
    O = oracle
    n, W, H, deg = 5000, 112, 72, 2                                          # ragged: 7 x 4.5 tiles
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 17)
    sc["scales"] = sc["scales"] + np.float32(1.2)                           # dense: several hundred entries per tile, pixels freeze
    dC = synthetic.make_dC(W, H, 17)
    res = {}
    for det in (True, False):
        c0 = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=t_min, deterministic=det, slab_mode=0)
        r0 = _frame(c0, dC, deg); c0.close()
        assert r0[0] == 1
        c1 = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=t_min, deterministic=det, slab_mode=1, slab_fractions=fractions)
        r1 = _frame(c1, dC, deg)
        assert r1[0] == rounds, r1[0]
        with pytest.raises(Exception):
            c1.get_array(13)                                                # lists are spread over the rounds
        c1.close()
        assert np.array_equal(r0[1], r1[1]) and np.array_equal(r0[2], r1[2])            # image, transmittance: bit-identical
        assert r1[2].min() >= 0.0                                                          # no sign flag left behind
        assert r0[4]["walked_fwd"] >= r1[4]["walked_fwd"] > 0 and r1[4]["walked_bwd"] == r1[4]["walked_fwd"]
        assert r1[4]["evaluated_fwd"] == r0[4]["evaluated_fwd"] == r1[4]["evaluated_bwd"]
        for k in GRADS:
            if det:
                assert np.array_equal(r0[3][k], r1[3][k]), k
            else:
                assert rel_l2(r1[3][k].reshape(-1), r0[3][k].reshape(-1)) <= 1e-5, k
        res[det] = r1
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=1, t_min=t_min)
    gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=t_min)
    img, tr = res[False][1], res[False][2]
    assert np.all(np.abs(img - ref["image"]) <= 1e-4 + 1e-4 * np.abs(ref["image"]))
    assert np.all(np.abs(tr - ref["trans"]) <= 1e-4 + 1e-4 * np.abs(ref["trans"]))
    for k in GRADS:
        assert rel_l2(res[False][3][k].reshape(-1), gref[k].reshape(-1)) <= 1e-3, k


def test_auto_slabs_after_a_dense_frame():
    """Automatic mode: the first frames of a ctx are classic (no history); once a frame walked less than the threshold share
    of its instances the binning switches to slabs; every frame gives the same bits.  The shipped threshold is 0.03 (with the
    two-level binning slabs no longer pay at C5's 0.06); the test raises it to round 2's 0.15 (gs_config.slab_max_ratio) to exercise the switch, then
    checks that the same scene and the headline workload C3 (28 %) stay classic at the shipped threshold."""
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 300_000, 800, 608, 3
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 77)
    sc["scales"] = sc["scales"] + np.float32(1.5)                           # dense: under 10 % of the instances are walked
    dC = synthetic.make_dC(W, H, 3)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, deterministic=True, slab_max_ratio=0.15)
    frames = [_frame(ctx, dC, deg) for _ in range(4)]
    share = frames[0][4]["walked_fwd"] / ctx.num_instances
    assert share < 0.15, share
    assert frames[0][0] == 1 and frames[-1][0] == 3, [f[0] for f in frames]
    for f in frames[1:]:
        assert np.array_equal(f[1], frames[0][1]) and np.array_equal(f[2], frames[0][2])
        for k in GRADS:
            assert np.array_equal(f[3][k], frames[0][3][k]), k
        assert f[4]["evaluated_fwd"] == frames[0][4]["evaluated_fwd"]
    ctx.close()
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, deterministic=True)
    again = [_frame(ctx, dC, deg) for _ in range(3)]
    assert [f[0] for f in again] == [1, 1, 1] and share > 0.03
    assert np.array_equal(again[-1][1], frames[0][1])
    ctx.close()
    n, W, H, deg = synthetic.CONFIGS["C3"]
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 1236)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
    dC = synthetic.make_dC(W, H, 3)
    assert [_frame(ctx, dC, deg)[0] for _ in range(3)] == [1, 1, 1]
    ctx.close()


def test_sparse_scene_stays_classic():
    """No tile saturates (tiny footprints): the walked share is ~1, so the automatic mode never leaves the single round."""
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 20000, 320, 208, 1
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 5)
    sc["scales"] = sc["scales"] - np.float32(1.5)
    dC = synthetic.make_dC(W, H, 5)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5)
    rounds = [_frame(ctx, dC, deg)[0] for _ in range(4)]
    assert rounds == [1, 1, 1, 1], rounds
    ctx.close()


def test_slab_resume_with_negative_transmittance():
    """A live pixel's T can go (slightly) negative: rounding in the exponent of a near-singular, elongated, fully opaque splat
    can push alpha past 1.  Round 2 carried the `frozen` flag of a pixel in the sign of its stored T between slab rounds, so
    such a pixel was resumed as frozen; the flags now travel in per-tile lane masks.  Elongated splats with sigmoid(o) = 1:
    forced slabs must stay bit-identical to the classic single list, including the pixels whose T is negative."""
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 6000, 160, 112, 0
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 23)
    sc = dict(sc)
    s = sc["scales"].copy()
    s[:, 0] += np.float32(3.0); s[:, 1:] -= np.float32(3.5)                  # needles: 600 : 1 axis ratios, conics close to singular
    sc["scales"] = s
    sc["opacities"] = np.full(n, 30.0, np.float32)                          # cusigmoid(30) rounds to 1.0f
    dC = synthetic.make_dC(W, H, 23)
    res = []
    for kw in (dict(slab_mode=0), dict(slab_mode=1, slab_fractions=(0.07, 0.3)), dict(slab_mode=1, slab_fractions=(0.01, 0.02, 0.5))):
        ctx = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-5, deterministic=True, **kw)
        res.append(_frame(ctx, dC, deg))
        ctx.close()
    assert res[0][0] == 1 and res[1][0] == 3 and res[2][0] == 4
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2], equal_nan=True)
        for k in GRADS:
            assert np.array_equal(r[3][k], res[0][3][k], equal_nan=True), k


def test_forced_slabs_with_super_tiles_of_16(oracle):
    """The slab rounds on the two-level path with super-tiles of 16 x 16 tiles (the default of 4K-class grids, forced here):
    completed super-tiles (super_done_kernel<4>) and completed tiles drop out of the later rounds; bits as with one round."""
    from gaussiansplat_amd import backend as B, synthetic
    n, W, H, deg = 9000, 600, 392, 1                                         # 38 x 25 tiles: 3 x 2 super-tiles of 16, ragged
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 23)
    sc["scales"] = sc["scales"] + np.float32(1.3)
    dC = synthetic.make_dC(W, H, 23)
    c0 = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-4, deterministic=True, slab_mode=0)
    r0 = _frame(c0, dC, deg); c0.close()
    for flags in (B.GS_DEBUG_SUPER16, 0):
        c1 = hip_context(sc, cam, T, P, W, H, deg, order=1, t_min=1e-4, deterministic=True, slab_mode=1, slab_fractions=(0.12, 0.4), debug_flags=flags)
        r1 = _frame(c1, dC, deg); c1.close()
        assert r1[0] == 3
        assert np.array_equal(r0[1], r1[1]) and np.array_equal(r0[2], r1[2])
        assert r1[4]["walked_bwd"] == r1[4]["walked_fwd"] and r1[4]["evaluated_fwd"] == r0[4]["evaluated_fwd"] == r1[4]["evaluated_bwd"]
        for k in GRADS:
            assert np.array_equal(r0[3][k], r1[3][k]), k
