"""Small grids: two or four waves per tile (2 or 1 pixels per lane; gs_composite.hip template parameter PX).

A frame with fewer tiles than the chip has wave slots is composited by several waves per tile, each walking the tile's list on its
own for 8 or 4 of its rows.  Same lists, same entries in the same order per pixel, same 64-entry batch boundaries for the
early-out; only the no-op test is per wave (finer), so: transmittance bit-identical to the one-wave-per-tile kernels, colour
within the no-op bound, gradients to float-atomic noise -- and all of it within the stated bars of the oracle."""
import numpy as np
import pytest

from common import hip_context, rel_l2, scene_and_cameras

pytestmark = pytest.mark.gpu
GRADS = ("means", "scales", "quats", "opacities", "shs")


def _frame(sc, cam, T, P, W, H, deg, dC, **kw):
    ctx = hip_context(sc, cam, T, P, W, H, deg, **kw)
    out = None
    for _ in range(2):                                                   # the second frame runs on speculative lists
        ctx.preprocess(); ctx.bin()
        img, tr = ctx.forward_host()
        g = ctx.grads_alloc(); ctx.backward(dC, g)
        out = (img, tr, ctx.grads_read(g, deg), ctx.work_counters_ex())
    ctx.close()
    return out


@pytest.mark.parametrize("n,W,H,deg,shift,t_min,waves", [
    (9_000, 256, 256, 0, 0.3, 1e-5, 4),       # config C1's grid: 256 tiles -> four waves per tile
    (12_000, 400, 300, 2, 0.8, 1e-5, 4),      # 25 x 19 tiles, ragged bottom and right edges (300 = 18.75 tiles)
    (12_000, 400, 300, 2, 0.8, 0.0, 4),       # literal reference: no early-out
    (40_000, 800, 720, 3, 0.5, 1e-5, 2),      # 50 x 45 = 2250 tiles -> two waves per tile
    (40_000, 800, 720, 1, 0.5, 1e-3, 2),
])
def test_waves_per_tile_agree_with_one_wave_per_tile_and_the_oracle(oracle, n, W, H, deg, shift, t_min, waves):
    from gaussiansplat_amd import backend as B, synthetic
    O = oracle
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 600 + deg)
    sc["scales"] = sc["scales"] + np.float32(shift)
    dC = synthetic.make_dC(W, H, 61)
    res = {}
    for det in (True, False):
        one = _frame(sc, cam, T, P, W, H, deg, dC, t_min=t_min, deterministic=det, debug_flags=B.GS_DEBUG_PX4)
        many = _frame(sc, cam, T, P, W, H, deg, dC, t_min=t_min, deterministic=det, debug_flags=0)
        ntiles = ((W + 15) // 16) * ((H + 15) // 16)
        assert (4 if ntiles * 4 <= 5120 else 2) == waves
        if t_min == 0.0:                                                 # every wave walks the whole list of its tile
            assert many[3]["walked_fwd"] == waves * one[3]["walked_fwd"] == many[3]["walked_bwd"]
        assert many[3]["evaluated_fwd"] == many[3]["evaluated_bwd"] <= waves * one[3]["evaluated_fwd"]
        assert np.array_equal(many[1], one[1])                           # transmittance: bit-identical
        assert np.all(np.abs(many[0] - one[0]) <= 2e-6 + 1e-6 * np.abs(one[0]))
        for k in GRADS:
            assert rel_l2(many[2][k].reshape(-1), one[2][k].reshape(-1)) <= 1e-5, k
        res[det] = many
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=1, t_min=t_min)
    gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=t_min)
    for det in (True, False):
        img, tr, grads, _ = res[det]
        assert np.all(np.abs(img - ref["image"]) <= 1e-4 + 1e-4 * np.abs(ref["image"]))
        assert np.all(np.abs(tr - ref["trans"]) <= 1e-4 + 1e-4 * np.abs(ref["trans"]))
        for k in GRADS:
            assert rel_l2(grads[k].reshape(-1), gref[k].reshape(-1)) <= 1e-3, k


def test_deterministic_mode_is_reproducible_with_several_waves_per_tile():
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 8_000, 320, 208, 2
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 77)
    dC = synthetic.make_dC(W, H, 5)
    a = _frame(sc, cam, T, P, W, H, deg, dC, t_min=1e-5, deterministic=True, debug_flags=0)
    b = _frame(sc, cam, T, P, W, H, deg, dC, t_min=1e-5, deterministic=True, debug_flags=0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for k in GRADS:
        assert np.array_equal(a[2][k], b[2][k]), k
