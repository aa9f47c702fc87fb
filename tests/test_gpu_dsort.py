"""Depth sort in two steps (gs_config.depth_sort: 256 key-range buckets, then one workgroup per bucket inside LDS) against the
oracle's depth order (reference src/forward.jl:103: sortperm(-tps[3,:], lt=isless), ties by gaussian index) and against the
four-pass radix sort it replaces: the SAME permutation, bit for bit, on every path -- including the depth distributions the
bucket map is bad at (all keys equal, a wall of gaussians at one depth plus an outlier: one oversize bucket, sorted through
global memory), non-finite depths, tiny and ragged sizes, and the automatic fall-back to the four-pass sort after such a frame.
"""
import numpy as np
import pytest

from common import hip_context, scene_and_cameras

pytestmark = pytest.mark.gpu

W, H, DEG = 160, 96, 0


def _scene(oracle, n, seed, edit=None):
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, DEG, seed)
    sc = dict(sc)
    if edit is not None:
        m = sc["means"].copy()
        edit(m)
        sc["means"] = m.astype(np.float32)
    pre = oracle.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], DEG, ocam)
    return sc, cam, T, P, pre


def _orders(oracle, sc, cam, T, P, pre, order, frames=2, **kw):
    from gaussiansplat_amd import backend as B
    want = oracle.depth_order(pre["tps"], order)
    out = {}
    for mode in (1, 2, 0):                                             # four-pass, buckets (forced), automatic
        ctx = hip_context(sc, cam, T, P, W, H, DEG, order=order, t_min=0.0, depth_sort=mode, **kw)
        for frame in range(frames):
            ctx.preprocess(); ctx.bin()
            got = ctx.get_array(B.ARR_SORT_IDXS)
            assert np.array_equal(got, want), (mode, order, frame, int(np.argmax(got != want)))
        out[mode] = got
        ctx.close()
    return out


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 257, 4095, 4096, 4097, 30_001, 300_000])
@pytest.mark.parametrize("order", [1, 2])
def test_bucket_sort_equals_oracle_order(oracle, n, order):
    sc, cam, T, P, pre = _scene(oracle, n, 4000 + n % 89)
    _orders(oracle, sc, cam, T, P, pre, order)


@pytest.mark.parametrize("rank_mode", [0, 1])
def test_bucket_sort_rank_modes(oracle, rank_mode):
    sc, cam, T, P, pre = _scene(oracle, 50_000, 4100)
    _orders(oracle, sc, cam, T, P, pre, 1, rank_mode=rank_mode)


def test_equal_keys_and_long_runs(oracle):
    """every gaussian at one point (one key: one bucket of 20 000, zero digits -> index order) and 300 distinct points (runs of 200
    equal keys inside the buckets)"""
    def one(m):
        m[:] = np.float32([0.1, 0.2, 0.5])
    sc, cam, T, P, pre = _scene(oracle, 20_000, 4200, one)            # > 8192 in ONE bucket: the global-memory path with zero passes
    assert np.unique(pre["tps"][:, 2]).size == 1
    _orders(oracle, sc, cam, T, P, pre, 1)

    def few(m):
        m[:] = m[np.arange(m.shape[0]) % 300]
    sc, cam, T, P, pre = _scene(oracle, 60_000, 4201, few)
    _orders(oracle, sc, cam, T, P, pre, 1)
    _orders(oracle, sc, cam, T, P, pre, 2)


@pytest.mark.parametrize("n", [40_000, 100_000])                      # 256-thread and 1024-thread finishing kernels
def test_oversize_bucket_goes_through_global_memory(oracle, n):
    """a wall: n gaussians within 1e-3 of one point and two outliers along the view axis stretch the key range, so all but two
    land in one bucket of many times a workgroup's capacity (several chunks x several digit passes through global memory)"""
    rng = np.random.default_rng(7)

    def wall(m):
        m[:] = np.float32([0.0, 0.0, 1.0]) + 1.0e-3 * rng.random(m.shape).astype(np.float32)
        m[5] = (0.0, 0.0, -3.9)
        m[77] = (0.0, 0.0, 3.9)
    sc, cam, T, P, pre = _scene(oracle, n, 4300, wall)
    assert np.unique(pre["tps"][:, 2]).size > 200                     # real sorting work inside the bucket
    _orders(oracle, sc, cam, T, P, pre, 1, frames=3)                  # automatic mode: frame 2 on runs the four-pass sort
    _orders(oracle, sc, cam, T, P, pre, 2, frames=2)


def test_non_finite_depths(oracle):
    """NaN and +-Inf depths stay out of the key range (they are clamped into the end buckets) and sort where isless puts them"""
    def bad(m):
        m[11, :] = np.nan
        m[500, 2] = np.nan
        m[900, 2] = np.inf
        m[901, 2] = -np.inf
        m[1500:1510, 2] = np.nan
    sc, cam, T, P, pre = _scene(oracle, 25_000, 4400, bad)
    _orders(oracle, sc, cam, T, P, pre, 1)
    _orders(oracle, sc, cam, T, P, pre, 2)


def test_camera_changes_between_frames(oracle):
    """the key range is per frame (two accumulator parities, re-armed by the sort): alternate two cameras on one ctx, preprocess
    twice before a bin, bin twice after one preprocess"""
    from gaussiansplat_amd import backend as B
    from gaussiansplat_amd import camera as gcam, synthetic
    from oracle import oracle as O
    n = 70_000
    sc, cam0, T0, P0, ocam0 = scene_and_cameras(n, W, H, DEG, 4500)
    cams = [(cam0, T0, P0, ocam0)]
    cam1 = synthetic.scene_camera(W, view=3)
    T1, P1 = gcam.compute_transform(cam1), gcam.compute_projection(cam1, W, H)
    cams.append((cam1, T1, P1, O.camera_from_arrays(T1, P1, np.float32(cam1.fx), np.float32(cam1.fy), np.float32(cam1.near), np.float32(cam1.far),
                                                    cam1.eye, cam1.lookAt, W, H)))
    want = []
    for cam, T, P, ocam in cams:
        pre = oracle.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], DEG, ocam)
        want.append(oracle.depth_order(pre["tps"], 1))
    assert not np.array_equal(want[0], want[1])
    ctx = hip_context(sc, cam0, T0, P0, W, H, DEG, order=1, t_min=0.0, depth_sort=2)

    def set_cam(k):
        cam, T, P, _ = cams[k]
        ctx.set_camera(T, P, float(np.float32(cam.fx)), float(np.float32(cam.fy)), float(np.float32(cam.near)), float(np.float32(cam.far)),
                       cam.eye, cam.lookAt, W, H)
    for k in (0, 1, 1, 0, 1, 0, 0):
        set_cam(k); ctx.preprocess(); ctx.bin()
        assert np.array_equal(ctx.get_array(B.ARR_SORT_IDXS), want[k]), k
    set_cam(0); ctx.preprocess()
    set_cam(1); ctx.preprocess(); ctx.bin()                            # two preprocesses, one bin
    assert np.array_equal(ctx.get_array(B.ARR_SORT_IDXS), want[1])
    ctx.bin()                                                          # a second bin of the same preprocess
    assert np.array_equal(ctx.get_array(B.ARR_SORT_IDXS), want[1])
    set_cam(0); ctx.preprocess(); ctx.preprocess(); ctx.preprocess(); ctx.bin()
    assert np.array_equal(ctx.get_array(B.ARR_SORT_IDXS), want[0])
    ctx.close()


def test_whole_frame_equal_on_both_sorts(oracle):
    """lists, image and deterministic gradients of a frame do not depend on which sort produced the depth order"""
    from gaussiansplat_amd import backend as B
    from gaussiansplat_amd import synthetic
    sc, cam, T, P, pre = _scene(oracle, 20_000, 4600)
    dC = synthetic.make_dC(W, H, 3)
    res = []
    for mode in (1, 2):
        ctx = hip_context(sc, cam, T, P, W, H, DEG, order=1, depth_sort=mode, deterministic=True)
        ctx.preprocess(); ctx.bin()
        img, tr = ctx.forward_host()
        g = ctx.grads_alloc(); ctx.backward(dC, g)
        gr = ctx.grads_read(g, DEG)
        res.append((ctx.get_array(B.ARR_TILE_RANGES), ctx.get_array(B.ARR_SORTED_IDS), img, tr, [gr[k] for k in sorted(gr)]))
        ctx.close()
    a, b = res
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    for x, y in zip(a[4], b[4]):
        assert np.array_equal(x, y)
