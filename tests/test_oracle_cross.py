"""The two independent CPU restatements (C and NumPy) must agree BIT-FOR-BIT."""
import numpy as np
import pytest

from common import scene_and_cameras
from oracle import gs_oracle_np as NP


@pytest.mark.parametrize("n,W,H,deg,seed", [(1500, 128, 96, 0, 1), (1200, 100, 70, 1, 2), (900, 96, 80, 2, 3), (1000, 112, 64, 3, 4)])
def test_c_vs_numpy_bit_exact(oracle, n, W, H, deg, seed):
    O = oracle
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, seed)
    T2, P2 = NP.camera_matrices(cam.eye, cam.lookAt, cam.up, cam.fx, cam.fy, cam.near, cam.far, W, H)
    oc2 = O.make_camera(cam.eye, cam.lookAt, cam.up, cam.fx, cam.fy, cam.near, cam.far, W, H)
    assert np.array_equal(T, T2) and np.array_equal(P, P2)                 # host mirror == numpy restatement
    assert np.array_equal(np.array(oc2.T[:], np.float32), T) and np.array_equal(np.array(oc2.P[:], np.float32), P)
    pre = O.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam)
    pre2 = NP.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, T, P, cam.fx, cam.fy,
                         cam.eye, cam.lookAt, W, H)
    for k in pre:
        assert np.array_equal(pre[k], pre2[k], equal_nan=True), k
    gx, gy = (W + 15) // 16, (H + 15) // 16
    for order in (0, 1, 2):
        r, i, k = O.bin_lists(pre["bbs"], pre["tps"], order, 16, gx, gy)
        r2, i2, k2 = NP.bin_lists(pre2["bbs"], pre2["tps"][:, 2], order, 16, gx, gy)
        assert np.array_equal(r, r2) and np.array_equal(i, i2) and np.array_equal(k, k2)
        assert np.all(k[1:] >= k[:-1])                                     # keys sorted
        for t_min in (0.0, 1e-3):
            img, tr = O.composite_forward(pre, r, i, ocam, 16, gx, gy, t_min=t_min)
            img2, tr2 = NP.composite_forward(pre2, r2, i2, cam.near, cam.far, W, H, 16, gx, gy, t_min=t_min)
            assert np.array_equal(img, img2, equal_nan=True) and np.array_equal(tr, tr2, equal_nan=True)


def test_early_out_rule_triggers_and_restatements_agree(oracle):
    """Dense scene (lists >> 64): pixels are frozen at 64-entry batch boundaries once T < t_min."""
    O = oracle
    n, W, H, deg = 4000, 64, 48, 1
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 9)
    sc["scales"] += 1.2                                            # bigger footprints -> long lists
    pre = O.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam)
    pre2 = NP.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, T, P, cam.fx, cam.fy,
                         cam.eye, cam.lookAt, W, H)
    gx, gy = W // 16, H // 16
    r, i, k = O.bin_lists(pre["bbs"], pre["tps"], 1, 16, gx, gy)
    assert int((r[:, 1] - r[:, 0]).min()) > 3 * 64
    lit, lit_t = O.composite_forward(pre, r, i, ocam, 16, gx, gy, t_min=0.0)
    for t_min in (0.2, 1e-3):
        img, tr = O.composite_forward(pre, r, i, ocam, 16, gx, gy, t_min=t_min)
        img2, tr2 = NP.composite_forward(pre2, r, i, cam.near, cam.far, W, H, 16, gx, gy, t_min=t_min)
        assert np.array_equal(img, img2) and np.array_equal(tr, tr2)
        assert not np.array_equal(tr, lit_t)                        # the rule did cut something
        assert np.abs(img - lit).max() <= t_min * 4.0               # what is cut is bounded by t_min * max|rgb|
        assert (tr[tr != lit_t] < t_min).all()                      # frozen pixels stopped below the threshold


def test_expf_spec_matches_and_is_accurate(oracle):
    xs = np.concatenate([np.linspace(-100, 100, 4001), [0.0, -0.0, 88.72283, 88.7229, -87.33654, -87.3366, np.nan, np.inf, -np.inf]]).astype(np.float32)
    c = np.array([oracle.expf(float(x)) for x in xs], np.float32)
    p = NP.expf_spec(xs)
    assert np.array_equal(c, p, equal_nan=True)
    ref = np.exp(xs.astype(np.float64))
    ok = np.isfinite(ref) & (ref > 1.2e-38) & (ref < 3.4e38)
    assert np.max(np.abs(p[ok] - ref[ok]) / ref[ok]) < 1.2e-7            # < 1 ulp
    assert p[np.isnan(xs)].size == 1 and np.isnan(p[np.isnan(xs)]).all()
    assert np.isinf(p[xs == np.inf]).all() and (p[xs == -np.inf] == 0).all()


def test_dense_literal_lists_equal_sparse_index_order(oracle):
    """binning.jl hits -> scan!(dims=3) -> compact.jl (literal, dense) gives per-tile lists in
    gaussian-INDEX order regardless of sortIdxs; the sparse builder must reproduce them."""
    O = oracle
    n, W, H, deg = 700, 96, 64, 0
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 11)
    pre = O.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam)
    gx, gy = W // 16, H // 16
    hit, max_hits = O.bin_dense_literal(pre["bbs"], 16, gx, gy)
    ranges, ids, _ = O.bin_lists(pre["bbs"], pre["tps"], O.ORDER_INDEX, 16, gx, gy)
    assert max_hits == int((ranges[:, 1] - ranges[:, 0]).max())
    for t in range(gx * gy):
        col = hit[:, t // gx, t % gx]
        lst = col[col != 0] - 1
        assert np.array_equal(lst, ids[ranges[t, 0]:ranges[t, 1]])


def test_openmp_native_build_is_bit_identical_to_the_serial_build(oracle):
    """The bench's cpu_baseline leg runs libgs_oracle_omp.so (-O3 -march=native -fopenmp, oracle/Makefile): same source,
    -ffp-contract=off, lists built by threads that own bands of tile rows.  Every output must equal the serial -O2 build
    whatever the thread count (ragged grid: more threads than tile rows included)."""
    import ctypes as C
    import os
    O = oracle
    if os.environ.get("GS_ORACLE_LIB"):
        pytest.skip("one library forced through GS_ORACLE_LIB (sanitizer run)")
    n, W, H, deg = 6000, 200, 72, 2
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 21)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    dC = np.random.default_rng(3).standard_normal((3, H, W)).astype(np.float32)
    omp = C.CDLL("libgomp.so.1")

    def run(par, threads):
        if par:
            omp.omp_set_num_threads(threads)
        pre = O.preprocess(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, omp=par)
        perm = O.depth_order(pre["tps"], 1)
        L = O.lib(par)
        fp, u32, u64 = (lambda a, t=t: a.ctypes.data_as(C.POINTER(t)) for t in (C.c_float, C.c_uint32, C.c_uint64))
        tot = L.gso_bin(n, fp(pre["bbs"]), fp(pre["tps"]), u32(perm), 1, 16, gx, gy, None, None, None, 0)
        rg = np.zeros((gx * gy, 2), np.uint32); ids = np.zeros(tot, np.uint32); keys = np.zeros(tot, np.uint64)
        L.gso_bin(n, fp(pre["bbs"]), fp(pre["tps"]), u32(perm), 1, 16, gx, gy, u32(rg), u32(ids), u64(keys), tot)
        img, tr = O.composite_forward(pre, rg, ids, ocam, 16, gx, gy, t_min=1e-4, omp=par)
        return pre, rg, ids, keys, img, tr

    ref = run(False, 1)
    for threads in (1, 3, 8, 13):
        got = run(True, threads)
        for k in ref[0]:
            assert np.array_equal(ref[0][k], got[0][k], equal_nan=True), (threads, k)
        for a, b in zip(ref[1:], got[1:]):
            assert np.array_equal(a, b, equal_nan=True), threads
    omp.omp_set_num_threads(os.cpu_count() or 1)
