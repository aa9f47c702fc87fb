"""Committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU oracle).

They are regression vectors of the build's own numeric spec -- the reference ships no tests or fixtures, parity stays
unpinned by it.  CPU: the C oracle and its NumPy twin must reproduce every stored fp32 / integer
output bit for bit.  GPU: the HIP path must reproduce the integer outputs bit for bit and the floats within the stated
bars, from the stored inputs alone."""
import glob
import os

import numpy as np
import pytest

from oracle import gs_oracle_np as ONP

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(HERE, "*.npz")))


def _load(path):
    return {k: v for k, v in np.load(path, allow_pickle=False).items()}


def _ocam(O, d):
    return O.camera_from_arrays(d["T"], d["P"], d["fx"], d["fy"], d["near"], d["far"], d["eye"], d["lookAt"], int(d["W"]), int(d["H"]))


def test_fixtures_exist():
    assert len(FILES) == 3


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_c_and_numpy_reproduce_golden(oracle, path):
    O = oracle
    d = _load(path)
    W, H = int(d["W"]), int(d["H"])
    gx, gy = (W + 15) // 16, (H + 15) // 16
    if str(d["kind"]) == "3d":
        deg, order, t_min = int(d["deg"]), int(d["order"]), float(d["t_min"])
        r = O.render(d["means"], d["scales"], d["quats"], d["opacities"], d["shs"], deg, _ocam(O, d), order=order, t_min=t_min)
        g = O.backward(d["means"], d["scales"], d["quats"], d["opacities"], d["shs"], deg, _ocam(O, d), r["ranges"], r["ids"], d["dC"], t_min=t_min)
        pre_np = ONP.preprocess(d["means"], d["scales"], d["quats"], d["opacities"], d["shs"], deg, d["T"], d["P"], d["fx"], d["fy"],
                                d["eye"], d["lookAt"], W, H)
        rn, idn, _ = ONP.bin_lists(pre_np["bbs"], pre_np["tps"][:, 2], order, 16, gx, gy)
        img_np, tr_np = ONP.composite_forward(pre_np, rn, idn, d["near"], d["far"], W, H, 16, gx, gy, t_min=t_min)
        gkeys = ("means", "scales", "quats", "opacities", "shs", "g2d")
    else:
        t_min = float(d["t_min"])
        r = O.render2d(d["means"], d["scales"], d["rots"], d["opacities"], d["colors"], W, H, t_min=t_min)
        g = O.backward2d(d["means"], d["scales"], d["rots"], d["opacities"], d["colors"], W, H, r["ranges"], r["ids"], d["dC"], t_min=t_min)
        pre_np = ONP.preprocess2d(d["means"], d["scales"], d["rots"], d["opacities"], d["colors"], W, H)
        rn, idn, _ = ONP.bin_lists(pre_np["bbs"], np.zeros(int(d["n"]), np.float32), ONP.ORDER_INDEX, 16, gx, gy)
        img_np, tr_np = ONP.composite_forward(pre_np, rn, idn, -1.0, 1.0, W, H, 16, gx, gy, t_min=t_min)
        gkeys = ("means", "scales", "rots", "opacities", "colors", "g2d")
    for k, v in r["pre"].items():
        assert np.ascontiguousarray(v).tobytes() == d["pre_" + k].tobytes(), ("C oracle", k)
        if k in pre_np:
            assert np.ascontiguousarray(pre_np[k], v.dtype).tobytes() == d["pre_" + k].tobytes(), ("NumPy oracle", k)
    for k in ("ranges", "ids", "keys", "image", "trans"):
        assert np.ascontiguousarray(r[k]).tobytes() == d[k].tobytes(), k
    assert np.array_equal(rn, d["ranges"]) and np.array_equal(idn, d["ids"])
    assert img_np.tobytes() == d["image"].tobytes() and tr_np.tobytes() == d["trans"].tobytes()
    for k in gkeys:                                  # fp64 adjoint: summation order inside libm / OpenMP-free build is fixed
        np.testing.assert_allclose(g[k], d["g_" + k], rtol=1e-12, atol=1e-14, err_msg=k)
    assert float(d["trans"].min()) < 0.95 and len(d["ids"]) > 100          # the fixture exercises the path


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_reproduces_golden(path):
    from gaussiansplat_amd import backend as B
    d = _load(path)
    W, H, n = int(d["W"]), int(d["H"]), int(d["n"])
    t_min = float(d["t_min"])
    if str(d["kind"]) == "3d":
        deg = int(d["deg"])
        ctx = B.Context(order=int(d["order"]), t_min=t_min, export_debug=True)
        ctx.set_model_host(d["means"], d["scales"], d["quats"], d["opacities"], d["shs"].reshape(n, -1), deg)
        ctx.set_camera(d["T"], d["P"], float(d["fx"]), float(d["fy"]), float(d["near"]), float(d["far"]), d["eye"], d["lookAt"], W, H)
        arrays = ((B.ARR_TS, "ts"), (B.ARR_TPS, "tps"), (B.ARR_MU, "mu"), (B.ARR_COV3D, "cov3d"), (B.ARR_COV2D, "cov2d"),
                  (B.ARR_INVCOV, "invcov"), (B.ARR_BBS, "bbs"), (B.ARR_RGB, "rgb"), (B.ARR_SIG, "sig"))
        gkeys = ("means", "scales", "quats", "opacities", "shs")
    else:
        ctx = B.Context(order=B.ORDER_INDEX, t_min=t_min, export_debug=True)
        ctx.set_model_2d_host(d["means"], d["scales"], d["rots"], d["opacities"], d["colors"])
        ctx.set_image_size(W, H)
        arrays = ((B.ARR_MU, "mu"), (B.ARR_COV2D, "cov2d"), (B.ARR_INVCOV, "invcov"), (B.ARR_BBS, "bbs"), (B.ARR_RGB, "rgb"), (B.ARR_SIG, "sig"))
        gkeys = ("means", "scales", "rots", "opacities", "colors")
    ctx.preprocess(); ctx.bin()
    for which, key in arrays:
        assert ctx.get_array(which).tobytes() == d["pre_" + key].tobytes(), key
    assert np.array_equal(ctx.get_array(B.ARR_TILE_RANGES), d["ranges"])
    assert np.array_equal(ctx.get_array(B.ARR_SORTED_IDS), d["ids"])
    assert np.array_equal(ctx.get_array(B.ARR_SORTED_KEYS), d["keys"])
    img, tr = ctx.forward_host()
    assert np.all(np.abs(img - d["image"]) <= 1e-4 + 1e-4 * np.abs(d["image"]))
    assert np.all(np.abs(tr - d["trans"]) <= 1e-4 + 1e-4 * np.abs(d["trans"]))
    g = ctx.grads_alloc()
    ctx.backward(d["dC"], g)
    got = ctx.grads_read(g, int(d["deg"])) if str(d["kind"]) == "3d" else ctx.grads_read_2d(g)
    for k in gkeys:
        a, b = got[k].reshape(-1).astype(np.float64), d["g_" + k].reshape(-1)
        assert np.linalg.norm(a - b) <= 1e-3 * max(np.linalg.norm(b), 1e-30), k
    ctx.close()
