"""SURVEY 8(f).1 on the GPU: the reference's entry point is getRenderer(:GAUSSIAN_3D, ..., plypath)
(src/examples/main.jl:14-27 -> src/renderer.jl:119-149 -> src/splat.jl:106-119) with a camera from cameras.json
(src/camera.jl:119-151).  These tests go PLY file + cameras.json -> getRenderer(path) -> preprocess / compactIdxs /
forward / backward on the HIP path and compare with the oracle fed from the same loaded arrays.
"""
import json

import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


def _write_inputs(tmp_path, n, W, H, seed, theta):
    from gaussiansplat_amd import ply, synthetic
    sc = synthetic.make_scene(n, W, H, 3, seed=seed)                         # a 3DGS file carries 45 f_rest columns
    ply_path = str(tmp_path / "point_cloud.ply")
    ply.save_ply(ply_path, sc)
    c, s = np.cos(theta), np.sin(theta)
    rot = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], np.float32)       # about +y
    eye = np.array([1.0, 3.0, 30.0], np.float32)
    # camera.jl:126-131: rotation = cat(rows..., dims=2) (the json rows become columns), eye = -rotation' * position
    position = (-(rot @ eye)).astype(np.float32)
    entry = dict(id=7, img_name="view_007", width=W, height=H, position=[float(x) for x in position],
                 rotation=[[float(x) for x in row] for row in rot.T], fx=float(3200.0 * W / 1920.0), fy=float(3100.0 * W / 1920.0))
    cam_path = str(tmp_path / "cameras.json")
    with open(cam_path, "w") as fh:
        json.dump([dict(entry, id=1, img_name="other", position=[0.0, 0.0, -5.0]), entry], fh)
    return sc, ply_path, cam_path


@pytest.mark.parametrize("sh_degree", [1, 3])
def test_renderer_from_ply_path_and_cameras_json(oracle, tmp_path, sh_degree):
    import torch
    from gaussiansplat_amd import backend as B, camera as gcam, ply, renderer as R, synthetic
    O = oracle
    n, W, H = 6001, 208, 144
    gx, gy = (W + 15) // 16, (H + 15) // 16
    sc, ply_path, cam_path = _write_inputs(tmp_path, n, W, H, 31, 0.04)
    cam = gcam.get_camera(cam_path, 2)                                       # 1-based like the reference (camera.jl:119)
    assert cam.id == 7 and cam.data == "view_007" and np.allclose(cam.eye, [1, 3, 30], atol=1e-4)
    assert abs(np.linalg.norm(cam.lookAt) - 1.0) < 1e-5                      # lookAt = -rotation' * [0,0,1] (camera.jl:131)
    if sh_degree == 1:
        # the reference's loader: shs = vcat(f_dc, f_rest[0:9]) = 12 floats (splat.jl:117) -> getRenderer(..., path)
        r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), ply_path, device=0, order=B.ORDER_DEPTH_DESC, t_min=0.0,
                          export_debug=True)
        data = ply.load_ply(ply_path)
        assert data["shs"].shape == (n, 4, 3)
        assert np.array_equal(data["shs"].reshape(n, 12), sc["shs"].reshape(n, 48)[:, :12])
    else:
        data = ply.load_ply(ply_path, sh_degree=3)                           # build extension: all 45 f_rest columns
        r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), data, device=0, order=B.ORDER_DEPTH_DESC, t_min=0.0,
                          export_debug=True)
    assert r.sh_degree == sh_degree and r.nGaussians == n
    for k in ("means", "scales", "quats", "opacities"):
        assert np.array_equal(data[k], sc[k]), k                             # splat.jl:110-116 field mapping
    tps = R.preprocess(r, cam)
    R.compactIdxs(r, (16, 16), (gx, gy))
    R.forward(r, tps, (16, 16), (gx, gy))
    dC = synthetic.make_dC(W, H, 32)
    R.backward(r, dC)
    torch.cuda.synchronize()
    ocam = O.camera_from_arrays(gcam.compute_transform(cam), gcam.compute_projection(cam, W, H), np.float32(cam.fx), np.float32(cam.fy),
                                np.float32(cam.near), np.float32(cam.far), cam.eye, cam.lookAt, W, H)
    ref = O.render(data["means"], data["scales"], data["quats"], data["opacities"], data["shs"], sh_degree, ocam, order=1, t_min=0.0)
    assert r.ctx.num_instances == len(ref["ids"]) > n
    assert np.array_equal(r.ctx.get_array(B.ARR_BBS), ref["pre"]["bbs"], equal_nan=True)
    assert np.array_equal(r.ctx.get_array(B.ARR_TPS), ref["pre"]["tps"], equal_nan=True)
    assert np.array_equal(r.ctx.get_array(B.ARR_RGB), ref["pre"]["rgb"], equal_nan=True)
    assert np.array_equal(r.ctx.get_array(B.ARR_TILE_RANGES), ref["ranges"])
    assert np.array_equal(r.ctx.get_array(B.ARR_SORTED_IDS), ref["ids"])
    assert np.array_equal(r.ctx.get_array(B.ARR_SORTED_KEYS), ref["keys"])
    img = r.imageData.cpu().numpy(); tr = r.transmittance.cpu().numpy()
    assert np.all(np.abs(img - ref["image"]) <= 1e-4 + 1e-4 * np.abs(ref["image"]))
    assert np.all(np.abs(tr - ref["trans"]) <= 1e-4 + 1e-4 * np.abs(ref["trans"]))
    assert float(tr.min()) < 0.5                                              # the camera really sees the scene
    g = O.backward(data["means"], data["scales"], data["quats"], data["opacities"], data["shs"], sh_degree, ocam, ref["ranges"], ref["ids"],
                   dC, t_min=0.0)
    for got, want in ((r.splatGrads.Δmeans, g["means"]), (r.splatGrads.Δscales, g["scales"]), (r.splatGrads.Δquaternions, g["quats"]),
                      (r.splatGrads.Δopacities, g["opacities"]), (r.splatGrads.Δshs, g["shs"])):
        assert rel_l2(got.cpu().numpy().reshape(-1), want.reshape(-1)) <= 1e-3
