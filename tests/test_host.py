"""Host-side mirror of the reference interface: camera, synthetic scenes, loaders (CPU only)."""
import json

import numpy as np

from gaussiansplat_amd import camera as gcam
from gaussiansplat_amd import synthetic


def test_default_camera_matches_reference_literals():
    c = gcam.default_camera()                          # camera.jl:24-47
    assert list(c.eye) == [1.0, 3.0, 30.0] and list(c.lookAt) == [0, 0, 0] and list(c.up) == [0, 1, 0]
    assert (c.fx, c.fy, c.near, c.far) == (3200.0, 3200.0, 0.1, 100.0)


def test_get_camera_from_cameras_json(tmp_path):
    rot = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    entry = dict(id=3, img_name="im0", width=640, height=480, position=[1.0, 2.0, 3.0], rotation=rot, fx=500.0, fy=510.0)
    p = tmp_path / "cameras.json"
    p.write_text(json.dumps([entry]))
    c = gcam.get_camera(str(p), 1)                     # 1-based like the reference (camera.jl:119)
    assert np.allclose(c.eye, [-1, -2, -3]) and np.allclose(c.lookAt, [0, 0, -1])   # camera.jl:130-131
    assert (c.fx, c.fy, c.near, c.far, c.id, c.data) == (500.0, 510.0, 0.010, 100.0, 3, "im0")


def test_synthetic_scene_is_deterministic_and_shaped():
    a = synthetic.make_scene(1000, 256, 256, 3, seed=1234)
    b = synthetic.make_scene(1000, 256, 256, 3, seed=1234)
    for k in a:
        assert np.array_equal(a[k], b[k]) and a[k].dtype == np.float32
    assert a["shs"].shape == (1000, 16, 3) and a["quats"].shape == (1000, 4)
    assert np.allclose(np.linalg.norm(a["quats"], axis=1), 1, atol=1e-6)
    assert (a["scales"] >= -4.5).all() and (a["scales"] <= -2.5).all()
    assert (a["opacities"] >= -2).all() and (a["opacities"] <= 4).all()
    c0 = synthetic.scene_camera(1920, 0); c2 = synthetic.scene_camera(1920, 2)
    assert c0.fx == 3200.0 and np.allclose(c0.eye, [1, 3, 30])
    assert np.isclose(np.linalg.norm(c2.eye), np.linalg.norm(c0.eye), rtol=1e-6) and np.isclose(c2.eye[1], 3.0)


def test_ply_round_trip_reference_field_mapping(tmp_path):
    from gaussiansplat_amd import ply
    sc = synthetic.make_scene(257, 128, 128, 3, seed=5)
    p = str(tmp_path / "scene.ply")
    ply.save_ply(p, sc)
    d1 = ply.load_ply(p)                                  # reference mapping: f_dc + f_rest[0:9] -> 12 floats
    assert d1["shs"].shape == (257, 4, 3)
    assert np.array_equal(d1["shs"].reshape(257, -1), sc["shs"].reshape(257, -1)[:, :12])
    d3 = ply.load_ply(p, sh_degree=3)
    for k in ("means", "scales", "quats", "opacities", "shs"):
        assert np.array_equal(d3[k], sc[k]), k


def test_export_image_like_reference_viewer(tmp_path):
    from gaussiansplat_amd import export
    img = np.zeros((3, 2, 4), np.float32)            # C=3, H=2, W=4
    img[0, 0, 3] = 2.0                               # clamps to 1
    img[1, 1, 0] = np.nan                            # scrubbed to 0
    img[2, 1, 2] = 0.5
    out = export.to_rgb8(img, rotate=False)
    assert out.shape == (4, 2, 3) and out.dtype == np.uint8        # Julia: W rows, H columns
    assert out[3, 0, 0] == 255 and out[0, 1, 1] == 0 and out[2, 1, 2] == 128
    rot = export.to_rgb8(img)
    assert rot.shape == (2, 4, 3) and np.array_equal(rot, np.rot90(out, 1, (0, 1)))
    p = tmp_path / "a.ppm"
    export.save_ppm(str(p), rot)
    assert p.read_bytes().startswith(b"P6\n4 2\n255\n") and len(p.read_bytes()) == 11 + 24


def test_bench_self_launch_command():
    """`python bench.py --gpus N` from a bare shell starts N ranks as CHILD processes of torch.distributed.run, before
    anything imports torch or touches HIP (GS_BENCH_DRY_LAUNCH shows the command instead of running it)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["GS_BENCH_DRY_LAUNCH"] = "1"
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '4', '--steps', '7', '--backend', 'gloo'];"
            "runpy.run_path(%r, run_name='__main__'); assert 'torch' not in sys.modules, 'torch imported before the launch'"
            % os.path.join(root, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--backend", "gloo"] and cmd[-7].endswith("bench.py")


def test_view_record_is_kept_with_the_camera_and_follows_its_fields():
    """renderer._set_view (the host side of preprocess(renderer, camera), forward.jl:35-53): a small frame is bound by its caller, so the
    view's matrices (computeTransform / computeProjection, camera.jl:53-111) are built once per camera and again whenever one of its
    fields -- or the image size -- changes, in place or by assignment."""
    from gaussiansplat_amd import renderer as R

    class Ctx:                                          # what _set_view needs of backend.Context
        def __init__(self): self.built, self.set, self.slots = [], [], []
        def camera_record(self, T, P, fx, fy, near, far, eye, lookAt, W, H):
            rec = (np.array(T, np.float32), np.array(P, np.float32), fx, fy, near, far, np.array(eye, np.float32), np.array(lookAt, np.float32), W, H)
            self.built.append(rec); return rec
        def set_camera_record(self, rec): self.set.append(rec)
        def set_view_slot(self, s): self.slots.append(s)

    class Fake:                                         # the two fields of GaussianRenderer3D it touches
        def __init__(self, W, H): self.ctx, self.camera, self.transmittance = Ctx(), None, np.ones((H, W), np.float32)
    f = Fake(64, 48)
    a, b = synthetic.scene_camera(64, view=0), synthetic.scene_camera(64, view=4)
    a.id, b.id = 0, 4
    for cam in (a, b, a, b, a):
        R.GaussianRenderer3D._set_view(f, cam)
    assert len(f.ctx.built) == 2 and len(f.ctx.set) == 5 and f.ctx.slots == [0, 4, 0, 4, 0]
    assert f.ctx.set[0] is f.ctx.set[2] and f.ctx.set[1] is f.ctx.set[3]
    assert np.array_equal(f.ctx.built[0][0], gcam.compute_transform(a)) and np.array_equal(f.ctx.built[1][1], gcam.compute_projection(b, 64, 48))
    a.eye[0] += np.float32(0.5)                          # in place
    R.GaussianRenderer3D._set_view(f, a)
    assert len(f.ctx.built) == 3 and np.array_equal(f.ctx.built[2][0], gcam.compute_transform(a))
    b.fx = b.fx * 1.25                                  # by assignment
    R.GaussianRenderer3D._set_view(f, b)
    assert len(f.ctx.built) == 4 and np.array_equal(f.ctx.built[3][1], gcam.compute_projection(b, 64, 48))
    g = Fake(80, 48)                                    # another image size: the same camera object is rebuilt for it
    R.GaussianRenderer3D._set_view(g, a)
    assert len(g.ctx.built) == 1 and g.ctx.built[0][8:] == (80, 48)
    R.GaussianRenderer3D._set_view(f, a)
    assert len(f.ctx.built) == 5                        # (one record per camera: the other size pushed it out)
