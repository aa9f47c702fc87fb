"""The Python mirror of the reference interface on the GPU (pytest -m gpu), plus edge cases."""
import numpy as np
import pytest

from common import hip_context, rel_l2, scene_and_cameras

pytestmark = pytest.mark.gpu


def test_reference_call_sequence_and_grad_accumulation_over_views(oracle):
    import torch
    from gaussiansplat_amd import distributed as D, renderer as R, synthetic
    from gaussiansplat_amd import camera as gcam
    O = oracle
    n, W, H, deg = 3000, 160, 112, 3
    gx, gy = (W + 15) // 16, (H + 15) // 16
    scene = synthetic.make_scene(n, W, H, deg, seed=3)
    r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, t_min=0.0)
    cams = [synthetic.scene_camera(W, view=v) for v in (0, 1)]
    dCs = [synthetic.make_dC(W, H, 50 + v) for v in (0, 1)]
    # main.jl:32-34 sequence, one view
    tps = R.preprocess(r, cams[0]); R.compactIdxs(r, (16, 16), (gx, gy)); R.forward(r, tps, (16, 16), (gx, gy))
    torch.cuda.synchronize()
    want = []
    for cam in cams:
        ocam = O.camera_from_arrays(gcam.compute_transform(cam), gcam.compute_projection(cam, W, H), np.float32(cam.fx), np.float32(cam.fy),
                                    np.float32(cam.near), np.float32(cam.far), cam.eye, cam.lookAt, W, H)
        want.append((ocam, O.render(scene["means"], scene["scales"], scene["quats"], scene["opacities"], scene["shs"], deg, ocam, t_min=0.0)))
    img = r.imageData.cpu().numpy()
    assert np.all(np.abs(img - want[0][1]["image"]) <= 1e-4 + 1e-4 * np.abs(want[0][1]["image"]))
    assert np.all(np.abs(r.transmittance.cpu().numpy() - want[0][1]["trans"]) <= 1e-4 + 1e-4 * want[0][1]["trans"])
    # two views through the data-parallel step (world 1): gradients accumulate, then resetGrads zeroes
    flat = D.multi_view_step(D.HipViewRenderer(r), cams, dCs)
    torch.cuda.synchronize()
    tot = None
    for (ocam, ref), dC in zip(want, dCs):
        g = O.backward(scene["means"], scene["scales"], scene["quats"], scene["opacities"], scene["shs"], deg, ocam, ref["ranges"], ref["ids"], dC)
        v = np.concatenate([g[k].reshape(-1) for k in ("means", "scales", "quats", "opacities", "shs")])
        tot = v if tot is None else tot + v
    assert rel_l2(flat.cpu().numpy(), tot) <= 1e-3
    assert rel_l2(r.splatGrads.Δshs.cpu().numpy().reshape(-1), tot[11 * n:]) <= 1e-3
    R.resetGrads(r)
    torch.cuda.synchronize()
    assert float(r.splatGrads.flat.abs().max()) == 0.0


def test_library_owned_grads_and_host_dC(oracle):
    """The Julia-style path: no torch, host dC, gs_grads_alloc / gs_grads_read."""
    from gaussiansplat_amd import synthetic
    O = oracle
    n, W, H, deg = 2000, 96, 80, 1
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 12)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=0.0)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    grads = ctx.grads_alloc()
    dC = synthetic.make_dC(W, H, 12)
    ctx.backward(dC, grads)
    got = ctx.grads_read(grads, deg)
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, t_min=0.0)
    g = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC)
    for k in ("means", "scales", "quats", "opacities", "shs"):
        assert rel_l2(got[k].reshape(-1), g[k].reshape(-1)) <= 1e-3, k
    ctx.close()


def test_error_behaviour():
    from gaussiansplat_amd import backend as B
    ctx = B.Context()
    with pytest.raises(B.GsError):
        ctx.preprocess()                       # no camera yet
    with pytest.raises(B.GsError):
        ctx.bin()
    with pytest.raises(B.GsError):
        B.Context(order=7)
    ctx.close()


@pytest.mark.parametrize("n", [0, 1, 31, 33])
def test_tiny_and_empty_models(oracle, n):
    """The reference launches div(n,32) blocks and silently drops the tail (forward.jl:72); the build must not."""
    O = oracle
    W, H, deg = 64, 48, 1
    sc, cam, T, P, ocam = scene_and_cameras(max(n, 1), W, H, deg, 5)
    sc = {k: v[:n] for k, v in sc.items()}
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=0.0)
    ctx.preprocess(); ctx.bin()
    img, tr = ctx.forward_host()
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, t_min=0.0)
    assert ctx.num_instances == len(ref["ids"])
    assert np.all(np.abs(img - ref["image"]) <= 1e-4 + 1e-4 * np.abs(ref["image"]))
    assert np.array_equal(tr == 1.0, ref["trans"] == 1.0)
    ctx.close()


def test_edge_gaussians_offscreen_huge_nan_and_depth_cull(oracle):
    from gaussiansplat_amd import backend as B
    O = oracle
    W, H, deg = 80, 72, 0
    sc, cam, T, P, ocam = scene_and_cameras(64, W, H, deg, 6)
    m = sc["means"]
    m[0] = [1e3, 0, 0]                 # far off-screen: touches no tile
    sc["scales"][1] = 1.5              # huge: covers the whole screen
    m[2] = [np.nan, 0, 0]              # NaN mean -> non-finite box -> dropped (reference would throw)
    m[3] = [1.0, 3.0, 29.95]           # just in front of the eye: clip z < near -> skipped (splat.jl:227)
    m[4] = [1.0, 3.0, 31.0]            # behind the camera
    sc["opacities"][5] = 40.0          # sigmoid saturates to exactly 1.0f (alpha can reach 1)
    sc["opacities"][6] = -120.0        # exp underflow -> alpha 0
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=0.0, export_debug=True)
    ctx.preprocess(); ctx.bin()
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, t_min=0.0)
    assert np.array_equal(ctx.get_array(B.ARR_BBS), ref["pre"]["bbs"], equal_nan=True)
    assert np.array_equal(ctx.get_array(B.ARR_TILE_RANGES), ref["ranges"])
    assert np.array_equal(ctx.get_array(B.ARR_SORTED_KEYS), ref["keys"])
    rect = ctx.get_array(B.ARR_TILE_RECT)
    assert rect[0, 0] == 0 and rect[2, 0] == 0 and tuple(rect[1]) == (1, (W + 15) // 16, 1, (H + 15) // 16)
    img, tr = ctx.forward_host()
    assert np.isfinite(img).all()
    assert np.all(np.abs(img - ref["image"]) <= 1e-4 + 1e-4 * np.abs(ref["image"]))
    assert np.all(np.abs(tr - ref["trans"]) <= 1e-4)
    ctx.close()


def test_deterministic_gradients_are_bitwise_reproducible(oracle):
    """gs_config.deterministic: fixed-point integer atomics -> identical bits run to run, still within
    tolerance of the fp64 adjoint; the float-atomics path is only required to be close."""
    from gaussiansplat_amd import synthetic
    O = oracle
    n, W, H, deg = 6000, 160, 128, 3
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 23)
    dC = synthetic.make_dC(W, H, 23)
    runs = []
    for rep in range(3):
        ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, deterministic=True)
        ctx.preprocess(); ctx.bin(); ctx.forward_host()
        g = ctx.grads_alloc(); ctx.backward(dC, g)
        runs.append(ctx.grads_read(g, deg))
        ctx.close()
    for k in runs[0]:
        assert np.array_equal(runs[0][k], runs[1][k]) and np.array_equal(runs[0][k], runs[2][k]), k
    ref = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, t_min=1e-5)
    gref = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, ref["ranges"], ref["ids"], dC, t_min=1e-5)
    for k in ("means", "scales", "quats", "opacities", "shs"):
        assert rel_l2(runs[0][k].reshape(-1), gref[k].reshape(-1)) <= 1e-3, k


def test_rccl_allreduce_through_c_abi_single_rank():
    """gs_comm_* / gs_allreduce_grads with a 1-rank RCCL communicator: the sum over one rank is the identity,
    issued as ONE collective on the flat buffer of gs_grads_alloc.  (N > 1 needs N GPUs: the driver's run.)"""
    from gaussiansplat_amd import backend as B
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 2000, 96, 64, 1
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 31)
    ctx = hip_context(sc, cam, T, P, W, H, deg)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    g = ctx.grads_alloc()
    ctx.backward(synthetic.make_dC(W, H, 31), g)
    before = ctx.grads_read(g, deg)
    ctx.comm_init(0, 1, B.Context.comm_unique_id())
    ctx.allreduce_grads(g)
    ctx.synchronize()
    after = ctx.grads_read(g, deg)
    for k in before:
        assert np.array_equal(before[k], after[k]), k
    ctx.close()


def test_colour_factored_exchange_equals_plain_accumulation(oracle):
    """distributed.multi_view_step(sync="factored") on one GPU (no process group): geometry gradients accumulate over
    the views, d rgb of every view is packed, and gs_sh_grads_from_views rebuilds the SH gradients -- the flat buffer
    must equal the plain accumulation over the same views, and (for one view) the fp64 oracle."""
    import torch
    from gaussiansplat_amd import distributed as D, renderer as R, synthetic
    n, W, H, deg = 3000, 160, 112, 3
    scene = synthetic.make_scene(n, W, H, deg, seed=61)
    cams = [synthetic.scene_camera(W, view=v) for v in range(3)]
    dCs = [synthetic.make_dC(W, H, 200 + v) for v in range(3)]
    flats = {}
    for sync in ("allreduce", "factored"):
        r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), None, scene, t_min=0.0, deterministic=True)   # bitwise reproducible sums
        flats[sync] = D.multi_view_step(D.HipViewRenderer(r), cams, dCs, sync=sync).clone()
        torch.cuda.synchronize()
    a, b = flats["allreduce"].cpu().numpy().astype(np.float64), flats["factored"].cpu().numpy().astype(np.float64)
    assert np.array_equal(a[:11 * n], b[:11 * n])                                   # geometry part: the same kernels
    assert np.linalg.norm(a[11 * n:] - b[11 * n:]) <= 2e-6 * np.linalg.norm(a[11 * n:])
    assert np.abs(a[11 * n:]).max() > 0
    # deterministic mode and accumulate-into (overwrite = False) through the raw ABI
    from common import hip_context, scene_and_cameras
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 61)
    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=0.0, deterministic=True)
    ctx.preprocess(); ctx.bin(); ctx.forward_host()
    g = ctx.grads_alloc()
    ctx.backward(dCs[0], g)
    want = ctx.grads_read(g, deg)["shs"].astype(np.float64)
    drgb = torch.empty((1, n, 3), dtype=torch.float32, device="cuda")
    ctx.color_grads_pack(drgb.data_ptr())
    out = torch.full((n, 3 * (deg + 1) ** 2), 1.0, dtype=torch.float32, device="cuda")
    ctx.sh_grads_from_views(D.view_records([cam], W, H), drgb.data_ptr(), out.data_ptr(), overwrite=False)
    ctx.synchronize()
    got = out.cpu().numpy().astype(np.float64) - 1.0
    assert np.linalg.norm(got - want) <= 1e-5 * np.linalg.norm(want)
    ctx.close()


def test_more_than_2_32_instances_is_refused_not_corrupted():
    """140 k screen-filling gaussians on a 4K tile grid = 4.5e9 tile instances: the 32-bit list offsets cannot hold
    that, gs_bin must say so (GS_ERR_UNSUPPORTED) instead of wrapping."""
    from common import hip_context, scene_and_cameras
    from gaussiansplat_amd import backend as B
    n, W, H, deg = 140_000, 3840, 2160, 0
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 5)
    sc["scales"] = np.full_like(sc["scales"], 3.0)
    ctx = hip_context(sc, cam, T, P, W, H, deg)
    ctx.preprocess()
    with pytest.raises(B.GsError) as e:
        ctx.bin()
    assert e.value.code == -5 and "tile instances" in str(e.value)
    with pytest.raises(B.GsError):
        ctx.forward_host()                       # nothing to draw: the frame has no lists
    ctx.close()


def test_plain_c_host_without_python_or_torch(tmp_path):
    """examples/render_c.c: the C ABI driven from a C program linked against /opt/rocm's HIP runtime only (what a Julia
    ccall host sees): must build with gcc, run a few fwd+bwd frames and exit 0 with finite checksums."""
    import os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "render_c")
    libdir = os.path.join(root, "gaussiansplat_amd", "lib")
    subprocess.run([shutil.which("gcc") or "gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "render_c.c"),
                    "-o", exe, "-L" + libdir, "-lgsplat_hip", "-lm", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe, "20000", "320", "208", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "render_c ok" in r.stdout and "instances=" in r.stdout


def test_split_backward_and_overlapped_factored_step_equal_plain_backward(oracle):
    """GS_BWD_COMPOSITE_ONLY + GS_BWD_PARAMS_ONLY == one gs_backward; distributed.factored_one_view_step (what bench.py
    runs per rank for N > 1; without a process group here) == the plain step's flat gradient buffer."""
    import torch
    from gaussiansplat_amd import backend as B, distributed as D, renderer as R, synthetic
    n, W, H, deg = 2500, 144, 96, 3
    scene = synthetic.make_scene(n, W, H, deg, seed=71)
    cam = synthetic.scene_camera(W, view=3)
    dC = torch.as_tensor(synthetic.make_dC(W, H, 71)).cuda()
    out = {}
    for mode in ("plain", "split", "factored"):
        r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), None, scene, t_min=1e-5, deterministic=True)
        R.resetGrads(r)
        if mode == "factored":
            hv = D.HipViewRenderer(r)
            gathered = torch.empty(3 * n, dtype=torch.float32, device="cuda")
            D.factored_one_view_step(hv, cam, dC, D.view_records([cam], W, H), gathered)
        else:
            tps = R.preprocess(r, cam); R.compactIdxs(r); R.forward(r, tps)
            if mode == "plain":
                R.backward(r, dC)
            else:
                with pytest.raises(B.GsError):
                    R.backward(r, dC, phase="params")            # needs the composite phase first
                R.backward(r, dC, phase="composite")
                R.backward(r, dC, phase="params")
        torch.cuda.synchronize()
        out[mode] = r.splatGrads.flat.cpu().numpy().astype(np.float64)
    assert np.array_equal(out["plain"], out["split"])
    assert np.array_equal(out["plain"][:11 * n], out["factored"][:11 * n])
    assert np.linalg.norm(out["plain"][11 * n:] - out["factored"][11 * n:]) <= 2e-6 * np.linalg.norm(out["plain"][11 * n:])


def test_bound_output_buffers(oracle):
    """gs_bind_outputs: the forward writes image / transmittance straight into caller-owned device buffers and the backward
    reads them back from there -- same bits as the ctx-owned buffers + copy, for the forward and (deterministic mode) the
    gradients; other destinations still get a copy; unbinding restores the ctx's own buffers."""
    import torch
    from gaussiansplat_amd import synthetic
    n, W, H, deg = 5000, 208, 144, 2
    sc, cam, T, P, ocam = scene_and_cameras(n, W, H, deg, 11)
    dC = torch.as_tensor(synthetic.make_dC(W, H, 4)).cuda()

    def run(ctx, img, tr, bind):
        if bind:
            ctx.bind_outputs(img.data_ptr(), tr.data_ptr())
        ctx.preprocess(); ctx.bin()
        ctx.forward_device(img.data_ptr(), tr.data_ptr())
        g = ctx.grads_alloc()
        ctx.backward(dC.data_ptr(), g, overwrite=True)
        ctx.synchronize()
        return img.cpu().numpy().copy(), tr.cpu().numpy().copy(), {k: v.copy() for k, v in ctx.grads_read(g, deg).items()}

    ctx = hip_context(sc, cam, T, P, W, H, deg, t_min=1e-5, deterministic=True)
    i0, t0 = torch.zeros((3, H, W), device="cuda"), torch.zeros((H, W), device="cuda")
    ref = run(ctx, i0, t0, bind=False)
    i1, t1 = torch.full((3, H, W), 7.0, device="cuda"), torch.full((H, W), 7.0, device="cuda")
    got = run(ctx, i1, t1, bind=True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    for k in ref[2]:
        assert np.array_equal(got[2][k], ref[2][k]), k
    # another destination while bound: it receives a copy, the bound buffers hold the result as well
    i2, t2 = torch.zeros((3, H, W), device="cuda"), torch.zeros((H, W), device="cuda")
    ctx.preprocess(); ctx.bin(); ctx.forward_device(i2.data_ptr(), t2.data_ptr()); ctx.synchronize()
    assert np.array_equal(i2.cpu().numpy(), ref[0]) and np.array_equal(i1.cpu().numpy(), ref[0])
    # a frame whose backward must read the BOUND transmittance: scribble over the ctx-independent copy first
    i2.zero_(); t2.zero_()
    g = ctx.grads_alloc(); ctx.backward(dC.data_ptr(), g, overwrite=True); ctx.synchronize()
    again = ctx.grads_read(g, deg)
    for k in ref[2]:
        assert np.array_equal(again[k], ref[2][k]), k
    # unbind: back to the ctx-owned buffers
    ctx.bind_outputs(0, 0)
    i3, t3 = torch.zeros((3, H, W), device="cuda"), torch.zeros((H, W), device="cuda")
    back = run(ctx, i3, t3, bind=False)
    assert np.array_equal(back[0], ref[0]) and np.array_equal(back[1], ref[1])
    with pytest.raises(Exception):
        ctx.bind_outputs(i3.data_ptr(), 0)                          # both or neither
    ctx.close()
