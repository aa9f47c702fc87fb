"""Config C4 as SURVEY 8(e) defines it: a batch of eight camera views on 1 / 2 / 4 / 8 ranks, every rank rendering its
8 / N views one after the other with the gradients accumulating, then the sum over the ranks
(gaussiansplat_amd.distributed.multi_view_step, what `bench.py --config C4 --gpus N` times).

Two ranks are rehearsed here on ONE GPU over gloo (RCCL refuses two ranks on one device: tools/rccl_same_device_probe.py):
in deterministic mode the two-rank result must equal, bit for bit, what one process gets when it accumulates the views
of each rank's share and adds the two partial buffers -- the same association of the float additions -- whether the
buffer is reduced by ONE all-reduce or as its two segments with the Δshs segment started behind the SH kernel."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, W, H, DEG, VIEWS = 30_000, 400, 304, 3, 8


def _setup():
    import torch
    from gaussiansplat_amd import renderer as R, synthetic
    scene = synthetic.make_scene(N, W, H, DEG, seed=404)
    scene["scales"] = (scene["scales"] + np.float32(0.5)).astype(np.float32)
    cams = [synthetic.scene_camera(W, view=v) for v in range(VIEWS)]
    dCs = [torch.as_tensor(synthetic.make_dC(W, H, 500 + v)).cuda() for v in range(VIEWS)]
    # one wave per tile: the comparisons below are BIT for bit, across runs with and without view-slot history, and how a tile is shared
    # between waves (pixel parts, list segments: gs_config.tile_parts) follows from that history on this small grid
    r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), None, scene, t_min=1e-5, deterministic=True, tile_parts=1)
    return r, cams, dCs


def _worker(rank, world, port, overlap, pipeline, nviews, out):
    import torch
    import torch.distributed as dist
    from gaussiansplat_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r, cams, dCs = _setup()
        cams, dCs = cams[:nviews], dCs[:nviews]
        hv = D.HipViewRenderer(r)
        for _ in range(2):                                              # twice: the first step overwrites lazily (resetGrads), the second runs on view-slot history and speculative lists
            flat = D.multi_view_step(hv, cams, dCs, overlap=overlap, pipeline=pipeline)
        torch.cuda.synchronize()
        if rank == 0:
            np.save(out, flat.cpu().numpy())
        mine = flat.cpu()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert all(torch.equal(both[0], b) for b in both)               # identical on every rank
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap,pipeline,nviews", [
    (True, True, 8),       # four views per rank, pipelined: ONE all-reduce
    (False, True, 8),      # the same with overlap off: ONE all-reduce
    (True, False, 8),      # four views per rank one after the other, the LAST in two steps: Δshs segment all-reduced (async) behind the SH
                           # kernel beside the geometry chain, then the geometry segment -- the split path on GPU tensors
    (True, True, 2),       # one view per rank (the 8-GPU shape): the split path on a rank's ONLY view, incl. the lazy overwrite of step 1
    (True, True, 3),       # shares 2 + 1: one rank pipelines, the other splits -- both post the same two segment all-reduces
])
def test_two_ranks_equal_the_sequential_batch_bitwise(tmp_path, overlap, pipeline, nviews):
    import torch
    import torch.multiprocessing as mp
    from gaussiansplat_amd import distributed as D
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "flat2.npy")
    mp.spawn(_worker, args=(2, port, overlap, pipeline, nviews, out), nprocs=2, join=True)
    got = np.load(out)
    r, cams, dCs = _setup()
    cams, dCs = cams[:nviews], dCs[:nviews]
    hv = D.HipViewRenderer(r)
    parts = []
    for share in (D.shard_views(nviews, 2, 0), D.shard_views(nviews, 2, 1)):
        D.multi_view_step(hv, [cams[v] for v in share], [dCs[v] for v in share])       # world 1: accumulates the share's views in order
        torch.cuda.synchronize()
        parts.append(hv.flat.clone())
    want = (parts[0] + parts[1]).cpu().numpy()                                       # the two-rank sum: ONE float addition per element
    assert np.array_equal(got, want)
    D.multi_view_step(hv, cams, dCs, pipeline=False)                                 # all eight views in one process: another association
    torch.cuda.synchronize()
    seq32 = hv.flat.cpu().numpy().copy()
    seq = seq32.astype(np.float64)
    assert np.linalg.norm(got - seq) <= 1e-6 * np.linalg.norm(seq)
    # two renderers / two streams over the same model (pipeline=True) keep the chains in view order: the same bits as one after the other
    for _ in range(2):
        D.multi_view_step(hv, cams, dCs, pipeline=True)
        torch.cuda.synchronize()
        assert np.array_equal(hv.flat.cpu().numpy(), seq32)
    assert np.abs(got[11 * N:]).max() > 0 and np.abs(got[:11 * N]).max() > 0


def test_chain_in_two_steps_equals_one_backward():
    """GS_BWD_PARAMS_SH then GS_BWD_PARAMS_GEOM == GS_BWD_PARAMS_ONLY == one gs_backward, overwrite and accumulate."""
    import torch
    from gaussiansplat_amd import backend as B, renderer as R
    r, cams, dCs = _setup()
    out = {}
    for mode in ("plain", "two_step"):
        R.resetGrads(r)
        for v in (2, 5):                                                # first view overwrites (lazy reset), second accumulates
            tps = R.preprocess(r, cams[v]); R.compactIdxs(r); R.forward(r, tps)
            if mode == "plain":
                R.backward(r, dCs[v])
            else:
                R.backward(r, dCs[v], phase="composite")
                R.backward(r, dCs[v], phase="params_sh")
                R.backward(r, dCs[v], phase="params_geom")
        torch.cuda.synchronize()
        out[mode] = r.splatGrads.flat.cpu().numpy().copy()
    assert np.array_equal(out["plain"], out["two_step"])
    # the flags need GS_BWD_PARAMS_ONLY
    g = r._grads
    import ctypes as C
    rc = r.ctx.L.gs_backward_ex(r.ctx.h, C.c_void_p(dCs[0].data_ptr()), B.GS_MEM_DEVICE, C.byref(g), 8)
    assert rc == -1
