"""world_size-2 gloo test of the multi-view path (CPU): view sharding + ONE flat all-reduce must
equal the single-process sum over all views.  The per-view renderer here is the oracle (tests may
use it); on the GPU the same multi_view_step drives HipViewRenderer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussiansplat_amd import camera as gcam
from gaussiansplat_amd import distributed as D
from gaussiansplat_amd import synthetic

N, W, H, DEG, VIEWS = 300, 64, 48, 1, 4


class OracleViewRenderer:
    def __init__(self):
        from oracle import oracle as O
        self.O = O
        self.sc = synthetic.make_scene(N, W, H, DEG, seed=77)
        k3 = 3 * (DEG + 1) ** 2
        self.flat = torch.zeros(N * (11 + k3), dtype=torch.float64)

    def reset(self):
        self.flat.zero_()

    def render_view(self, cam, dC):
        O, sc = self.O, self.sc
        ocam = O.camera_from_arrays(gcam.compute_transform(cam), gcam.compute_projection(cam, W, H), np.float32(cam.fx), np.float32(cam.fy),
                                    np.float32(cam.near), np.float32(cam.far), cam.eye, cam.lookAt, W, H)
        r = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], DEG, ocam)
        g = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], DEG, ocam, r["ranges"], r["ids"], dC)
        self.flat += torch.from_numpy(np.concatenate([g[k].reshape(-1) for k in ("means", "scales", "quats", "opacities", "shs")]))


def _views():
    cams = [synthetic.scene_camera(W, view=v) for v in range(VIEWS)]
    dCs = [synthetic.make_dC(W, H, 100 + v) for v in range(VIEWS)]
    return cams, dCs


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, dCs = _views()
        flat = D.multi_view_step(OracleViewRenderer(), cams, dCs)
        if rank == 0:
            np.save(out, flat.numpy())
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)       # identical on every rank
    finally:
        dist.destroy_process_group()


def test_shard_views_partitions_exactly():
    for world in (1, 2, 3, 4, 8):
        got = sum((D.shard_views(8, world, r) for r in range(world)), [])
        assert got == list(range(8))
    assert D.shard_views(8, 8, 5) == [5] and D.shard_views(8, 2, 1) == [4, 5, 6, 7]


@pytest.mark.timeout(300)
def test_two_rank_gloo_allreduce_equals_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "flat.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    cams, dCs = _views()
    single = OracleViewRenderer()
    D.multi_view_step(single, cams, dCs)                                 # world 1: plain sum over 4 views
    got = np.load(out)
    assert np.allclose(got, single.flat.numpy(), rtol=1e-12, atol=1e-14)
    assert np.abs(got).max() > 0
