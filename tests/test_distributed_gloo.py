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


class OracleSplitRenderer(OracleViewRenderer):
    """The two-step last view of distributed.multi_view_step(overlap=True): Δshs is final after the SH step (its all-reduce
    is started there), the geometry part lands afterwards."""

    @property
    def geometry_floats(self):
        return 11 * N

    def render_view_until_sh(self, cam, dC):
        before = self.flat.clone()
        OracleViewRenderer.render_view(self, cam, dC)
        self._geo = self.flat[:11 * N] - before[:11 * N]
        self.flat[:11 * N] = before[:11 * N]                          # the geometry chain has not run yet

    def finish_geometry(self):
        self.flat[:11 * N] += self._geo


class OracleFactoredRenderer(OracleViewRenderer):
    """The colour-factored protocol of distributed.multi_view_step on the CPU: geometry gradients and d rgb per view
    from the oracle's adjoint, SH gradients rebuilt as sum_v basis(dir_v) (x) d rgb_v in NumPy (fp64)."""

    @property
    def geometry_floats(self):
        return 11 * N

    def color_slots(self, nviews):
        return torch.zeros((nviews, N, 3), dtype=torch.float64)

    def render_view_factored(self, cam, dC, slot):
        O, sc = self.O, self.sc
        ocam = O.camera_from_arrays(gcam.compute_transform(cam), gcam.compute_projection(cam, W, H), np.float32(cam.fx), np.float32(cam.fy),
                                    np.float32(cam.near), np.float32(cam.far), cam.eye, cam.lookAt, W, H)
        r = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], DEG, ocam)
        g = O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], DEG, ocam, r["ranges"], r["ids"], dC)
        self.flat[:11 * N] += torch.from_numpy(np.concatenate([g[k].reshape(-1) for k in ("means", "scales", "quats", "opacities")]))
        slot.copy_(torch.from_numpy(g["g2d"][:, :3]))

    def sh_from_views(self, cameras, drgb_all):
        from oracle import gs_oracle_np as ONP
        m = self.sc["means"].astype(np.float64)
        K = (DEG + 1) ** 2
        acc = np.zeros((N, K, 3))
        for cam, d in zip(cameras, drgb_all.numpy()):
            T = np.asarray(gcam.compute_transform(cam), np.float64).reshape(4, 4, order="F")
            P = np.asarray(gcam.compute_projection(cam, W, H), np.float64).reshape(4, 4, order="F")
            p = (P @ (T @ np.concatenate([m, np.ones((N, 1))], 1).T)).T
            v = p[:, :3] - (np.asarray(cam.lookAt, np.float64) - np.asarray(cam.eye, np.float64))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            B = np.stack([np.broadcast_to(np.asarray(b, np.float64), (N,)) for b in ONP.sh_basis(DEG, v[:, 0], v[:, 1], v[:, 2])], 1)   # [N, K]
            acc += B[:, :, None] * d[:, None, :]
        self.flat[11 * N:] = torch.from_numpy(acc.reshape(-1))


class OraclePipelinedSplitRenderer(OracleSplitRenderer):
    """What HipViewRenderer offers: a pipelined path for a rank with several views AND the two-step last view."""

    def render_views_pipelined(self, cams, dCs):
        for cam, dC in zip(cams, dCs):
            OracleViewRenderer.render_view(self, cam, dC)


def _views(nviews=VIEWS):
    cams = [synthetic.scene_camera(W, view=v) for v in range(nviews)]
    dCs = [synthetic.make_dC(W, H, 100 + v) for v in range(nviews)]
    return cams, dCs


def _worker_uneven(rank, world, port, out, nviews, kind):
    """unequal shares: every rank must post the same collectives whatever its own view count is"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, dCs = _views(nviews)
        r = {"plain": OracleViewRenderer, "split": OracleSplitRenderer, "pipelined": OraclePipelinedSplitRenderer}[kind]()
        flat = D.multi_view_step(r, cams, dCs, overlap=True, pipeline=True)
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        if rank == world - 1:
            np.save(out, flat.numpy())
    finally:
        dist.destroy_process_group()


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


def _worker_split(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, dCs = _views()
        flat = D.multi_view_step(OracleSplitRenderer(), cams, dCs, overlap=True)
        if rank == 1:
            np.save(out, flat.numpy())
    finally:
        dist.destroy_process_group()


def _worker_factored(rank, world, port, out, sync="factored"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cams, dCs = _views()
        flat = D.multi_view_step(OracleFactoredRenderer(), cams, dCs, sync=sync)
        if rank == 1:
            np.save(out, flat.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_factored_sync_equals_plain_sum(tmp_path):
    """all-reduce of the geometry part + all-gather of d rgb + local rebuild of the SH gradients == the plain sum."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "flat_f.npy")
    mp.spawn(_worker_factored, args=(2, port, out), nprocs=2, join=True)
    cams, dCs = _views()
    single = OracleViewRenderer()
    D.multi_view_step(single, cams, dCs)
    got, want = np.load(out), single.flat.numpy()
    assert np.allclose(got[:11 * N], want[:11 * N], rtol=1e-12, atol=1e-14)
    assert np.linalg.norm(got[11 * N:] - want[11 * N:]) <= 1e-6 * np.linalg.norm(want[11 * N:])     # fp32 camera matrices vs fp64 here
    one = OracleFactoredRenderer()
    D.multi_view_step(one, cams, dCs, sync="factored")                   # world 1 takes the same path without collectives
    assert np.allclose(one.flat.numpy(), got, rtol=1e-12, atol=1e-14)


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


@pytest.mark.timeout(300)
def test_two_rank_gloo_segment_allreduce_equals_one_allreduce(tmp_path):
    """overlap=True reduces the flat buffer as its two segments (Δshs started before the geometry part of the last view exists):
    the result must equal the single all-reduce of the whole buffer bit for bit (an all-reduce is element-wise)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out_a, out_b = str(tmp_path / "flat_split.npy"), str(tmp_path / "flat_one.npy")
    mp.spawn(_worker_split, args=(2, port, out_a), nprocs=2, join=True)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, out_b), nprocs=2, join=True)
    a, b = np.load(out_a), np.load(out_b)
    assert np.array_equal(a[11 * N:], b[11 * N:])                        # the Δshs segment: the same two-rank sums
    assert np.allclose(a[:11 * N], b[:11 * N], rtol=1e-13, atol=1e-15)   # the split renderer forms the geometry part as (x + g) - x + ...
    assert np.abs(a).max() > 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("nviews,kind", [(3, "split"), (3, "pipelined"), (3, "plain"), (1, "split"), (1, "pipelined")])
def test_two_rank_gloo_unequal_shares_post_matching_collectives(tmp_path, nviews, kind):
    """3 views on 2 ranks (shares 2 + 1: one rank would pipeline, the other split its only view) and fewer views than ranks
    (shares 1 + 0): ADVICE round 3 -- the old code chose its collectives from the rank's own share and gloo aborted with a size
    mismatch.  The result must be the plain sum over all views on every rank."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "flat_u.npy")
    mp.spawn(_worker_uneven, args=(2, port, out, nviews, kind), nprocs=2, join=True)
    cams, dCs = _views(nviews)
    single = OracleViewRenderer()
    D.multi_view_step(single, cams, dCs)
    got, want = np.load(out), single.flat.numpy()
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14)
    assert np.abs(got).max() > 0


@pytest.mark.timeout(300)
def test_two_rank_gloo_touched_rows_exchange_equals_factored_and_plain_sum(tmp_path):
    """the colour gradients of a view travel as a bitmap + the rows of the touched gaussians only.  Same result as the
    colour-factored exchange bit for bit (the receiver rebuilds the very same [views, N, 3] array), and as the plain sum to fp rounding."""
    outs = {}
    for sync in ("factored", "touched"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        outs[sync] = str(tmp_path / f"flat_{sync}.npy")
        mp.spawn(_worker_factored, args=(2, port, outs[sync], sync), nprocs=2, join=True)
    a, b = np.load(outs["factored"]), np.load(outs["touched"])
    assert np.array_equal(a, b)
    cams, dCs = _views()
    single = OracleViewRenderer()
    D.multi_view_step(single, cams, dCs)
    want = single.flat.numpy()
    assert np.allclose(b[:11 * N], want[:11 * N], rtol=1e-12, atol=1e-14)
    assert np.linalg.norm(b[11 * N:] - want[11 * N:]) <= 1e-6 * np.linalg.norm(want[11 * N:])


def test_touched_rows_pack_unpack_round_trip_and_bytes():
    rng = np.random.default_rng(3)
    for n in (1, 31, 32, 33, 300, 1000):
        sl = torch.from_numpy(rng.standard_normal((3, n, 3)))
        sl[rng.random((3, n)) < 0.6] = 0.0                                   # most gaussians untouched
        sl[1] = 0.0                                                          # a view that touched nothing
        bits, counts, rows = D.pack_touched_rows(sl)
        assert bits.shape == (3, (n + 31) // 32) and bits.dtype == torch.int32
        assert [int(c) for c in counts] == [int((sl[v] != 0).any(dim=1).sum()) for v in range(3)]
        cap = max(int(counts.max()), 1)
        padded = torch.zeros((3, cap, 3), dtype=sl.dtype)
        for v in range(3):
            padded[v, :rows[v].shape[0]] = rows[v]
        assert torch.equal(D.unpack_touched_rows(bits, counts, padded, n, sl.dtype), sl)
    # the paper figures of DESIGN.md section 6: 1 M gaussians, one view per rank, eight ranks, 37 % touched (C3)
    by = D.touched_exchange_bytes(1_000_000, 1, 8, 0.37)
    assert round(by["allreduce_flat"] / 1e6) == 413 and round(by["factored"] / 1e6) == 161 and 100 < by["touched"] / 1e6 < 115
