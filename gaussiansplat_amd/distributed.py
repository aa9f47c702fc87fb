"""Multi-view data parallelism: one camera view per GPU, ONE RCCL all-reduce per step.

The reference has no distributed code at all (no collective call sites); this is
the build's scaling axis from BASELINE.json's north_star.  The path shards naturally across
camera views: every rank holds a full replica of the gaussians, renders its share of the view
batch (preprocess -> compactIdxs -> forward -> backward, gradients ACCUMULATING across its views),
and the only exchange is the sum of the per-gaussian parameter gradients:

    flat = [Δmeans 3N | Δscales 3N | Δquats 4N | Δopac N | Δshs 3K·N]      (59 N floats at SH3)

one contiguous fp32 buffer, one `all_reduce(SUM)` (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" in the CPU tests).  xGMI is point-to-point, so a single large collective (236 MB at
1 M gaussians) is the right shape: RCCL can split it over all 7 links per GPU.
"""
from __future__ import annotations

from typing import Callable, Protocol, Sequence


class ViewRenderer(Protocol):
    """Anything that can render one view and accumulate d(loss)/d(params) into `flat`."""
    flat: "torch.Tensor"

    def reset(self) -> None: ...
    def render_view(self, camera, dC) -> None: ...


def shard_views(num_views: int, world: int, rank: int) -> list[int]:
    """Contiguous block partition of the view batch (8 views: 1/2/4/8 GPUs -> 8/4/2/1 views each)."""
    base, rem = divmod(num_views, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def multi_view_step(r: ViewRenderer, cameras: Sequence, dCs: Sequence, group=None) -> "torch.Tensor":
    """One data-parallel step over a view batch.  Returns the all-reduced flat gradient buffer
    (identical on every rank; the caller applies its optimiser and the next step starts with
    reset())."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    r.reset()
    for v in shard_views(len(cameras), world, rank):
        r.render_view(cameras[v], dCs[v])
    if world > 1:
        dist.all_reduce(r.flat, op=dist.ReduceOp.SUM, group=group)      # the ONE collective of the step
    return r.flat


class HipViewRenderer:
    """Adapter of a GaussianRenderer3D (HIP path) to the ViewRenderer protocol."""

    def __init__(self, renderer):
        self.r = renderer

    @property
    def flat(self):
        return self.r.splatGrads.flat

    def reset(self) -> None:
        from . import renderer as R
        R.resetGrads(self.r)

    def render_view(self, camera, dC) -> None:
        from . import renderer as R
        tps = R.preprocess(self.r, camera)
        R.compactIdxs(self.r)
        R.forward(self.r, tps)
        R.backward(self.r, dC)
