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

`sync="factored"` produces the SAME buffer with ~2.6x less xGMI traffic at SH degree 3.  The SH gradient of one view
is rank one, d sh[k][c] = basis_k(dir_view) * d rgb[c], and dir_view depends only on the gaussian's mean and the
view's camera, which every rank knows.  So the ranks all-reduce only the geometry part [Δmeans|Δscales|Δquats|Δopac]
(11 N floats), all-gather the three floats d rgb per (view, gaussian), and rebuild Δshs = sum_v basis(dir_v) (x) d rgb_v
locally (gs_sh_grads_from_views): 44 MB all-reduced + 12 MB per view gathered instead of 236 MB all-reduced at 1 M
gaussians.  Sums over views are taken in view order, so the result is also reproducible run to run.

`sync="touched"` (round 5; gloo-tested, never run on more than one GPU) is "factored" with the colour gradients of a view sent
as the rows of the gaussians the view TOUCHED only: a view's composite adjoint leaves d rgb = 0 for every gaussian no pixel
evaluated (63 % of them at C3, 90 % at C5: tools/touched_rows.py), so a view travels as a bitmap of N bits plus 12 bytes per
touched gaussian -- 0.125 + 4.4 MB instead of 12 MB at C3 -- padded to the largest count among the views (two small
all-gathers: counts, then bitmaps and rows).  The receiver scatters the rows back and rebuilds the SH gradients as "factored"
does, so the result is the same, bit for bit.
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


def multi_view_step(r: ViewRenderer, cameras: Sequence, dCs: Sequence, group=None, sync: str = "allreduce",
                    overlap: bool = True, pipeline: bool = True, exchange: bool = True) -> "torch.Tensor":
    """One data-parallel step over a view batch (8 views on 1 / 2 / 4 / 8 GPUs = 8 / 4 / 2 / 1 views per rank, rendered
    one after the other with the gradients ACCUMULATING, then the exchange).  Returns the flat gradient buffer summed over all
    views (identical on every rank; the caller applies its optimiser and the next step starts with reset()).
    sync = "allreduce": the sum of the whole flat buffer.  With `overlap`, a renderer that can split its last backward
    (render_view_until_sh / finish_geometry: the HIP renderer) and a batch that leaves some rank exactly ONE view (8 views on
    8 GPUs), the buffer is reduced as its two contiguous segments ON EVERY RANK: the Δshs segment (81 % of the bytes at SH
    degree 3) is final as soon as the last view's SH kernel has run, so its all-reduce is started there (async: RCCL's own
    stream, event-ordered behind the kernel) and runs beside the geometry chain; the all-reduce of the 11 N geometry floats
    follows.  Same sums, bit for bit, as ONE all-reduce of the whole buffer -- an all-reduce is element-wise.  Otherwise
    (overlap = False, every rank pipelines two or more views, or some rank has no view at all): literally one collective.
    Which of the two it is follows from the view counts of ALL ranks, so every rank posts the same collectives in the same order
    whatever its own share is (3 views on 2 ranks, 12 on 8, fewer views than ranks).
    pipeline (a rank with two or more views, HIP renderer): consecutive views alternate between `pipeline_depth` (3) renderers that share the model and
    the gradient buffer but own their per-view scratch and HIP stream (HipViewRenderer.render_views_pipelined), so the short,
    latency-bound kernels of view k+1 (preprocess, depth sort, tile lists) run beside the composite kernels of view k; the
    per-gaussian chains stay in view order (event-chained), so the sums are the same bits as one view after the other.
    sync = "factored": see the module docstring (needs a renderer with render_view_factored / color_slots / sh_from_views, and
    the same number of views on every rank).
    exchange = False (measurement only, bench.py's compute_ms): this rank's share of the views exactly as in a real step -- the same
    kernels on the same streams -- but no collective is posted; the buffer then holds the rank's partial sums."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_views(len(cameras), world, rank)
    if not exchange:
        class _NoExchange:                                                  # stands in for torch.distributed below: every collective is a no-op
            class ReduceOp: SUM = None
            class _Done:
                def wait(self): pass
            @staticmethod
            def all_reduce(*a, **k): return _NoExchange._Done()
            @staticmethod
            def all_gather_into_tensor(out, inp, **k):
                out.view(-1)[rank * inp.numel():(rank + 1) * inp.numel()].copy_(inp.view(-1)); return _NoExchange._Done()
        dist = _NoExchange
    r.reset()
    if sync == "allreduce":
        # How many collectives a step posts, and over which segments, is decided from numbers EVERY rank computes alike -- the
        # view counts of all ranks and the renderer's type -- never from this rank's own share: unequal shares (3 views on 2
        # ranks, fewer views than ranks) would otherwise post mismatched collectives (gloo aborts, RCCL hangs).
        counts = [len(shard_views(len(cameras), world, k)) for k in range(world)]
        can_pipeline = pipeline and hasattr(r, "render_views_pipelined")
        can_split = overlap and hasattr(r, "render_view_until_sh") and hasattr(r, "geometry_floats")
        # two segments (Δshs first) when some rank renders exactly one view the two-step way -- then every rank reduces the same
        # two segments, whatever path its own views took; a rank without a view has nothing to split, so min(counts) >= 1
        two_segments = world > 1 and can_split and min(counts) >= 1 and (min(counts) == 1 or not can_pipeline)
        if can_pipeline and len(mine) > 1:
            r.render_views_pipelined([cameras[v] for v in mine], [dCs[v] for v in mine])
            last_split = False
        else:
            last_split = two_segments and len(mine) > 0
            for v in (mine[:-1] if last_split else mine):
                r.render_view(cameras[v], dCs[v])
        if two_segments:
            geo = r.geometry_floats
            if last_split:
                r.render_view_until_sh(cameras[mine[-1]], dCs[mine[-1]])    # ... composite adjoint, SH kernel: Δshs is final
            work = dist.all_reduce(r.flat[geo:], op=dist.ReduceOp.SUM, group=group, async_op=True)
            if last_split:
                r.finish_geometry()                                         # the geometry chain runs beside the collective
            dist.all_reduce(r.flat[:geo], op=dist.ReduceOp.SUM, group=group)
            work.wait()
        elif world > 1:
            dist.all_reduce(r.flat, op=dist.ReduceOp.SUM, group=group)      # the ONE collective of the step
        return r.flat
    if sync not in ("factored", "touched"):
        raise ValueError(sync)
    if len(cameras) % world:
        raise ValueError("factored / touched sync needs the same number of views on every rank")
    slots = r.color_slots(len(mine))                                        # [len(mine), n, 3] on the renderer's device
    for i, v in enumerate(mine):
        r.render_view_factored(cameras[v], dCs[v], slots[i])                # geometry grads accumulate, d rgb -> slot i
    geo = r.flat[:r.geometry_floats]
    if world > 1 and sync == "touched":
        dist.all_reduce(geo, op=dist.ReduceOp.SUM, group=group)             # 11 N floats
        allc = exchange_touched_rows(slots, world, group, dist)             # per view: N bits + 3 floats per touched gaussian
    elif world > 1:
        dist.all_reduce(geo, op=dist.ReduceOp.SUM, group=group)             # 11 N floats
        allc = torch.empty(world * slots.numel(), dtype=slots.dtype, device=slots.device)
        dist.all_gather_into_tensor(allc, slots.reshape(-1), group=group)   # 3 N floats per view
        allc = allc.reshape((len(cameras),) + tuple(slots.shape[1:]))       # contiguous block partition: rank-major == view order
    else:
        allc = slots
    r.sh_from_views(list(cameras), allc)                                    # overwrites the Δshs part of flat
    return r.flat


def pack_touched_rows(slots):
    """slots [views, N, 3] -> (bits [views, ceil(N / 32)] int32: bit g % 32 of word g / 32 set when view v left gaussian g a non-zero
    colour gradient; counts [views] int64; rows: list of [count_v, 3] tensors, the touched rows in gaussian order).  Host-side
    torch plumbing (boolean indexing synchronises); a device kernel would pack in one pass -- not built: no N > 1 hardware to time it on."""
    import torch
    V, N, _ = slots.shape
    touched = (slots != 0).any(dim=2)                                       # an exactly-zero row adds nothing to the SH sums either way
    words = (N + 31) // 32
    pad = torch.zeros((V, words * 32), dtype=torch.int64, device=slots.device)
    pad[:, :N] = touched.to(torch.int64)
    weights = (torch.ones(32, dtype=torch.int64, device=slots.device) << torch.arange(32, dtype=torch.int64, device=slots.device))
    bits = (pad.reshape(V, words, 32) * weights).sum(dim=2)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)      # (two's complement: the 32 bits as stored)
    rows = [slots[v][touched[v]] for v in range(V)]
    return bits, touched.sum(dim=1), rows


def unpack_touched_rows(bits, counts, rows_padded, N, dtype):
    """inverse of pack_touched_rows for the gathered views: -> [views, N, 3] with zeros where a view touched nothing"""
    import torch
    V, words = bits.shape
    b = bits.to(torch.int64) & 0xFFFFFFFF
    shifts = torch.arange(32, dtype=torch.int64, device=bits.device)
    touched = (((b.unsqueeze(2) >> shifts) & 1) != 0).reshape(V, words * 32)[:, :N]
    out = torch.zeros((V, N, 3), dtype=dtype, device=bits.device)
    for v in range(V):
        out[v][touched[v]] = rows_padded[v, :int(counts[v])]
    return out


def exchange_touched_rows(slots, world, group, dist):
    """all ranks' views' colour gradients [world * views, N, 3] from this rank's `slots` [views, N, 3], sending per view its bitmap of
    touched gaussians and their rows only (padded to the largest count among all views of the step)"""
    import torch
    V, N, _ = slots.shape
    bits, counts, rows = pack_touched_rows(slots)
    all_counts = torch.empty(world * V, dtype=torch.int64, device=slots.device)
    dist.all_gather_into_tensor(all_counts, counts.to(torch.int64).contiguous(), group=group)
    cap = max(int(all_counts.max()), 1)
    mine = torch.zeros((V, cap, 3), dtype=slots.dtype, device=slots.device)
    for v in range(V):
        mine[v, :rows[v].shape[0]] = rows[v]
    all_bits = torch.empty((world * V, bits.shape[1]), dtype=torch.int32, device=slots.device)
    all_rows = torch.empty((world * V, cap, 3), dtype=slots.dtype, device=slots.device)
    dist.all_gather_into_tensor(all_bits.reshape(-1), bits.contiguous().reshape(-1), group=group)       # N / 8 bytes per view
    dist.all_gather_into_tensor(all_rows.reshape(-1), mine.reshape(-1), group=group)                    # 12 bytes per touched gaussian (padded)
    return unpack_touched_rows(all_bits, all_counts, all_rows, N, slots.dtype)


def touched_exchange_bytes(n: int, views_per_rank: int, world: int, touched_share: float, k3: int = 48):
    """bytes leaving each GPU per step for the three exchanges (ring all-reduce: 2 (w - 1) / w of the buffer; all-gather: w - 1 times the
    rank's share): {flat, factored, touched} -- the paper figures of DESIGN.md section 6"""
    ar = lambda b: 2.0 * (world - 1) / world * b
    ag = lambda b: (world - 1) * b
    geo = 11 * n * 4
    return {"allreduce_flat": ar((11 + k3) * n * 4),
            "factored": ar(geo) + ag(views_per_rank * 3 * n * 4),
            "touched": ar(geo) + ag(views_per_rank * (n / 8 + 8 + 12 * touched_share * n))}


class HipViewRenderer:
    """Adapter of a GaussianRenderer3D (HIP path) to the ViewRenderer protocol."""

    def __init__(self, renderer, pipeline_depth: int = 3, own_renderer_in_flight: bool = False):
        self.r = renderer
        # True: the rank's own renderer is one of the `pipeline_depth` in flight (one ctx less).  Measured SLOWER on MI355X with the
        # runtime's default of four hardware queues (8-view batch, one GPU, same box: 9.8 ms against 9.0 ms with `pipeline_depth` new
        # renderers; the relation flips with GPU_MAX_HW_QUEUES=8 -- it is a matter of which HSA queue each stream lands on, which
        # neither torch nor HIP lets a host choose; profiles/r04i_c4_twins.log), so the default keeps the own renderer out of it.
        self.own_renderer_in_flight = own_renderer_in_flight
        self.last_ctx = renderer.ctx            # the ctx that rendered the most recent view (introspection: counters, instance counts)
        self.pipeline_depth = max(2, int(pipeline_depth))      # views in flight in render_views_pipelined

    def contexts(self):
        """every gs_ctx views are rendered through (the renderer's own, and the twins of the pipelined mode once they exist)"""
        return [self.r.ctx] + [t.ctx for t, _ in (getattr(self, "_tw", None) or []) if t is not self.r]

    @property
    def flat(self):
        # Between render_view_until_sh and finish_geometry of a rank's ONLY view the lazy reset is still pending for the geometry
        # segment (its chain overwrites it) while the Δshs segment is already final: looking at the buffer then must not
        # materialise the zero fill (renderer.splatGrads does), or the async all-reduce would sum a wiped Δshs segment.
        if getattr(self, "_mid_split", False):
            return self.r._splatGrads.flat
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
        self.last_ctx = self.r.ctx

    # ---- several views of one rank on `pipeline_depth` streams (multi_view_step(pipeline=True))
    def _twins(self):
        """`pipeline_depth` renderers in flight over the SAME parameter tensors and the SAME flat gradient buffer (borrowed device
        pointers: no copy of the model, no gradient buffer of their own): twins of the rank's renderer with their own ctx -- i.e.
        their own per-view scratch (payload rows, depth order, tile lists, image; 288 GB of HBM hold many) -- created from a copy of
        the WHOLE gs_config, so that every view takes the same code paths.  Each has its own HIP stream.  (own_renderer_in_flight:
        the rank's own renderer is the first of them.)"""
        if getattr(self, "_tw", None) is None:
            import torch
            from . import renderer as R
            r = self.r
            H, W = r.transmittance.shape
            dev = r.imageData.device
            tw = [(r, torch.cuda.Stream(device=dev))] if self.own_renderer_in_flight else []
            for _ in range(self.pipeline_depth - len(tw)):
                t = R.GaussianRenderer3D(r.splatData, (W, H), r.sh_degree, device=dev.index or 0, share_grads_with=r, cfg=r.ctx.cfg)
                tw.append((t, torch.cuda.Stream(device=dev)))
            self._tw = tw
        return self._tw

    def render_views_pipelined(self, cameras, dCs) -> None:
        """Views alternate between the twins.  On a twin's stream: preprocess, lists, forward, composite adjoint of its view --
        nothing there touches what the other twin uses -- then the per-gaussian chain, which accumulates into the shared
        gradient buffer and therefore waits (event) for the chain of the view before it.  The first view overwrites (the lazy
        reset), the others accumulate, in view order: the same sums, bit for bit, as render_view in a loop."""
        import numpy as np
        import torch
        from . import renderer as R
        r = self.r
        dev = r.imageData.device
        dCs = [(d if isinstance(d, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(d, np.float32))).to(dev, torch.float32).contiguous()
               for d in dCs]                                                # (uploads, if any, on the caller's stream: before `start`)
        for d in dCs:                                                       # the checks R.backward makes, before anything is enqueued
            if tuple(d.shape) != tuple(r.imageData.shape):
                raise ValueError(f"dC has shape {tuple(d.shape)}, the image is {tuple(r.imageData.shape)}")
        cur = torch.cuda.current_stream(dev)
        start = torch.cuda.Event(); start.record(cur)
        overwrite = r._grads_lazy_zero
        prev_chain = None
        for i, (cam, dC) in enumerate(zip(cameras, dCs)):
            t, st = self._twins()[i % self.pipeline_depth]
            with torch.cuda.stream(st):
                if i < self.pipeline_depth:
                    st.wait_event(start)                                   # inputs and the previous step's readers of the gradient buffer
                tps = R.preprocess(t, cam)
                R.compactIdxs(t)
                R.forward(t, tps)
                t._dC_keepalive = dC
                t.ctx.backward(dC.data_ptr(), t._grads, overwrite=False, phase="composite")
                if prev_chain is not None:
                    st.wait_event(prev_chain)
                t.ctx.backward(dC.data_ptr(), t._grads, overwrite=overwrite, phase="params")
                prev_chain = torch.cuda.Event(); prev_chain.record(st)
            overwrite = False
            self.last_ctx = t.ctx
        r._grads_lazy_zero = False
        cur.wait_event(prev_chain)                                         # the chains are chained: the last one is the end of the batch

    # ---- the last view of a rank in two steps (multi_view_step(overlap=True))
    def render_view_until_sh(self, camera, dC) -> None:
        from . import renderer as R
        tps = R.preprocess(self.r, camera)
        R.compactIdxs(self.r)
        R.forward(self.r, tps)
        self._dC = dC
        R.backward(self.r, dC, phase="composite")
        R.backward(self.r, dC, phase="params_sh")
        self._mid_split = True
        self.last_ctx = self.r.ctx

    def finish_geometry(self) -> None:
        from . import renderer as R
        R.backward(self.r, self._dC, phase="params_geom")
        self._mid_split = False

    # ---- colour-factored exchange
    @property
    def geometry_floats(self) -> int:
        return 11 * self.r.nGaussians

    def color_slots(self, nviews: int):
        import torch
        key = (nviews, self.r.nGaussians)
        if getattr(self, "_slots_key", None) != key:
            self._slots = torch.empty((nviews, self.r.nGaussians, 3), dtype=torch.float32, device=self.r.imageData.device)
            self._slots_key = key
        return self._slots

    def render_view_factored(self, camera, dC, slot) -> None:
        from . import renderer as R
        tps = R.preprocess(self.r, camera)
        R.compactIdxs(self.r)
        R.forward(self.r, tps)
        R.backward(self.r, dC, skip_shs=True)
        self.r.ctx.color_grads_pack(slot.data_ptr())

    def sh_from_views(self, cameras, drgb_all) -> None:
        R_ = self.r
        H, W = R_.transmittance.shape
        R_._begin()
        R_.ctx.sh_grads_from_views(view_records(cameras, W, H), drgb_all.contiguous().data_ptr(), R_.splatGrads.Δshs.data_ptr(), overwrite=True)


def factored_one_view_step(hv: "HipViewRenderer", camera, dC, cam_records, gathered, group=None) -> None:
    """The colour-factored step for ONE view per rank (the 8-GPU batch of BASELINE.json), with the all-gather of the
    colour gradients started right after the composite adjoint so that it overlaps the per-gaussian chain:
        preprocess, bin, forward, composite backward -> pack d rgb -> all_gather (async)
        per-gaussian chain (no Δshs)                 -> all_reduce of the 11N geometry floats
        wait for the gather                          -> Δshs from all views.
    cam_records: view_records of ALL ranks' cameras in rank order (host, computed once); gathered: [world * 3N] float32
    device buffer (allocated once).  Without a process group it degenerates to the single-view factored path."""
    import torch.distributed as dist
    from . import renderer as R
    r = hv.r
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    tps = R.preprocess(r, camera)
    R.compactIdxs(r)
    R.forward(r, tps)
    R.backward(r, dC, phase="composite")
    slot = hv.color_slots(1)
    r.ctx.color_grads_pack(slot.data_ptr())
    work = dist.all_gather_into_tensor(gathered, slot.reshape(-1), group=group, async_op=True) if world > 1 else None
    R.backward(r, dC, skip_shs=True, phase="params")
    flat = r.splatGrads.flat
    if world > 1:
        dist.all_reduce(flat[:hv.geometry_floats], group=group)
        work.wait()
        src = gathered
    else:
        src = slot
    r.ctx.sh_grads_from_views(cam_records, src.data_ptr(), r.splatGrads.Δshs.data_ptr(), overwrite=True)


def view_records(cameras, W: int, H: int):
    """[nviews, 38] float32 {T16, P16, eye3, lookAt3} (column-major matrices, as gs_set_camera takes them)."""
    import numpy as np
    from .camera import compute_projection, compute_transform
    out = np.empty((len(cameras), 38), np.float32)
    for i, cam in enumerate(cameras):
        out[i, :16] = np.asarray(compute_transform(cam), np.float32).reshape(-1, order="F")
        out[i, 16:32] = np.asarray(compute_projection(cam, W, H), np.float32).reshape(-1, order="F")
        out[i, 32:35] = np.asarray(cam.eye, np.float32)
        out[i, 35:38] = np.asarray(cam.lookAt, np.float32)
    return out
