"""Host-side mirror of the reference renderer API (src/renderer.jl, src/forward.jl,
src/backward.jl, src/splat.jl) over the C ABI of libgsplat_hip.so.

The reference's call sequence (src/examples/main.jl:14-34) works unchanged in spirit:

    renderer = getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), path_or_scene)
    tps = preprocess(renderer)
    compactIdxs(renderer, (16, 16), (gx, gy))
    forward(renderer, tps, (16, 16), (gx, gy))
    backward(renderer, dC); ...optimiser...; resetGrads(renderer.splatGrads)

Names, argument meaning and in-place semantics follow the reference: outputs land in
`renderer.imageData` / `renderer.transmittance`, gradients ACCUMULATE in
`renderer.splatGrads.*` until `resetGrads`.  Arrays are torch tensors on the renderer's GPU
with the reference's memory layout ([n, comp] row-major == Julia [comp, n] column-major;
imageData [3, H, W] == Julia [W, H, 3]).  torch is plumbing only (device memory, streams);
all compute is in the HIP library and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import backend as B
from .camera import Camera, compute_projection, compute_transform, default_camera


class RendererType(enum.Enum):          # renderer.jl:20-24
    GAUSSIAN_2D = 0
    GAUSSIAN_3D = 1
    OPTIMAL_PROJECTION_3D = 2


@dataclass
class SplatData3D:                      # splat.jl:36-43
    means: "torch.Tensor"               # [n, 3]
    scales: "torch.Tensor"              # [n, 3]  log-space
    shs: "torch.Tensor"                 # [n, 3K] element [3k + c]
    quaternions: "torch.Tensor"         # [n, 4]  (w, x, y, z), not normalised by the kernels
    opacities: "torch.Tensor"           # [n, 1]  logit
    features: object = None


@dataclass
class SplatGrads3D:                     # splat.jl:45-52 -- views into ONE flat buffer (single all-reduce)
    flat: "torch.Tensor"
    Δmeans: "torch.Tensor"
    Δscales: "torch.Tensor"
    Δquaternions: "torch.Tensor"
    Δopacities: "torch.Tensor"
    Δshs: "torch.Tensor"
    Δfeatures: object = None


@dataclass
class SplatData2D:                      # splat.jl:20-26
    means: "torch.Tensor"               # [n, 2]  fractions of the image, pixel position (W*mx, H*my)
    scales: "torch.Tensor"              # [n, 2]  log-space
    rotations: "torch.Tensor"           # [n, 1]  theta
    opacities: "torch.Tensor"           # [n, 1]  used raw (splat.jl:341)
    colors: "torch.Tensor"              # [n, 3]


@dataclass
class SplatGrads2D:                     # splat.jl:28-34 -- views into ONE flat buffer
    flat: "torch.Tensor"
    Δmeans: "torch.Tensor"
    Δscales: "torch.Tensor"
    Δrotations: "torch.Tensor"
    Δopacities: "torch.Tensor"
    Δcolors: "torch.Tensor"


def initGrads2D(splatData: SplatData2D) -> SplatGrads2D:
    """initGrads(::SplatData2D), splat.jl:121-135, as one flat buffer [Δmeans 2N | Δscales 2N | Δrot N | Δopac N | Δcolors 3N]."""
    import torch
    n = splatData.means.shape[0]
    sizes = [2 * n, 2 * n, n, n, 3 * n]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=splatData.means.device)
    parts, o = [], 0
    for sz in sizes:
        parts.append(flat[o:o + sz]); o += sz
    return SplatGrads2D(flat, parts[0].view(n, 2), parts[1].view(n, 2), parts[2].view(n, 1), parts[3].view(n, 1), parts[4].view(n, 3))


def initGrads(splatData: SplatData3D) -> SplatGrads3D:
    """splat.jl:137-156; laid out as one flat fp32 buffer
    [Δmeans 3N | Δscales 3N | Δquats 4N | Δopac N | Δshs 3K·N] so that a multi-view step needs a
    single RCCL all-reduce."""
    import torch
    n = splatData.means.shape[0]
    k3 = splatData.shs.shape[1]
    sizes = [3 * n, 3 * n, 4 * n, n, k3 * n]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=splatData.means.device)
    parts, o = [], 0
    for s in sizes:
        parts.append(flat[o:o + s]); o += s
    return SplatGrads3D(flat, parts[0].view(n, 3), parts[1].view(n, 3), parts[2].view(n, 4), parts[3].view(n, 1),
                        parts[4].view(n, k3))


class GaussianRenderer3D:               # renderer.jl:205-219
    def __init__(self, splatData: SplatData3D, imgSize, sh_degree: int, device: int = 0, order: int = B.ORDER_DEPTH_DESC,
                 t_min: float = 1e-5, export_debug: bool = False, profile_stages: bool = False, deterministic: bool = False,
                 alpha_cull: bool = True, rank_mode: int = 1, slab_mode: int = 1, schedule: int = 0, share_grads_with=None, **ctx_kw):
        """share_grads_with: another GaussianRenderer3D over the same splatData whose gradient buffer this one accumulates into
        (a second view in flight, distributed.HipViewRenderer): no buffer of its own is allocated."""
        import torch
        self.splatData = splatData
        self._splatGrads = share_grads_with._splatGrads if share_grads_with is not None else initGrads(splatData)
        self._grads_lazy_zero = False       # resetGrads() pending: the next backward overwrites
        W, H = int(imgSize[0]), int(imgSize[1])
        dev = splatData.means.device
        self.imageData = torch.zeros((3, H, W), dtype=torch.float32, device=dev)      # CUDA.zeros(imgSize...)
        self.transmittance = torch.ones((H, W), dtype=torch.float32, device=dev)      # CUDA.ones(imgSize[1:2])
        self.nGaussians = splatData.means.shape[0]
        self.sh_degree = sh_degree
        self.camera: Camera | None = None
        self.ctx = B.Context(device=device, order=order, t_min=t_min, export_debug=export_debug, profile_stages=profile_stages,
                             deterministic=deterministic, alpha_cull=alpha_cull, rank_mode=rank_mode, slab_mode=slab_mode, schedule=schedule, **ctx_kw)
        # the library enqueues on the caller's CURRENT torch stream (re-read at every API call), like any torch
        # op: no cross-stream fences, so consecutive calls run back to back on the GPU
        self._stream_handle = None
        self._begin()
        self.ctx.set_model_device(self.nGaussians, sh_degree,
                                  [t.data_ptr() for t in (splatData.means, splatData.scales, splatData.quaternions,
                                                          splatData.opacities, splatData.shs)])
        g = self._splatGrads
        self._grads = B.GsGrads(g.Δmeans.data_ptr(), g.Δscales.data_ptr(), g.Δquaternions.data_ptr(),
                                g.Δopacities.data_ptr(), g.Δshs.data_ptr())

    @property
    def splatGrads(self) -> SplatGrads3D:
        """renderer.splatGrads (renderer.jl:207).  resetGrads is lazy: the zero fill is skipped when the
        next backward overwrites anyway, and materialised here if somebody looks first."""
        if self._grads_lazy_zero:
            import torch
            self._splatGrads.flat.zero_()
            self._grads_lazy_zero = False
        return self._splatGrads

    def _begin(self):
        """Bind the ctx to torch's current stream (handle 0 = the legacy default stream = GS_STREAM_LEGACY)."""
        h = _current_stream_handle(self.imageData.device) or 1
        if h != self._stream_handle:
            self.ctx.set_stream(h)
            self._stream_handle = h

    def _end(self):                 # kept for callers written against the fenced version: nothing to do
        pass

    def _set_view(self, camera):
        cam = camera or self.camera or default_camera()
        self.camera = cam
        H, W = self.transmittance.shape
        # A small frame is bound by its caller (C1: ten thousand gaussians take the GPU 70 us, tools/host_rate.py): the view's matrices
        # (computeTransform / computeProjection, camera.jl:53-111: 60 us of numpy) are kept with the camera object and rebuilt when
        # one of its fields, or the image size, has changed
        snap = (np.asarray(cam.eye, np.float32).tobytes(), np.asarray(cam.lookAt, np.float32).tobytes(), np.asarray(cam.up, np.float32).tobytes(),
                cam.fx, cam.fy, cam.near, cam.far, W, H)
        ent = getattr(cam, "_gs_view", None)
        if ent is None or ent[0] != snap:
            ent = (snap, self.ctx.camera_record(compute_transform(cam), compute_projection(cam, W, H), float(np.float32(cam.fx)), float(np.float32(cam.fy)),
                                                float(np.float32(cam.near)), float(np.float32(cam.far)), cam.eye, cam.lookAt, W, H))
            try:
                cam._gs_view = ent
            except AttributeError:                  # (a camera type without room for it: rebuilt every time)
                pass
        self.ctx.set_camera_record(ent[1])
        # the camera's id (camera.jl:10-22; `id` of cameras.json) names the view slot: a training loop cycles over a fixed
        # camera set and the library launches the forward's tiles heaviest-first by what the same view measured last time
        slot = getattr(cam, "id", None)
        slot = int(slot) if isinstance(slot, (int, np.integer)) and 0 <= int(slot) < B.GS_MAX_VIEW_SLOTS else -1
        if slot != getattr(self, "_view_slot", None):
            self.ctx.set_view_slot(slot)
            self._view_slot = slot

    # scratch arrays of the reference struct, fetched on demand (export_debug for the fp32 ones)
    @property
    def sortIdxs(self): return self.ctx.get_array(B.ARR_SORT_IDXS)
    @property
    def positions(self): return self.ctx.get_array(B.ARR_MU)
    @property
    def cov2ds(self): return self.ctx.get_array(B.ARR_COV2D)
    @property
    def cov3ds(self): return self.ctx.get_array(B.ARR_COV3D)
    @property
    def invCov2ds(self): return self.ctx.get_array(B.ARR_INVCOV)
    @property
    def bbs(self): return self.ctx.get_array(B.ARR_BBS)


class GaussianRenderer2D:               # renderer.jl:7-18
    """The 2-D image-fitting renderer: same binning and composite kernels as the 3-D one behind a different
    preprocess (cov2d.jl:3-28).  The reference's own 2-D constructors reference undefined names (renderer.jl:38-82)
    and its backward (splat.jl:271-396) is not the adjoint of any one forward; this is the consistent version
    (DESIGN.md, 2-D renderer)."""

    def __init__(self, splatData: SplatData2D, imgSize, device: int = 0, t_min: float = 1e-5, export_debug: bool = False,
                 profile_stages: bool = False, deterministic: bool = False, alpha_cull: bool = True):
        import torch
        self.splatData = splatData
        self._splatGrads = initGrads2D(splatData)
        self._grads_lazy_zero = False
        W, H = int(imgSize[0]), int(imgSize[1])
        dev = splatData.means.device
        self.imageData = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
        self.transmittance = torch.ones((H, W), dtype=torch.float32, device=dev)
        self.nGaussians = splatData.means.shape[0]
        self.camera = None
        self.ctx = B.Context(device=device, order=B.ORDER_INDEX, t_min=t_min, export_debug=export_debug, profile_stages=profile_stages,
                             deterministic=deterministic, alpha_cull=alpha_cull)
        self._stream_handle = None
        self._begin()
        self.ctx.set_model_2d_device(self.nGaussians, [t.data_ptr() for t in (splatData.means, splatData.scales, splatData.rotations,
                                                                              splatData.opacities, splatData.colors)])
        g = self._splatGrads
        self._grads = B.GsGrads(g.Δmeans.data_ptr(), g.Δscales.data_ptr(), g.Δrotations.data_ptr(), g.Δopacities.data_ptr(),
                                g.Δcolors.data_ptr())

    splatGrads = GaussianRenderer3D.splatGrads
    _begin = GaussianRenderer3D._begin
    _end = GaussianRenderer3D._end

    def _set_view(self, camera=None):
        H, W = self.transmittance.shape
        self.ctx.set_image_size(W, H)

    @property
    def positions(self): return self.ctx.get_array(B.ARR_MU)
    @property
    def cov2ds(self): return self.ctx.get_array(B.ARR_COV2D)
    @property
    def invCov2ds(self): return self.ctx.get_array(B.ARR_INVCOV)
    @property
    def bbs(self): return self.ctx.get_array(B.ARR_BBS)


def initData2D(nGaussians: int, seed: int = 0) -> dict:
    """initData(Val(SPLAT2D), n), splat.jl:74-87: everything uniform [0,1), rotations pi/2*(U-0.5)."""
    rng = np.random.default_rng(seed)
    r = lambda *sh: rng.random(sh, dtype=np.float32)
    return dict(means=r(nGaussians, 2), scales=r(nGaussians, 2), rots=np.float32(np.pi / 2) * (r(nGaussians) - np.float32(0.5)),
                opacities=r(nGaussians), colors=r(nGaussians, 3))


def _to_device_data_2d(scene: dict, device) -> SplatData2D:
    import torch
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(device).contiguous()
    n = np.asarray(scene["means"]).shape[0]
    return SplatData2D(means=t(scene["means"]), scales=t(scene["scales"]), rotations=t(np.reshape(scene["rots"], (n, 1))),
                       opacities=t(np.reshape(scene["opacities"], (n, 1))), colors=t(scene["colors"]))


def _to_device_data(scene: dict, device) -> SplatData3D:
    import torch
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(device).contiguous()
    n = scene["means"].shape[0]
    return SplatData3D(means=t(scene["means"]), scales=t(scene["scales"]), shs=t(np.reshape(scene["shs"], (n, -1))),
                       quaternions=t(scene["quats"]), opacities=t(np.reshape(scene["opacities"], (n, 1))))


def initData(nGaussians: int, seed: int = 0) -> dict:
    """splat.jl:90-104: uniform [0,1) parameters (the reference draws a 9xN `shs`, inconsistent
    with the 12 floats splatDraw reads; 12 are drawn here = SH degree 1)."""
    rng = np.random.default_rng(seed)
    r = lambda *s: rng.random(s, dtype=np.float32)
    return dict(means=r(nGaussians, 3), quats=r(nGaussians, 4), scales=r(nGaussians, 3), shs=r(nGaussians, 4, 3),
                opacities=r(nGaussians))


def getRenderer(rendererType, imgSize, threads, blocks, source=None, *, device: int = 0, **kw):
    """renderer.jl:164-186.  `source`: a PLY path (splat.jl:106-119), an int nGaussians
    (splat.jl:90-104) or a dict of arrays (means, scales, quats, opacities, shs[n,K,3])."""
    import torch
    if isinstance(rendererType, str):
        rendererType = RendererType[rendererType.lstrip(":")]
    if rendererType == RendererType.OPTIMAL_PROJECTION_3D:
        raise NotImplementedError("OPTIMAL_PROJECTION_3D: the reference has an enum value and a forwarding method "
                                  "(renderer.jl:23,189-195) but no implementation to follow")
    if tuple(threads) != (16, 16):
        raise ValueError("threads must be (16, 16) (tile size of the reference's example, main.jl:9)")
    W, H = int(imgSize[0]), int(imgSize[1])
    if blocks is not None and tuple(blocks) != ((W + 15) // 16, (H + 15) // 16):
        raise ValueError("blocks must be ceil(imgSize/threads)")
    if not torch.cuda.is_available():
        raise RuntimeError("gaussiansplat_amd needs a HIP device (no CPU fallback)")
    if rendererType == RendererType.GAUSSIAN_2D:         # renderer.jl:38-82: nGaussians or arrays (no PLY form)
        if isinstance(source, int):
            scene = initData2D(source)
        elif isinstance(source, dict):
            scene = source
        else:
            raise TypeError("GAUSSIAN_2D: source must be an int (nGaussians) or a dict of arrays (means, scales, rots, opacities, colors)")
        kw.pop("order", None)
        return GaussianRenderer2D(_to_device_data_2d(scene, torch.device("cuda", device)), (W, H), device=device, **kw)
    if isinstance(source, str):
        from .ply import load_ply
        scene = load_ply(source)
    elif isinstance(source, int):
        scene = initData(source)
    elif isinstance(source, dict):
        scene = source
    else:
        raise TypeError("source must be a PLY path, an int or a dict of arrays")
    shs = np.asarray(scene["shs"])
    K = shs.reshape(shs.shape[0], -1).shape[1] // 3
    deg = {1: 0, 4: 1, 9: 2, 16: 3}[K]
    data = _to_device_data(scene, torch.device("cuda", device))
    return GaussianRenderer3D(data, (W, H), deg, device=device, **kw)


def preprocess(renderer, camera: Camera | None = None):
    """forward.jl:35-111 (3-D; the reference hard-codes defaultCamera(), forward.jl:53 -- a camera may be passed
    instead) and forward.jl:9-33 (2-D: no camera).  Returns `tps` lazily (the reference returns the clip-space
    positions only to hand them to forward()); None for the 2-D renderer."""
    renderer._set_view(camera)
    renderer._begin()
    renderer.ctx.preprocess()
    return _LazyTps(renderer) if isinstance(renderer, GaussianRenderer3D) else None


def _current_stream_handle(device) -> int:
    """torch's current stream on `device` as a raw handle (0: the legacy default stream).  The private fast path costs 0.3 us, the
    public one (a Stream object per call) 2.5 us -- four times per frame."""
    import torch
    try:
        return int(torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device()))
    except Exception:
        return int(torch.cuda.current_stream(device).cuda_stream)


class _LazyTps:
    def __init__(self, r): self._r = r
    def numpy(self): return self._r.ctx.get_array(B.ARR_TPS)


def compactIdxs(renderer, threads=(16, 16), blocks=None):
    """forward.jl:118-161: builds the per-tile splat lists (tile|depth keys, radix sort, ranges)."""
    gx, gy = blocks if blocks is not None else (0, 0)
    renderer._begin()
    renderer.ctx.bin(int(gx), int(gy))


def _bind_outputs(renderer):
    """renderer.imageData / renderer.transmittance ARE the library's output buffers (gs_bind_outputs: no copy).  The raw
    pointers are re-read at every forward / backward: if the caller replaced a tensor since, the new one is bound (and a
    backward then refuses to run on a forward that wrote somewhere else)."""
    img, tr = renderer.imageData, renderer.transmittance
    H, W = tr.shape
    if tuple(img.shape) != (3, H, W) or not img.is_contiguous() or not tr.is_contiguous():
        raise ValueError("renderer.imageData must be a contiguous [3, H, W] tensor matching renderer.transmittance [H, W]")
    key = (img.data_ptr(), tr.data_ptr())
    if getattr(renderer, "_bound_out", None) != key:
        renderer.ctx.bind_outputs(*key)                         # (invalidates the ctx's forward state: did_fwd = false)
        renderer._bound_out = key
    return key


def forward(renderer, tps=None, threads=(16, 16), blocks=None):
    """forward.jl:163-198: writes renderer.imageData and renderer.transmittance in place.
    Contract: the two tensors stay the library's buffers until the frame's last backward -- the backward reads the rendered
    colours from renderer.imageData, so editing it in place between forward and backward changes the gradients, and
    replacing it makes backward() raise (render again first)."""
    renderer._begin()
    ip, tp = _bind_outputs(renderer)
    renderer.ctx.forward_device(ip, tp)


def backward(renderer, ΔC, skip_shs: bool = False, phase: str = "all"):
    """backward.jl:3-38: ΔC has the shape of imageData; accumulates into renderer.splatGrads.
    skip_shs (3-D renderer, colour-factored multi-GPU exchange): leave Δshs alone -- the caller rebuilds it from the
    per-view colour gradients (distributed.multi_view_step(sync="factored")).
    phase: "all" (default); "composite" then "params" (split backward); or "composite", "params_sh", "params_geom": the
    per-gaussian chain in two steps, so that the all-reduce of Δshs can start while the geometry chain still runs."""
    import torch
    dC = ΔC if isinstance(ΔC, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(ΔC, np.float32))
    dC = dC.to(renderer.imageData.device, torch.float32).contiguous()
    assert dC.shape == renderer.imageData.shape
    renderer._dC_keepalive = dC
    renderer._begin()
    _bind_outputs(renderer)                                     # a replaced imageData is caught here: the ctx then reports "gs_forward first"
    grads = renderer._grads
    if skip_shs:
        grads = B.GsGrads(grads.d_means, grads.d_scales, grads.d_quats, grads.d_opacities, None)
    renderer.ctx.backward(dC.data_ptr(), grads, overwrite=renderer._grads_lazy_zero, phase=phase)
    if phase not in ("composite", "params_sh"):                 # "params_sh" is followed by "params_geom" with the same overwrite flag
        renderer._grads_lazy_zero = False


def resetGrads(renderer_or_grads):
    """splat.jl:158-173."""
    import torch
    if isinstance(renderer_or_grads, (GaussianRenderer3D, GaussianRenderer2D)):
        renderer_or_grads._grads_lazy_zero = True      # zero fill deferred: see GaussianRenderer3D.splatGrads
    else:
        renderer_or_grads.flat.zero_()
