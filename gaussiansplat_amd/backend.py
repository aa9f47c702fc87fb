"""ctypes binding of libgsplat_hip.so (include/gsplat.h) -- the Python twin of julia/backend.jl.

This is the only way the package reaches the GPU: there is NO CPU fallback here.  If the
shared library is missing or the HIP device is absent every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSPLAT_HIP_LIB") or os.path.join(_HERE, "lib", "libgsplat_hip.so")   # override: diagnostic builds

GS_MEM_HOST, GS_MEM_DEVICE = 0, 1
ORDER_INDEX, ORDER_DEPTH_DESC, ORDER_DEPTH_ASC = 0, 1, 2

(ARR_TS, ARR_TPS, ARR_MU, ARR_COV3D, ARR_COV2D, ARR_INVCOV, ARR_BBS, ARR_RGB, ARR_SIG, ARR_DEPTH_KEY,
 ARR_TILE_RECT, ARR_SORT_IDXS, ARR_TILE_RANGES, ARR_SORTED_IDS, ARR_SORTED_KEYS, ARR_GRAD2D) = range(16)

STAGES = ("preprocess", "depth_sort", "count_scan", "emit", "tile_sort", "ranges", "composite_fwd",
          "composite_bwd", "preprocess_bwd")

# every symbol include/gsplat.h declares (tests/test_abi.py checks the header against this)
SYMBOLS = ("gs_default_config", "gs_abi_version", "gs_create", "gs_destroy", "gs_last_error", "gs_set_stream",
           "gs_synchronize", "gs_set_model", "gs_set_model_2d", "gs_set_image_size", "gs_set_camera", "gs_preprocess", "gs_bin", "gs_bind_outputs", "gs_forward", "gs_backward_sgd",
           "gs_backward", "gs_backward_ex", "gs_reset_grads", "gs_loss_l1_dssim", "gs_sgd_step", "gs_comm_unique_id",
           "gs_comm_init", "gs_allreduce_grads", "gs_comm_destroy", "gs_color_grads_pack", "gs_sh_grads_from_views", "gs_grads_alloc", "gs_grads_read", "gs_num_gaussians", "gs_num_instances", "gs_get_array",
           "gs_get_stage_times", "gs_get_stage_stats", "gs_get_work_counters", "gs_get_work_counters_ex", "gs_debug_time_composite",
           "gs_debug_tile_clock", "gs_debug_clock_mhz", "gs_rank_probe_result", "gs_num_rounds", "gs_set_view_slot", "gs_num_coarse_instances",
           "gs_get_list_stats", "gs_get_tile_parts", "gs_get_bin_path", "gs_debug_set_window", "gs_debug_tile_clock_rows")

GS_ABI_VERSION = 3          # include/gsplat.h; load() refuses a library that reports another version
GS_DEBUG_WIDE_CURSORS = 1
GS_DEBUG_ALWAYS_ORDER = 2     # launch orders + side stream also on small frames (tests)
GS_DEBUG_SUPER16 = 8          # two-level binning: super-tiles of 16 x 16 tiles whatever the grid (tests)
GS_DEBUG_SUPER8 = 16          # ... of 8 x 8 tiles whatever the grid
GS_DEBUG_TINY_CAPS = 4        # capped lists with the minimum cap on every tile (tests: every busy tile extends its list in the composite kernel)
GS_MAX_VIEW_SLOTS = 4096


class GsConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("abi_version", C.c_int32), ("tile_size", C.c_int32), ("order", C.c_int32), ("t_min", C.c_float),
                ("deterministic", C.c_int32), ("export_debug", C.c_int32), ("profile_stages", C.c_int32),
                ("bin_path", C.c_int32), ("rank_mode", C.c_int32), ("alpha_cull", C.c_int32), ("schedule", C.c_int32),
                ("slab_mode", C.c_int32), ("slab_max_ratio", C.c_float), ("slab_fractions", C.c_float * 3), ("debug_flags", C.c_int32),
                ("depth_sort", C.c_int32), ("list_cap", C.c_int32), ("tile_parts", C.c_int32), ("reserved", C.c_int32 * 3)]


class GsGrads(C.Structure):
    _fields_ = [("d_means", C.c_void_p), ("d_scales", C.c_void_p), ("d_quats", C.c_void_p),
                ("d_opacities", C.c_void_p), ("d_shs", C.c_void_p)]


class GsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libgsplat_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """dlopen the library; raises if it has not been built (python -m gaussiansplat_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # ONE HIP runtime per process: torch bundles its own libamdhip64.so.7 / libhsa-runtime64;
        # loading it first makes our library's NEEDED libamdhip64.so.7 resolve to the same copy
        # (loading /opt/rocm's copy first leaves torch with "No HIP GPUs are available").
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not found: build it with `python -m gaussiansplat_amd.build` "
                                "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, fp = C.c_void_p, C.POINTER(C.c_float)
    L.gs_default_config.argtypes = [C.POINTER(GsConfig)]; L.gs_default_config.restype = None
    L.gs_abi_version.restype = C.c_int
    L.gs_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(GsConfig)]
    L.gs_destroy.argtypes = [vp]
    L.gs_last_error.argtypes = [vp]; L.gs_last_error.restype = C.c_char_p
    L.gs_set_stream.argtypes = [vp, vp]
    L.gs_synchronize.argtypes = [vp]
    L.gs_set_model.argtypes = [vp, C.c_int64, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    L.gs_set_camera.argtypes = [vp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, C.c_int32, C.c_int32]
    L.gs_preprocess.argtypes = [vp]
    L.gs_bin.argtypes = [vp, C.c_int32, C.c_int32]
    L.gs_bind_outputs.argtypes = [vp, vp, vp]
    L.gs_backward_sgd.argtypes = [vp, vp, C.c_int, C.c_float]
    L.gs_forward.argtypes = [vp, vp, vp, C.c_int]
    L.gs_backward.argtypes = [vp, vp, C.c_int, C.POINTER(GsGrads)]
    L.gs_backward_ex.argtypes = [vp, vp, C.c_int, C.POINTER(GsGrads), C.c_int]
    L.gs_reset_grads.argtypes = [vp, C.POINTER(GsGrads)]
    L.gs_loss_l1_dssim.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp, C.POINTER(C.c_double), C.c_int]
    L.gs_sgd_step.argtypes = [vp, C.c_float, C.POINTER(GsGrads)]
    L.gs_comm_unique_id.argtypes = [vp]
    L.gs_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.gs_allreduce_grads.argtypes = [vp, C.POINTER(GsGrads)]
    L.gs_comm_destroy.argtypes = [vp]
    L.gs_grads_alloc.argtypes = [vp, C.POINTER(GsGrads)]
    L.gs_grads_read.argtypes = [vp, C.POINTER(GsGrads), vp, vp, vp, vp, vp]
    L.gs_num_gaussians.argtypes = [vp]; L.gs_num_gaussians.restype = C.c_int64
    L.gs_num_instances.argtypes = [vp]; L.gs_num_instances.restype = C.c_int64
    L.gs_get_array.argtypes = [vp, C.c_int, vp, C.c_int64]
    L.gs_get_stage_times.argtypes = [vp, fp]
    L.gs_get_stage_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.gs_get_work_counters.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.gs_set_model_2d.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int]
    L.gs_set_image_size.argtypes = [vp, C.c_int32, C.c_int32]
    L.gs_color_grads_pack.argtypes = [vp, vp]
    L.gs_sh_grads_from_views.argtypes = [vp, C.c_int32, vp, vp, vp, C.c_int]
    L.gs_get_work_counters_ex.argtypes = [vp, C.POINTER(C.c_int64)]
    L.gs_debug_time_composite.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.gs_debug_tile_clock.argtypes = [vp, C.c_int, C.c_int, vp]
    L.gs_debug_clock_mhz.argtypes = [vp, C.POINTER(C.c_float)]
    L.gs_rank_probe_result.argtypes = [vp]
    L.gs_num_rounds.argtypes = [vp]
    L.gs_set_view_slot.argtypes = [vp, C.c_int32]
    L.gs_num_coarse_instances.argtypes = [vp]; L.gs_num_coarse_instances.restype = C.c_int64
    L.gs_get_list_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.gs_get_tile_parts.argtypes = [vp]
    L.gs_get_bin_path.argtypes = [vp]
    L.gs_debug_set_window.argtypes = [vp, C.c_int32, C.c_int32]
    L.gs_debug_tile_clock_rows.argtypes = [vp]
    if L.gs_abi_version() != GS_ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {L.gs_abi_version()}, this binding is written for {GS_ABI_VERSION}: "
                           "rebuild with `python -m gaussiansplat_amd.build --force`")
    _lib = L
    return L


def default_config() -> GsConfig:
    cfg = GsConfig()
    load().gs_default_config(C.byref(cfg))
    return cfg


class Context:
    """Owns one gs_ctx (one GPU, one stream)."""

    def __init__(self, device: int = 0, order: int = ORDER_DEPTH_DESC, t_min: float = 1e-5, export_debug: bool = False,
                 profile_stages: bool = False, deterministic: bool = False, bin_path: int = 0, rank_mode: int = 1,
                 alpha_cull: bool = True, schedule: int = 0, slab_mode: int = 1, slab_fractions=(), slab_max_ratio: float = 0.0,
                 debug_flags: int = 0, depth_sort: int = 0, list_cap: int = 0, tile_parts: "int | None" = None,
                 cfg: "GsConfig | None" = None):
        """schedule 0 = the library default (3); slab_fractions / slab_max_ratio / debug_flags: gs_config fields for tests;
        depth_sort 0 automatic, 1 the four-pass radix sort, 2 always key-range buckets + LDS (same permutation);
        list_cap 0 automatic (tile lists written as far as the view slot's previous frame walked them), 1 never, 2 also on small grids.
        cfg: a complete gs_config to copy instead (every field: a second ctx that must take the same code paths as the first)."""
        self.L = load()
        if cfg is not None:
            twin = GsConfig()
            C.memmove(C.byref(twin), C.byref(cfg), C.sizeof(GsConfig))
            self.cfg = twin
            self.h = C.c_void_p()
            rc = self.L.gs_create(C.byref(self.h), device, C.byref(twin))
            if rc != 0:
                raise GsError(rc, (self.L.gs_last_error(None) or b"").decode())
            self._keep = []
            return
        cfg = default_config()
        assert cfg.struct_size == C.sizeof(GsConfig) and cfg.abi_version == GS_ABI_VERSION
        cfg.schedule = int(schedule)
        cfg.slab_mode = int(slab_mode)
        cfg.slab_max_ratio = float(slab_max_ratio)
        for i, f in enumerate(tuple(slab_fractions)[:3]):
            cfg.slab_fractions[i] = float(f)
        cfg.debug_flags = int(debug_flags)
        cfg.depth_sort = int(depth_sort)
        cfg.list_cap = int(list_cap)
        # waves per tile on small grids: 0 = automatic; None = GSPLAT_TILE_PARTS (the test suite pins 1 there: its cross-mode tests
        # assert bit-identical results, which holds under one partition of the pixels only) or automatic
        cfg.tile_parts = int(os.environ.get("GSPLAT_TILE_PARTS", "0")) if tile_parts is None else int(tile_parts)
        cfg.order, cfg.t_min = int(order), float(t_min)
        cfg.export_debug, cfg.profile_stages, cfg.deterministic = int(export_debug), int(profile_stages), int(deterministic)
        cfg.bin_path, cfg.rank_mode, cfg.alpha_cull = int(bin_path), int(rank_mode), int(alpha_cull)
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.L.gs_create(C.byref(self.h), device, C.byref(cfg))
        if rc != 0:
            raise GsError(rc, (self.L.gs_last_error(None) or b"").decode())
        self._keep = []

    def _chk(self, rc: int):
        if rc != 0:
            raise GsError(rc, (self.L.gs_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.gs_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing
    def set_stream(self, hip_stream: int | None):
        self._chk(self.L.gs_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        self._chk(self.L.gs_synchronize(self.h))

    # -- model / camera
    def set_model_host(self, means, scales, quats, opacities, shs, sh_degree: int):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (means, scales, quats, opacities, shs)]
        n = arrs[0].shape[0]
        self._chk(self.L.gs_set_model(self.h, n, sh_degree, *(C.c_void_p(a.ctypes.data) for a in arrs), GS_MEM_HOST))

    def set_model_device(self, n: int, sh_degree: int, ptrs):
        """ptrs: five device pointers (means, scales, quats, opacities, shs); borrowed, not copied."""
        self._chk(self.L.gs_set_model(self.h, n, sh_degree, *(C.c_void_p(int(p)) for p in ptrs), GS_MEM_DEVICE))

    def set_model_2d_host(self, means, scales, rots, opacities, colors):
        """SplatData2D (splat.jl:20-26): means [n,2], scales [n,2], rotations [n], opacities [n], colors [n,3]."""
        arrs = [np.ascontiguousarray(a, np.float32) for a in (means, scales, rots, opacities, colors)]
        n = arrs[0].shape[0]
        self._chk(self.L.gs_set_model_2d(self.h, n, *(C.c_void_p(a.ctypes.data) for a in arrs), GS_MEM_HOST))

    def set_model_2d_device(self, n: int, ptrs):
        self._chk(self.L.gs_set_model_2d(self.h, n, *(C.c_void_p(int(p)) for p in ptrs), GS_MEM_DEVICE))

    def set_image_size(self, W: int, H: int):
        self._chk(self.L.gs_set_image_size(self.h, int(W), int(H)))
        self.W, self.H = int(W), int(H)

    def set_camera(self, T, P, fx, fy, near, far, eye, lookAt, W, H):
        fp = C.POINTER(C.c_float)
        a = [np.ascontiguousarray(v, np.float32).reshape(-1) for v in (T, P, eye, lookAt)]
        self._chk(self.L.gs_set_camera(self.h, a[0].ctypes.data_as(fp), a[1].ctypes.data_as(fp), fx, fy, near, far,
                                       a[2].ctypes.data_as(fp), a[3].ctypes.data_as(fp), int(W), int(H)))
        self.W, self.H = int(W), int(H)

    def camera_record(self, T, P, fx, fy, near, far, eye, lookAt, W, H):
        """The arguments of gs_set_camera converted once (ctypes arrays): set_camera_record(rec) then costs one foreign call."""
        f16, f3 = C.c_float * 16, C.c_float * 3
        a = [np.ascontiguousarray(v, np.float32).reshape(-1) for v in (T, P, eye, lookAt)]
        return (f16(*a[0]), f16(*a[1]), float(fx), float(fy), float(near), float(far), f3(*a[2]), f3(*a[3]), int(W), int(H))

    def set_camera_record(self, rec):
        self._chk(self.L.gs_set_camera(self.h, *rec))
        self.W, self.H = rec[8], rec[9]

    def set_view_slot(self, slot: int):
        """gs_set_view_slot: name the view about to be rendered (e.g. the camera's id); -1 = none."""
        self._chk(self.L.gs_set_view_slot(self.h, int(slot)))

    # -- pipeline
    def preprocess(self):
        self._chk(self.L.gs_preprocess(self.h))

    def bin(self, gx: int = 0, gy: int = 0):
        self._chk(self.L.gs_bin(self.h, gx, gy))

    def forward_host(self):
        img = np.empty((3, self.H, self.W), np.float32)
        tr = np.empty((self.H, self.W), np.float32)
        self._chk(self.L.gs_forward(self.h, C.c_void_p(img.ctypes.data), C.c_void_p(tr.ctypes.data), GS_MEM_HOST))
        return img, tr

    def bind_outputs(self, image_ptr: int = 0, trans_ptr: int = 0):
        """gs_bind_outputs: the forward writes straight into these device buffers (0, 0 unbinds); they must stay valid and
        unmodified until the frame's last backward."""
        self._chk(self.L.gs_bind_outputs(self.h, C.c_void_p(image_ptr), C.c_void_p(trans_ptr)))

    def forward_device(self, image_ptr: int = 0, trans_ptr: int = 0):
        self._chk(self.L.gs_forward(self.h, C.c_void_p(image_ptr), C.c_void_p(trans_ptr), GS_MEM_DEVICE))

    def backward(self, dC_ptr_or_array, grads: GsGrads, overwrite: bool = False, phase: str = "all"):
        """phase: "all", "composite" (GS_BWD_COMPOSITE_ONLY), "params" (GS_BWD_PARAMS_ONLY), or the chain in two steps:
        "params_sh" (GS_BWD_PARAMS_ONLY | GS_BWD_PARAMS_SH) then "params_geom" (GS_BWD_PARAMS_ONLY | GS_BWD_PARAMS_GEOM)."""
        flags = (1 if overwrite else 0) | {"all": 0, "composite": 2, "params": 4, "params_sh": 4 | 8, "params_geom": 4 | 16}[phase]   # GS_BWD_*
        if isinstance(dC_ptr_or_array, np.ndarray):
            a = np.ascontiguousarray(dC_ptr_or_array, np.float32)
            self._chk(self.L.gs_backward_ex(self.h, C.c_void_p(a.ctypes.data), GS_MEM_HOST, C.byref(grads), flags))
        else:
            self._chk(self.L.gs_backward_ex(self.h, C.c_void_p(int(dC_ptr_or_array)), GS_MEM_DEVICE, C.byref(grads), flags))

    def color_grads_pack(self, drgb_ptr: int):
        """d rgb of the last backward, packed [n, 3] into a device buffer (colour-factored exchange)."""
        self._chk(self.L.gs_color_grads_pack(self.h, C.c_void_p(int(drgb_ptr))))

    def sh_grads_from_views(self, cam_records: np.ndarray, drgb_ptr: int, d_shs_ptr: int, overwrite: bool = True):
        """cam_records [nviews, 38] host float32 {T16, P16, eye3, lookAt3}; drgb [nviews, n, 3] device."""
        cr = np.ascontiguousarray(cam_records, np.float32)
        assert cr.ndim == 2 and cr.shape[1] == 38
        self._chk(self.L.gs_sh_grads_from_views(self.h, cr.shape[0], C.c_void_p(cr.ctypes.data), C.c_void_p(int(drgb_ptr)),
                                                C.c_void_p(int(d_shs_ptr)), 1 if overwrite else 0))

    def backward_sgd(self, dC_ptr: int, lr: float):
        """gs_backward_sgd: backward and `param .-= lr * grad` (train.jl:42-46) in one pass on the resident model; dC on the device."""
        self._chk(self.L.gs_backward_sgd(self.h, C.c_void_p(dC_ptr), GS_MEM_DEVICE, C.c_float(lr)))

    def grads_alloc(self) -> GsGrads:
        """Library-owned flat gradient buffer (for hosts without a device allocator, e.g. plain Julia)."""
        g = GsGrads()
        self._chk(self.L.gs_grads_alloc(self.h, C.byref(g)))
        return g

    def grads_read(self, grads: GsGrads, sh_degree: int) -> dict:
        n, k3 = self.num_gaussians, 3 * (sh_degree + 1) ** 2
        out = dict(means=np.empty((n, 3), np.float32), scales=np.empty((n, 3), np.float32), quats=np.empty((n, 4), np.float32),
                   opacities=np.empty((n,), np.float32), shs=np.empty((n, k3), np.float32))
        self._chk(self.L.gs_grads_read(self.h, C.byref(grads), *(C.c_void_p(out[k].ctypes.data) for k in
                                                                   ("means", "scales", "quats", "opacities", "shs"))))
        return out

    def grads_read_2d(self, grads: GsGrads) -> dict:
        """SplatGrads2D (splat.jl:28-34) from a gs_grads whose slots are read as (means, scales, rotations, opacities, colors)."""
        n = self.num_gaussians
        out = dict(means=np.empty((n, 2), np.float32), scales=np.empty((n, 2), np.float32), rots=np.empty((n,), np.float32),
                   opacities=np.empty((n,), np.float32), colors=np.empty((n, 3), np.float32))
        self._chk(self.L.gs_grads_read(self.h, C.byref(grads), *(C.c_void_p(out[k].ctypes.data) for k in
                                                                   ("means", "scales", "rots", "opacities", "colors"))))
        return out

    def loss_host(self, img: np.ndarray, gt: np.ndarray, lam: float = 0.1):
        """(loss, dC) for host images [C, H, W] (src/loss.jl:62-72 and its gradient)."""
        img = np.ascontiguousarray(img, np.float32); gt = np.ascontiguousarray(gt, np.float32)
        dC = np.empty_like(img)
        out = C.c_double()
        Cn, H, W = img.shape
        self._chk(self.L.gs_loss_l1_dssim(self.h, C.c_void_p(img.ctypes.data), C.c_void_p(gt.ctypes.data), W, H, Cn, lam,
                                          C.c_void_p(dC.ctypes.data), C.byref(out), GS_MEM_HOST))
        return float(out.value), dC

    def loss_device(self, img_ptr: int, gt_ptr: int, dC_ptr: int, W: int, H: int, Cn: int, lam: float = 0.1, want_loss: bool = True):
        out = C.c_double()
        self._chk(self.L.gs_loss_l1_dssim(self.h, C.c_void_p(img_ptr), C.c_void_p(gt_ptr), W, H, Cn, lam, C.c_void_p(dC_ptr),
                                          C.byref(out) if want_loss else None, GS_MEM_DEVICE))
        return float(out.value) if want_loss else None

    # -- RCCL directly through the C ABI (what the Julia glue uses; the Python mirror defaults to torch.distributed)
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = load().gs_comm_unique_id(buf)
        if rc != 0:
            raise GsError(rc, (load().gs_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, rank: int, nranks: int, unique_id: bytes):
        self._chk(self.L.gs_comm_init(self.h, rank, nranks, C.c_char_p(unique_id)))

    def allreduce_grads(self, grads: GsGrads):
        self._chk(self.L.gs_allreduce_grads(self.h, C.byref(grads)))

    def sgd_step(self, lr: float, grads: GsGrads):
        self._chk(self.L.gs_sgd_step(self.h, lr, C.byref(grads)))

    def reset_grads(self, grads: GsGrads):
        self._chk(self.L.gs_reset_grads(self.h, C.byref(grads)))

    # -- introspection
    @property
    def num_gaussians(self) -> int:
        return int(self.L.gs_num_gaussians(self.h))

    @property
    def num_instances(self) -> int:
        return int(self.L.gs_num_instances(self.h))

    @property
    def num_coarse_instances(self) -> int:
        """two-level binning: (super-tile, gaussian) instances of the last bin()"""
        return int(self.L.gs_num_coarse_instances(self.h))

    @property
    def num_rounds(self) -> int:
        """binning rounds of the last bin(): 1 = classic full lists, 2..4 = depth slabs"""
        return int(self.L.gs_num_rounds(self.h))

    def get_array(self, which: int, gx: int | None = None, gy: int | None = None) -> np.ndarray:
        n, ni = self.num_gaussians, self.num_instances
        spec = {ARR_TS: ((n, 4), np.float32), ARR_TPS: ((n, 4), np.float32), ARR_MU: ((n, 2), np.float32),
                ARR_COV3D: ((n, 9), np.float32), ARR_COV2D: ((n, 4), np.float32), ARR_INVCOV: ((n, 4), np.float32),
                ARR_BBS: ((n, 4), np.float32), ARR_RGB: ((n, 3), np.float32), ARR_SIG: ((n,), np.float32),
                ARR_DEPTH_KEY: ((n,), np.uint32), ARR_TILE_RECT: ((n, 4), np.uint16), ARR_SORT_IDXS: ((n,), np.uint32),
                ARR_SORTED_IDS: ((ni,), np.uint32), ARR_SORTED_KEYS: ((ni,), np.uint64), ARR_GRAD2D: ((n, 10), np.float32)}
        if which == ARR_TILE_RANGES:
            ntiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
            shape, dt = (ntiles, 2), np.uint32
        else:
            shape, dt = spec[which]
        out = np.empty(shape, dt)
        self._chk(self.L.gs_get_array(self.h, which, C.c_void_p(out.ctypes.data), out.nbytes))
        return out

    def stage_stats(self, reset: bool = False) -> dict:
        """{stage: (sum_ms, launches)} accumulated by hipEvents on the ctx stream since the last reset."""
        sm = (C.c_double * len(STAGES))()
        ct = (C.c_int64 * len(STAGES))()
        self._chk(self.L.gs_get_stage_stats(self.h, sm, ct, int(reset)))
        return {k: (float(sm[i]), int(ct[i])) for i, k in enumerate(STAGES)}

    def work_counters(self):
        a, b = C.c_int64(), C.c_int64()
        self._chk(self.L.gs_get_work_counters(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def work_counters_ex(self) -> dict:
        """walked = list entries staged; evaluated = those that survived the alpha_cull no-op test."""
        o = (C.c_int64 * 4)()
        self._chk(self.L.gs_get_work_counters_ex(self.h, o))
        return dict(walked_fwd=int(o[0]), walked_bwd=int(o[1]), evaluated_fwd=int(o[2]), evaluated_bwd=int(o[3]))

    def list_stats(self) -> dict:
        """entries written into the tile lists of the last frame, list segments appended by composite waves, capped or not"""
        o = (C.c_int64 * 3)()
        self._chk(self.L.gs_get_list_stats(self.h, o))
        return dict(listed=int(o[0]), extended_segments=int(o[1]), capped=bool(o[2]))

    def tile_parts_of_frame(self) -> int:
        """waves per tile (1, 2, 4) of the last frame's composite launches (gs_config.tile_parts)"""
        rc = self.L.gs_get_tile_parts(self.h)
        if rc < 0:
            self._chk(rc)
        return int(rc)

    def bin_path_of_frame(self) -> int:
        """the path that built the last frame's lists: 0 two-level, 1 / 2 radix, 3 small-frame path (gs_config.bin_path)"""
        rc = self.L.gs_get_bin_path(self.h)
        if rc < 0:
            self._chk(rc)
        return int(rc)

    def time_composite(self, which: int, variant: int, reps: int = 10) -> float:
        ms = C.c_float()
        self._chk(self.L.gs_debug_time_composite(self.h, which, variant, reps, C.byref(ms)))
        return float(ms.value)

    def tile_clock(self, which: int, variant: int = 0) -> np.ndarray:
        """[ntiles, 15] uint64 per tile {start, end (100 MHz ticks), HW_ID | XCC_ID << 32, walked << 32 | evaluated, shader cycles
        inside the per-entry loops, shader cycles outside them, strip slots executed << 32 | slots with live pixels packed,
        strips with a live pixel << 32 | live pixels} of one composite launch (which: 0 forward, 1 backward)."""
        ntiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        rows = ntiles
        if variant < 0:                                     # one record per workgroup of the launch (split tiles as production runs them)
            rows = int(self.L.gs_debug_tile_clock_rows(self.h))
            if rows <= 0:
                raise GsError(-1, "tile_clock: records by workgroup need a frame with a launch order")
        out = np.zeros((rows, 15), np.uint64)
        self._chk(self.L.gs_debug_tile_clock(self.h, which, variant, C.c_void_p(out.ctypes.data)))
        return out

    def set_debug_window(self, start: int = 0, length: int = 0):
        """debug launches cover only order[start : start + length] of the frame's launch order (0, 0: all of it)"""
        self._chk(self.L.gs_debug_set_window(self.h, start, length))

    def clock_mhz(self) -> float:
        """the shader clock the chip runs at right now (one wave counting cycles over 20 us; waits for the stream)"""
        mhz = C.c_float()
        self._chk(self.L.gs_debug_clock_mhz(self.h, C.byref(mhz)))
        return float(mhz.value)

    @property
    def rank_probe_result(self) -> int:
        return int(self.L.gs_rank_probe_result(self.h))

    def stage_times(self) -> dict:
        ms = (C.c_float * len(STAGES))()
        self._chk(self.L.gs_get_stage_times(self.h, ms))
        return {k: float(ms[i]) for i, k in enumerate(STAGES)}
