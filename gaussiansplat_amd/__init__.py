"""gaussiansplat_amd -- MI355X (gfx950) differentiable Gaussian-splat rasterizer behind the
arhik/GaussianSplat renderer API.  Compute lives in lib/libgsplat_hip.so (hand-written HIP,
C ABI in include/gsplat.h); this package is the host-side mirror of the reference interface.
"""
from .camera import Camera, compute_projection, compute_transform, default_camera, get_camera  # noqa: F401

__all__ = ["Camera", "default_camera", "compute_transform", "compute_projection", "get_camera"]
