"""Image export of the reference's viewer script (src/examples/main.jl:35-45), SURVEY 8(f) rank 4.

    imageData[isnan] .= 0 ; clamp to [0,1] ; N0f8 (round(x*255)) ; colorview(RGB) ; imrotate(pi/2)

`to_rgb8` takes the renderer's image in the [3, H, W] layout (== Julia's [W, H, 3]) and returns an
uint8 array [rows, cols, 3] as the reference would display it: Julia's image has W rows and H columns
(first array axis = image row), then it is rotated by +90 degrees.
"""
from __future__ import annotations

import numpy as np


def to_rgb8(image_chw, rotate: bool = True) -> np.ndarray:
    img = np.asarray(image_chw.detach().cpu().numpy() if hasattr(image_chw, "detach") else image_chw, np.float32)
    img = np.where(np.isnan(img), np.float32(0), img)                  # main.jl:35
    img = np.clip(img, 0.0, 1.0)                                         # main.jl:40
    u8 = np.floor(img * 255.0 + 0.5).astype(np.uint8)                    # n0f8: nearest of 255 levels
    julia = np.transpose(u8, (2, 1, 0))                                  # [W, H, 3]: Julia rows = x, columns = y
    if rotate:
        julia = np.rot90(julia, k=1, axes=(0, 1))                        # imrotate(img, pi/2): counter-clockwise
    return np.ascontiguousarray(julia)


def save_ppm(path: str, rgb8: np.ndarray) -> None:
    """Minimal dependency-free writer (binary PPM) for the exported image."""
    h, w, _ = rgb8.shape
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (w, h))
        fh.write(np.ascontiguousarray(rgb8, np.uint8).tobytes())
