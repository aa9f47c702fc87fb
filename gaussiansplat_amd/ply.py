"""3DGS PLY scene I/O with the reference's field mapping (src/splat.jl:106-119).

    points       <- x, y, z
    scales       <- scale_0..2          (log space)
    quaternions  <- rot_0..3            (w, x, y, z; NOT normalised, as in the reference)
    opacities    <- opacity             (logit)
    shs          <- vcat(f_dc_0..2, f_rest_0..8)   = 12 floats = 4 coefficients x rgb (degree 1)

The reference reads f_rest in FILE order as [coefficient][channel] (splat.jl:117 + splat.jl:248-250),
i.e. shs[3k + c]; that mapping is kept.  `sh_degree` > 1 (build extension) takes the first
3K-3 f_rest values the same way.  Only the binary_little_endian / ascii `vertex` element with
scalar properties is supported (what 3DGS trainers write).
"""
from __future__ import annotations

import numpy as np

_PLY_T = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1",
          "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4",
          "uint": "<u4", "uint32": "<u4"}


def _read_vertex_table(path: str):
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, props, n, in_vertex = None, [], 0, False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_T[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            return np.fromfile(fh, dtype=np.dtype(props), count=n)
        if fmt == "ascii":
            raw = np.loadtxt(fh, max_rows=n, ndmin=2)
            out = np.empty(n, dtype=np.dtype(props))
            for i, (name, _) in enumerate(props):
                out[name] = raw[:, i]
            return out
        raise ValueError(f"{path}: unsupported PLY format {fmt!r}")


def load_ply(path: str, sh_degree: int = 1) -> dict:
    v = _read_vertex_table(path)
    K = (sh_degree + 1) ** 2
    col = lambda names: np.stack([np.asarray(v[n], np.float32) for n in names], axis=1)
    n = len(v)
    shs = np.concatenate([col(["f_dc_0", "f_dc_1", "f_dc_2"]),
                          col([f"f_rest_{i}" for i in range(3 * K - 3)]) if K > 1 else np.zeros((n, 0), np.float32)], axis=1)
    return dict(means=col(["x", "y", "z"]), scales=col(["scale_0", "scale_1", "scale_2"]),
                quats=col(["rot_0", "rot_1", "rot_2", "rot_3"]), opacities=np.asarray(v["opacity"], np.float32),
                shs=shs.reshape(n, K, 3))


def save_ply(path: str, scene: dict) -> None:
    """Write a scene in the 3DGS layout (45 f_rest columns, zero padded) -- used by tests/tools."""
    n = scene["means"].shape[0]
    shs = np.asarray(scene["shs"], np.float32).reshape(n, -1)
    names = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + ["opacity"]
             + ["scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    tab = np.zeros(n, dtype=[(k, "<f4") for k in names])
    for i, k in enumerate("xyz"):
        tab[k] = scene["means"][:, i]
    for i in range(3):
        tab[f"f_dc_{i}"] = shs[:, i]
        tab[f"scale_{i}"] = scene["scales"][:, i]
    for i in range(shs.shape[1] - 3):
        tab[f"f_rest_{i}"] = shs[:, 3 + i]
    for i in range(4):
        tab[f"rot_{i}"] = scene["quats"][:, i]
    tab["opacity"] = np.asarray(scene["opacities"]).reshape(-1)
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n).encode())
        fh.write("".join(f"property float {k}\n" for k in names).encode())
        fh.write(b"end_header\n")
        tab.tofile(fh)
