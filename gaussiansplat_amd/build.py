"""Build libgsplat_hip.so in-tree with hipcc for gfx950 (no CMake, no JIT cache).

    python -m gaussiansplat_amd.build [--force] [--tag NAME -DMACRO=... ...]

--tag NAME builds lib_NAME/libgsplat_hip.so with the given -D macros (e.g. --tag seg1k -DL2_SEG=1024): a variant for a same-box
A/B run, loaded with GSPLAT_HIP_LIB=.../lib_NAME/libgsplat_hip.so (tools/ab_libs.sh).  The kernel variants that lost their A/B
in rounds 1-3 (persistent ticket queues, the reduction tree, the software-pipelined backward, the 64-VGPR forward) and their
environment switches are no longer in the tree: profiles/HISTORY.md names the commit that last had them.

The preprocess translation units are compiled with -ffp-contract=off (numeric spec: tile ids
and depth keys must be bit-identical to the CPU oracle).  The composite kernels turn contraction
off themselves (`#pragma clang fp contract(off)` at the top of gs_composite.hip: only the fmaf()
calls written there fuse, so that every instantiation rounds a pixel's T the same way) and are
built without the SLP vectoriser; their pixels and gradients are tolerance-checked.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libgsplat_hip.so")
ARCH = "gfx950"

SOURCES = {
    "gs_preprocess.hip": ["-ffp-contract=off"],
    "gs_preprocess2d.hip": ["-ffp-contract=off"],
    "gs_preprocess_bwd.hip": [],
    "gs_sort.hip": [],
    "gs_bin2.hip": [],
    "gs_bin3.hip": [],
    "gs_bin_small.hip": [],
    # no SLP vectoriser: a v_pk_*_f32 issues at the cost of two plain operations on gfx950, and forming the pairs costs moves
    # (C3 1.328 -> 1.294 ms, C5 4.38 -> 4.26, same box, profiles/r04q_ab_no_slp.log; round 2's kernels had measured the opposite)
    "gs_composite.hip": ["-fno-slp-vectorize"],
    # the SLP vectoriser turns the stencil into v_pk_* (no faster than two plain ops on gfx950) plus 270 register moves per loop body
    "gs_loss.hip": ["-fno-slp-vectorize"],
    "gs_api.hip": [],
    "gs_api_bin.hip": [],
    "gs_api_composite.hip": [],
    "gs_api_comm.hip": [],
    "gs_api_debug.hip": [],
}
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _deps() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "gsplat.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force: bool = False, verbose: bool = False, tag: str = "", defines=()) -> str:
    """tag / defines: a variant build for same-box A/B runs (e.g. tag="p48", defines=["-DGS_PAYLOAD_QUADS=3"] -> lib_p48/)."""
    out_dir = os.path.join(HERE, "lib_" + tag) if tag else OUT_DIR
    lib = os.path.join(out_dir, "libgsplat_hip.so")
    os.makedirs(out_dir, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = _deps()
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        extra = [*extra, *defines]
        s = os.path.join(CSRC, src)
        o = os.path.join(out_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append([hipcc, *COMMON, *extra, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(lib):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs, "-ldl"])
    return lib


if __name__ == "__main__":
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose=True, tag=tag,
                defines=[a for a in sys.argv[1:] if a.startswith("-D") or a.startswith("-f") or a.startswith("-mllvm") or a.startswith("-amdgpu")]))
