#!/usr/bin/env python3
"""Instruction mix and issue-cost model of the composite inner loops, from the disassembly of THIS build.

    python3 tools/loop_cost.py [out.json]        (runs here: hipcc cross-compiles gfx950 without a GPU)

For the default forward and backward kernels it takes the per-entry loop (the innermost loop that holds v_exp_f32),
counts the wave64 instructions by form and prices them with the per-form issue costs measured on MI355X by
tools/valu_ubench3.hip (cycles per wave-instruction per SIMD, eight waves per SIMD, independent chains):

    plain VOP (fma / mul / add / sub / fmac, VGPR or inline-constant sources)   2.3
    v_med3 / v_min / v_max / v_cndmask                                        4.2
    v_pk_* (two results)                                                      4.3
    any SGPR source (v_mov from SGPR, v_mad_u64_u32 with SGPR)                 4.4
    v_exp / v_rcp / v_log (quarter rate)                                      8.2
    DPP forms (+ hazard nop)                                                  4.9
    v_readfirstlane                                                           6.3

`model_cycles_per_entry` is the VALU-issue floor of one evaluated (tile, splat) entry if every instruction issued at its
microbenchmark cost; bench.py / DESIGN.md compare it with the measured cycles per entry (kernel time x clock x 1024 SIMDs
/ evaluated entries).  The backward loop skips dead 16x4 strips with wave-uniform branches: the count is given for four
live strips and per strip, the model uses the measured mean of live strips per entry (default 3.02, C3).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = [
    (r"v_(exp|rcp|log|rsq|sqrt)_f32", "transcendental", 8.2),
    (r"v_(med3|min3|max3|min|max)_f32|v_cndmask_b32", "select_minmax", 4.2),
    (r"v_pk_", "packed", 4.3),
    (r"v_readfirstlane", "readfirstlane", 6.3),
    (r"v_mad_u64_u32", "mad64", 4.5),
    (r"_dpp|v_mov_b32_dpp", "dpp", 4.9),
    (r"v_", "plain", 2.3),
]


def classify(line):
    op = line.split()[0]
    sgpr_src = op.startswith("v_mov_b32") and re.search(r",\s*s\d+", line) is not None
    if sgpr_src:
        return "sgpr_source", 4.4
    if "_dpp" in line or " quad_perm" in line or " row_" in line:
        return "dpp", 4.9
    for pat, name, c in COST:
        if re.match(pat, op):
            return name, c
    return None, 0.0


def kernel_body(asm, name):
    m = re.search(r"^" + re.escape(name) + r":.*?s_endpgm", asm, re.S | re.M)
    return m.group(0) if m else None


def loop_with_exp(body):
    """lines of the innermost loop (by the compiler's Depth annotation) that contains v_exp_f32"""
    lines = body.splitlines()
    depth_of = []
    cur = 0
    for ln in lines:
        m = re.search(r"Depth=(\d+)", ln)
        if m and (ln.strip().startswith(".LBB") or "Loop Header" in ln or ln.strip().startswith(";")):
            cur = int(m.group(1))
        depth_of.append(cur)
    best = max((d for d, ln in zip(depth_of, lines) if "v_exp_f32" in ln), default=0)
    # contiguous region at that depth around the exp instructions
    idx = [i for i, (d, ln) in enumerate(zip(depth_of, lines)) if d == best and "v_exp_f32" in ln]
    lo, hi = idx[0], idx[-1]
    while lo > 0 and depth_of[lo - 1] >= best:
        lo -= 1
    while hi + 1 < len(lines) and depth_of[hi + 1] >= best:
        hi += 1
    return [ln for ln in lines[lo:hi + 1] if re.match(r"^\s+[vsdg][a-z_0-9]*\s", ln) or re.match(r"^\s+(v_|s_|ds_|global_)", ln)]


def summarise(lines, live_strips=None):
    mix, cycles, nv, lds, vmem, salu = {}, 0.0, 0, 0, 0, 0
    for ln in lines:
        t = ln.strip()
        if t.startswith("ds_"):
            lds += 1
        elif t.startswith("global_") or t.startswith("buffer_"):
            vmem += 1
        elif t.startswith("s_"):
            salu += 1
        elif t.startswith("v_"):
            name, c = classify(t)
            mix[name] = mix.get(name, 0) + 1
            cycles += c
            nv += 1
    return {"valu": nv, "lds": lds, "vmem": vmem, "salu_and_waits": salu, "mix": mix, "model_cycles": round(cycles, 1),
            "model_cycles_per_valu": round(cycles / max(nv, 1), 2)}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    src = os.path.join(ROOT, "gaussiansplat_amd", "csrc", "gs_composite.hip")
    with tempfile.TemporaryDirectory() as td:
        s_path = os.path.join(td, "c.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", s_path, src] + os.environ.get("LOOP_COST_FLAGS", "").split(),
                       check=True, capture_output=True)
        asm = open(s_path).read()
    names = re.findall(r"^(_Z20composite_(?:fwd|bwd)_kernel\S*?):", asm, re.M)
    # the production instantiations: early-out, 5 waves, cull; forward <EARLY=1, 5, CULL=1, CLK=0, SLAB=0, SNAP=0>, backward <EARLY=1, 6, DET=0, CULL=1, CLK=0, PAIR=0>
    pick = {"composite_fwd": [n for n in names if "fwd_kernelILb1ELi5ELb1ELb0ELb0ELb0EE" in n],
            "composite_bwd": [n for n in names if "bwd_kernelILb1ELi6ELb0ELb1ELb0ELb0EE" in n]}
    missing = [k for k, v in pick.items() if not v]
    if missing:
        sys.exit("loop_cost.py: production instantiation not found for %s (template parameters changed?): %s" % (missing, names))
    res = {}
    for k, cand in pick.items():
        if not cand:
            continue
        body = kernel_body(asm, cand[0])
        lines = loop_with_exp(body)
        info = summarise(lines)
        info["kernel"] = cand[0]
        meta = re.search(r"\.name:\s+" + re.escape(cand[0]) + r"\n.*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", asm, re.S)
        if meta:
            info["sgpr"], info["vgpr"] = int(meta.group(1)), int(meta.group(2))
        n_exp = sum(1 for ln in lines if "v_exp_f32" in ln)
        info["entries_per_loop_iteration"] = max(1, n_exp // 4)
        info["model_cycles_per_entry_all_strips"] = round(info["model_cycles"] / info["entries_per_loop_iteration"], 1)
        info["valu_per_entry_all_strips"] = round(info["valu"] / info["entries_per_loop_iteration"], 1)
        if k == "composite_fwd" and info["entries_per_loop_iteration"] > 2:
            # since round 4 the forward has three per-entry loops (4 / 2 / 1 pixel slots, two entries per iteration) at one nesting depth and this
            # region spans them: the counts are of all three together, not of one entry
            info["note"] = "spans the forward's 4-, 2- and 1-slot loops: per-entry figures are NOT those of one loop"
        res[k] = info
    sys.path.insert(0, ROOT)
    import bench
    res["csrc_sha"] = bench.csrc_sha()
    if out:
        with open(out, "w") as fh:
            json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
