"""julia/backend.jl against include/gsplat.h -- the only verification possible without a Julia toolchain.

Every `ccall((:name, libgs), Ret, (Args...), ...)` is parsed and checked against the C prototype of the same name:
arity, scalar widths, pointer-ness and pointee types; the Julia mirror structs must have the header's fields in the
header's order (and gs_config must stay 64 bytes); every function the header declares must be bound at least once.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _c_prototypes():
    src = _strip_comments(open(os.path.join(ROOT, "include", "gsplat.h")).read())
    protos = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?[A-Za-z_0-9]+\s*\**)\s*(gs_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        alist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, alist)
    return protos


def _c_kind(decl):
    """(kind, pointee) of a C parameter / return declaration."""
    d = re.sub(r"\bconst\b", "", decl).strip()
    arr = "[" in d
    d = re.sub(r"\[.*?\]", "", d).strip()
    stars = d.count("*")
    d = d.replace("*", " ").split()
    base = d[0] if d else ""
    if len(d) > 1 and d[0] in ("unsigned", "long"):
        base = " ".join(d[:-1])
    nptr = stars + (1 if arr else 0)
    if nptr == 0:
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "float": "f32", "double": "f64", "void": "void"}[base], None
    return "ptr" * 1, (base, nptr)


_J_SCALAR = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cfloat": "f32", "Float32": "f32", "Cdouble": "f64", "Float64": "f64",
             "Cvoid": "void"}
_J_POINTEE = {"Float32": "float", "Cfloat": "float", "Float64": "double", "Cdouble": "double", "Int64": "int64_t", "UInt64": "uint64_t",
              "UInt8": "void", "Cvoid": "void", "GsGrads": "gs_grads", "GsConfig": "gs_config", "Int32": "int32_t", "Cint": "int"}


def _j_kind(t):
    t = t.strip()
    if t in _J_SCALAR:
        return _J_SCALAR[t], None
    if t == "Cstring":
        return "ptr", ("char", 1)
    m = re.fullmatch(r"(Ptr|Ref)\{(.*)\}", t)
    assert m, f"unknown Julia ccall type {t!r}"
    inner = m.group(2).strip()
    if inner.startswith("Ptr{"):
        return "ptr", (_j_kind(inner)[1][0], 2)
    return "ptr", (_J_POINTEE[inner], 1)


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def _julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "backend.jl")).read()
    src = re.sub(r"(?m)#.*$", "", src)
    calls = []
    for m in re.finditer(r"ccall\(\(:(gs_[a-z_0-9]+),\s*(?:HipBackend\.)?libgs\),", src):
        i, depth = m.end(), 1                       # parse the rest of the ccall argument list
        j = i
        while depth:
            depth += src[j] in "({["
            depth -= src[j] in ")}]"
            j += 1
        parts = _split_top(src[i:j - 1])
        ret, argt = parts[0], parts[1]
        assert argt.startswith("(") and argt.endswith(")"), (m.group(1), argt)
        inner = argt[1:-1].strip()
        if inner.endswith(","):
            inner = inner[:-1]
        types = _split_top(inner) if inner else []
        calls.append((m.group(1), ret, types, len(parts) - 2))
    return calls


def _compatible(ck, jk, name, what):
    kind_c, pt_c = ck
    kind_j, pt_j = jk
    assert kind_c == kind_j, f"{name}: {what}: C {ck} vs Julia {jk}"
    if kind_c != "ptr":
        return
    base_c, n_c = pt_c
    base_j, n_j = pt_j
    assert n_c == n_j, f"{name}: {what}: pointer depth C {pt_c} vs Julia {pt_j}"
    if base_c == "gs_ctx":
        assert base_j == "void", f"{name}: {what}: gs_ctx* must be Ptr{{Cvoid}}"
    elif base_c == "void" or base_j == "void":
        pass                                          # void* takes any pointer; Ptr{Cvoid} may carry any device pointer
    elif base_c == "char":
        assert base_j == "char"
    else:
        assert base_c == base_j, f"{name}: {what}: pointee C {base_c} vs Julia {base_j}"


def test_every_ccall_matches_its_prototype():
    protos = _c_prototypes()
    calls = _julia_ccalls()
    assert len(protos) >= 30 and len(calls) >= 35
    for name, ret, types, nvals in calls:
        assert name in protos, f"ccall of {name}: not declared in include/gsplat.h"
        cret, cargs = protos[name]
        assert len(types) == len(cargs), f"{name}: {len(types)} ccall argument types, prototype has {len(cargs)}"
        assert nvals == len(types), f"{name}: {nvals} values passed for {len(types)} argument types"
        _compatible(_c_kind(cret), _j_kind(ret), name, "return")
        for k, (ca, jt) in enumerate(zip(cargs, types)):
            _compatible(_c_kind(ca), _j_kind(jt), name, f"argument {k} ({ca})")


def test_every_exported_function_is_bound():
    protos = _c_prototypes()
    bound = {c[0] for c in _julia_ccalls()}
    missing = sorted(set(protos) - bound)
    assert not missing, f"declared in include/gsplat.h but never ccall'ed in julia/backend.jl: {missing}"


def _c_struct_fields(name):
    src = _strip_comments(open(os.path.join(ROOT, "include", "gsplat.h")).read())
    m = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*" + name + r"\s*;", src)
    fields = []
    for line in m.group(1).split(";"):
        line = line.strip()
        if not line:
            continue
        fm = re.fullmatch(r"([A-Za-z_0-9]+)\s*(\*?)\s*([A-Za-z_0-9]+)(?:\[(\d+)\])?", line)
        assert fm, line
        fields.append((fm.group(3), fm.group(1) + fm.group(2), int(fm.group(4) or 1)))
    return fields


def _julia_struct_fields(name):
    src = open(os.path.join(ROOT, "julia", "backend.jl")).read()
    m = re.search(r"struct\s+" + name + r"\b[^\n]*\n(.*?)\nend", src, flags=re.S)
    fields = []
    for line in m.group(1).splitlines():
        line = re.sub(r"#.*$", "", line).strip()
        if line:
            fname, ftype = [x.strip() for x in line.split("::")]
            fields.append((fname, ftype))
    return fields


def test_config_and_grads_structs_mirror_the_header():
    size = {"int32_t": 4, "float": 4, "float*": 8}
    cf = _c_struct_fields("gs_config")
    jf = _julia_struct_fields("GsConfig")
    assert [f[0] for f in cf] == [f[0] for f in jf], "gs_config field names/order differ"
    total = 0
    for (cn, ct, cnt), (jn, jt) in zip(cf, jf):
        total += size[ct] * cnt
        if cnt > 1:
            assert jt == f"NTuple{{{cnt}, {'Int32' if ct == 'int32_t' else 'Float32'}}}", (cn, jt)
        else:
            assert jt == {"int32_t": "Int32", "float": "Float32"}[ct], (cn, jt)
    assert total == 96, "sizeof(gs_config) is 96 in ABI version 2 (struct_size + abi_version guard)"
    src = open(os.path.join(ROOT, "julia", "backend.jl")).read()
    ctor = re.search(r"GsConfig\(([^\n]*)\)\n", src).group(1)
    assert len(_split_top(ctor)) == len(jf), "defaultConfig(): GsConfig constructor arity"
    gf = _c_struct_fields("gs_grads")
    jg = _julia_struct_fields("GsGrads")
    assert [f[0] for f in gf] == [f[0] for f in jg]
    assert all(t == "Ptr{Float32}" for _, t in jg) and all(t == "float*" for _, t, _ in gf)
    # the Python mirror agrees as well
    import ctypes as C
    from gaussiansplat_amd import backend
    assert [f[0] for f in backend.GsConfig._fields_] == [f[0] for f in cf] and C.sizeof(backend.GsConfig) == 96
    assert [f[0] for f in backend.GsGrads._fields_] == [f[0] for f in gf]


# ---- the reference's own signatures: julia/backend.jl must define METHODS on the reference's types so that
# src/examples/main.jl:32-34 runs unchanged.  The strings below are the reference's (held here: /root/reference does not travel):
#   src/forward.jl:35   function preprocess(renderer::GaussianRenderer3D)
#   src/forward.jl:118  function compactIdxs(renderer, threads, blocks)
#   src/forward.jl:163  function forward(renderer, tps, threads, blocks)
#   src/backward.jl:3   function backward(renderer, ΔC)
#   src/splat.jl:158    function resetGrads(splatData::SplatGrads2D) / (splatData::SplatData3D)
#   src/renderer.jl:151 function getRenderer(rendererTypeVal::Val{GAUSSIAN_3D}, path::String, imgSize::Tuple, threads::Tuple, blocks::Tuple)
REFERENCE_METHODS = {
    "preprocess": ["renderer::GaussianRenderer3D"],
    "compactIdxs": ["renderer::GaussianRenderer3D", "threads", "blocks"],
    "forward": ["renderer::GaussianRenderer3D", "tps", "threads", "blocks"],
    "backward": ["renderer::GaussianRenderer3D", "ΔC"],
    "resetGrads": ["grads::SplatGrads3D"],
}
# src/renderer.jl:205-219 (GaussianRenderer3D), src/splat.jl:36-52 (SplatData3D, SplatGrads3D)
RENDERER3D_FIELDS = ["splatData", "splatGrads", "imageData", "positions", "transmittance", "cov2ds", "cov3ds", "bbs", "invCov2ds",
                     "nGaussians", "hitIdxs", "camera", "sortIdxs"]
SPLATDATA3D_FIELDS = ["means", "scales", "shs", "quaternions", "opacities", "features"]
SPLATGRADS3D_FIELDS = ["Δmeans", "Δscales", "Δshs", "Δquaternions", "Δopacities", "Δfeatures"]


def _reference_half():
    src = open(os.path.join(ROOT, "julia", "backend.jl")).read()
    i = src.index("if @isdefined(GaussianRenderer3D)")
    return re.sub(r"(?m)#.*$", "", src[i:])


def test_reference_signatures_are_defined_on_the_reference_types():
    half = _reference_half()
    for name, args in REFERENCE_METHODS.items():
        m = re.search(r"(?m)^function " + name + r"\(([^)]*)\)", half)
        assert m, f"julia/backend.jl defines no method {name}(...) on the reference's types"
        got = [a.strip() for a in m.group(1).split(",")]
        assert got == args, f"{name}: arguments {got}, the reference's call sites need {args}"
    # the two getRenderer methods main.jl:14-27 / renderer.jl:164-186 dispatch to
    for second in ("path::String", "nGaussians::Int"):
        assert re.search(r"getRenderer\(rendererTypeVal::Val\{GAUSSIAN_3D\}, " + re.escape(second) + r", imgSize::Tuple, threads::Tuple, blocks::Tuple\)", half), second


def test_reference_methods_use_the_reference_field_names():
    half = _reference_half()
    used_r = set(re.findall(r"\brenderer\.([A-Za-z0-9_]+)", half))
    assert used_r and used_r <= set(RENDERER3D_FIELDS), f"not fields of GaussianRenderer3D: {sorted(used_r - set(RENDERER3D_FIELDS))}"
    for need in ("splatData", "splatGrads", "imageData", "transmittance", "camera", "nGaussians"):
        assert need in used_r, need
    used_d = set(re.findall(r"\b(?:d|splatData|renderer\.splatData)\.([a-z]+)\b", half)) - {"hr"}
    assert {"means", "scales", "quaternions", "opacities", "shs"} <= used_d <= set(SPLATDATA3D_FIELDS), used_d
    used_g = set(re.findall(r"\b(?:g|grads)\.(Δ[a-z]+)", half))
    assert set(SPLATGRADS3D_FIELDS) - {"Δfeatures"} <= used_g <= set(SPLATGRADS3D_FIELDS), used_g
    # the constructors are called with one value per field, in the reference's field order
    m = re.search(r"GaussianRenderer3D\((.*?)\)\nend", half, flags=re.S)
    assert m and len(_split_top(m.group(1))) == len(RENDERER3D_FIELDS)
    m = re.search(r"grads = SplatGrads3D\((.*?)\)\n", half, flags=re.S)
    order = re.findall(r"splatData\.([a-z]+)\)\)", m.group(1))
    assert order == ["means", "scales", "shs", "quaternions", "opacities"], order      # = SPLATGRADS3D_FIELDS without the Δ
    # forward writes in place into the renderer's own arrays; backward accumulates (+=) into the renderer's gradient arrays
    assert "hip_forward!(hipSide(renderer).hr, renderer.imageData, renderer.transmittance)" in half
    for f in SPLATGRADS3D_FIELDS[:-1]:
        assert re.search(r"g\." + f + r" \.\+= ", half), f
