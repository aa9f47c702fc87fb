# backend.jl -- drop-in body for the reference's EMPTY src/backend.jl (/root/reference/src/backend.jl:1).
#
# Thin ccall glue over libgsplat_hip.so (include/gsplat.h).  NOT executed in the build container
# (no Julia toolchain there); the Python ctypes binding gaussiansplat_amd/backend.py binds the same
# symbols 1:1 and is what the tests drive.  Arrays keep the reference's column-major layouts, so
# `pointer(A)` is passed unpermuted: means 3xN, scales 3xN, quaternions 4xN, opacities 1xN,
# shs (3K)xN, image WxHx3, T/P 4x4.
#
# A maintainer copies this file over the reference's empty src/backend.jl and adds ONE line, `include("backend.jl")`, at the
# end of src/GaussianSplat.jl (after `include("projection.jl")`, i.e. after renderer.jl has pulled in splat.jl, forward.jl and
# backward.jl).  The second half of this file then defines METHODS WITH THE REFERENCE'S OWN SIGNATURES on the reference's own types
#   getRenderer(Val(GAUSSIAN_3D), path | nGaussians, imgSize, threads, blocks)      src/renderer.jl:119-186
#   preprocess(renderer::GaussianRenderer3D)                                        src/forward.jl:35
#   compactIdxs(renderer::GaussianRenderer3D, threads, blocks)                      src/forward.jl:118
#   forward(renderer::GaussianRenderer3D, tps, threads, blocks)                     src/forward.jl:163
#   backward(renderer::GaussianRenderer3D, ΔC)                                      src/backward.jl:3
#   resetGrads(grads::SplatGrads3D)                                                 src/splat.jl:158-173
# which replace the CUDA.jl bodies (later definitions of the same signature win), so src/examples/main.jl:14-34 runs unchanged:
# the renderer's fields are host `Array`s (there is no CUDA.jl on this target) written in place, and the HipRenderer that holds the
# resident model lives in a side table keyed by the renderer object.  The `hip_*` functions of module HipBackend remain the
# explicit, allocation-free interface (device pointers, view slots, fused SGD, multi-GPU).

module HipBackend

const libgs = get(ENV, "GSPLAT_HIP_LIB", "libgsplat_hip.so")

const GS_MEM_HOST = Cint(0)
const GS_MEM_DEVICE = Cint(1)
@enum GsOrder::Cint GS_ORDER_INDEX = 0 GS_ORDER_DEPTH_DESC = 1 GS_ORDER_DEPTH_ASC = 2

const GS_ABI_VERSION = Cint(3)               # include/gsplat.h: the header this glue is written against

mutable struct GsConfig                      # must mirror gs_config (96 bytes)
    struct_size::Int32
    abi_version::Int32
    tile_size::Int32
    order::Int32
    t_min::Float32
    deterministic::Int32
    export_debug::Int32
    profile_stages::Int32
    bin_path::Int32
    rank_mode::Int32
    alpha_cull::Int32
    schedule::Int32
    slab_mode::Int32
    slab_max_ratio::Float32
    slab_fractions::NTuple{3, Float32}
    debug_flags::Int32
    depth_sort::Int32
    list_cap::Int32                          # capped tile lists: 0 automatic, 1 never, 2 also on small grids
    tile_parts::Int32                        # waves per tile on small grids: 0 automatic, 1, 2, 4
    reserved::NTuple{3, Int32}
end

struct GsGrads                               # gs_grads: device pointers, may be C_NULL
    d_means::Ptr{Float32}
    d_scales::Ptr{Float32}
    d_quats::Ptr{Float32}
    d_opacities::Ptr{Float32}
    d_shs::Ptr{Float32}
end

mutable struct HipRenderer
    ctx::Ptr{Cvoid}
    n::Int
    shDegree::Int
    W::Int
    H::Int
end

function check(r::HipRenderer, rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:gs_last_error, libgs), Cstring, (Ptr{Cvoid},), r.ctx))
    error("libgsplat_hip error $rc: $msg")
end

function defaultConfig()
    cfg = GsConfig(0, 0, 0, 0, 0f0, 0, 0, 0, 0, 0, 0, 0, 0, 0f0, ntuple(_ -> 0f0, 3), 0, 0, 0, 0, ntuple(_ -> Int32(0), 3))
    ccall((:gs_default_config, libgs), Cvoid, (Ref{GsConfig},), cfg)
    # a library built from another header would read this struct with shifted fields: refuse it here, loudly
    (hip_abiVersion() == GS_ABI_VERSION && cfg.abi_version == GS_ABI_VERSION && cfg.struct_size == sizeof(GsConfig)) ||
        error("libgsplat_hip reports ABI $(hip_abiVersion()), this glue is written for $(GS_ABI_VERSION): rebuild one of them")
    return cfg
end

# getRenderer(...)  (src/renderer.jl:119-149): host arrays are copied once and stay resident
function hip_getRenderer(means::Matrix{Float32}, scales::Matrix{Float32}, quaternions::Matrix{Float32},
                         opacities::Matrix{Float32}, shs::Matrix{Float32}, imgSize; device = 0, cfg = defaultConfig())
    ctxref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:gs_create, libgs), Cint, (Ref{Ptr{Cvoid}}, Cint, Ref{GsConfig}), ctxref, device, cfg)
    rc == 0 || error("gs_create failed: " * unsafe_string(ccall((:gs_last_error, libgs), Cstring, (Ptr{Cvoid},), C_NULL)))
    n = size(means, 2)
    K = div(size(shs, 1), 3)
    deg = isqrt(K) - 1
    r = HipRenderer(ctxref[], n, deg, imgSize[1], imgSize[2])
    check(r, ccall((:gs_set_model, libgs), Cint,
                   (Ptr{Cvoid}, Int64, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cint),
                   r.ctx, n, deg, means, scales, quaternions, opacities, shs, GS_MEM_HOST))
    finalizer(x -> ccall((:gs_destroy, libgs), Cint, (Ptr{Cvoid},), x.ctx), r)
    return r
end

# getRenderer(Val(GAUSSIAN_2D), ...)  (src/renderer.jl:38-82) with SplatData2D arrays (src/splat.jl:20-26): means 2xN in
# [0,1]^2, scales 2xN (log), rotations 1xN, opacities 1xN, colors 3xN.  Gradient slots of GsGrads are then read as
# SplatGrads2D: d_means 2xN, d_scales 2xN, d_quats -> rotations 1xN, d_opacities 1xN, d_shs -> colors 3xN.
function hip_getRenderer2D(means::Matrix{Float32}, scales::Matrix{Float32}, rotations::Matrix{Float32},
                           opacities::Matrix{Float32}, colors::Matrix{Float32}, imgSize; device = 0, cfg = defaultConfig())
    ctxref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:gs_create, libgs), Cint, (Ref{Ptr{Cvoid}}, Cint, Ref{GsConfig}), ctxref, device, cfg)
    rc == 0 || error("gs_create failed: " * unsafe_string(ccall((:gs_last_error, libgs), Cstring, (Ptr{Cvoid},), C_NULL)))
    n = size(means, 2)
    r = HipRenderer(ctxref[], n, 0, imgSize[1], imgSize[2])
    check(r, ccall((:gs_set_model_2d, libgs), Cint,
                   (Ptr{Cvoid}, Int64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cint),
                   r.ctx, n, means, scales, rotations, opacities, colors, GS_MEM_HOST))
    check(r, ccall((:gs_set_image_size, libgs), Cint, (Ptr{Cvoid}, Int32, Int32), r.ctx, r.W, r.H))
    finalizer(x -> ccall((:gs_destroy, libgs), Cint, (Ptr{Cvoid},), x.ctx), r)
    return r
end

# preprocess(renderer::GaussianRenderer2D)  (src/forward.jl:9-33): no camera
hip_preprocess2D(r::HipRenderer) = check(r, ccall((:gs_preprocess, libgs), Cint, (Ptr{Cvoid},), r.ctx))

# preprocess(renderer)  (src/forward.jl:35-111); T, P from computeTransform/computeProjection (.linear)
function hip_preprocess(r::HipRenderer, camera, T::AbstractMatrix, P::AbstractMatrix)
    Tm = Matrix{Float32}(T); Pm = Matrix{Float32}(P)
    eye = Vector{Float32}(camera.eye); lookAt = Vector{Float32}(camera.lookAt)
    check(r, ccall((:gs_set_camera, libgs), Cint,
                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float32}, Int32, Int32),
                   r.ctx, Tm, Pm, camera.fx, camera.fy, camera.near, camera.far, eye, lookAt, r.W, r.H))
    check(r, ccall((:gs_preprocess, libgs), Cint, (Ptr{Cvoid},), r.ctx))
end

# optional, before hip_preprocess: name the view about to be rendered (e.g. the `id` of the camera, src/camera.jl:10-22,119-151).
# A training loop cycles over a fixed camera set; the library launches the forward's tiles heaviest-first by what the last frame
# rendered under the same slot measured.  Speed only; slot < 0: none.
hip_setViewSlot(r::HipRenderer, slot::Integer) =
    check(r, ccall((:gs_set_view_slot, libgs), Cint, (Ptr{Cvoid}, Int32), r.ctx, slot))

# compactIdxs(renderer, threads, blocks)  (src/forward.jl:118-161)
hip_compactIdxs(r::HipRenderer, threads, blocks) =
    check(r, ccall((:gs_bin, libgs), Cint, (Ptr{Cvoid}, Int32, Int32), r.ctx, blocks[1], blocks[2]))

# optional: the forward writes directly into caller-owned DEVICE buffers (e.g. a ROCArray's pointer); C_NULL, C_NULL unbinds
function hip_bind_outputs(r::HipRenderer, image_dev::Ptr{Float32}, transmittance_dev::Ptr{Float32})
    check(r, ccall((:gs_bind_outputs, libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), r.ctx, image_dev, transmittance_dev))
end

# forward(renderer, tps, threads, blocks)  (src/forward.jl:163-198): fills imageData (W x H x 3), transmittance (W x H)
function hip_forward!(r::HipRenderer, imageData::Array{Float32, 3}, transmittance::Array{Float32, 2})
    check(r, ccall((:gs_forward, libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Cint),
                   r.ctx, imageData, transmittance, GS_MEM_HOST))
end

# backward + `param .-= lr * grad` (src/train.jl:42-46) fused: the resident model is updated in place, no gradient arrays
function hip_backward_sgd!(r::HipRenderer, ΔC::Array{Float32, 3}, lr::Float32)
    check(r, ccall((:gs_backward_sgd, libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}, Cint, Cfloat), r.ctx, ΔC, GS_MEM_HOST, lr))
end

# initGrads (src/splat.jl:137-156): one flat zeroed device buffer owned by the library
function hip_initGrads(r::HipRenderer)
    g = Ref(GsGrads(C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
    check(r, ccall((:gs_grads_alloc, libgs), Cint, (Ptr{Cvoid}, Ref{GsGrads}), r.ctx, g))
    return g[]
end

# copy the accumulated gradients into host arrays shaped like the parameters
function hip_readGrads!(r::HipRenderer, grads::GsGrads, Δmeans, Δscales, Δquaternions, Δopacities, Δshs)
    check(r, ccall((:gs_grads_read, libgs), Cint,
                   (Ptr{Cvoid}, Ref{GsGrads}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                   r.ctx, grads, Δmeans, Δscales, Δquaternions, Δopacities, Δshs))
end

# backward(renderer, ΔC)  (src/backward.jl:3-38): accumulates into DEVICE gradient arrays
hip_backward!(r::HipRenderer, ΔC::Array{Float32, 3}, grads::GsGrads) =
    check(r, ccall((:gs_backward, libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}, Cint, Ref{GsGrads}), r.ctx, ΔC, GS_MEM_HOST, grads))

# multi-GPU (one process per GPU): rank 0 creates the id, every rank joins, one all-reduce per step
function hip_commUniqueId()
    id = zeros(UInt8, 128)
    rc = ccall((:gs_comm_unique_id, libgs), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("gs_comm_unique_id failed")
    return id
end
hip_commInit(r::HipRenderer, rank, nranks, id::Vector{UInt8}) =
    check(r, ccall((:gs_comm_init, libgs), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), r.ctx, rank, nranks, id))
hip_allreduceGrads!(r::HipRenderer, grads::GsGrads) =
    check(r, ccall((:gs_allreduce_grads, libgs), Cint, (Ptr{Cvoid}, Ref{GsGrads}), r.ctx, grads))

# colour-factored exchange (optional, same gradients with ~2.6x less xGMI traffic): per view hip_backward! with
# grads.d_shs = C_NULL and hip_packColorGrads!; per step all-reduce the 11N geometry floats, all-gather the packed
# d rgb (3N per view) and rebuild the SH gradients from all views.  camRecords: 38 x nviews Float32 {T16, P16, eye3, lookAt3}.
hip_packColorGrads!(r::HipRenderer, drgb::Ptr{Float32}) =
    check(r, ccall((:gs_color_grads_pack, libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.ctx, drgb))
hip_shGradsFromViews!(r::HipRenderer, camRecords::Matrix{Float32}, drgbAll::Ptr{Float32}, Δshs::Ptr{Float32}; overwrite = true) =
    check(r, ccall((:gs_sh_grads_from_views, libgs), Cint, (Ptr{Cvoid}, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cint),
                   r.ctx, size(camRecords, 2), camRecords, drgbAll, Δshs, overwrite ? 1 : 0))

# resetGrads(renderer.splatGrads)  (src/splat.jl:158-173)
hip_resetGrads!(r::HipRenderer, grads::GsGrads) =
    check(r, ccall((:gs_reset_grads, libgs), Cint, (Ptr{Cvoid}, Ref{GsGrads}), r.ctx, grads))

# ---- the step after backward (src/train.jl:39-46, src/loss.jl:62-72) ---------------------------------------------

# backward with flags: GS_BWD_OVERWRITE (first backward after a reset: store instead of accumulate), GS_BWD_COMPOSITE_ONLY /
# GS_BWD_PARAMS_ONLY (split backward of the colour-factored multi-GPU step).  ΔC and grads are DEVICE pointers here.
const GS_BWD_OVERWRITE = Cint(1)
const GS_BWD_COMPOSITE_ONLY = Cint(2)
const GS_BWD_PARAMS_ONLY = Cint(4)
const GS_BWD_PARAMS_SH = Cint(8)        # with GS_BWD_PARAMS_ONLY: only the SH / colour kernel (then all-reduce Δshs while ...)
const GS_BWD_PARAMS_GEOM = Cint(16)     # ... with GS_BWD_PARAMS_ONLY: only the geometry chain
hip_backwardEx!(r::HipRenderer, ΔC::Ptr{Float32}, grads::GsGrads, flags::Integer) =
    check(r, ccall((:gs_backward_ex, libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}, Cint, Ref{GsGrads}, Cint),
                   r.ctx, ΔC, GS_MEM_DEVICE, grads, flags))

# loss = 0.9 L1/2 + 0.1 DSSIM/2 (src/loss.jl:62-72, lam = 0.1) and its gradient ΔC w.r.t. the rendered image; host arrays W x H x C
function hip_lossL1Dssim!(r::HipRenderer, img::Array{Float32, 3}, gt::Array{Float32, 3}, ΔC::Array{Float32, 3}; lam = 0.1f0)
    loss = Ref{Cdouble}(0.0)
    check(r, ccall((:gs_loss_l1_dssim, libgs), Cint,
                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Int32, Int32, Cfloat, Ptr{Float32}, Ref{Cdouble}, Cint),
                   r.ctx, img, gt, size(img, 1), size(img, 2), size(img, 3), lam, ΔC, loss, GS_MEM_HOST))
    return loss[]
end

# params .-= lr .* grads on the resident model (src/train.jl:42-46)
hip_sgdStep!(r::HipRenderer, lr::Real, grads::GsGrads) =
    check(r, ccall((:gs_sgd_step, libgs), Cint, (Ptr{Cvoid}, Cfloat, Ref{GsGrads}), r.ctx, lr, grads))

# ---- plumbing and introspection -----------------------------------------------------------------------------------

# enqueue on an existing hipStream_t (C_NULL: the ctx's own stream)
hip_setStream(r::HipRenderer, stream::Ptr{Cvoid}) =
    check(r, ccall((:gs_set_stream, libgs), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), r.ctx, stream))
hip_synchronize(r::HipRenderer) = check(r, ccall((:gs_synchronize, libgs), Cint, (Ptr{Cvoid},), r.ctx))
hip_commDestroy(r::HipRenderer) = check(r, ccall((:gs_comm_destroy, libgs), Cint, (Ptr{Cvoid},), r.ctx))

hip_numGaussians(r::HipRenderer) = ccall((:gs_num_gaussians, libgs), Int64, (Ptr{Cvoid},), r.ctx)
hip_numInstances(r::HipRenderer) = ccall((:gs_num_instances, libgs), Int64, (Ptr{Cvoid},), r.ctx)
hip_numCoarseInstances(r::HipRenderer) = ccall((:gs_num_coarse_instances, libgs), Int64, (Ptr{Cvoid},), r.ctx)
hip_abiVersion() = ccall((:gs_abi_version, libgs), Cint, ())
hip_numRounds(r::HipRenderer) = ccall((:gs_num_rounds, libgs), Cint, (Ptr{Cvoid},), r.ctx)
# tile lists of the last frame: (entries written, list segments appended by composite waves, 1 if the lists were capped)
function hip_listStats(r::HipRenderer)
    out = zeros(Int64, 3)
    check(r, ccall((:gs_get_list_stats, libgs), Cint, (Ptr{Cvoid}, Ptr{Int64}), r.ctx, out))
    return out
end

# waves per tile (1, 2, 4) of the last frame's composite launches (GsConfig.tile_parts)
hip_tileParts(r::HipRenderer) = ccall((:gs_get_tile_parts, libgs), Cint, (Ptr{Cvoid},), r.ctx)
# the path that built the last frame's tile lists: 0 two-level, 1 / 2 radix, 3 the small-frame path (GsConfig.bin_path)
hip_binPath(r::HipRenderer) = ccall((:gs_get_bin_path, libgs), Cint, (Ptr{Cvoid},), r.ctx)

# renderer scratch arrays (gs_array ids of include/gsplat.h; e.g. 11 = sortIdxs, 12 = tile ranges, 13 = sorted ids) into a host array
function hip_getArray!(r::HipRenderer, which::Integer, dst::Array)
    check(r, ccall((:gs_get_array, libgs), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64), r.ctx, which, dst, sizeof(dst)))
    return dst
end

# per-stage hipEvent times of the last frame (gs_config.profile_stages = 1), GS_STAGE_COUNT = 9 entries, milliseconds
function hip_stageTimes(r::HipRenderer)
    ms = zeros(Float32, 9)
    check(r, ccall((:gs_get_stage_times, libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.ctx, ms))
    return ms
end
function hip_stageStats(r::HipRenderer; reset = false)
    sums = zeros(Float64, 9); counts = zeros(Int64, 9)
    check(r, ccall((:gs_get_stage_stats, libgs), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Cint), r.ctx, sums, counts, reset ? 1 : 0))
    return sums, counts
end
function hip_workCounters(r::HipRenderer)
    out = zeros(Int64, 4)           # walked_fwd, walked_bwd, evaluated_fwd, evaluated_bwd
    check(r, ccall((:gs_get_work_counters_ex, libgs), Cint, (Ptr{Cvoid}, Ptr{Int64}), r.ctx, out))
    return out
end
function hip_workCountersWalked(r::HipRenderer)
    f = Ref{Int64}(0); b = Ref{Int64}(0)
    check(r, ccall((:gs_get_work_counters, libgs), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), r.ctx, f, b))
    return f[], b[]
end
function hip_debugTimeComposite(r::HipRenderer, which::Integer, variant::Integer, reps::Integer)
    ms = Ref{Cfloat}(0f0)
    check(r, ccall((:gs_debug_time_composite, libgs), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ref{Cfloat}), r.ctx, which, variant, reps, ms))
    return ms[]
end
function hip_debugTileClock(r::HipRenderer, which::Integer, variant::Integer, ntiles::Integer)
    out = zeros(UInt64, 6, ntiles)
    check(r, ccall((:gs_debug_tile_clock, libgs), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt64}), r.ctx, which, variant, out))
    return out
end
hip_debugTileClockRows(r::HipRenderer) = ccall((:gs_debug_tile_clock_rows, libgs), Cint, (Ptr{Cvoid},), r.ctx)
function hip_clockMHz(r::HipRenderer)
    mhz = Ref{Cfloat}(0f0)
    check(r, ccall((:gs_debug_clock_mhz, libgs), Cint, (Ptr{Cvoid}, Ref{Cfloat}), r.ctx, mhz))
    return mhz[]
end

hip_rankProbeResult(r::HipRenderer) = ccall((:gs_rank_probe_result, libgs), Cint, (Ptr{Cvoid},), r.ctx)

# profiling: the debug launches cover only order[start+1 : start+len] of the frame's launch order (0, 0: all of it)
hip_debugSetWindow(r::HipRenderer, start::Integer, len::Integer) =
    check(r, ccall((:gs_debug_set_window, libgs), Cint, (Ptr{Cvoid}, Int32, Int32), r.ctx, start, len))

end # module

# ======================================================================================================================
# The reference's own API on the reference's own types (see the header of this file).  Only defined when this file is
# included where those types exist, i.e. at the end of src/GaussianSplat.jl.
# ======================================================================================================================
if @isdefined(GaussianRenderer3D)

using .HipBackend: HipRenderer, GsGrads, hip_getRenderer, hip_preprocess, hip_compactIdxs, hip_forward!, hip_initGrads,
                   hip_readGrads!, hip_resetGrads!, hip_getArray!, defaultConfig

# state the reference keeps in CuArrays hung off the renderer: here one HipRenderer (resident model, scratch, gradient buffer) per
# reference renderer, dropped with it (weak keys; HipRenderer's finalizer calls gs_destroy)
mutable struct HipSide
    hr::HipRenderer
    grads::Union{Nothing, GsGrads}
    uploaded::Bool
end
const HIP_SIDE = WeakKeyDict{Any, HipSide}()
# The reference uploads the model on every call (`|> CuArray`, src/forward.jl:63-69,169-170), so host-side edits of
# renderer.splatData between frames (src/train.jl:42-46) are always seen.  true keeps that; false uploads once and relies on
# hipModelChanged!(renderer) after an edit.
const HIP_REUPLOAD_EVERY_FRAME = Ref(true)
# true: preprocess / compactIdxs also fill the reference's scratch fields (positions, cov2ds, cov3ds, invCov2ds, bbs, sortIdxs)
# and return the real tps (gs_config.export_debug: 124 more bytes per gaussian written per frame)
const HIP_EXPORT_SCRATCH = Ref(false)

hipModelChanged!(renderer::GaussianRenderer3D) = (haskey(HIP_SIDE, renderer) && (HIP_SIDE[renderer].uploaded = false); nothing)

function hipSide(renderer::GaussianRenderer3D)
    get!(HIP_SIDE, renderer) do
        d = renderer.splatData
        cfg = defaultConfig()
        cfg.export_debug = HIP_EXPORT_SCRATCH[] ? 1 : 0
        hr = hip_getRenderer(Matrix{Float32}(d.means), Matrix{Float32}(d.scales), Matrix{Float32}(d.quaternions),
                             Matrix{Float32}(d.opacities), Matrix{Float32}(d.shs), size(renderer.imageData); cfg = cfg)
        HipSide(hr, nothing, true)
    end
end

# getRenderer (src/renderer.jl:119-149 from a gaussian count, :151-186 from a .ply): same fields, host arrays instead of CuArrays
function hostRenderer3D(splatData::SplatData3D, imgSize::Tuple)
    n = length(splatData.opacities)
    feat = splatData.features === nothing ? nothing : zeros(Float32, size(splatData.features))
    # by FIELD name (the reference's initGrads passes them in another order than SplatGrads3D declares, src/splat.jl:137-156)
    grads = SplatGrads3D(zeros(Float32, size(splatData.means)), zeros(Float32, size(splatData.scales)), zeros(Float32, size(splatData.shs)),
                         zeros(Float32, size(splatData.quaternions)), zeros(Float32, size(splatData.opacities)), feat)
    return GaussianRenderer3D(splatData, grads, zeros(Float32, imgSize...), nothing, ones(Float32, imgSize[1:end-1]...),
                              zeros(Float32, 2, 2, n), zeros(Float32, 3, 3, n), zeros(Float32, 2, 2, n), zeros(Float32, 2, 2, n),
                              n, nothing, nothing, nothing)
end
getRenderer(rendererTypeVal::Val{GAUSSIAN_3D}, path::String, imgSize::Tuple, threads::Tuple, blocks::Tuple) =
    hostRenderer3D(initData(Val(SPLAT3D), path), imgSize)
function getRenderer(rendererTypeVal::Val{GAUSSIAN_3D}, nGaussians::Int, imgSize::Tuple, threads::Tuple, blocks::Tuple)
    # initData(Val(SPLAT3D), n) (src/splat.jl:91-104) with host arrays
    data = SplatData3D(rand(Float32, 3, nGaussians), rand(Float32, 3, nGaussians), rand(Float32, 9, nGaussians),
                       rand(Float32, 4, nGaussians), rand(Float32, 1, nGaussians), zeros(Float32, 0, nGaussians))
    return hostRenderer3D(data, imgSize)
end

# preprocess(renderer)  src/forward.jl:35-111
function preprocess(renderer::GaussianRenderer3D)
    side = hipSide(renderer)
    hr = side.hr
    d = renderer.splatData
    if !side.uploaded || HIP_REUPLOAD_EVERY_FRAME[]
        HipBackend.check(hr, ccall((:gs_set_model, HipBackend.libgs), Cint,
                                   (Ptr{Cvoid}, Int64, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cint),
                                   hr.ctx, hr.n, hr.shDegree, d.means, d.scales, d.quaternions, d.opacities, d.shs, HipBackend.GS_MEM_HOST))
        side.uploaded = true
    end
    (w, h) = size(renderer.imageData)[1:2]
    camera = defaultCamera()                                        # src/forward.jl:53 (the cameras.json path there is commented out)
    hip_preprocess(hr, camera, computeTransform(camera).linear, computeProjection(camera, w, h).linear)
    renderer.camera = camera
    n = renderer.nGaussians
    tps = zeros(Float32, 4, n)
    if HIP_EXPORT_SCRATCH[]
        hip_getArray!(hr, 1, tps)                                   # GS_ARR_TPS
        renderer.positions = hip_getArray!(hr, 2, zeros(Float32, 2, n))            # GS_ARR_MU
        hip_getArray!(hr, 3, renderer.cov3ds); hip_getArray!(hr, 4, renderer.cov2ds)
        hip_getArray!(hr, 5, renderer.invCov2ds); hip_getArray!(hr, 6, renderer.bbs)
    end
    return tps          # (forward ignores it: the per-view payload stays on the device)
end

# compactIdxs(renderer, threads, blocks)  src/forward.jl:118-161: depth order + per-tile lists (kept on the device as ranges + ids;
# the reference's dense hitIdxs[tx, ty, slot] is not materialised)
function compactIdxs(renderer::GaussianRenderer3D, threads, blocks)
    hr = hipSide(renderer).hr
    hip_compactIdxs(hr, threads, blocks)
    if HIP_EXPORT_SCRATCH[]
        renderer.sortIdxs = hip_getArray!(hr, 11, zeros(UInt32, renderer.nGaussians)) .+ UInt32(1)      # GS_ARR_SORT_IDXS is 0-based
    end
    return nothing
end

# forward(renderer, tps, threads, blocks)  src/forward.jl:163-198: writes renderer.imageData and renderer.transmittance in place
function forward(renderer::GaussianRenderer3D, tps, threads, blocks)
    hip_forward!(hipSide(renderer).hr, renderer.imageData, renderer.transmittance)
    return nothing
end

# backward(renderer, ΔC)  src/backward.jl:3-38: accumulates into renderer.splatGrads.Δ* (same shapes as the parameters)
function backward(renderer::GaussianRenderer3D, ΔC)
    side = hipSide(renderer)
    hr = side.hr
    g = renderer.splatGrads
    d = renderer.splatData
    for (Δ, p, name) in ((g.Δmeans, d.means, "Δmeans"), (g.Δscales, d.scales, "Δscales"), (g.Δquaternions, d.quaternions, "Δquaternions"),
                         (g.Δopacities, d.opacities, "Δopacities"), (g.Δshs, d.shs, "Δshs"))
        size(Δ) == size(p) || error("renderer.splatGrads.$name has size $(size(Δ)), its parameter $(size(p)): build the gradients by " *
                                    "field name (hostRenderer3D); the reference's initGrads passes them in another order (src/splat.jl:148-155)")
    end
    side.grads === nothing && (side.grads = hip_initGrads(hr))
    dev = side.grads
    # one frame's gradients on the device (stored, not accumulated: GS_BWD_OVERWRITE = 1), then += on the host arrays
    HipBackend.check(hr, ccall((:gs_backward_ex, HipBackend.libgs), Cint, (Ptr{Cvoid}, Ptr{Float32}, Cint, Ref{GsGrads}, Cint),
                               hr.ctx, Array{Float32, 3}(ΔC), HipBackend.GS_MEM_HOST, dev, HipBackend.GS_BWD_OVERWRITE))
    tm, ts, tq, to, tsh = similar(g.Δmeans), similar(g.Δscales), similar(g.Δquaternions), similar(g.Δopacities), similar(g.Δshs)
    hip_readGrads!(hr, dev, tm, ts, tq, to, tsh)
    g.Δmeans .+= tm; g.Δscales .+= ts; g.Δquaternions .+= tq; g.Δopacities .+= to; g.Δshs .+= tsh
    return nothing
end

# resetGrads(renderer.splatGrads)  src/splat.jl:158-173 (the reference's 3-D method is typed on SplatData3D, which has no Δ fields)
function resetGrads(grads::SplatGrads3D)
    grads.Δmeans .= 0; grads.Δquaternions .= 0; grads.Δscales .= 0; grads.Δshs .= 0; grads.Δopacities .= 0
    grads.Δfeatures === nothing || (grads.Δfeatures .= 0)
    return nothing
end

end # if @isdefined(GaussianRenderer3D)
