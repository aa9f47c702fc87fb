/*
 * gsplat.h -- C ABI of libgsplat_hip.so, the MI355X (gfx950) differentiable Gaussian-splat
 * rasterizer that sits behind arhik/GaussianSplat's (empty) backend.jl.
 *
 * The reference defines no FFI: src/backend.jl:1 is empty and the de-facto surface is five
 * Julia functions over a mutable renderer struct.  Each entry point below names the
 * reference function it replaces (paths relative to /root/reference):
 *
 *   gs_create / gs_destroy    getRenderer(...)                    src/renderer.jl:119-149,164-186
 *   gs_set_model              initData + `|> CuArray` uploads     src/splat.jl:106-119, src/forward.jl:63-69,169-170
 *   gs_set_camera             defaultCamera/computeTransform/...  src/forward.jl:53-62, src/camera.jl:88-111
 *   gs_set_model_2d           initData(Val(SPLAT2D)) / SplatData2D src/splat.jl:20-26,74-87 (2-D image-fitting renderer)
 *   gs_set_image_size         size(renderer.imageData)            src/forward.jl:29
 *   gs_preprocess             preprocess(renderer)                src/forward.jl:35-111 (3-D), :9-33 (2-D)
 *   gs_bin                    compactIdxs(renderer,threads,blocks) src/forward.jl:118-161
 *   gs_forward                forward(renderer,tps,threads,blocks) src/forward.jl:163-198
 *   gs_backward               backward(renderer, dC)              src/backward.jl:3-38
 *   gs_reset_grads            resetGrads(renderer.splatGrads)     src/splat.jl:158-173
 *
 * Conventions
 *   - plain C: pointers + sizes, no exceptions; every function returns 0 (GS_OK) or a
 *     negative gs_status; gs_last_error(ctx) gives the message of the last failure.
 *   - array layouts are the reference's column-major ones so a Julia `pointer(A)` can be
 *     passed unpermuted: parameters [component, gaussian] (component fastest), images
 *     [x, y, channel] (x fastest, channel planes), T and P 4x4 column-major.
 *   - `mem` says where a caller buffer lives: GS_MEM_HOST (copied over PCIe) or
 *     GS_MEM_DEVICE (a HIP device pointer on the ctx's device; used in place).
 *   - the caller owns every buffer it passes; the library owns the ctx scratch (grow-only,
 *     no per-frame allocation once sizes are stable).
 *   - one ctx = one GPU = one stream; a ctx is not thread-safe, different ctxs are.
 *   - all work is enqueued on the ctx stream; functions that return data to HOST buffers
 *     synchronise that stream before returning, device-buffer variants do not.  gs_bin does not wait for the GPU either
 *     (from a ctx's second frame on): the tile lists are enqueued with the buffer sizes of the previous frame while the
 *     frame's instance counts are still on their way to the host; gs_forward reads them after it has enqueued the
 *     composite and, in the rare case that a list outgrew its buffer, rebuilds lists and image with larger ones.
 *   - there is NO CPU fallback: without a HIP device gs_create fails with GS_ERR_NO_DEVICE.
 */
#ifndef GSPLAT_H
#define GSPLAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: gs_config carries abi_version (gs_create refuses any other value: a caller built against another header fails loudly
 *    instead of running with shifted fields); schedule 0 now means "library default"; view slots (gs_set_view_slot);
 *    speculative binning (no host wait inside gs_bin); GS_BWD_PARAMS_SH / _GEOM; 64-byte payload and gradient rows.
 * 3: gs_config.list_cap (capped tile lists from the view slot's history; a former reserved word, sizeof unchanged); 4096 view
 *    slots; gs_get_list_stats.  (gs_config.tile_parts, a later reserved word: 0 = the library's choice, as reserved words must be.) */
#define GS_ABI_VERSION 3

typedef enum {
    GS_OK = 0,
    GS_ERR_INVALID = -1,      /* bad argument / call order                                  */
    GS_ERR_NO_DEVICE = -2,    /* no usable HIP device                                       */
    GS_ERR_HIP = -3,          /* a HIP runtime call failed (see gs_last_error)              */
    GS_ERR_OOM = -4,          /* device allocation failed                                   */
    GS_ERR_UNSUPPORTED = -5   /* e.g. tile size != 16, sh_degree > 3                        */
} gs_status;

typedef enum { GS_MEM_HOST = 0, GS_MEM_DEVICE = 1 } gs_mem;

/* Order of every per-tile splat list (what the radix sort realises).
 *   INDEX       literal reference behaviour: compact.jl:3-21 places a gaussian at slot
 *               inclusive-scan(hits) over the GAUSSIAN INDEX axis, so sortIdxs has no effect.
 *   DEPTH_DESC  the order forward.jl:103 computes (sortperm(-tps[3,:]), far -> near) and the
 *               TODO at forward.jl:123-131 intends; ties by gaussian index (stable).
 *   DEPTH_ASC   near -> far (extension).                                                      */
typedef enum { GS_ORDER_INDEX = 0, GS_ORDER_DEPTH_DESC = 1, GS_ORDER_DEPTH_ASC = 2 } gs_order;

typedef struct {
    int32_t struct_size;      /* = sizeof(gs_config); ABI guard                              */
    int32_t abi_version;      /* = GS_ABI_VERSION of the header the caller was built with    */
    int32_t tile_size;        /* threads=(16,16) of examples/main.jl:9; only 16 is supported */
    int32_t order;            /* gs_order; default GS_ORDER_DEPTH_DESC                       */
    float   t_min;            /* transmittance early-out: a pixel stops taking splats once
                                 its T < t_min.  0 = literal reference (splat.jl:224-261 has
                                 no early-out).  Default 1e-5 (pixel error <= t_min*max|rgb|) */
    int32_t deterministic;    /* 1: the per-(tile,splat) gradient sums are accumulated as 2^-40 fixed
                                 point (2^-28 for the second-order moments) with integer atomics (order independent ->
                                 bitwise reproducible run to run; absolute resolution 9e-13 / 3.7e-9 per add);
                                 0: float atomics (run-to-run differences in the last bits)    */
    int32_t export_debug;     /* 1: gs_preprocess also materialises the reference's scratch
                                 arrays (ts, tps, mu', cov3ds, cov2ds, invCov2ds, bbs) for
                                 gs_get_array; costs 124 extra bytes/gaussian of HBM writes   */
    int32_t profile_stages;   /* 1: record hipEvents around every stage (gs_get_stage_times; costs ~3 % of a C3 frame);
                                 2 + s: around stage s (gs_stage) only; 0: none */
    int32_t bin_path;         /* tile lists (same lists on every path, bit for bit): 0 (default) two-level binning -- lists per
                                 super-tile of 8 x 8 tiles from an LDS bitmap, tile lists as filtered copies -- and, for SMALL frames
                                 (up to 16 384 gaussians, 1024 tiles, 4 M gaussian x tile pairs: BASELINE C1), the whole of gs_bin in two
                                 launches: depth order + tile ranges inside one workgroup's LDS, lists by one workgroup per tile;
                                 3: two-level whatever the size (tests, A/B); 2: radix sort of instances generated in-pass (32-bit
                                 words); 1: explicit 64-bit tile|id instances + two radix passes */
    int32_t rank_mode;        /* radix-sort stable ranks: 1 (default) = wave64 ballots (portable); 0 = one LDS atomic-add-return per
                                 key -- its pre-values come back in ascending lane order on gfx950, which is an observed, not a
                                 documented property: gs_create CHECKS it on the device with a probe kernel and falls back to 1 if
                                 the check fails.  Same lists either way.  Since the two-level binning only the depth sort ranks
                                 keys at all; there the atomic form saves 4 us of a 1.44 ms C3 frame, so ballots are the default */
    int32_t alpha_cull;       /* 1 (default): while staging a tile's list the composite kernels drop every
                                 (tile, splat) entry whose largest alpha over that tile's pixels is below 2^-27
                                 -- a no-op in the reference's own fp32 arithmetic (T*(1-alpha) == T, colour term
                                 < 7.5e-9*|rgb|).  The lists (gs_bin) are unchanged.  0: evaluate every entry. */
    int32_t schedule;         /* composite kernels (speed only: every mode gives the same image and, up to atomic order, gradients):
                                 0 the library default (= 3);
                                 3 one wave per tile, plain launch over a longest-first permutation of the tiles (groups of 8 x 8
                                   tiles dealt to the XCDs): the backward by the forward's per-tile count of evaluated
                                   entries of the same frame (C3: 0.84 -> 0.74 ms); the forward by the count the last forward
                                   rendered under the same VIEW SLOT measured (gs_set_view_slot: training cycles over a fixed
                                   camera set), in tile order when the slot has no history yet;
                                 4 as 3, and a forward without slot history uses the work of this ctx's PREVIOUS forward
                                   (pays when consecutive frames see similar views);
                                 1 one wave per tile in launch (tile) order.
                                 (Persistent waves pulling tiles from per-XCD ticket counters were measured slower and are gone:
                                 profiles/HISTORY.md.)                                                                          */
    int32_t slab_mode;        /* binning in depth slabs (speed only; image, transmittance and deterministic-mode gradients are
                                 bit-identical either way): 1 (default) automatic -- when the previous frame walked under 3 % of
                                 its tile instances before the transmittance early-out stopped every tile (with the two-level
                                 binning a single round is faster above that share), the next frame is
                                 binned in three rounds over slabs of the depth order and only the tiles still open take the later
                                 slabs; 0 always one round (the classic full lists).  In a slab frame the per-tile lists are
                                 spread over the rounds, so GS_ARR_TILE_RANGES / SORTED_IDS / SORTED_KEYS are unavailable (a
                                 ctx's first frame is always a classic one).                                              */
    float   slab_max_ratio;   /* slab_mode 1 engages when the previous frame walked less than this share of its tile instances;
                                 0 = the library default (0.03, where three rounds start to beat one with the two-level binning)  */
    float   slab_fractions[3];/* tests / experiments: != 0 forces depth slabs that end at these fractions of the depth order
                                 (0 < f1 [< f2 [< f3]] < 1), whatever the previous frame walked                               */
    int32_t debug_flags;      /* GS_DEBUG_* bits; 0 in production                                                              */
    int32_t depth_sort;       /* the depth order (CUDA.sortperm of forward.jl:103; same permutation on every path, bit for bit):
                                 0 (default) automatic -- two steps (256 buckets over the frame's key range, then one workgroup
                                 per bucket: inside LDS up to 8192 pairs, else through global memory in chunks; 4 launches) for 3-D frames
                                 of up to 8.4 M gaussians, the four-pass LSD radix sort (12 launches) otherwise and for 64 frames after
                                 a frame whose depths piled up in one bucket (more than eight chunks); 1 always the four-pass sort; 2 always the two-step sort (tests)                          */
    int32_t list_cap;         /* capped tile lists (speed only; image, transmittance and deterministic-mode gradients are bit-identical
                                 either way).  The reference allocates and fills every tile's list in full (forward.jl:139-142:
                                 hitIdxs of maxBinSize per tile), although with the transmittance early-out a dense scene walks a
                                 fraction of it (C3: 28 %, C5: 6 % of the entries).  0 (default) automatic: when the frame's VIEW SLOT
                                 (gs_set_view_slot) has rendered this image size before, gs_bin writes every tile's list only as far
                                 as that frame walked it (+ 25 % + 128 entries, rounded up to a segment of the two-level binning); a
                                 composite wave that gets to the written end with pixels still taking entries writes the next
                                 segment of its tile's list itself and goes on, so nothing depends on the history being right.
                                 Engages when the ctx's previous frame walked less than 15 % of its list entries on a grid of more
                                 than 5120 tiles (measured: at 28 % walked the shorter write pass only pays for the extra cap pass).
                                 1 never; 2 also on small grids (tests).  GS_ARR_TILE_RANGES is always the full ranges; asking for
                                 GS_ARR_SORTED_IDS / _KEYS of a capped frame first writes the unwritten rest.                   */
    int32_t tile_parts;       /* how a tile is shared between waves (speed only).  The reference runs one 16 x 16 thread block per tile
                                 (splat.jl:224-231, threads = (16, 16)); here one wave64 composites a tile, four pixels per lane in four
                                 16 x 4 strips.  0 (default) automatic:
                                 (a) a grid with fewer tiles than the chip has wave slots (256 CUs x 4 SIMDs x 5) leaves slots idle and every
                                     wave runs alone on its SIMD: 4 waves share a tile when 4 x tiles fit the slots, else 2 when 2 x tiles fit,
                                     each owning two strips or one and walking the tile's list on its own (the entries a wave evaluates are
                                     tested against ITS pixels, so pixels may differ from the one-wave result by contributions below 2^-27
                                     -- the no-op rule of alpha_cull).  Once the frame's VIEW SLOT has history, the BACKWARD of every tile
                                     runs instead as two segments of its list (x pixel parts where they still fit): the forward's waves
                                     leave a snapshot (C, T) of their pixels at the boundary, the second segment starts from it (a list of
                                     two or three batches is cut after its first batch, longer ones at 5/8 of the slot's previous walk), and
                                     a wave keeps two entries in flight (it is alone on its SIMD there: the per-entry chain is latency);
                                 (b) on larger grids the launch order (schedule 3 / 4) gives the FEW tiles whose work stands far above an
                                     even share -- a trained scene's heavy tail -- two or four waves (by strips) in the forward, and up to
                                     eight list segments in the backward (snapshots as above); a spatially uniform scene splits nothing.
                                 Gradients of a tile shared this way differ from the one-wave result by the order of the additions
                                 (<= 1e-5 relative; bitwise reproducible run to run in deterministic mode).
                                 1: always one wave per tile (what cross-mode bit-identity tests pin); 2, 4: that many pixel parts on every
                                 tile (small grids only).  Always 1 for frames without the early-out (t_min = 0), frames binned in depth
                                 slabs and capped lists.  With several waves per tile the forward's per-tile work counters are those of the
                                 tile's first part.                                                                                   */
    int32_t reserved[3];      /* sizeof(gs_config) == 96                                                                       */
} gs_config;
#define GS_DEBUG_WIDE_CURSORS 1   /* two-level binning: 64-bit list cursors although the lists fit 32-bit byte offsets (tests)    */
#define GS_DEBUG_ALWAYS_ORDER 2    /* longest-first launch orders (and their side stream) also on grids with fewer tiles than wave slots (tests) */
#define GS_DEBUG_SUPER16 8          /* two-level binning with super-tiles of 16 x 16 tiles whatever the grid (default: on grids whose 8 x 8 super-tiles
                                      would be more than 256, i.e. 4K-class images); GS_DEBUG_SUPER8 16: 8 x 8 whatever the grid (tests) */
#define GS_DEBUG_SUPER8 16
#define GS_DEBUG_TINY_CAPS 4       /* capped lists with the minimum cap on every tile, history or not: every tile that walks more than its
                                      first segment extends its list in the composite kernel (tests)                               */

typedef struct gs_ctx gs_ctx;

/* Fill cfg with the defaults above. */
void gs_default_config(gs_config *cfg);

int gs_abi_version(void);

/* getRenderer: create a renderer context on HIP device `device`. */
int gs_create(gs_ctx **out, int device, const gs_config *cfg);
int gs_destroy(gs_ctx *ctx);
const char *gs_last_error(const gs_ctx *ctx);   /* ctx may be NULL: last gs_create error */

/* Use an existing hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream; GS_STREAM_LEGACY
 * (= hipStreamLegacy) = the device's default (null) stream.  Switching streams synchronises the old one; setting
 * the stream that is already in use returns at once, so a host may call this before every frame. */
#define GS_STREAM_LEGACY ((void *)1)
int gs_set_stream(gs_ctx *ctx, void *hip_stream);
int gs_synchronize(gs_ctx *ctx);

/* Model (SplatData3D, splat.jl:36-43).  means 3xN, scales 3xN (log), quats 4xN (w,x,y,z, not
 * normalised by the kernels), opacities 1xN (logit), shs (3*K)xN with K=(sh_degree+1)^2 and
 * element [c + 3k] = channel c of coefficient k (splat.jl:117 for degree 1).
 * GS_MEM_HOST: copied once and kept resident.  GS_MEM_DEVICE: borrowed (the caller keeps them
 * alive and may update them in place between frames, e.g. an optimiser step). */
int gs_set_model(gs_ctx *ctx, int64_t n, int sh_degree,
                 const float *means, const float *scales, const float *quats,
                 const float *opacities, const float *shs, int mem);

/* The 2-D image-fitting renderer (RendererType GAUSSIAN_2D; renderer.jl:7-18, SplatData2D splat.jl:20-26): means 2xN
 * in [0,1]^2 (pixel position (W*mx, H*my), splat.jl:337-339), scales 2xN (log, cov2d.jl:14-17), rotations 1xN (theta,
 * cov2d.jl:5), opacities 1xN (used raw, splat.jl:341: alpha = opacity*exp(-dist/2), clamped to [0, 1 - 2^-24] so that a step
 * of the optimiser cannot push alpha to 1 or below 0; the clamp has zero gradient outside), colors 3xN.  After this call gs_preprocess runs cov2d.jl:3-45 + boundingbox.jl on the
 * pixel position, gs_bin builds the lists in gaussian-index order (there is no depth), gs_forward / gs_backward use the
 * same composite kernels, and every gs_grads argument is read as SplatGrads2D (splat.jl:28-34):
 *   d_means 2xN, d_scales 2xN, d_quats -> d_rotations 1xN, d_opacities 1xN, d_shs -> d_colors 3xN.
 * Needs gs_set_image_size (or gs_set_camera, of which only W and H are used). */
int gs_set_model_2d(gs_ctx *ctx, int64_t n, const float *means, const float *scales, const float *rotations,
                    const float *opacities, const float *colors, int mem);
int gs_set_image_size(gs_ctx *ctx, int32_t W, int32_t H);

/* Camera + image size (forward.jl:41,53-62).  T, P: 16 floats each, column-major. */
int gs_set_camera(gs_ctx *ctx, const float T[16], const float P[16], float fx, float fy,
                  float near_, float far_, const float eye[3], const float lookAt[3],
                  int32_t W, int32_t H);

/* Optional: name the view about to be rendered (call before gs_preprocess; e.g. the `id` of the camera in cameras.json,
 * camera.jl:119-151).  A training loop cycles over a fixed camera set, and how the work of a frame is spread over the tiles is a
 * property of the view: the ctx keeps, per slot 0 .. GS_MAX_VIEW_SLOTS - 1, the longest-first tile order of the last frame
 * rendered under that slot and launches the next forward of the same slot in that order (gs_config.schedule 3 / 4), and how
 * many entries of its list each tile walked, which caps the lists gs_bin writes for the slot's next frame (gs_config.list_cap).  Speed
 * only -- a stale or wrong slot costs nothing but the benefit.  slot < 0 (default): the frame belongs to no slot.  A slot costs
 * three arrays of one word per tile (100 KB at 1080p), allocated when it is first rendered. */
#define GS_MAX_VIEW_SLOTS 4096
int gs_set_view_slot(gs_ctx *ctx, int32_t slot);

/* preprocess(renderer): projection, 2-D covariance + inverse, bounding box, SH colour,
 * sigmoid, depth key, tile rectangle. */
int gs_preprocess(gs_ctx *ctx);

/* compactIdxs(renderer, threads, blocks): per-tile splat lists = tile|depth keys, stable
 * radix sort, tile ranges.  gx, gy = the reference's `blocks`; pass 0,0 for ceil(W/16),
 * ceil(H/16). */
int gs_bin(gs_ctx *ctx, int32_t gx, int32_t gy);

/* Optional: the forward writes image [3,H,W] and transmittance [H,W] DIRECTLY into these caller-owned DEVICE buffers instead of
 * the ctx's own (and gs_backward reads them back from there), which saves the two device-to-device copies of
 * gs_forward(..., GS_MEM_DEVICE).  Contract: the buffers stay valid and unmodified from gs_forward until the last
 * gs_backward of the frame; size them for the image set with gs_set_camera.  Passing the bound pointers to gs_forward is
 * allowed (no copy).  NULL, NULL unbinds.  (The reference keeps imageData / transmittance in the renderer, renderer.jl:89-117;
 * this is the same ownership with the host's allocator.) */
int gs_bind_outputs(gs_ctx *ctx, float *image_dev, float *transmittance_dev);

/* forward(renderer, tps, threads, blocks): per-tile alpha composite.  image: W*H*3 floats
 * (planar), transmittance: W*H floats.  Either may be NULL. */
int gs_forward(gs_ctx *ctx, float *image, float *transmittance, int mem);

typedef struct {              /* SplatGrads3D, splat.jl:45-52; any pointer may be NULL       */
    float *d_means;           /* 3 x N                                                        */
    float *d_scales;          /* 3 x N                                                        */
    float *d_quats;           /* 4 x N                                                        */
    float *d_opacities;       /* 1 x N                                                        */
    float *d_shs;             /* 3K x N                                                       */
} gs_grads;

/* backward(renderer, dC): dC is W*H*3 (same layout as the image).  Gradients ACCUMULATE (+=)
 * into `grads` (DEVICE pointers), exactly as the reference's grads persist until resetGrads.
 * Requires gs_forward on the same frame. */
int gs_backward(gs_ctx *ctx, const float *dC, int mem, const gs_grads *grads);

/* gs_backward with flags.  GS_BWD_OVERWRITE: store the gradients instead of accumulating -- for
 * the first backward after a reset, so the caller can skip the zero fill and this pass skips
 * the read of the old values (the result equals reset + accumulate). */
#define GS_BWD_OVERWRITE 1
/* Split backward, so a multi-GPU host can start exchanging the colour gradients while the per-gaussian chain still runs:
 * GS_BWD_COMPOSITE_ONLY stops after the composite adjoint (the per-gaussian 2-D sums exist: gs_color_grads_pack,
 * GS_ARR_GRAD2D; grads may be NULL); GS_BWD_PARAMS_ONLY runs only the chain from those sums to `grads` (dC may be NULL). */
#define GS_BWD_COMPOSITE_ONLY 2
#define GS_BWD_PARAMS_ONLY 4
/* With GS_BWD_PARAMS_ONLY, the chain in two steps so that a multi-GPU host can start the all-reduce of d_shs (81 % of the
 * gradient buffer at SH degree 3) while the geometry chain still runs: GS_BWD_PARAMS_SH runs only the SH / colour kernel
 * (d_shs and the colour -> position term), GS_BWD_PARAMS_GEOM only the geometry chain (after a GS_BWD_PARAMS_SH call of the
 * same frame).  Neither bit = both, in that order. */
#define GS_BWD_PARAMS_SH 8
#define GS_BWD_PARAMS_GEOM 16
int gs_backward_ex(gs_ctx *ctx, const float *dC, int mem, const gs_grads *grads, int flags);
/* Backward AND the optimiser step of train.jl:39-46 in one pass (3-D renderer, single-view steps): the per-gaussian kernels apply
 * param = fma(-lr, gradient, param) to the resident model instead of storing gradients -- bit-identical to gs_backward
 * (GS_BWD_OVERWRITE) followed by gs_sgd_step, at one read-modify-write of the 59 N parameter floats instead of three passes. */
int gs_backward_sgd(gs_ctx *ctx, const float *dC, int mem, float lr);

/* resetGrads: zero the arrays of `grads` (DEVICE pointers) on the ctx stream. */
int gs_reset_grads(gs_ctx *ctx, const gs_grads *grads);

/* initGrads (splat.jl:137-156) for hosts without their own device allocator: allocates ONE flat
 * zeroed device buffer [d_means 3N | d_scales 3N | d_quats 4N | d_opac N | d_shs 3K*N] owned by
 * the ctx (freed by gs_destroy or the next gs_grads_alloc) and points `out` into it.
 * gs_grads_read copies gradient arrays back to HOST buffers (any may be NULL); synchronises. */
int gs_grads_alloc(gs_ctx *ctx, gs_grads *out);
int gs_grads_read(gs_ctx *ctx, const gs_grads *grads, float *h_means, float *h_scales, float *h_quats,
                  float *h_opacities, float *h_shs);

/* ---- multi-GPU: one camera view per GPU, ONE all-reduce of the gradients -------- */

/* RCCL (librccl.so.1, loaded on first use) over xGMI.  Rank 0 calls gs_comm_unique_id and hands
 * the 128 bytes to the other ranks by any host transport; every rank then calls gs_comm_init on its
 * own ctx (one process per GPU).  gs_allreduce_grads sums the gradient arrays across ranks on the ctx
 * stream: a single ncclAllReduce when `grads` is one contiguous buffer
 * [d_means 3N | d_scales 3N | d_quats 4N | d_opac N | d_shs 3K*N] (the layout of gs_grads_alloc and of
 * the Python mirror), else one per array. */
/* Colour-factored exchange (optional; same gradients, ~2.6x less xGMI traffic at SH degree 3).  The SH gradient of one
 * view is basis_k(dir_view) * d rgb[c], so instead of all-reducing the 3K floats per gaussian (81 % of the buffer) the
 * ranks all-gather the THREE floats d rgb per (view, gaussian) and every rank rebuilds sum_v basis(dir_v) (x) d rgb_v:
 *   per view:  gs_backward with grads.d_shs = NULL (everything else accumulates as usual), then
 *              gs_color_grads_pack(ctx, drgb_view)            3 x N floats, DEVICE
 *   per step:  all-reduce of [d_means | d_scales | d_quats | d_opacities] (11 N floats), all-gather of the packed d rgb,
 *              gs_sh_grads_from_views(ctx, nviews, cams, drgb_all, d_shs, flags)
 * cams: HOST, nviews records of GS_VIEW_RECORD_FLOATS floats {T[16], P[16], eye[3], lookAt[3]} in the order of drgb_all
 * ([nviews][3 x N], DEVICE); flags: GS_BWD_OVERWRITE stores, 0 accumulates into d_shs (DEVICE, 3K x N). */
#define GS_VIEW_RECORD_FLOATS 38
int gs_color_grads_pack(gs_ctx *ctx, float *drgb);
int gs_sh_grads_from_views(gs_ctx *ctx, int32_t nviews, const float *cams, const float *drgb, float *d_shs, int flags);

#define GS_COMM_ID_BYTES 128
int gs_comm_unique_id(void *id128);
int gs_comm_init(gs_ctx *ctx, int rank, int nranks, const void *id128);
int gs_allreduce_grads(gs_ctx *ctx, const gs_grads *grads);
int gs_comm_destroy(gs_ctx *ctx);

/* ---- the step after backward ---------------------------------------- */

/* Loss of src/loss.jl:62-72 and its gradient w.r.t. the rendered image:
 *   loss = (1-lam) * sum|img-gt| / (2*length) + lam * (1 - mean ssim) / 2,   11x11 window of loss.jl:5-12.
 * img, gt, dC: W*H*C floats in the image layout (x fastest, channel planes); dC may be NULL.
 * loss_out (HOST, may be NULL): receives the loss; when non-NULL the call synchronises. */
int gs_loss_l1_dssim(gs_ctx *ctx, const float *img, const float *gt, int32_t W, int32_t H, int32_t C, float lam,
                     float *dC, double *loss_out, int mem);

/* train.jl:42-46: param .-= lr * grad on the ctx's resident model arrays (DEVICE gradients). */
int gs_sgd_step(gs_ctx *ctx, float lr, const gs_grads *grads);

/* ---- introspection (parity tests, profiling) ------------------------------------------- */

typedef enum {
    GS_ARR_TS = 0,            /* 4 x N  f32   view-space position            (export_debug) */
    GS_ARR_TPS = 1,           /* 4 x N  f32   clip-space position            (export_debug) */
    GS_ARR_MU = 2,            /* 2 x N  f32   renderer.positions             (export_debug) */
    GS_ARR_COV3D = 3,         /* 9 x N  f32   renderer.cov3ds                (export_debug) */
    GS_ARR_COV2D = 4,         /* 4 x N  f32   renderer.cov2ds                (export_debug) */
    GS_ARR_INVCOV = 5,        /* 4 x N  f32   renderer.invCov2ds             (export_debug) */
    GS_ARR_BBS = 6,           /* 4 x N  f32   renderer.bbs [xmin ymin xmax ymax] (export_debug) */
    GS_ARR_RGB = 7,           /* 3 x N  f32   sh2color result                                */
    GS_ARR_SIG = 8,           /* 1 x N  f32   cusigmoid(opacity)                             */
    GS_ARR_DEPTH_KEY = 9,     /* 1 x N  u32   radix key of clip z for the ctx order          */
    GS_ARR_TILE_RECT = 10,    /* 4 x N  u16   x0 x1 y0 y1, 1-based inclusive; x0=0: no tile  */
    GS_ARR_SORT_IDXS = 11,    /* 1 x N  u32   renderer.sortIdxs (0-based)                    */
    GS_ARR_TILE_RANGES = 12,  /* 2 x gx*gy u32 [start,end) into the sorted instance list     */
    GS_ARR_SORTED_IDS = 13,   /* 1 x I  u32   gaussian id per sorted instance (0-based)      */
    GS_ARR_SORTED_KEYS = 14,  /* 1 x I  u64   tile<<32 | depth key (or | id in INDEX order)  */
    GS_ARR_GRAD2D = 15        /* 10 x N f32   d{rgb3,sig,mu2,inv4} after gs_backward          */
} gs_array;

int64_t gs_num_gaussians(const gs_ctx *ctx);
int64_t gs_num_instances(gs_ctx *ctx);            /* I of the last gs_bin (waits for the frame's totals if they are still in flight) */
int64_t gs_num_coarse_instances(gs_ctx *ctx);      /* two-level binning: (super-tile, gaussian) instances of the last gs_bin, else 0 */
int gs_num_rounds(const gs_ctx *ctx);             /* binning rounds of the last gs_bin: 1 = classic full lists, 2..4 = depth slabs */

/* Copy an internal array to a HOST buffer of `bytes` bytes (synchronises). */
int gs_get_array(gs_ctx *ctx, int which, void *host_dst, int64_t bytes);

typedef enum {
    GS_STAGE_PREPROCESS = 0, GS_STAGE_DEPTH_SORT, GS_STAGE_COUNT_SCAN, GS_STAGE_EMIT,
    GS_STAGE_TILE_SORT, GS_STAGE_RANGES, GS_STAGE_COMPOSITE_FWD, GS_STAGE_COMPOSITE_BWD,
    GS_STAGE_PREPROCESS_BWD, GS_STAGE_COUNT
} gs_stage;

/* Milliseconds of the last execution of each stage (profile_stages=1); synchronises. */
int gs_get_stage_times(gs_ctx *ctx, float ms[GS_STAGE_COUNT]);

/* Accumulated hipEvent time (ms) and launch count per stage since the last reset
 * (profile_stages=1); synchronises.  reset != 0 clears the accumulators afterwards. */
int gs_get_stage_stats(gs_ctx *ctx, double sum_ms[GS_STAGE_COUNT], int64_t count[GS_STAGE_COUNT], int reset);

/* Work counters of the last frame: list entries actually walked by the composite kernels
 * (== gs_num_instances when t_min == 0; fewer with the transmittance early-out). */
int gs_get_work_counters(gs_ctx *ctx, int64_t *walked_fwd, int64_t *walked_bwd);

/* Tile lists of the last frame: out = {entries actually written into the tile lists (== gs_num_instances unless the lists were
 * capped, gs_config.list_cap; includes what composite waves appended), list segments appended by composite waves (0 when the
 * slot's history covered the frame), 1 if the frame's lists were capped else 0}.  Call after gs_forward; synchronises. */
int gs_get_list_stats(gs_ctx *ctx, int64_t out[3]);

/* Waves per tile (1, 2 or 4) of the last frame's composite launches (gs_config.tile_parts); negative: error.  Call after gs_forward.
 * No counterpart in the reference (one thread block per tile, splat.jl:224-231). */
int gs_get_tile_parts(gs_ctx *ctx);

/* The path that built the last frame's tile lists (gs_config.bin_path says what was asked for): 0 two-level binning, 1 / 2 the radix
 * paths, 3 the small-frame path (gs_bin_small.hip: depth order and tile ranges in one workgroup, the lists by one workgroup per tile;
 * taken by bin_path 0 for up to 16 384 gaussians x 1024 tiles).  Same lists on every path.  Negative: error.  Call after gs_bin. */
int gs_get_bin_path(gs_ctx *ctx);

/* out = {walked_fwd, walked_bwd, evaluated_fwd, evaluated_bwd}: `evaluated` counts the walked entries that
 * survived the alpha_cull no-op test and were evaluated per pixel (== walked when alpha_cull == 0). */
int gs_get_work_counters_ex(gs_ctx *ctx, int64_t out[4]);

/* Profiling aid: re-launch the composite forward (which=0) or backward (which=1) kernel of the
 * current frame `reps` times with kernel variant `variant` and return the mean hipEvent time.
 * Backward gradients of these launches go to the internal 2-D gradient scratch only. */
int gs_debug_time_composite(gs_ctx *ctx, int which, int variant, int reps, float *mean_ms);

/* Profiling aid: one launch of the composite forward (which=0) or backward (which=1) kernel of the current frame with
 * per-tile clocks.  out (HOST): 15 x gx*gy uint64 per tile {start, end (100 MHz s_memrealtime ticks), HW_ID | XCC_ID << 32,
 * walked << 32 | evaluated, shader cycles inside the per-entry loops, shader cycles outside them (staging a batch and waiting
 * for its gathers), per-entry strip slots executed << 32 | the slots needed if the tile's live pixels were packed 64 to a slot,
 * strips holding a live pixel << 32 | live pixels (the last four summed over the evaluated entries), and (forward) six words: the
 * evaluated entries by the number of 64-lane slots they would need with live pixels packed anywhere / by whole rows / inside columns,
 * and the entries that would survive the no-op test against the rectangle of the live pixels instead of the whole tile}.
 * tools/tile_tail.py turns it into the occupancy-over-time, tail and frozen-pixel summaries under profiles/. */
int gs_debug_tile_clock(gs_ctx *ctx, int which, int variant, uint64_t *out);
/* A NEGATIVE variant v runs variant -v with one record per WORKGROUP of the launch instead of per tile -- the launch as production runs
 * it, heavy tiles split into two or four waves (the parts of a split tile would overwrite each other's per-tile record): out then
 * holds gs_debug_tile_clock_rows(ctx) rows of 15 words, rows of workgroups without a tile all zero.  0: the frame has no launch order. */
int gs_debug_tile_clock_rows(gs_ctx *ctx);

/* Profiling hook: the shader clock (MHz) the chip runs at right now, from one wave that counts s_memtime cycles over 20 us of
 * s_memrealtime on the ctx stream (waits for the stream).  tools/frames_probe.py samples it between frames. */
int gs_debug_clock_mhz(gs_ctx *ctx, float *mhz);

/* Profiling aid: restrict the debug launches above (gs_debug_time_composite, gs_debug_tile_clock) to the `len` workgroups of
 * the frame's longest-first launch order that start at entry `start` (a multiple of 8, so that a tile keeps its XCD); len = 0:
 * the whole order again.  Needs a frame with a launch order (gs_config.schedule 3 / 4 on a grid above the wave slots).  With it
 * tools/occupancy_curve.py launches 1024, 2048 .. 5120 tiles at once -- one to five waves per SIMD -- and reads the SIMD
 * throughput by resident waves directly instead of inferring it from tile lifetimes. */
int gs_debug_set_window(gs_ctx *ctx, int32_t start, int32_t len);

/* -1: the lane-order probe of the LDS-atomic rank was not run (rank_mode = 1 was asked for); 0: it ran at gs_create and
 * passed; 1: it failed on this device and ballots were forced. */
int gs_rank_probe_result(const gs_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_H */
