// gs_api.hip -- host side of the C ABI declared in include/gsplat.h.
//
// Mirrors the reference's host drivers -- getRenderer (src/renderer.jl:119-149), preprocess
// (src/forward.jl:35-111), compactIdxs (src/forward.jl:118-161), forward (src/forward.jl:163-198),
// backward (src/backward.jl:3-38), resetGrads (src/splat.jl:158-173) -- without their
// per-frame costs: the model stays resident (the reference re-uploads it with `|> CuArray`
// on every call, forward.jl:63-69,169-170), scratch is grow-only (the reference re-allocates
// hitIdxs/hits/hitScans each frame, forward.jl:120,137,142), every stage is enqueued on one
// stream with a single host read-back (the instance count; the reference syncs after every
// kernel and reads maxHits back, forward.jl:72-156).
// (The ctx itself and the helpers shared by the gs_api_*.hip files: gs_ctx.h.)
#include "gs_ctx.h"

std::string g_create_error;

extern "C" {

void gs_default_config(gs_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(gs_config);
    cfg->abi_version = GS_ABI_VERSION;
    cfg->tile_size = GS_TILE;
    cfg->order = GS_ORDER_DEPTH_DESC;
    cfg->t_min = 1e-5f;
    cfg->deterministic = 0;
    cfg->export_debug = 0;
    cfg->profile_stages = 0;
    cfg->rank_mode = 1;                  // ballots: with the two-level binning the LDS-atomic rank only serves the depth sort (0.3 % of a C3 frame)
    cfg->alpha_cull = 1;
    cfg->schedule = 3;
    cfg->slab_mode = 1;
}

int gs_abi_version(void) { return GS_ABI_VERSION; }

const char *gs_last_error(const gs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int gs_create(gs_ctx **out, int device, const gs_config *cfg) {
    if (!out) return fail(nullptr, GS_ERR_INVALID, "gs_create: out is NULL");
    *out = nullptr;
    gs_config c0;
    gs_default_config(&c0);
    if (cfg) {
        if (cfg->struct_size != (int32_t)sizeof(gs_config) || cfg->abi_version != GS_ABI_VERSION)
            return fail(nullptr, GS_ERR_INVALID, "gs_create: gs_config.struct_size / abi_version mismatch (caller built against another gsplat.h)");
        c0 = *cfg;
    }
    if (c0.schedule == 0) c0.schedule = 3;                               // 0 = the library default
    if (c0.tile_size != GS_TILE) return fail(nullptr, GS_ERR_UNSUPPORTED, "gs_create: only tile_size 16 is supported (reference threads=(16,16))");
    if (c0.order < GS_ORDER_INDEX || c0.order > GS_ORDER_DEPTH_ASC) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad order");
    if (!(c0.t_min >= 0.0f)) return fail(nullptr, GS_ERR_INVALID, "gs_create: t_min must be >= 0");
    if (c0.schedule != 1 && c0.schedule != 3 && c0.schedule != 4) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad schedule (0, 1, 3 or 4)");
    if (c0.slab_mode < 0 || c0.slab_mode > 1) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad slab_mode");
    if (c0.bin_path < 0 || c0.bin_path > 3) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad bin_path");
    if (!(c0.slab_max_ratio >= 0.0f && c0.slab_max_ratio <= 1.0f)) return fail(nullptr, GS_ERR_INVALID, "gs_create: slab_max_ratio must be in [0, 1]");
    for (int i = 0; i < 3; ++i)
        if (!(c0.slab_fractions[i] >= 0.0f && c0.slab_fractions[i] < 1.0f)) return fail(nullptr, GS_ERR_INVALID, "gs_create: slab_fractions must be in [0, 1)");
    if (c0.debug_flags & ~(GS_DEBUG_WIDE_CURSORS | GS_DEBUG_ALWAYS_ORDER | GS_DEBUG_TINY_CAPS | GS_DEBUG_SUPER16 | GS_DEBUG_SUPER8)) return fail(nullptr, GS_ERR_INVALID, "gs_create: unknown debug_flags");
    if (c0.depth_sort < 0 || c0.depth_sort > 2) return fail(nullptr, GS_ERR_INVALID, "gs_create: depth_sort must be 0, 1 or 2");
    if (c0.list_cap < 0 || c0.list_cap > 2) return fail(nullptr, GS_ERR_INVALID, "gs_create: list_cap must be 0, 1 or 2");
    if (c0.tile_parts != 0 && c0.tile_parts != 1 && c0.tile_parts != 2 && c0.tile_parts != 4) return fail(nullptr, GS_ERR_INVALID, "gs_create: tile_parts must be 0, 1, 2 or 4");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, GS_ERR_NO_DEVICE, "gs_create: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, GS_ERR_INVALID, "gs_create: device index out of range");
    gs_ctx *c = new (std::nothrow) gs_ctx();
    if (!c) return fail(nullptr, GS_ERR_OOM, "gs_create: host allocation failed");
    c->device = device;
    c->cfg = c0;
    c->slots.resize(GS_MAX_VIEW_SLOTS + 1);
    if ((e = hipSetDevice(device)) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipSetDevice"); }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipStreamCreate"); }
    c->own_stream = true;
    {   // wave slots of the composite kernels (their __launch_bounds__ ask for five waves per SIMD): what "the grid fills the chip" means
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->wave_slots = prop.multiProcessorCount * 4 * 5;
        else (void)hipGetLastError();
    }
    if ((e = hipHostMalloc((void **)&c->pinned, 512, hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return hipfail(nullptr, e, "hipHostMalloc"); }
    if ((e = hipHostMalloc((void **)&c->pinned_split, sizeof(uint32_t) * 2 * (GS_MAX_VIEW_SLOTS + 1), hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess) {
        (void)hipGetLastError(); c->pinned_split = nullptr;              // (speed only: without it every order counts as "may hold split tiles")
    } else std::memset(c->pinned_split, 0, sizeof(uint32_t) * 2 * (GS_MAX_VIEW_SLOTS + 1));
    for (int s = 0; s < GS_STAGE_COUNT; ++s)
        for (int k = 0; k < 2; ++k)
            if ((e = hipEventCreate(&c->ev[s][k])) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipEventCreate"); }
    if ((e = hipEventCreateWithFlags(&c->ev_count, hipEventDisableTiming | hipEventReleaseToSystem)) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipEventCreate"); }
    if ((e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking)) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipStreamCreate"); }
    for (hipEvent_t *ev : {&c->ev_main, &c->ev_order})
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipEventCreate"); }
    if (c->cfg.rank_mode == 0) {
        // The one-instruction stable rank (ds_add_rtn pre-values in ascending lane order) is a measured property of
        // gfx950's LDS, not an architectural guarantee: check it on THIS device before relying on it; ballots otherwise.
        int bad = 0;
        if ((e = gs_probe_lds_atomic_order(c->stream, &bad)) != hipSuccess) { (void)gs_destroy(c); return hipfail(nullptr, e, "gs_probe_lds_atomic_order"); }
        c->rank_probe = bad ? 1 : 0;
        if (bad) c->cfg.rank_mode = 1;
    }
    *out = c;
    return GS_OK;
}

int gs_destroy(gs_ctx *c) {
    if (!c) return GS_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    comm_release(c);
    DevBuf *bufs[] = {&c->payload, &c->depth_key, &c->rect, &c->pairs_a, &c->pairs_b, &c->perm, &c->offsets, &c->block_sums,
                      &c->inst_a, &c->inst_b, &c->table, &c->digit_total, &c->ranges, &c->image, &c->trans, &c->g2d, &c->stage_in,
                      &c->counters, &c->snap, &c->snap_walked, &c->grads_flat, &c->dpc, &c->ids, &c->words, &c->cs, &c->diff,
                      &c->tile_work, &c->tile_clock,
                      &c->tile_pos, &c->tile_done, &c->live2d, &c->rect_r, &c->offsets_r, &c->live_total,
                      &c->rect_sorted, &c->l1_table, &c->l1_rows, &c->l1_partials, &c->cids, &c->clr, &c->cranges, &c->segcnt, &c->sdone, &c->tilecnt,
                      &c->ranges_r[0], &c->ranges_r[1], &c->ranges_r[2], &c->ranges_r[3],
                      &c->invcov, &c->loss_maps, &c->loss_acc, &c->loss_in[0], &c->loss_in[1], &c->loss_dc, &c->view_cams, &c->tile_dead, &c->key_range, &c->tile_walked, &c->tile_walked_b, &c->tile_work_b};
    for (DevBuf *b : bufs) b->release();
    for (auto &v : c->slots) { v.order[0].release(); v.order[1].release(); v.walkbuf[0].release(); v.walkbuf[1].release(); }
    for (DevBuf *b : {&c->tile_nopen, &c->smax, &c->tile_ext, &c->zero_tiles}) b->release();
    for (auto &b : c->model) b.release();
    for (auto &b : c->dbg) b.release();
    for (int s = 0; s < GS_STAGE_COUNT; ++s)
        for (int k = 0; k < 2; ++k)
            if (c->ev[s][k]) (void)hipEventDestroy(c->ev[s][k]);
    if (c->ev_count) (void)hipEventDestroy(c->ev_count);
    for (hipEvent_t ev : {c->ev_main, c->ev_order}) if (ev) (void)hipEventDestroy(ev);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_split) (void)hipHostFree(c->pinned_split);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GS_OK;
}

int gs_set_stream(gs_ctx *c, void *hip_stream) {
    if (!c) return GS_ERR_INVALID;
    // GS_STREAM_LEGACY (= hipStreamLegacy) names the device's default (null) stream, which a NULL argument cannot
    hipStream_t want = hip_stream == (void *)1 ? (hipStream_t) nullptr : (hipStream_t)hip_stream;
    if (hip_stream && !c->own_stream && c->stream == want && c->borrowed_stream) return GS_OK;      // unchanged: no sync
    if (bind_device(c)) return GS_ERR_HIP;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->side) HIPCHK(c, hipStreamSynchronize(c->side));
    if (c->own_stream && c->stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    if (hip_stream) { c->stream = want; c->borrowed_stream = true; }
    else { HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; c->borrowed_stream = false; }
    return GS_OK;
}

int gs_synchronize(gs_ctx *c) {
    if (!c) return GS_ERR_INVALID;
    if (bind_device(c)) return GS_ERR_HIP;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->side) HIPCHK(c, hipStreamSynchronize(c->side));
    return GS_OK;
}

int gs_set_model(gs_ctx *c, int64_t n, int sh_degree, const float *means, const float *scales, const float *quats,
                 const float *opacities, const float *shs, int mem) {
    if (!c) return GS_ERR_INVALID;
    if (n < 0 || n > 0x7FFFFFF0LL) return fail(c, GS_ERR_INVALID, "gs_set_model: n out of range");
    if (sh_degree < 0 || sh_degree > 3) return fail(c, GS_ERR_UNSUPPORTED, "gs_set_model: sh_degree must be 0..3");
    if (n > 0 && (!means || !scales || !quats || !opacities || !shs)) return fail(c, GS_ERR_INVALID, "gs_set_model: NULL array");
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(c, GS_ERR_INVALID, "gs_set_model: bad mem");
    if (bind_device(c)) return GS_ERR_HIP;
    const int K = (sh_degree + 1) * (sh_degree + 1);
    const float *src[5] = {means, scales, quats, opacities, shs};
    const size_t width[5] = {3, 3, 4, 1, (size_t)3 * K};
    const float *dst[5];
    if (mem == GS_MEM_HOST) {
        for (int i = 0; i < 5; ++i) {
            const size_t bytes = sizeof(float) * width[i] * (size_t)n;
            HIPCHK(c, c->model[i].ensure(bytes ? bytes : 4));
            if (bytes) HIPCHK(c, hipMemcpyAsync(c->model[i].p, src[i], bytes, hipMemcpyHostToDevice, c->stream));
            dst[i] = c->model[i].as<float>();
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));            // caller may free its host arrays on return
    } else {
        for (int i = 0; i < 5; ++i) dst[i] = src[i];
    }
    c->n = n; c->sh_degree = sh_degree; c->kind = 0;
    for (int i = 0; i < 5; ++i) c->width[i] = width[i];
    c->means = dst[0]; c->scales = dst[1]; c->quats = dst[2]; c->opac = dst[3]; c->shs = dst[4];
    c->did_pre = c->did_bin = c->did_fwd = c->did_bwd = false;
    return GS_OK;
}

int gs_set_model_2d(gs_ctx *c, int64_t n, const float *means, const float *scales, const float *rotations,
                    const float *opacities, const float *colors, int mem) {
    if (!c) return GS_ERR_INVALID;
    if (n < 0 || n > 0x7FFFFFF0LL) return fail(c, GS_ERR_INVALID, "gs_set_model_2d: n out of range");
    if (n > 0 && (!means || !scales || !rotations || !opacities || !colors)) return fail(c, GS_ERR_INVALID, "gs_set_model_2d: NULL array");
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(c, GS_ERR_INVALID, "gs_set_model_2d: bad mem");
    if (bind_device(c)) return GS_ERR_HIP;
    const float *src[5] = {means, scales, rotations, opacities, colors};
    const size_t width[5] = {2, 2, 1, 1, 3};
    const float *dst[5];
    if (mem == GS_MEM_HOST) {
        for (int i = 0; i < 5; ++i) {
            const size_t bytes = sizeof(float) * width[i] * (size_t)n;
            HIPCHK(c, c->model[i].ensure(bytes ? bytes : 4));
            if (bytes) HIPCHK(c, hipMemcpyAsync(c->model[i].p, src[i], bytes, hipMemcpyHostToDevice, c->stream));
            dst[i] = c->model[i].as<float>();
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        for (int i = 0; i < 5; ++i) dst[i] = src[i];
    }
    c->n = n; c->sh_degree = 0; c->kind = 1;
    for (int i = 0; i < 5; ++i) c->width[i] = width[i];
    c->means = dst[0]; c->scales = dst[1]; c->quats = dst[2]; c->opac = dst[3]; c->shs = dst[4];
    c->did_pre = c->did_bin = c->did_fwd = c->did_bwd = false;
    return GS_OK;
}

int gs_set_image_size(gs_ctx *c, int32_t W, int32_t H) {
    if (!c) return GS_ERR_INVALID;
    if (W <= 0 || H <= 0 || W > 32767 || H > 32767) return fail(c, GS_ERR_INVALID, "gs_set_image_size: image size must be in 1..32767");
    if (!c->have_cam) {                                  // a harmless camera, so the shared paths have one
        std::memset(&c->cam, 0, sizeof(c->cam));
        c->cam.near_ = -1.0f; c->cam.far_ = 1.0f;
    }
    c->cam.W = W; c->cam.H = H;
    c->have_cam = true;
    c->did_pre = c->did_bin = c->did_fwd = c->did_bwd = false;
    return GS_OK;
}

int gs_set_camera(gs_ctx *c, const float T[16], const float P[16], float fx, float fy, float near_, float far_,
                  const float eye[3], const float lookAt[3], int32_t W, int32_t H) {
    if (!c) return GS_ERR_INVALID;
    if (!T || !P || !eye || !lookAt) return fail(c, GS_ERR_INVALID, "gs_set_camera: NULL argument");
    if (W <= 0 || H <= 0 || W > 32767 || H > 32767) return fail(c, GS_ERR_INVALID, "gs_set_camera: image size must be in 1..32767");
    std::memcpy(c->cam.T, T, sizeof(float) * 16);
    std::memcpy(c->cam.P, P, sizeof(float) * 16);
    c->cam.fx = fx; c->cam.fy = fy; c->cam.near_ = near_; c->cam.far_ = far_;
    for (int i = 0; i < 3; ++i) { c->cam.eye[i] = eye[i]; c->cam.lookAt[i] = lookAt[i]; }
    c->cam.W = W; c->cam.H = H;
    c->have_cam = true;
    c->did_pre = c->did_bin = c->did_fwd = c->did_bwd = false;
    return GS_OK;
}

int gs_preprocess(gs_ctx *c) {
    if (!c) return GS_ERR_INVALID;
    if (!c->have_cam) return fail(c, GS_ERR_INVALID, "gs_preprocess: gs_set_camera first");
    if (bind_device(c)) return GS_ERR_HIP;
    if (int rc = settle_totals(c, nullptr, false)) return rc;               // totals of a frame that was binned but never rendered
    c->frame_id += 1;
    const size_t n = (size_t)c->n, n1 = n ? n : 1;
    HIPCHK(c, c->payload.ensure(sizeof(GsPayload) * n1));
    HIPCHK(c, c->invcov.ensure(sizeof(float) * 4 * n1));
    HIPCHK(c, c->depth_key.ensure(sizeof(uint32_t) * n1));
    HIPCHK(c, c->rect.ensure(sizeof(uint16_t) * 4 * n1));
    // the tile grid is fixed by the image: the reference passes blocks = size/threads (main.jl:9-11)
    c->gx = (c->cam.W + GS_TILE - 1) / GS_TILE;
    c->gy = (c->cam.H + GS_TILE - 1) / GS_TILE;
    if (c->kind == 1) {                                  // preprocess(::GaussianRenderer2D), forward.jl:9-33
        c->range_valid = false;
        GsPreprocess2DArgs a2{};
        a2.n = c->n; a2.W = c->cam.W; a2.H = c->cam.H; a2.gx = c->gx; a2.gy = c->gy;
        a2.means = c->means; a2.scales = c->scales; a2.rots = c->quats; a2.opac = c->opac; a2.colors = c->shs;
        a2.payload = c->payload.as<GsPayload>(); a2.invcov = c->invcov.as<float>(); a2.depth_key = c->depth_key.as<uint32_t>(); a2.rect = c->rect.as<uint16_t>();
        if (c->cfg.export_debug) {
            const size_t w[7] = {4, 4, 2, 9, 4, 4, 4};
            for (int i = 0; i < 7; ++i) HIPCHK(c, c->dbg[i].ensure(sizeof(float) * w[i] * n1));
            a2.dbg.mu = c->dbg[2].as<float>(); a2.dbg.cov2d = c->dbg[4].as<float>(); a2.dbg.invcov = c->dbg[5].as<float>();
            a2.dbg.bbs = c->dbg[6].as<float>();
        }
        {
            StageTimer t(c, GS_STAGE_PREPROCESS);
            HIPCHK(c, gs_launch_preprocess2d(a2, c->stream));
        }
        c->did_pre = true; c->did_bin = c->did_fwd = c->did_bwd = false;
        return GS_OK;
    }
    GsPreprocessArgs a{};
    a.n = c->n; a.sh_degree = c->sh_degree; a.order = c->cfg.order;
    a.gx = c->gx; a.gy = c->gy;
    a.means = c->means; a.scales = c->scales; a.quats = c->quats; a.opac = c->opac; a.shs = c->shs;
    a.payload = c->payload.as<GsPayload>();
    a.invcov = c->invcov.as<float>();
    a.depth_key = c->depth_key.as<uint32_t>();
    a.rect = c->rect.as<uint16_t>();
    c->range_valid = false;
    // the key range is folded only when this frame's depth sort can take the bucket path: while the classic sort runs (the 64-frame
    // fallback, N beyond the bucket path's limit) nothing would reset the accumulators and the atomics would be wasted
    if (c->dsort_can_bucket() && c->order() != GS_ORDER_INDEX && c->n > 0 && !c->small_bin_possible()) {
        if (!c->key_range.p) {
            HIPCHK(c, c->key_range.ensure(sizeof(uint32_t) * gs_depth_range_words()));
            HIPCHK(c, gs_depth_range_reset(c->key_range.as<uint32_t>(), c->stream));
        }
        c->range_parity ^= 1;
        a.key_range = c->key_range.as<uint32_t>() + (size_t)c->range_parity * gs_depth_range_parity_words();
        c->range_valid = true;
    }
    if (c->cfg.export_debug) {
        const size_t w[7] = {4, 4, 2, 9, 4, 4, 4};
        for (int i = 0; i < 7; ++i) HIPCHK(c, c->dbg[i].ensure(sizeof(float) * w[i] * n1));
        a.dbg.ts = c->dbg[0].as<float>(); a.dbg.tps = c->dbg[1].as<float>(); a.dbg.mu = c->dbg[2].as<float>();
        a.dbg.cov3d = c->dbg[3].as<float>(); a.dbg.cov2d = c->dbg[4].as<float>(); a.dbg.invcov = c->dbg[5].as<float>();
        a.dbg.bbs = c->dbg[6].as<float>();
    }
    {
        StageTimer t(c, GS_STAGE_PREPROCESS);
        HIPCHK(c, gs_launch_preprocess(a, c->cam, c->stream));
    }
    c->did_pre = true; c->did_bin = c->did_fwd = c->did_bwd = false;
    return GS_OK;
}

int gs_loss_l1_dssim(gs_ctx *c, const float *img, const float *gt, int32_t W, int32_t H, int32_t C, float lam, float *dC,
                     double *loss_out, int mem) {
    if (!c || !img || !gt) return GS_ERR_INVALID;
    if (W <= 0 || H <= 0 || C <= 0) return fail(c, GS_ERR_INVALID, "gs_loss_l1_dssim: bad image size");
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(c, GS_ERR_INVALID, "gs_loss_l1_dssim: bad mem");
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)W * H * C;
    HIPCHK(c, c->loss_maps.ensure(sizeof(float) * 3 * n));
    HIPCHK(c, c->loss_acc.ensure(sizeof(double) * GS_LOSS_SLOTS * GS_LOSS_SLOT_STRIDE));
    const float *d_img = img, *d_gt = gt;
    float *d_dc = dC;
    if (mem == GS_MEM_HOST) {
        HIPCHK(c, c->loss_in[0].ensure(sizeof(float) * n));
        HIPCHK(c, c->loss_in[1].ensure(sizeof(float) * n));
        HIPCHK(c, hipMemcpyAsync(c->loss_in[0].p, img, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->loss_in[1].p, gt, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
        d_img = c->loss_in[0].as<float>(); d_gt = c->loss_in[1].as<float>();
    }
    if (mem == GS_MEM_HOST || !dC) { HIPCHK(c, c->loss_dc.ensure(sizeof(float) * n)); d_dc = c->loss_dc.as<float>(); }
    // window of loss.jl:5-12: exp(-r)/sqrt(2 sigma^2), r = distance from (6,6), normalised (sigma cancels); Float64 -> Float32
    float win[121];
    {
        double k[121], sum = 0.0;
        for (int j = 0; j < 11; ++j)
            for (int i = 0; i < 11; ++i) { k[j * 11 + i] = std::exp(-std::sqrt((double)((5 - j) * (5 - j) + (5 - i) * (5 - i)))); sum += k[j * 11 + i]; }
        for (int i = 0; i < 121; ++i) win[i] = (float)(k[i] / sum);
    }
    HIPCHK(c, gs_loss_run(W, H, C, d_img, d_gt, c->loss_maps.as<float>(), c->loss_acc.as<double>(), d_dc, lam, win, c->stream));
    if (mem == GS_MEM_HOST && dC) HIPCHK(c, hipMemcpyAsync(dC, d_dc, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    if (loss_out) {
        double slots[GS_LOSS_SLOTS * GS_LOSS_SLOT_STRIDE], acc[2] = {0.0, 0.0};
        HIPCHK(c, hipMemcpyAsync(slots, c->loss_acc.p, sizeof(slots), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < GS_LOSS_SLOTS; ++k) { acc[0] += slots[k * GS_LOSS_SLOT_STRIDE]; acc[1] += slots[k * GS_LOSS_SLOT_STRIDE + 1]; }
        *loss_out = (1.0 - (double)lam) * acc[0] / (2.0 * (double)n) + (double)lam * (1.0 - acc[1] / (double)n) / 2.0;
    } else if (mem == GS_MEM_HOST) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return GS_OK;
}

int gs_sgd_step(gs_ctx *c, float lr, const gs_grads *g) {
    if (!c || !g) return GS_ERR_INVALID;
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)c->n;
    float *p[5] = {const_cast<float *>(c->means), const_cast<float *>(c->scales), const_cast<float *>(c->quats),
                   const_cast<float *>(c->opac), const_cast<float *>(c->shs)};
    const float *gr[5] = {g->d_means, g->d_scales, g->d_quats, g->d_opacities, g->d_shs};
    const size_t *w = c->width;
    for (int i = 0; i < 5; ++i) HIPCHK(c, gs_launch_sgd(p[i], gr[i], lr, w[i] * n, c->stream));
    c->did_pre = c->did_bin = c->did_fwd = c->did_bwd = false;       // the model changed
    return GS_OK;
}

int gs_reset_grads(gs_ctx *c, const gs_grads *g) {
    if (!c || !g) return GS_ERR_INVALID;
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)c->n;
    float *p[5] = {g->d_means, g->d_scales, g->d_quats, g->d_opacities, g->d_shs};
    for (int i = 0; i < 5; ++i)
        if (p[i] && n) HIPCHK(c, hipMemsetAsync(p[i], 0, sizeof(float) * c->width[i] * n, c->stream));
    return GS_OK;
}

int gs_grads_alloc(gs_ctx *c, gs_grads *out) {
    if (!c || !out) return GS_ERR_INVALID;
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)c->n;
    size_t off[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) off[i + 1] = off[i] + c->width[i] * n;
    const size_t total = off[5];
    HIPCHK(c, c->grads_flat.ensure(sizeof(float) * (total ? total : 1)));
    HIPCHK(c, hipMemsetAsync(c->grads_flat.p, 0, sizeof(float) * total, c->stream));
    float *f = c->grads_flat.as<float>();
    out->d_means = f; out->d_scales = f + off[1]; out->d_quats = f + off[2]; out->d_opacities = f + off[3]; out->d_shs = f + off[4];
    return GS_OK;
}

int gs_grads_read(gs_ctx *c, const gs_grads *g, float *h_means, float *h_scales, float *h_quats, float *h_opacities, float *h_shs) {
    if (!c || !g) return GS_ERR_INVALID;
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)c->n;
    const float *src[5] = {g->d_means, g->d_scales, g->d_quats, g->d_opacities, g->d_shs};
    float *dst[5] = {h_means, h_scales, h_quats, h_opacities, h_shs};
    const size_t *w = c->width;
    for (int i = 0; i < 5; ++i)
        if (dst[i] && src[i] && n) HIPCHK(c, hipMemcpyAsync(dst[i], src[i], sizeof(float) * w[i] * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

}  // extern "C"
