// gs_api_composite.hip -- gs_forward / gs_backward (reference src/forward.jl:163-198, src/backward.jl:3-38): the composite launches,
// the per-gaussian chain behind the backward, and the longest-first launch orders kept per view slot.
#include "gs_ctx.h"

// Launch order of the frame's composite kernels (gs_config.schedule 3 / 4): what the last forward under the same view slot
// measured, else (schedule 4) what this ctx's previous slot-less forward measured; null = no history: tile order for the forward.
const uint32_t *forward_order(gs_ctx *c) {
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    if (!lpt_schedule(c) || ntiles <= 0 || ntiles > GS_LPT_MAX_TILES) return nullptr;
    const int k = order_index(c);
    if (k == GS_MAX_VIEW_SLOTS && c->cfg.schedule != 4) return nullptr;
    if (c->slots[k].tiles != (((int64_t)c->gx << 32) | (int64_t)c->gy)) return nullptr;         // (the order's length and groups belong to one grid)
    if (c->order_pending) {                                              // (long complete by now; an event wait on the stream costs nothing)
        if (hipStreamWaitEvent(c->stream, c->ev_order, 0) != hipSuccess) return nullptr;
        c->order_pending = false;
    }
    c->frame_order_split = c->pinned_split ? c->pinned_split + 2 * k + c->slots[k].sel : nullptr;
    return c->slots[k].order[c->slots[k].sel].as<uint32_t>();
}

// After the frame's (last) forward: ONE order kernel turns its per-tile work into a launch order.  When the forward already ran
// on the slot's history, the backward uses the same order and the kernel runs on the SIDE stream, beside the backward, into the
// slot's other buffer -- for the slot's next frame; nothing of this frame waits for it.  Without history (a slot's first frame)
// the backward waits for it: it is its only source of a longest-first order.
int build_frame_order(gs_ctx *c, const uint32_t *used) {
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    c->frame_order = used;
    if (!lpt_schedule(c) || ntiles <= 0 || ntiles > GS_LPT_MAX_TILES) return GS_OK;
    const int k = order_index(c);
    const int dst = used ? 1 - c->slots[k].sel : c->slots[k].sel;
    DevBuf &ob = c->slots[k].order[dst];
    HIPCHK(c, ob.ensure(sizeof(uint32_t) * ((size_t)lpt_order_entries(c) + GS_SEG_SLOTS + 16)));
    // heavy tiles: the order kernel also sizes the backward's list segments (from what this forward walked) and re-arms the walked
    // lengths of the NEXT frame's snapshots (this frame's are being read by its backward)
    const uint32_t *walked = lpt_front(c) > 0 ? c->last_walked : nullptr;
    uint32_t *rearm = nullptr, *nsplit = c->pinned_split ? c->pinned_split + 2 * k + dst : nullptr;
    if (nsplit) *nsplit = 0xFFFFFFFFu;                                     // unknown until the order kernel has stored its count
    if (lpt_front(c) > 0) {
        if (!c->snap_walked.p) {
            HIPCHK(c, c->snap_walked.ensure(sizeof(uint32_t) * 2 * GS_SEG_SLOTS));
            HIPCHK(c, hipMemsetAsync(c->snap_walked.p, 0, sizeof(uint32_t) * 2 * GS_SEG_SLOTS, c->stream));
        }
        rearm = c->snap_walked.as<uint32_t>() + (size_t)(c->snap_parity ^ 1) * GS_SEG_SLOTS;
    }
    if (used && use_side_stream(c)) {
        if (c->order_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_order, 0));      // (never two in flight)
        HIPCHK(c, hipEventRecord(c->ev_main, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0));
        HIPCHK(c, gs_launch_tile_lpt_order(c->tile_work.as<uint32_t>(), 0, c->gx, c->gy, ob.as<uint32_t>(), c->side, nullptr, 0, lpt_front(c), lpt_split_div(c), walked, rearm, nsplit));
        HIPCHK(c, hipEventRecord(c->ev_order, c->side));
        c->order_pending = true;
        c->slots[k].sel = dst;
        // (tile_work is rewritten by the next forward of this ctx: it waits for ev_order first, see forward_order / gs_forward)
    } else {
        HIPCHK(c, gs_launch_tile_lpt_order(c->tile_work.as<uint32_t>(), 0, c->gx, c->gy, ob.as<uint32_t>(), c->stream, nullptr, 0, lpt_front(c), lpt_split_div(c), walked, rearm, nsplit));
        c->frame_order = ob.as<uint32_t>();
        c->frame_order_split = nsplit;
    }
    c->slots[k].tiles = ((int64_t)c->gx << 32) | (int64_t)c->gy;
    return GS_OK;
}

extern "C" {

int gs_bind_outputs(gs_ctx *c, float *image, float *transmittance) {
    if (!c) return GS_ERR_INVALID;
    if ((image == nullptr) != (transmittance == nullptr)) return fail(c, GS_ERR_INVALID, "gs_bind_outputs: bind both buffers or neither");
    c->bound_image = image; c->bound_trans = transmittance;
    c->did_fwd = c->did_bwd = false;                                      // the forward's result lives in the buffers bound at its time
    return GS_OK;
}

int gs_set_view_slot(gs_ctx *c, int32_t slot) {
    if (!c) return GS_ERR_INVALID;
    if (slot >= GS_MAX_VIEW_SLOTS) return fail(c, GS_ERR_INVALID, "gs_set_view_slot: slot must be below GS_MAX_VIEW_SLOTS (or negative: none)");
    c->view_slot = slot < 0 ? -1 : (int)slot;
    return GS_OK;
}

// waves per tile of this frame's composite launches (gs_config.tile_parts)
static int composite_parts(const gs_ctx *c) {
    const int want = c->cfg.tile_parts;
    if (want == 1 || !(c->cfg.t_min > 0.0f) || c->n_rounds > 1 || c->frame_capped) return 1;
    if (want == 2 || want == 4) return want;
    const long long ntiles = (long long)c->gx * c->gy, slots = c->wave_slots;   // 256 CUs x 4 SIMDs x five waves (GS_FWD_MINW, GS_BWD_MINW)
    return 4 * ntiles <= slots ? 4 : 2 * ntiles <= slots ? 2 : 1;
}

// the composite launch of round r of the frame (r = 0 unless the frame is binned in depth slabs)
static int enqueue_forward_round(gs_ctx *c, int r, const uint32_t *order) {
    const int R = c->n_rounds;
    GsCompositeArgs a{};
    a.W = c->cam.W; a.H = c->cam.H; a.gx = c->gx; a.gy = c->gy; a.t_min = c->cfg.t_min;
    a.ranges = r == 0 ? c->ranges.as<uint32_t>() : c->ranges_r[r].as<uint32_t>();
    a.ids = c->ids.as<uint32_t>() + c->round_ids_off[r]; a.payload = c->payload.as<GsPayload>();
    a.image = c->img(); a.trans = c->tr();
    a.walked = nullptr;                                                     // per tile: tile_walked / tile_work (GsCompositeArgs.walked)
    a.cull = c->cfg.alpha_cull != 0;
    a.resume = r > 0; a.final_round = r == R - 1;
    a.tile_work = c->tile_work.as<uint32_t>(); a.tile_walked = const_cast<uint32_t *>(c->last_walked);
    a.tile_order = order; a.order_len = order ? lpt_order_entries(c) : 0;
    a.parts = c->frame_parts;
    a.split_ok = order && lpt_front(c) > 0 && c->frame_parts == 1 && !c->frame_capped && R == 1;   // heavy tiles run as the order's entries say
    c->snap_order = nullptr;
    if (c->frame_seg_n && R == 1) {                                         // small grid: every tile leaves snapshots for its backward's list segments
        const size_t nt = (size_t)c->gx * c->gy;
        HIPCHK(c, c->snap.ensure(sizeof(float) * nt * (size_t)(c->frame_seg_n - 1) * 4 * 256));
        HIPCHK(c, c->tile_walked_b.ensure(sizeof(uint32_t) * (nt ? nt : 1)));
        HIPCHK(c, c->tile_work_b.ensure(sizeof(uint32_t) * (nt ? nt : 1)));
        a.snap = c->snap.as<float>(); a.seg_hist = c->seg_hist; a.seg_n = c->frame_seg_n;
        a.bw_walked = c->tile_walked_b.as<uint32_t>(); a.bw_work = c->tile_work_b.as<uint32_t>();
    }
    const bool may_split = !c->frame_order_split || *c->frame_order_split != 0u;   // (the order kernel's count; 0xFFFFFFFF: not reported yet)
    if (a.split_ok && c->snap_walked.p && may_split) {                      // ... and leave snapshots for the list segments of their backward
        HIPCHK(c, c->snap.ensure(sizeof(float) * (size_t)GS_SEG_SLOTS * GS_SEG_SNAP_FLOATS));
        const size_t nt = (size_t)c->gx * c->gy;
        HIPCHK(c, c->tile_walked_b.ensure(sizeof(uint32_t) * (nt ? nt : 1)));
        HIPCHK(c, c->tile_work_b.ensure(sizeof(uint32_t) * (nt ? nt : 1)));
        a.snap = c->snap.as<float>(); a.seg_len = order_seg_len(c, order); a.front = lpt_front(c);
        a.snap_walked = c->snap_walked.as<uint32_t>() + (size_t)c->snap_parity * GS_SEG_SLOTS;
        a.bw_walked = c->tile_walked_b.as<uint32_t>(); a.bw_work = c->tile_work_b.as<uint32_t>();
        c->snap_order = order;
    }
    if (c->frame_capped && R == 1) {                                        // capped lists: the wave extends its tile's list when it must
        a.tile_ext = c->tile_ext.as<uint2>(); a.cranges = c->cranges.as<uint32_t>(); a.cids = c->cids.as<uint32_t>(); a.clr = c->clr.as<uint16_t>();
        a.ids_w = c->ids.as<uint32_t>(); a.sgx = c->sgx; a.sbs = c->sbs; a.ext_count = c->ext_count();
    }
    if (R > 1) { a.tile_pos = c->tile_pos.as<uint32_t>(); a.tile_done = c->tile_done.as<uint8_t>(); a.tile_dead = c->tile_dead.as<unsigned long long>(); }
    StageTimer t(c, GS_STAGE_COMPOSITE_FWD);                               // the kernel alone
    HIPCHK(c, gs_launch_composite_fwd(a, c->stream));
    return GS_OK;
}

int gs_forward(gs_ctx *c, float *image, float *transmittance, int mem) {
    if (!c) return GS_ERR_INVALID;
    if (!c->did_bin) return fail(c, GS_ERR_INVALID, "gs_forward: gs_bin first");
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(c, GS_ERR_INVALID, "gs_forward: bad mem");
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t px = (size_t)c->cam.W * c->cam.H;
    const size_t ntiles = (size_t)c->gx * c->gy;
    if (!c->bound_image) {
        HIPCHK(c, c->image.ensure(sizeof(float) * 3 * px));
        HIPCHK(c, c->trans.ensure(sizeof(float) * px));
    }
    HIPCHK(c, c->counters.ensure(GS_COUNTER_BYTES));
    HIPCHK(c, c->tile_work.ensure(sizeof(uint32_t) * (ntiles ? ntiles : 1)));
    c->counters_grid = ((int64_t)c->gx << 32) | (int64_t)c->gy;
    c->frame_seg_n = 0; c->seg_hist = nullptr;
    int flip_slot = -1;
    {   // the per-tile walked counts go to the frame's view slot (the caps of the slot's next lists; double buffered: this frame may read the
        // previous walk while it writes its own), else to the ctx's own array
        const int k = order_index(c);
        if (k < GS_MAX_VIEW_SLOTS || c->cfg.schedule == 4) {
            gs_ctx::ViewSlot &vs = c->slots[k];
            DevBuf &wb = vs.walkbuf[vs.wsel ^ 1];
            HIPCHK(c, wb.ensure(sizeof(uint32_t) * (ntiles ? ntiles : 1)));
            const bool hist = vs.walked_grid == c->counters_grid && vs.walked().p;
            // small grids (no launch order, fewer tiles than half the wave slots): with the slot's previous walk at hand EVERY tile's backward
            // runs as 2 / 4 list segments; the forward's waves leave the snapshots
            if (hist && c->cfg.tile_parts == 0 && c->cfg.t_min > 0.0f && c->n_rounds == 1 && !c->frame_capped && !lpt_schedule(c) && c->kind == 0 &&
                2 * (int64_t)ntiles <= c->wave_slots && ntiles > 0) {
                // (C1, lists of two or three batches: four segments instead of four pixel parts lost, 0.049 against 0.034 ms; three segments x four
                // parts ran like two, 0.0239 against 0.0243 -- a lone wave needs 0.33 us per evaluated entry and the launch lasts as long as the
                // batch with the most of them, which no cut at a batch boundary shortens)
                c->frame_seg_n = 2;
                c->seg_hist = vs.walked().as<uint32_t>();
            }
            c->last_walked = wb.as<uint32_t>();
            flip_slot = k;
        } else {
            HIPCHK(c, c->tile_walked.ensure(sizeof(uint32_t) * (ntiles ? ntiles : 1)));
            c->last_walked = c->tile_walked.as<uint32_t>();
        }
    }
    const int R = c->n_rounds;
    if (R > 1) {
        HIPCHK(c, c->tile_pos.ensure(sizeof(uint32_t) * (ntiles ? ntiles : 1)));
        HIPCHK(c, c->tile_done.ensure(ntiles ? ntiles : 1));
        HIPCHK(c, c->tile_dead.ensure(sizeof(unsigned long long) * 4 * (ntiles ? ntiles : 1)));
        HIPCHK(c, hipMemsetAsync(c->tile_pos.p, 0, sizeof(uint32_t) * ntiles, c->stream));
    }
    c->frame_parts = composite_parts(c);
    c->snap_parity ^= 1;                                                   // (the order kernel behind the previous forward re-armed this parity)
    const uint32_t *order = forward_order(c);
    if (c->order_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_order, 0)); c->order_pending = false; }     // the order kernel in flight reads tile_work
    for (int r = 0; r < R; ++r) {
        if (r > 0) { if (int rc = bin_round(c, r)) return rc; }
        if (int rc = enqueue_forward_round(c, r, order)) return rc;
        if (r == 0) {
            // the frame's totals: by now the composite is enqueued behind the lists, so the GPU has work while the host looks
            bool redo = false;
            if (int rc = settle_totals(c, &redo, true)) return rc;
            if (redo) {                                                    // the lists outgrew a buffer and were rebuilt: composite again
                if (int rc = enqueue_forward_round(c, 0, order)) return rc;
            }
        }
    }
    if (flip_slot >= 0) { c->slots[flip_slot].wsel ^= 1; c->slots[flip_slot].walked_grid = c->counters_grid; }   // this frame's walk is the slot's history now
    if (int rc = build_frame_order(c, order)) return rc;
    const hipMemcpyKind kind = mem == GS_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (image && image != c->img()) HIPCHK(c, hipMemcpyAsync(image, c->img(), sizeof(float) * 3 * px, kind, c->stream));
    if (transmittance && transmittance != c->tr()) HIPCHK(c, hipMemcpyAsync(transmittance, c->tr(), sizeof(float) * px, kind, c->stream));
    if (mem == GS_MEM_HOST && (image || transmittance)) HIPCHK(c, hipStreamSynchronize(c->stream));
    c->did_fwd = true; c->did_bwd = false; c->did_bwd_composite = false;
    c->prev_counters_valid = true; c->prev_n_inst = c->n_inst;
    return GS_OK;
}

int gs_backward(gs_ctx *c, const float *dC, int mem, const gs_grads *grads) { return gs_backward_ex(c, dC, mem, grads, 0); }

static int backward_impl(gs_ctx *c, const float *dC, int mem, const gs_grads *grads, int flags, float sgd_scale);
int gs_backward_ex(gs_ctx *c, const float *dC, int mem, const gs_grads *grads, int flags) { return backward_impl(c, dC, mem, grads, flags, 0.0f); }

// backward + SGD in one pass: the per-gaussian kernels apply param = fma(-lr, gradient, param) to the resident model instead
// of storing the gradient (the same fma gs_sgd_step applies to the stored float): one read-modify-write of the parameters
// instead of gradient write + gradient read + parameter read-modify-write.  Single-view steps only (nothing is accumulated).
int gs_backward_sgd(gs_ctx *c, const float *dC, int mem, float lr) {
    if (!c) return GS_ERR_INVALID;
    if (c->kind != 0) return fail(c, GS_ERR_UNSUPPORTED, "gs_backward_sgd: 3-D renderer only");
    if (!(lr != 0.0f)) return fail(c, GS_ERR_INVALID, "gs_backward_sgd: lr must be non-zero");
    gs_grads g{const_cast<float *>(c->means), const_cast<float *>(c->scales), const_cast<float *>(c->quats),
               const_cast<float *>(c->opac), const_cast<float *>(c->shs)};
    const int rc = backward_impl(c, dC, mem, &g, 0, -lr);
    if (rc == GS_OK) c->did_pre = c->did_bin = c->did_fwd = c->did_bwd = false;       // the model changed
    return rc;
}

static int backward_impl(gs_ctx *c, const float *dC, int mem, const gs_grads *grads, int flags, float sgd_scale) {
    if (!c) return GS_ERR_INVALID;
    if (!c->did_fwd) return fail(c, GS_ERR_INVALID, "gs_backward: gs_forward first");
    const bool params_only = (flags & GS_BWD_PARAMS_ONLY) != 0, composite_only = (flags & GS_BWD_COMPOSITE_ONLY) != 0;
    if (params_only && composite_only) return fail(c, GS_ERR_INVALID, "gs_backward: COMPOSITE_ONLY and PARAMS_ONLY exclude each other");
    const int chain = (flags & (GS_BWD_PARAMS_SH | GS_BWD_PARAMS_GEOM)) >> 3;             // bit 0: SH kernel, bit 1: geometry chain; 0 = both
    if (chain && !params_only) return fail(c, GS_ERR_INVALID, "gs_backward: GS_BWD_PARAMS_SH / _GEOM need GS_BWD_PARAMS_ONLY");
    if (chain && c->kind != 0) return fail(c, GS_ERR_UNSUPPORTED, "gs_backward: GS_BWD_PARAMS_SH / _GEOM: 3-D renderer only");
    if (params_only && !c->did_bwd_composite) return fail(c, GS_ERR_INVALID, "gs_backward: GS_BWD_PARAMS_ONLY needs a GS_BWD_COMPOSITE_ONLY call on this frame");
    if ((!dC && !params_only) || (!grads && !composite_only)) return fail(c, GS_ERR_INVALID, "gs_backward: NULL argument");
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(c, GS_ERR_INVALID, "gs_backward: bad mem");
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t px = (size_t)c->cam.W * c->cam.H, n1 = c->n ? (size_t)c->n : 1;
    const float *dC_dev = dC;
    if (mem == GS_MEM_HOST && !params_only) {
        HIPCHK(c, c->stage_in.ensure(sizeof(float) * 3 * px));
        HIPCHK(c, hipMemcpyAsync(c->stage_in.p, dC, sizeof(float) * 3 * px, hipMemcpyHostToDevice, c->stream));
        dC_dev = c->stage_in.as<float>();
    }
    const bool det = c->cfg.deterministic != 0;
    HIPCHK(c, c->g2d.ensure((det ? sizeof(long long) : sizeof(float)) * GS_G2D_STRIDE * n1));
    GsCompositeArgs a{};
    a.W = c->cam.W; a.H = c->cam.H; a.gx = c->gx; a.gy = c->gy; a.t_min = c->cfg.t_min;
    a.ranges = c->ranges.as<uint32_t>(); a.ids = c->ids.as<uint32_t>(); a.payload = c->payload.as<GsPayload>();
    a.image = c->img(); a.trans = c->tr();
    a.nseg = 0;
    for (int r = 0; r < c->n_rounds; ++r) {
        if (r > 0 && c->round_gen[r] == 0) continue;                       // a round that listed nothing
        a.seg_ranges[a.nseg] = r == 0 ? c->ranges.as<uint32_t>() : c->ranges_r[r].as<uint32_t>();
        a.seg_ids[a.nseg] = c->ids.as<uint32_t>() + c->round_ids_off[r];
        ++a.nseg;
    }
    a.dC = dC_dev; a.g2d = det ? nullptr : c->g2d.as<float>(); a.g2d_fixed = det ? c->g2d.as<long long>() : nullptr;
    {
        const size_t nt = (size_t)c->gx * c->gy;
        HIPCHK(c, c->tile_walked_b.ensure(sizeof(uint32_t) * (nt ? nt : 1)));
        HIPCHK(c, c->tile_work_b.ensure(sizeof(uint32_t) * (nt ? nt : 1)));
    }
    a.walked = nullptr; a.tile_walked = c->tile_walked_b.as<uint32_t>(); a.tile_work = c->tile_work_b.as<uint32_t>();
    if (c->frame_capped && c->n_rounds == 1) a.tile_ext = c->tile_ext.as<uint2>();      // the lists end where gs_bin / the forward stopped writing them
    a.cull = c->cfg.alpha_cull != 0;
    a.parts = c->frame_parts;                                              // as the frame's forward
    if (c->frame_seg_n && c->n_rounds == 1 && !c->frame_capped) {          // small grid: list segments instead of pixel parts (the forward left the snapshots)
        // two list segments, and as many pixel parts on top as still fit the wave slots (C1: 4 x 2 waves per tile; C2: 1 x 2)
        const long long nt = (long long)c->gx * c->gy;
        a.snap = c->snap.as<float>(); a.seg_hist = c->seg_hist; a.seg_n = c->frame_seg_n;
        a.parts = 4 * a.seg_n * nt <= c->wave_slots ? 4 : 2 * a.seg_n * nt <= c->wave_slots ? 2 : 1;
    }
    if (!params_only) {
        c->last_dC = dC_dev;
        // zero fill of the gradient rows (64 B per gaussian), in line: on a side stream beside the forward composite it cost more than
        // it hid (C3, same box, interleaved: 1.433 ms with the fill beside the forward, 1.412 ms in line -- its workgroups take wave slots
        // from the forward's tiles, and the two event hand-shakes per frame delay the depth sort's launches)
        // (a fill with streaming stores, which would spare the caches 64 B per gaussian, was measured: the atomics of the composite backward then
        // find their rows in memory instead of the cache, C3 + 1 %; profiles/r04t_ab_nontemporal.log)
        // (a small frame's gs_bin kernel has cleared them already: gs_bin_small.hip)
        if (!c->g2d_clean) HIPCHK(c, hipMemsetAsync(c->g2d.p, 0, (det ? sizeof(long long) : sizeof(float)) * GS_G2D_STRIDE * n1, c->stream));
        c->g2d_clean = false;
        // launch order: the one the frame's forward used, or (no history) what the order kernel made of that forward
        if (lpt_schedule(c)) { a.tile_order = c->frame_order; a.order_len = c->frame_order ? lpt_order_entries(c) : 0; }
        a.split_ok = a.tile_order && lpt_front(c) > 0 && c->frame_parts == 1 && !c->frame_capped && c->n_rounds == 1;
        if (a.split_ok && c->snap_order && c->snap_order == a.tile_order) {      // the forward left snapshots for this order's split tiles: list segments
            a.snap = c->snap.as<float>(); a.seg_len = order_seg_len(c, a.tile_order); a.front = lpt_front(c);
            a.snap_walked = c->snap_walked.as<uint32_t>() + (size_t)c->snap_parity * GS_SEG_SLOTS;
        }
        {
            StageTimer t(c, GS_STAGE_COMPOSITE_BWD);                   // the kernel alone (what rocprof reports for it)
            HIPCHK(c, gs_launch_composite_bwd(a, c->stream));
        }
        c->did_bwd_composite = true;
    }
    if (composite_only) {
        if (mem == GS_MEM_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
        c->did_bwd = true;                                             // the 2-D gradient sums exist (gs_color_grads_pack, GS_ARR_GRAD2D)
        return GS_OK;
    }
    if (c->kind == 1) {                                  // SplatGrads2D, splat.jl:28-34
        GsPreprocess2DBwdArgs b2{};
        b2.n = c->n; b2.W = c->cam.W; b2.H = c->cam.H;
        b2.scales = c->scales; b2.rots = c->quats; b2.opac = c->opac;
        b2.g2d = det ? nullptr : c->g2d.as<float>(); b2.g2d_fixed = det ? c->g2d.as<long long>() : nullptr;
        b2.overwrite = (flags & GS_BWD_OVERWRITE) ? 1 : 0;
        b2.d_means = grads->d_means; b2.d_scales = grads->d_scales; b2.d_rots = grads->d_quats;
        b2.d_opac = grads->d_opacities; b2.d_colors = grads->d_shs;
        {
            StageTimer t(c, GS_STAGE_PREPROCESS_BWD);
            HIPCHK(c, gs_launch_preprocess2d_bwd(b2, c->stream));
        }
        if (mem == GS_MEM_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
        c->did_bwd = true;
        return GS_OK;
    }
    GsPreprocessBwdArgs b{};
    b.n = c->n; b.sh_degree = c->sh_degree;
    b.means = c->means; b.scales = c->scales; b.quats = c->quats; b.opac = c->opac; b.shs = c->shs;
    b.g2d = det ? nullptr : c->g2d.as<float>(); b.g2d_fixed = det ? c->g2d.as<long long>() : nullptr;
    HIPCHK(c, c->dpc.ensure(sizeof(float) * 4 * n1));
    b.dpc = c->dpc.as<float>();
    b.overwrite = (flags & GS_BWD_OVERWRITE) ? 1 : 0;
    b.sgd_scale = sgd_scale;
    b.d_means = grads->d_means; b.d_scales = grads->d_scales; b.d_quats = grads->d_quats;
    b.d_opac = grads->d_opacities; b.d_shs = grads->d_shs;
    {
        StageTimer t(c, GS_STAGE_PREPROCESS_BWD);
        HIPCHK(c, gs_launch_preprocess_bwd(b, c->cam, c->stream, chain ? chain : 3));
    }
    if (mem == GS_MEM_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));   // dC host buffer no longer needed
    c->did_bwd = true;
    return GS_OK;
}

int gs_color_grads_pack(gs_ctx *c, float *drgb) {
    if (!c || !drgb) return GS_ERR_INVALID;
    if (c->kind != 0) return fail(c, GS_ERR_UNSUPPORTED, "gs_color_grads_pack: 3-D renderer only");
    if (!c->did_bwd) return fail(c, GS_ERR_INVALID, "gs_color_grads_pack: gs_backward first");
    if (bind_device(c)) return GS_ERR_HIP;
    const bool det = c->cfg.deterministic != 0;
    HIPCHK(c, gs_launch_pack_drgb(det ? nullptr : c->g2d.as<float>(), det ? c->g2d.as<long long>() : nullptr, drgb, c->n, c->stream));
    return GS_OK;
}

int gs_sh_grads_from_views(gs_ctx *c, int32_t nviews, const float *cams, const float *drgb, float *d_shs, int flags) {
    if (!c || !cams || !drgb || !d_shs || nviews <= 0) return GS_ERR_INVALID;
    if (c->kind != 0) return fail(c, GS_ERR_UNSUPPORTED, "gs_sh_grads_from_views: 3-D renderer only");
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t bytes = sizeof(float) * GS_VIEW_RECORD_FLOATS * (size_t)nviews;
    HIPCHK(c, c->view_cams.ensure(bytes));
    HIPCHK(c, hipMemcpyAsync(c->view_cams.p, cams, bytes, hipMemcpyHostToDevice, c->stream));   // pageable source: staged before return
    HIPCHK(c, gs_launch_sh_from_views(c->n, c->sh_degree, c->means, nviews, c->view_cams.as<float>(), drgb, d_shs,
                                      (flags & GS_BWD_OVERWRITE) ? 1 : 0, c->stream));
    return GS_OK;
}

}  // extern "C"
