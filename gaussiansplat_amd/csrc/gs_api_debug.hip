// gs_api_debug.hip -- introspection for the parity tests (gs_get_array: the reference's scratch arrays and the tile lists) and the
// profiling hooks (per-stage hipEvent timers, work counters, per-tile clocks, isolated composite launches).
#include "gs_ctx.h"

extern "C" {

int64_t gs_num_gaussians(const gs_ctx *c) { return c ? c->n : 0; }
int64_t gs_num_instances(gs_ctx *c) { if (!c) return 0; (void)settle_totals(c, nullptr, true); return c->n_inst; }
int64_t gs_num_coarse_instances(gs_ctx *c) { if (!c) return 0; (void)settle_totals(c, nullptr, true); return c->two_level ? c->n_coarse : 0; }
int gs_num_rounds(const gs_ctx *c) { return c ? c->n_rounds : 0; }

int gs_get_array(gs_ctx *c, int which, void *dst, int64_t bytes) {
    if (!c || !dst) return GS_ERR_INVALID;
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)c->n;
    const void *src = nullptr;
    size_t need = 0;
    auto need_dbg = [&](int i, size_t w) -> int {
        if (!c->cfg.export_debug) return fail(c, GS_ERR_INVALID, "gs_get_array: needs gs_config.export_debug = 1");
        src = c->dbg[i].p; need = sizeof(float) * w * n; return GS_OK;
    };
    if (which <= GS_ARR_TILE_RECT && !c->did_pre) return fail(c, GS_ERR_INVALID, "gs_get_array: gs_preprocess first");
    if (which >= GS_ARR_SORT_IDXS && which <= GS_ARR_SORTED_KEYS && !c->did_bin) return fail(c, GS_ERR_INVALID, "gs_get_array: gs_bin first");
    if (which >= GS_ARR_TILE_RANGES && which <= GS_ARR_SORTED_KEYS) { if (int rc = settle_totals(c, nullptr, true)) return rc; }
    if (which >= GS_ARR_TILE_RANGES && which <= GS_ARR_SORTED_KEYS && c->n_rounds > 1)
        return fail(c, GS_ERR_INVALID, "gs_get_array: this frame was binned in depth slabs (lists spread over rounds); use gs_config.slab_mode = 0");
    if ((which == GS_ARR_SORTED_IDS || which == GS_ARR_SORTED_KEYS) && c->frame_capped && c->have_l2) {
        // capped lists: only the part of every list the view slot's history says is walked has been written.  Write the rest now (the
        // same kernel with the caps off: same positions, same order); from here on the frame's lists are complete.
        GsBin3Args a = c->last_l2;
        a.cap_src = nullptr; a.tile_nopen = nullptr; a.smax = nullptr; a.tile_ext = nullptr;
        HIPCHK(c, gs_bin3_write_lists(a, c->stream));
        c->frame_capped = false;
    }
    switch (which) {
        case GS_ARR_TS: if (int r = need_dbg(0, 4)) return r; break;
        case GS_ARR_TPS: if (int r = need_dbg(1, 4)) return r; break;
        case GS_ARR_COV3D: if (int r = need_dbg(3, 9)) return r; break;
        case GS_ARR_COV2D: if (int r = need_dbg(4, 4)) return r; break;
        case GS_ARR_BBS: if (int r = need_dbg(6, 4)) return r; break;
        case GS_ARR_INVCOV: src = c->invcov.p; need = sizeof(float) * 4 * n; break;
        case GS_ARR_MU: case GS_ARR_RGB: case GS_ARR_SIG: {
            const size_t w = which == GS_ARR_MU ? 2 : which == GS_ARR_RGB ? 3 : 1;
            if ((size_t)bytes != sizeof(float) * w * n) return fail(c, GS_ERR_INVALID, "gs_get_array: size mismatch");
            std::vector<GsPayload> h(n ? n : 1);
            HIPCHK(c, hipMemcpyAsync(h.data(), c->payload.p, sizeof(GsPayload) * n, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            float *o = static_cast<float *>(dst);
            for (size_t g = 0; g < n; ++g) {
                const GsPayload &p = h[g];
                if (which == GS_ARR_MU) { o[2 * g] = p.mx; o[2 * g + 1] = p.my; }
                else if (which == GS_ARR_RGB) { o[3 * g] = p.r; o[3 * g + 1] = p.g; o[3 * g + 2] = p.b; }
                else o[g] = p.sig;
            }
            return GS_OK;
        }
        case GS_ARR_DEPTH_KEY: src = c->depth_key.p; need = sizeof(uint32_t) * n; break;
        case GS_ARR_TILE_RECT: src = c->rect.p; need = sizeof(uint16_t) * 4 * n; break;
        case GS_ARR_SORT_IDXS: {
            if ((size_t)bytes != sizeof(uint32_t) * n) return fail(c, GS_ERR_INVALID, "gs_get_array: size mismatch");
            if (c->perm_pending) {                                  // a frame binned by the small path (gs_bin_small.hip) sorted nothing globally
                uint32_t *perm = nullptr;
                if (int rc = depth_order(c, &perm)) return rc;
                c->perm_ptr = c->perm_all = perm; c->perm_pending = false;
            }
            if (c->perm_ptr) { src = c->perm_ptr; need = sizeof(uint32_t) * n; break; }
            uint32_t *o = static_cast<uint32_t *>(dst);
            for (size_t g = 0; g < n; ++g) o[g] = (uint32_t)g;
            return GS_OK;
        }
        case GS_ARR_TILE_RANGES: src = c->ranges.p; need = sizeof(uint32_t) * 2 * (size_t)c->gx * c->gy; break;
        case GS_ARR_SORTED_IDS: case GS_ARR_SORTED_KEYS: {
            const size_t ni = (size_t)c->n_inst, nt = (size_t)c->gx * c->gy;
            const size_t w = which == GS_ARR_SORTED_IDS ? sizeof(uint32_t) : sizeof(uint64_t);
            if ((size_t)bytes != w * ni) return fail(c, GS_ERR_INVALID, "gs_get_array: size mismatch");
            std::vector<uint32_t> h(ni ? ni : 1), dk(n ? n : 1), rg(2 * (nt ? nt : 1));
            HIPCHK(c, hipMemcpyAsync(h.data(), c->ids.p, sizeof(uint32_t) * ni, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(dk.data(), c->depth_key.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(rg.data(), c->ranges.p, sizeof(uint32_t) * 2 * nt, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (which == GS_ARR_SORTED_IDS) {
                std::memcpy(dst, h.data(), sizeof(uint32_t) * ni);
            } else {                                   // tile<<32 | depth key (or | id): the key the list order realises
                uint64_t *o = static_cast<uint64_t *>(dst);
                const bool by_index = c->order() == GS_ORDER_INDEX;
                for (size_t t = 0; t < nt; ++t)
                    for (size_t p = rg[2 * t]; p < rg[2 * t + 1] && p < ni; ++p)
                        o[p] = ((uint64_t)t << 32) | (by_index ? h[p] : dk[h[p]]);
            }
            return GS_OK;
        }
        case GS_ARR_GRAD2D: {
            if (!c->did_bwd) return fail(c, GS_ERR_INVALID, "gs_get_array: gs_backward first");
            if ((size_t)bytes != sizeof(float) * 10 * n) return fail(c, GS_ERR_INVALID, "gs_get_array: size mismatch");
            float *o = static_cast<float *>(dst);
            if (c->cfg.deterministic) {
                std::vector<long long> fx((size_t)GS_G2D_STRIDE * (n ? n : 1));
                if (n) HIPCHK(c, hipMemcpyAsync(fx.data(), c->g2d.p, sizeof(long long) * GS_G2D_STRIDE * n, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                for (size_t g = 0; g < n; ++g)
                    for (int i = 0; i < 10; ++i) o[10 * g + i] = (float)((double)fx[GS_G2D_STRIDE * g + i] * gs_fixed_inv(i));
            } else {
                std::vector<float> fl((size_t)GS_G2D_STRIDE * (n ? n : 1));
                if (n) HIPCHK(c, hipMemcpyAsync(fl.data(), c->g2d.p, sizeof(float) * GS_G2D_STRIDE * n, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                for (size_t g = 0; g < n; ++g)
                    for (int i = 0; i < 10; ++i) o[10 * g + i] = fl[GS_G2D_STRIDE * g + i];
            }
            // the device rows hold raw moments: apply the per-gaussian factors with the view's payload (sig, conic)
            std::vector<GsPayload> pay(n ? n : 1);
            std::vector<float> ic(4 * (n ? n : 1));
            if (n) HIPCHK(c, hipMemcpyAsync(pay.data(), c->payload.p, sizeof(GsPayload) * n, hipMemcpyDeviceToHost, c->stream));
            if (n) HIPCHK(c, hipMemcpyAsync(ic.data(), c->invcov.p, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (size_t g = 0; g < n; ++g) {
                float row[10];
                for (int i = 0; i < 10; ++i) row[i] = o[10 * g + i];
                gs_g2d_to_grads(row, pay[g].sig, ic[4 * g], 0.5f * (ic[4 * g + 1] + ic[4 * g + 2]), ic[4 * g + 3]);
                for (int i = 0; i < 10; ++i) o[10 * g + i] = row[i];
            }
            return GS_OK;
        }
        default: return fail(c, GS_ERR_INVALID, "gs_get_array: unknown array");
    }
    if ((size_t)bytes != need) return fail(c, GS_ERR_INVALID, "gs_get_array: size mismatch");
    if (need) HIPCHK(c, hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_get_stage_times(gs_ctx *c, float ms[GS_STAGE_COUNT]) {
    if (!c || !ms) return GS_ERR_INVALID;
    if (!c->cfg.profile_stages) return fail(c, GS_ERR_INVALID, "gs_get_stage_times: needs gs_config.profile_stages = 1");
    if (bind_device(c)) return GS_ERR_HIP;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < GS_STAGE_COUNT; ++s) {
        ms[s] = 0.0f;
        if (c->ev_valid[s]) HIPCHK(c, hipEventElapsedTime(&ms[s], c->ev[s][0], c->ev[s][1]));
    }
    return GS_OK;
}

int gs_get_stage_stats(gs_ctx *c, double sum_ms[GS_STAGE_COUNT], int64_t count[GS_STAGE_COUNT], int reset) {
    if (!c || !sum_ms || !count) return GS_ERR_INVALID;
    if (!c->cfg.profile_stages) return fail(c, GS_ERR_INVALID, "gs_get_stage_stats: needs gs_config.profile_stages = 1");
    if (bind_device(c)) return GS_ERR_HIP;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    harvest_events(c);
    for (int s = 0; s < GS_STAGE_COUNT; ++s) {
        sum_ms[s] = c->ev_sum[s]; count[s] = c->ev_cnt[s];
        if (reset) { c->ev_sum[s] = 0.0; c->ev_cnt[s] = 0; }
    }
    return GS_OK;
}

static int debug_composite_args(gs_ctx *c, int which, int variant, GsCompositeArgs &a) {
    a.W = c->cam.W; a.H = c->cam.H; a.gx = c->gx; a.gy = c->gy; a.t_min = c->cfg.t_min;
    a.ranges = c->ranges.as<uint32_t>(); a.ids = c->ids.as<uint32_t>(); a.payload = c->payload.as<GsPayload>();
    a.image = c->img(); a.trans = c->tr();
    a.dC = c->last_dC; a.walked = nullptr;
    if (variant >= 10000) {                                                  // + 10000: with the two work-counter atomics per tile of a real frame (scratch words; tools/atomics_tail.py)
        variant -= 10000;
        a.walked = reinterpret_cast<unsigned long long *>(static_cast<char *>(c->counters.p) + 160);
    }
    a.final_round = 1;
    a.nseg = 0;
    for (int r = 0; r < c->n_rounds; ++r) {
        if (r > 0 && c->round_gen[r] == 0) continue;
        a.seg_ranges[a.nseg] = r == 0 ? c->ranges.as<uint32_t>() : c->ranges_r[r].as<uint32_t>();
        a.seg_ids[a.nseg] = c->ids.as<uint32_t>() + c->round_ids_off[r];
        ++a.nseg;
    }
    a.g2d = c->cfg.deterministic ? nullptr : c->g2d.as<float>(); a.g2d_fixed = c->cfg.deterministic ? c->g2d.as<long long>() : nullptr;
    c->g2d_clean = false;
    if (c->frame_capped && c->n_rounds == 1) {
        a.tile_ext = c->tile_ext.as<uint2>(); a.cranges = c->cranges.as<uint32_t>(); a.cids = c->cids.as<uint32_t>(); a.clr = c->clr.as<uint16_t>();
        a.ids_w = c->ids.as<uint32_t>(); a.sgx = c->sgx; a.sbs = c->sbs; a.ext_count = c->ext_count();
    }
    a.variant = variant % 100;
    a.cull = (c->cfg.alpha_cull != 0) != (variant >= 1000);                  // +1000: the other cull setting
    // variant tens digit (gs_composite.hip: apply_sched_variant): 0 the frame's own launch order (what production uses for the
    // backward and for the next forward of the slot), 1 tile order, 3 = 0 explicitly
    a.tile_order = lpt_schedule(c) ? c->frame_order : nullptr;
    a.order_len = a.tile_order ? lpt_order_entries(c) : 0;
    a.split_ok = a.tile_order && lpt_front(c) > 0 && c->frame_parts == 1 && !c->frame_capped && c->n_rounds == 1;
    if (which == 1 && a.split_ok && c->snap_order && c->snap_order == a.tile_order) {          // the backward's list segments, as the frame ran them
        a.snap = c->snap.as<float>(); a.seg_len = order_seg_len(c, a.tile_order); a.front = lpt_front(c);
        a.snap_walked = c->snap_walked.as<uint32_t>() + (size_t)c->snap_parity * GS_SEG_SLOTS;
    }
    a.parts = c->frame_parts;                                                // as the frame's own launches (tile clocks: one wave per tile only)
    if (which == 1 && c->frame_seg_n && c->n_rounds == 1 && !c->frame_capped && !a.tile_order) {   // small grid: the backward's list segments
        const long long nt = (long long)c->gx * c->gy;
        a.snap = c->snap.as<float>(); a.seg_hist = c->seg_hist; a.seg_n = c->frame_seg_n;
        a.parts = 4 * a.seg_n * nt <= c->wave_slots ? 4 : 2 * a.seg_n * nt <= c->wave_slots ? 2 : 1;
    }
    if (c->dbg_win_len > 0) {                                                // gs_debug_set_window: a slice of the launch order
        if (!a.tile_order || a.parts > 1) return fail(c, GS_ERR_INVALID, "gs_debug_set_window: the frame has no launch order (or several waves per tile)");
        if (c->dbg_win_start + c->dbg_win_len > a.order_len) return fail(c, GS_ERR_INVALID, "gs_debug_set_window: beyond the launch order");
        a.tile_order += c->dbg_win_start; a.order_len = c->dbg_win_len;
    }
    return GS_OK;
}

int gs_debug_set_window(gs_ctx *c, int32_t start, int32_t len) {
    if (!c) return GS_ERR_INVALID;
    if (start < 0 || len < 0 || (start & 7) || (len & 7)) return fail(c, GS_ERR_INVALID, "gs_debug_set_window: start and len must be non-negative multiples of 8");
    c->dbg_win_start = start; c->dbg_win_len = len;
    return GS_OK;
}

int gs_debug_time_composite(gs_ctx *c, int which, int variant, int reps, float *mean_ms) {
    if (!c || !mean_ms || reps == 0) return GS_ERR_INVALID;
    const bool cold = reps < 0;                                               // negative: -reps launches WITHOUT the warm-up launch (tools/cold_fwd.py)
    if (cold) reps = -reps;
    if (!c->did_fwd) return fail(c, GS_ERR_INVALID, "gs_debug_time_composite: gs_forward first");
    if (which == 1 && !c->did_bwd) return fail(c, GS_ERR_INVALID, "gs_debug_time_composite: gs_backward first");
    if (bind_device(c)) return GS_ERR_HIP;
    GsCompositeArgs a{};
    if (int rc = debug_composite_args(c, which, variant, a)) return rc;
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    auto launch = [&]() -> hipError_t { return which == 0 ? gs_launch_composite_fwd(a, c->stream) : gs_launch_composite_bwd(a, c->stream); };
    if (!cold) HIPCHK(c, launch());   // warm
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; ++i) HIPCHK(c, launch());
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    *mean_ms = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return GS_OK;
}

int gs_debug_tile_clock(gs_ctx *c, int which, int variant, uint64_t *out) {
    if (!c || !out) return GS_ERR_INVALID;
    if (!c->did_fwd) return fail(c, GS_ERR_INVALID, "gs_debug_tile_clock: gs_forward first");
    if (which == 1 && !c->did_bwd) return fail(c, GS_ERR_INVALID, "gs_debug_tile_clock: gs_backward first");
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t ntiles = (size_t)c->gx * c->gy;
    GsCompositeArgs a{};
    const bool by_block = variant < 0;                                        // negative variant: records per WORKGROUP (launches with split tiles):
    if (by_block) variant = -variant;                                         // out holds gs_debug_tile_clock_rows() rows
    if (int rc = debug_composite_args(c, which, variant, a)) return rc;
    if (by_block) {
        // a frame with a launch order, or a small grid whose tiles are shared between waves (pixel parts / list segments)
        if ((!a.tile_order && a.parts <= 1 && !a.seg_hist) || (a.variant / 10) % 10 == 1)
            return fail(c, GS_ERR_INVALID, "gs_debug_tile_clock: records by workgroup need the frame's launch order (or a small grid's tile parts)");
        a.clock_by_block = 1;
    } else { a.split_ok = 0; a.snap = nullptr; }                              // records by tile: whole tiles only
    const size_t rows = !a.clock_by_block ? ntiles : a.tile_order ? (size_t)(lpt_order_entries(c) + gs_seg_units(lpt_front(c))) : (size_t)gs_debug_tile_clock_rows(c);
    if (a.clock_by_block && (size_t)gs_composite_grid_blocks(a, which) > rows) return fail(c, GS_ERR_INVALID, "gs_debug_tile_clock: launch larger than its record");
    HIPCHK(c, c->tile_clock.ensure(sizeof(uint64_t) * GS_TILE_CLOCK_WORDS * (rows ? rows : 1)));
    HIPCHK(c, hipMemsetAsync(c->tile_clock.p, 0, sizeof(uint64_t) * GS_TILE_CLOCK_WORDS * rows, c->stream));
    a.tile_clock = c->tile_clock.as<unsigned long long>();
    for (int rep = 0; rep < 2; ++rep) {                                       // the second launch (warm) is the one kept
        HIPCHK(c, which == 0 ? gs_launch_composite_fwd(a, c->stream) : gs_launch_composite_bwd(a, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(out, c->tile_clock.p, sizeof(uint64_t) * GS_TILE_CLOCK_WORDS * rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_debug_tile_clock_rows(gs_ctx *c) {
    if (!c) return GS_ERR_INVALID;
    if (lpt_schedule(c)) return lpt_order_entries(c) + gs_seg_units(lpt_front(c));
    // small grids: blocks of one part x the most waves a tile is given (pixel parts x list segments of the backward); 0: one wave per tile
    const int len = ((c->gx * c->gy + 7) / 8) * 8;
    const int units = std::max(c->frame_parts, 1) * (c->frame_seg_n ? 4 * c->frame_seg_n : 1);
    return units > 1 ? len * units : 0;
}

int gs_debug_clock_mhz(gs_ctx *c, float *mhz) {
    if (!c || !mhz) return GS_ERR_INVALID;
    if (bind_device(c)) return GS_ERR_HIP;
    HIPCHK(c, c->counters.ensure(GS_COUNTER_BYTES));
    unsigned long long *d = reinterpret_cast<unsigned long long *>(static_cast<char *>(c->counters.p) + 160), h[2] = {0, 1};
    HIPCHK(c, gs_launch_clock_probe(d, c->stream));
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *mhz = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.0f;       // s_memrealtime ticks at 100 MHz
    return GS_OK;
}

int gs_rank_probe_result(const gs_ctx *c) { return c ? c->rank_probe : -1; }

// counters[0..3] = {walked, evaluated} of the last forward and of the last composite backward, summed from the per-tile arrays
static int sum_work_counters(gs_ctx *c) {
    unsigned long long *w = c->counters.as<unsigned long long>();
    const int nt = c->gx * c->gy;
    HIPCHK(c, gs_launch_sum_tiles(c->last_walked, c->tile_work.as<uint32_t>(), nt, w, c->stream));
    if (c->did_bwd_composite && c->tile_walked_b.p && c->tile_work_b.p)
        HIPCHK(c, gs_launch_sum_tiles(c->tile_walked_b.as<uint32_t>(), c->tile_work_b.as<uint32_t>(), nt, w + 2, c->stream));
    else HIPCHK(c, hipMemsetAsync(w + 2, 0, 16, c->stream));
    return GS_OK;
}

int gs_get_work_counters(gs_ctx *c, int64_t *walked_fwd, int64_t *walked_bwd) {
    if (!c) return GS_ERR_INVALID;
    if (!c->did_fwd) return fail(c, GS_ERR_INVALID, "gs_get_work_counters: gs_forward first");
    if (bind_device(c)) return GS_ERR_HIP;
    unsigned long long h[4] = {0, 0, 0, 0};
    if (int rc = sum_work_counters(c)) return rc;
    HIPCHK(c, hipMemcpyAsync(h, c->counters.p, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (walked_fwd) *walked_fwd = (int64_t)h[0];
    if (walked_bwd) *walked_bwd = (int64_t)h[2];
    return GS_OK;
}

int gs_get_list_stats(gs_ctx *c, int64_t out[3]) {
    if (!c || !out) return GS_ERR_INVALID;
    if (!c->did_fwd) return fail(c, GS_ERR_INVALID, "gs_get_list_stats: gs_forward first");
    if (bind_device(c)) return GS_ERR_HIP;
    if (int rc = settle_totals(c, nullptr, true)) return rc;
    out[0] = c->n_inst; out[1] = 0; out[2] = 0;
    if (!c->frame_capped || c->n_rounds != 1) return GS_OK;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(static_cast<char *>(c->counters.p) + 160), h = 0;
    uint32_t e = 0;
    HIPCHK(c, gs_launch_sum_listed(c->tile_ext.as<uint2>(), c->gx * c->gy, d, c->stream));
    HIPCHK(c, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&e, c->ext_count(), sizeof(e), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out[0] = (int64_t)h; out[1] = (int64_t)e; out[2] = 1;
    return GS_OK;
}

int gs_get_tile_parts(gs_ctx *c) {
    if (!c) return GS_ERR_INVALID;
    if (!c->did_fwd) return fail(c, GS_ERR_INVALID, "gs_get_tile_parts: gs_forward first");
    return c->frame_parts;
}

int gs_get_bin_path(gs_ctx *c) {
    if (!c) return GS_ERR_INVALID;
    if (!c->did_bin) return fail(c, GS_ERR_INVALID, "gs_get_bin_path: gs_bin first");
    return c->small_bin ? 3 : c->two_level ? 0 : c->fast_bin ? 2 : 1;
}

int gs_get_work_counters_ex(gs_ctx *c, int64_t out[4]) {
    if (!c || !out) return GS_ERR_INVALID;
    if (!c->did_fwd) return fail(c, GS_ERR_INVALID, "gs_get_work_counters_ex: gs_forward first");
    if (bind_device(c)) return GS_ERR_HIP;
    unsigned long long h[4] = {0, 0, 0, 0};
    if (int rc = sum_work_counters(c)) return rc;
    HIPCHK(c, hipMemcpyAsync(h, c->counters.p, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out[0] = (int64_t)h[0]; out[1] = (int64_t)h[2]; out[2] = (int64_t)h[1]; out[3] = (int64_t)h[3];
    return GS_OK;
}

}  // extern "C"
