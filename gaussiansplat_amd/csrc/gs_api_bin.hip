// gs_api_bin.hip -- gs_bin (compactIdxs, reference src/forward.jl:118-161): the depth order and the per-tile splat lists.
// Two-level tile lists enqueued speculatively against the capacities at hand (no host wait inside a frame), written only as far as
// the view slot's history says they are walked (capped lists), optionally in rounds over depth slabs; the radix paths behind them.
#include "gs_ctx.h"

// ---------------------------------------------------------------- binning in depth slabs
// A dense scene walks only the front of every tile's list before the transmittance early-out stops it (C3: 28 % of the
// 30 M instances, C5: 6 % of 507 M), yet the classic path sorts all of them.  With gs_config.slab_mode the frame is binned
// in rounds over slabs of the depth order: round 0 lists the front slab for every tile and composites it; a tile whose
// 256 pixels are all frozen is complete; round r lists the next slab only for the tiles still open (a gaussian whose
// rectangle holds no open tile drops out, instances of completed tiles inside the other rectangles are dropped while they
// are generated) and the forward resumes each open tile where it stopped -- same entries, same order, same 64-entry batch
// boundaries as the single list, so image, transmittance and (deterministic mode) gradients are bit-identical to the
// classic path.  The slab bounds come from the share of the instances the previous frame walked; a first frame, a sparse
// scene (share >= GS_SLAB_MAX_RATIO) or t_min = 0 take the classic single round.
// Measured on MI355X.  With the radix binning of round 1 (16 B of traffic per instance): at C3 (share 0.28) two rounds cost
// more than they save, at C5 (share 0.06) three rounds cut the frame from 10.4 to 6.8 ms.  With the two-level binning
// (gs_bin3.hip: 4 B written per instance, no pass over the instances) the single round wins at C5 as well (5.54 vs 5.67 ms:
// three forward launches with their tails and three level-1 passes cost more than the 0.4 ms of list writes they save), so
// the automatic mode now engages only below a share of 0.03 (GS_SLAB_MAX_RATIO overrides; the tests use 0.15).
#define GS_SLAB_MAX_RATIO 0.03
static double slab_max_ratio(const gs_ctx *c) { return c->cfg.slab_max_ratio > 0.0f ? (double)c->cfg.slab_max_ratio : GS_SLAB_MAX_RATIO; }
static int plan_rounds(gs_ctx *c) {
    c->n_rounds = 1;
    c->slab_lo[0] = 0; c->slab_lo[1] = c->n;
    if (!c->fast_bin || c->cfg.t_min <= 0.0f || c->n < 1024 || c->order() == GS_ORDER_INDEX) return 1;
    double f[GS_MAX_ROUNDS] = {1.0, 1.0, 1.0, 1.0};
    int R = 1;
    if (c->cfg.slab_fractions[0] > 0.0f) {                                  // tests / experiments: explicit fractions
        for (int k = 0; k < 3 && k + 1 < GS_MAX_ROUNDS && c->cfg.slab_fractions[k] > 0.0f; ++k) { f[k] = c->cfg.slab_fractions[k]; R = k + 2; }
    } else if (c->cfg.slab_mode == 1 && c->walked_ratio >= 0.0 && c->walked_ratio < slab_max_ratio(c)) {
        const double rho = c->walked_ratio;
        f[0] = std::min(0.9, std::max(0.02, 2.0 * rho + 0.02));
        f[1] = std::min(0.95, std::max(f[0] + 0.05, 6.0 * rho + 0.05));
        R = 3;
    }
    if (R == 1) return 1;
    int64_t prev = 0;
    int r = 0;
    for (int k = 0; k + 1 < R; ++k) {
        int64_t b = (int64_t)(f[k] * (double)c->n);
        b = std::min(c->n, std::max(prev, b));
        if (b > prev && b < c->n) { c->slab_lo[++r] = b; prev = b; }
    }
    c->slab_lo[++r] = c->n;
    c->n_rounds = r;
    return r;
}

// Two-level binning of one round (gs_bin3.hip).  two_level_count enqueues the level-1 histogram of the slab's gaussians
// (after it the round's three totals are on the device: coarse instances listed, fine instances of the slab, of all n);
// two_level_lists enqueues the super-tile lists and the tile lists for buffers that hold `coarse` / `fine` entries -- the
// actual totals once the host knows them, or (speculative launch) the capacities of the buffers at hand: the kernels compare
// the totals on the device with these numbers and list nothing when a buffer would overflow.
static GsBin3L1 two_level_args(gs_ctx *c, const uint32_t *perm_slab, int64_t n_all, int64_t nr, const uint8_t *sdone, size_t cap_coarse, size_t cap_fine) {
    GsBin3L1 b{};
    b.rect = c->rect.as<uint16_t>(); b.perm = perm_slab; b.sdone = sdone; b.n = n_all; b.n_slab = nr; b.sgx = c->sgx; b.ns = c->sgx * c->sgy; b.sbs = c->sbs;
    b.rect_sorted = c->rect_sorted.as<uint32_t>(); b.table = c->l1_table.as<uint32_t>(); b.row_total = c->l1_rows.as<uint32_t>();
    b.partials = c->l1_partials.as<uint32_t>(); b.totals = c->bin_totals(); b.cranges = c->cranges.as<uint32_t>();
    b.cids = c->cids.as<uint32_t>(); b.clr = c->clr.as<uint16_t>();
    b.tilecnt = c->tilecnt.as<uint32_t>(); b.ntiles = c->gx * c->gy;
    b.cap_coarse = (uint32_t)std::min<size_t>(cap_coarse, 0xFFFFFFFEu); b.cap_fine = (uint32_t)std::min<size_t>(cap_fine, 0xFFFFFFFEu);
    return b;
}
static int two_level_count(gs_ctx *c, const uint32_t *perm_slab, int64_t n_all, int64_t nr, const uint8_t *sdone, bool to_host = false) {
    const int ns = c->sgx * c->sgy;
    HIPCHK(c, c->rect_sorted.ensure(sizeof(uint32_t) * 2 * (size_t)(nr ? nr : 1)));
    HIPCHK(c, c->l1_table.ensure(sizeof(uint32_t) * gs_bin3_table_words(nr, ns)));
    HIPCHK(c, c->l1_rows.ensure(sizeof(uint32_t) * (size_t)ns));
    HIPCHK(c, c->l1_partials.ensure(sizeof(uint32_t) * gs_bin3_partial_words(n_all, ns)));
    HIPCHK(c, c->counters.ensure(GS_COUNTER_BYTES));
    HIPCHK(c, c->cranges.ensure(sizeof(uint32_t) * 2 * (size_t)ns));
    HIPCHK(c, c->tilecnt.ensure(sizeof(uint32_t) * (size_t)c->gx * c->gy));
    GsBin3L1 b = two_level_args(c, perm_slab, n_all, nr, sdone, 0, 0);
    if (to_host) {                                                          // the layout settle_totals reads: counter block at pinned + 8
        b.host_totals = c->pinned + 8 + 32; b.host_walked = c->pinned + 8; b.walked_src = c->counters.as<uint32_t>();
        // the previous forward's walked entries, per tile (valid only if that forward ran on this grid: prev_counters_valid)
        const bool same_grid = c->counters_grid == (((int64_t)c->gx << 32) | (int64_t)c->gy) && c->last_walked;
        if (!same_grid) c->prev_counters_valid = false;
        b.tile_walked = c->prev_counters_valid ? c->last_walked : nullptr; b.n_tile_walked = c->gx * c->gy;
    }
    HIPCHK(c, gs_bin3_l1_count(b, c->stream));
    return GS_OK;
}
// cap_src (round 0 of a one-round frame only): per-tile walked counts of the view slot's previous forward -> capped lists
static int two_level_lists(gs_ctx *c, const uint32_t *perm_slab, int64_t n_all, int64_t nr, size_t coarse, size_t fine, uint32_t *ranges, uint32_t *ids_out,
                           const uint8_t *done, const uint8_t *sdone, const uint32_t *cap_src = nullptr) {
    const int ns = c->sgx * c->sgy;
    if (!done) { c->frame_capped = false; c->have_l2 = false; }
    if (coarse == 0) {                                      // nothing listed: every tile range of the round is empty
        HIPCHK(c, hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)c->gx * c->gy, c->stream));
        return GS_OK;
    }
    const int64_t max_work = gs_bin3_max_work((int64_t)coarse, ns);
    HIPCHK(c, c->cids.ensure(sizeof(uint32_t) * coarse));
    HIPCHK(c, c->clr.ensure(sizeof(uint16_t) * coarse));
    HIPCHK(c, c->segcnt.ensure(sizeof(uint32_t) * ((size_t)1 << (2 * c->sbs)) * (size_t)max_work));
    HIPCHK(c, gs_bin3_l1_scatter(two_level_args(c, perm_slab, n_all, nr, sdone, coarse, fine), c->stream));
    GsBin3Args a{};
    a.cranges = c->cranges.as<uint32_t>(); a.cids = c->cids.as<uint32_t>(); a.clr = c->clr.as<uint16_t>(); a.ranges = ranges; a.tilecnt = c->tilecnt.as<uint32_t>();
    a.done = done; a.segcnt = c->segcnt.as<uint32_t>(); a.ids_out = ids_out;
    a.gx = c->gx; a.gy = c->gy; a.sgx = c->sgx; a.ns = ns; a.sbs = c->sbs; a.max_work = (int)max_work;
    a.wide = (uint64_t)fine * 4ull >= (1ull << 32) || (c->cfg.debug_flags & GS_DEBUG_WIDE_CURSORS) != 0;
    a.totals = c->bin_totals(); a.cap_coarse = (uint32_t)std::min<size_t>(coarse, 0xFFFFFFFEu); a.cap_fine = (uint32_t)std::min<size_t>(fine, 0xFFFFFFFEu);
    if (cap_src && !done) {
        const size_t nt = (size_t)c->gx * c->gy;
        HIPCHK(c, c->tile_nopen.ensure(sizeof(uint32_t) * nt));
        HIPCHK(c, c->smax.ensure(sizeof(uint32_t) * (size_t)ns));
        HIPCHK(c, c->tile_ext.ensure(sizeof(uint2) * nt));
        a.cap_src = cap_src; a.tile_nopen = c->tile_nopen.as<uint32_t>(); a.smax = c->smax.as<uint32_t>(); a.tile_ext = c->tile_ext.as<uint2>();
        a.ext_count = c->ext_count();
        c->frame_capped = true;
    }
    HIPCHK(c, gs_bin3_build_lists(a, c->stream));
    if (!done) { c->last_l2 = a; c->have_l2 = true; }
    return GS_OK;
}

// Capped lists: whose history, if any, caps the lists of the frame being binned (null: every list is written in full).  Engages on
// one-round frames of the two-level path with the early-out on, when the frame's view slot has rendered this grid before, and --
// unless gs_config.list_cap = 2 -- only when the ctx's previous frame walked less than GS_LIST_CAP_MAX_RATIO of its list entries on a
// grid with more tiles than wave slots.
#define GS_LIST_CAP_MAX_RATIO 0.15
static int list_cap_source(gs_ctx *c, int rounds, const uint32_t **out) {
    *out = nullptr;
    const int64_t ntiles = (int64_t)c->gx * c->gy, grid = ((int64_t)c->gx << 32) | (int64_t)c->gy;
    if (!c->two_level || rounds != 1 || c->cfg.list_cap == 1 || !(c->cfg.t_min > 0.0f) || ntiles <= 0) return GS_OK;
    if (c->cfg.debug_flags & GS_DEBUG_TINY_CAPS) {          // tests: every tile capped at the minimum, whatever it walked
        if (c->zero_tiles.cap < sizeof(uint32_t) * (size_t)ntiles) {
            HIPCHK(c, c->zero_tiles.ensure(sizeof(uint32_t) * (size_t)ntiles));
            HIPCHK(c, hipMemsetAsync(c->zero_tiles.p, 0, c->zero_tiles.cap, c->stream));
        }
        *out = c->zero_tiles.as<uint32_t>();
        return GS_OK;
    }
    // Engages where it pays (measured, profiles/r04b_kernel_stats_*): the write pass is bound by its entries only when most of them
    // are never walked -- C5 (6 % walked): 591 -> 129 us; at C3 (28 % walked) the pass goes 45 -> 37 us and the cap pass costs that.
    if (c->cfg.list_cap != 2 && (ntiles <= c->wave_slots || !(c->walked_ratio >= 0.0 && c->walked_ratio < GS_LIST_CAP_MAX_RATIO))) return GS_OK;
    const int k = order_index(c);
    if (k == GS_MAX_VIEW_SLOTS && c->cfg.schedule != 4) return GS_OK;          // frames without a slot: history only under schedule 4
    if (c->slots[k].walked_grid != grid || !c->slots[k].walked().p) return GS_OK;
    *out = c->slots[k].walked().as<uint32_t>();
    return GS_OK;
}

// later round of a slab frame on the two-level path
static int bin_round_two_level(gs_ctx *c, int r) {
    const int64_t lo = c->slab_lo[r], nr = c->slab_lo[r + 1] - lo;
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    const int ns = c->sgx * c->sgy;
    const uint32_t *perm = c->perm_all + lo;
    HIPCHK(c, c->ranges_r[r].ensure(sizeof(uint32_t) * 2 * (size_t)ntiles));
    HIPCHK(c, c->sdone.ensure((size_t)ns));
    {
        StageTimer t(c, GS_STAGE_COUNT_SCAN);
        HIPCHK(c, gs_launch_super_done(c->tile_done.as<uint8_t>(), c->gx, c->gy, c->sgx, c->sgy, c->sbs, c->sdone.as<uint8_t>(), c->stream));
        if (int rc = two_level_count(c, perm, nr, nr, c->sdone.as<uint8_t>())) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(c->pinned, c->bin_totals(), 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_count, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev_count));
    harvest_events(c);
    const int64_t coarse = (int64_t)c->pinned[0];
    c->round_gen[r] = (int64_t)c->pinned[1];
    c->round_ids_off[r] = c->round_ids_off[r - 1] + (size_t)c->round_gen[r - 1];
    if (c->round_ids_off[r] + (size_t)c->round_gen[r] > (size_t)c->n_inst) return fail(c, GS_ERR_HIP, "gs_forward: slab instance accounting out of range");
    {
        StageTimer t(c, GS_STAGE_TILE_SORT);
        if (int rc = two_level_lists(c, perm, nr, nr, (size_t)coarse, (size_t)c->round_gen[r], c->ranges_r[r].as<uint32_t>(), c->ids.as<uint32_t>() + c->round_ids_off[r],
                                     c->tile_done.as<uint8_t>(), c->sdone.as<uint8_t>())) return rc;
    }
    return GS_OK;
}

// Lists of round r (r >= 1) for the tiles still open; called from gs_forward after the forward of round r - 1.
int bin_round(gs_ctx *c, int r) {
    if (c->two_level) return bin_round_two_level(c, r);
    const int64_t lo = c->slab_lo[r], nr = c->slab_lo[r + 1] - lo;
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    const uint32_t *perm = c->perm_all + lo;
    HIPCHK(c, c->live2d.ensure(sizeof(uint32_t) * 2 * (size_t)(c->gx + 1) * (c->gy + 1)));      // the table + the row-pass scratch
    HIPCHK(c, c->rect_r.ensure(sizeof(uint16_t) * 4 * (size_t)(c->n ? c->n : 1)));
    HIPCHK(c, c->offsets_r.ensure(sizeof(uint32_t) * ((size_t)nr + 1)));
    HIPCHK(c, c->live_total.ensure(sizeof(uint32_t) * GS_MAX_ROUNDS));
    HIPCHK(c, c->ranges_r[r].ensure(sizeof(uint32_t) * 2 * (size_t)ntiles));
    {
        StageTimer t(c, GS_STAGE_COUNT_SCAN);
        HIPCHK(c, gs_launch_live_prefix(c->tile_done.as<uint8_t>(), c->gx, c->gy, c->live2d.as<uint32_t>(),
                                        c->live2d.as<uint32_t>() + (size_t)(c->gx + 1) * (c->gy + 1), c->stream));
        HIPCHK(c, gs_launch_count_scan_live(c->rect.as<uint16_t>(), perm, c->live2d.as<uint32_t>(), c->gx, c->rect_r.as<uint16_t>(),
                                            c->offsets_r.as<uint32_t>(), c->block_sums.as<uint32_t>(), nr, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->pinned, c->offsets_r.as<uint32_t>() + nr, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_count, c->stream));
    {
        StageTimer t(c, GS_STAGE_RANGES);                               // does not need the count: keeps the GPU busy while the host waits
        HIPCHK(c, gs_launch_tile_ranges(c->rect_r.as<uint16_t>(), perm, nr, c->diff.as<int>(), c->gx, c->gy, c->ranges_r[r].as<uint32_t>(),
                                        c->tile_done.as<uint8_t>(), c->stream));
    }
    HIPCHK(c, hipEventSynchronize(c->ev_count));
    harvest_events(c, GS_STAGE_RANGES);
    c->round_gen[r] = (int64_t)c->pinned[0];
    c->round_ids_off[r] = c->round_ids_off[r - 1] + (size_t)c->round_gen[r - 1];
    if (c->round_gen[r] == 0) return GS_OK;
    if (c->round_ids_off[r] + (size_t)c->round_gen[r] > (size_t)c->n_inst) return fail(c, GS_ERR_HIP, "gs_forward: slab instance accounting out of range");
    {
        StageTimer t(c, GS_STAGE_TILE_SORT);
        GsBin2Args b{};
        b.n = nr; b.n_inst = c->round_gen[r]; b.gx = c->gx; b.lo_bits = c->lo_bits; b.hi_bits = c->hi_bits; b.gid_bits = c->gid_bits;
        b.offsets = c->offsets_r.as<uint32_t>(); b.perm = perm; b.rect = c->rect_r.as<uint16_t>();
        b.cs = c->cs.as<uint32_t>(); b.block_hist = c->table.as<uint32_t>(); b.digit_total = c->digit_total.as<uint32_t>();
        b.buf_a = c->words.as<uint32_t>(); b.ids_out = c->ids.as<uint32_t>() + c->round_ids_off[r]; b.ballot_ranks = c->cfg.rank_mode != 0;
        b.done = c->tile_done.as<uint8_t>(); b.live_total = c->live_total.as<uint32_t>() + r;
        HIPCHK(c, gs_bin2_build_lists(b, c->stream));
    }
    return GS_OK;
}

// The frame's totals arrive in pinned memory behind ev_count.  On the two-level path the host does not wait for them inside
// gs_bin (speculative launch): it enqueues the lists against the CAPACITIES of the buffers it already has, the kernels compare
// the totals on the device with those capacities (gs_bin3.hip: lists_overflow) and list nothing if a buffer is too small;
// settle_totals, called once the host needs the numbers (after gs_forward has enqueued the composite), reads them and -- in
// the rare frame whose lists outgrew a buffer -- grows the buffers and enqueues the lists again (returns 1: the caller
// re-enqueues what it had enqueued on top of the empty lists).  The GPU never idles while the host wakes up, and there is no
// stream synchronisation between gs_preprocess and the end of the frame.
// The bucket path of the depth sort reported a bucket beyond a workgroup's capacity (it was sorted through global memory: correct,
// slow): the next 64 frames use the classic sort, then the bucket path is tried again.  Read once the frame's ev_count has passed.
static void dsort_feedback(gs_ctx *c) {
    if (!c->dsort_buckets_used) return;
    c->dsort_buckets_used = false;
    if (*c->dsort_stat() != 0u && c->cfg.depth_sort != 2) c->dsort_classic_until = c->frame_id + 64;
}

static int bin_frame(gs_ctx *c, bool special_paths);

int settle_totals(gs_ctx *c, bool *redo, bool may_relist) {
    if (redo) *redo = false;
    if (!c->pending_totals) return GS_OK;
    HIPCHK(c, hipEventSynchronize(c->ev_count));
    c->pending_totals = false;
    harvest_events(c);
    dsort_feedback(c);
    // pinned + 8: the counter block {walked_f, evaluated_f, walked_b, evaluated_b (u64) ... | byte 128: coarse listed, fine of the slab, fine of all}
    unsigned long long walked_prev = 0;
    std::memcpy(&walked_prev, c->pinned + 8, sizeof(walked_prev));
    const uint32_t coarse = c->pinned[8 + 32], fine_slab = c->pinned[8 + 33], fine_all = c->pinned[8 + 34];
    if (fine_all == 0xFFFFFFFFu)
        return fail(c, GS_ERR_UNSUPPORTED, "gs_bin: more than 2^32 - 2 tile instances (32-bit list offsets); reduce the scene or the image");
    if (c->prev_counters_valid && c->prev_n_inst > 0) c->walked_ratio = (double)walked_prev / (double)c->prev_n_inst;
    c->prev_counters_valid = false;
    c->n_inst = (int64_t)fine_all;
    c->n_coarse = (int64_t)coarse;
    c->round_gen[0] = c->n_rounds > 1 ? (int64_t)fine_slab : c->n_inst;
    c->round_ids_off[0] = 0;
    c->coarse_listed = (int64_t)coarse;
    if (!c->spec_lists) return GS_OK;
    c->spec_lists = false;
    if ((size_t)coarse <= c->spec_cap_coarse && (size_t)fine_slab <= c->spec_cap_fine) return GS_OK;
    if (!may_relist) { c->did_bin = false; return GS_OK; }                  // the frame is being abandoned (a new gs_preprocess / gs_bin follows)
    // a list outgrew its buffer: nothing was listed (all ranges empty).  Grow and list again with the real totals -- in FULL: the caps'
    // source is the slot's walked array, and the forward that just ran on the empty lists has overwritten it with zeros (rare frame:
    // a model that outgrew its buffers; full lists are the fast choice there)
    HIPCHK(c, c->ids.ensure(sizeof(uint32_t) * (size_t)(c->n_inst ? c->n_inst : 1)));
    c->cap_src = nullptr;
    {
        StageTimer t(c, GS_STAGE_TILE_SORT);
        if (int rc = two_level_lists(c, c->perm_ptr, c->n, c->slab_lo[1], (size_t)coarse, (size_t)fine_slab, c->ranges.as<uint32_t>(), c->ids.as<uint32_t>(), nullptr, nullptr,
                                     nullptr)) return rc;
    }
    if (redo) *redo = true;
    return GS_OK;
}

// The depth order of the frame (CUDA.sortperm, forward.jl:103) -> c->perm.  Also called by gs_get_array for a frame binned by the small
// path, which needs no global order.
int depth_order(gs_ctx *c, uint32_t **perm_out) {
    const size_t n1 = c->n ? (size_t)c->n : 1;
    StageTimer t(c, GS_STAGE_DEPTH_SORT);
    HIPCHK(c, c->pairs_a.ensure(sizeof(uint64_t) * n1));
    HIPCHK(c, c->pairs_b.ensure(sizeof(uint64_t) * n1));
    HIPCHK(c, c->perm.ensure(sizeof(uint32_t) * n1));
    HIPCHK(c, c->table.ensure(sizeof(uint32_t) * gs_sort_table_entries(c->n)));
    HIPCHK(c, c->digit_total.ensure(sizeof(uint32_t) * 4 * 256));
    int in_b = 0;
    uint32_t *perm = c->perm.as<uint32_t>();            // the last pass writes the permutation itself (low word of the pairs)
    // Two steps (256 key-range buckets, then one workgroup per bucket in LDS: 4 launches) when this frame's preprocess left the
    // key range, the mean bucket is well inside a workgroup's capacity, and no oversize bucket was reported lately; else the
    // classic four LSD passes (12 launches).  Same permutation either way.
    const bool buckets = c->range_valid && c->dsort_can_bucket();
    c->dsort_buckets_used = buckets;
    if (c->range_valid && !buckets)                     // folded but not consumed (cannot happen with one predicate; kept so that a stale union never survives)
        HIPCHK(c, gs_depth_range_reset(c->key_range.as<uint32_t>() + (size_t)c->range_parity * gs_depth_range_parity_words(), c->stream, 1));
    if (buckets) {
        uint32_t *range = c->key_range.as<uint32_t>() + (size_t)c->range_parity * gs_depth_range_parity_words();
        uint32_t *other = c->key_range.as<uint32_t>() + (size_t)(c->range_parity ^ 1) * gs_depth_range_parity_words();
        c->dsort_stat_parity = c->range_parity;
        *c->dsort_stat() = 0u;                          // (no kernel of an earlier frame writes this parity's word any more: two frames back)
        HIPCHK(c, gs_depth_sort_buckets(c->depth_key.as<uint32_t>(), c->pairs_a.as<uint64_t>(), c->pairs_b.as<uint64_t>(), c->n, c->table.as<uint32_t>(),
                                        c->digit_total.as<uint32_t>(), perm, range, other, c->dsort_stat(), c->stream, c->cfg.rank_mode != 0));
    } else {
        // (depth | id) pairs are formed by the first pass from the 32-bit keys; the last pass writes only the ids
        HIPCHK(c, gs_radix_sort_u64(c->pairs_a.as<uint64_t>(), c->pairs_b.as<uint64_t>(), c->n, 32, 64, c->table.as<uint32_t>(),
                                    c->digit_total.as<uint32_t>(), &in_b, c->stream, c->cfg.rank_mode != 0, perm, c->depth_key.as<uint32_t>()));
    }
    *perm_out = perm;
    return GS_OK;
}

// Small frames (gs_bin_small.hip): the whole of gs_bin in one launch, one workgroup per tile.  The ids buffer holds every (gaussian, tile)
// pair the path admits, so nothing is speculative; the totals travel to the host as on the two-level path.
static int bin_small(gs_ctx *c) {
    const size_t n = (size_t)c->n, nt = (size_t)c->gx * c->gy;
    c->n_rounds = 1; c->slab_lo[0] = 0; c->slab_lo[1] = c->n;
    c->frame_capped = false; c->have_l2 = false; c->cap_src = nullptr; c->spec_lists = false;
    HIPCHK(c, c->ids.ensure(sizeof(uint32_t) * n * nt));
    if (c->range_valid)                                                     // (gs_preprocess judged otherwise and folded the key range: nobody will consume it)
        HIPCHK(c, gs_depth_range_reset(c->key_range.as<uint32_t>() + (size_t)c->range_parity * gs_depth_range_parity_words(), c->stream, 1));
    GsBinSmallArgs a{};
    a.depth_key = c->order() != GS_ORDER_INDEX ? c->depth_key.as<uint32_t>() : nullptr; a.rect = c->rect.as<uint2>();
    a.n = (int)c->n; a.gx = c->gx; a.gy = c->gy; a.ntiles = (int)nt;
    a.ranges = c->ranges.as<uint32_t>(); a.ids = c->ids.as<uint32_t>(); a.totals = c->bin_totals();
    a.host_totals = c->pinned + 8 + 32; a.host_walked = c->pinned + 8; a.walked_src = c->counters.as<uint32_t>();
    const bool same_grid = c->counters_grid == (((int64_t)c->gx << 32) | (int64_t)c->gy) && c->last_walked;
    if (!same_grid) c->prev_counters_valid = false;
    a.tile_walked = c->prev_counters_valid ? c->last_walked : nullptr; a.n_tile_walked = (int)nt;
    {   // the rows the composite backward accumulates into (64 B per gaussian; nothing touches them between here and that kernel)
        const size_t bytes = (c->cfg.deterministic ? sizeof(long long) : sizeof(float)) * GS_G2D_STRIDE * n;
        HIPCHK(c, c->g2d.ensure(bytes));
        a.zero = c->g2d.as<uint4>(); a.zero_words16 = bytes / sizeof(uint4);
    }
    {
        StageTimer t(c, GS_STAGE_TILE_SORT);
        HIPCHK(c, gs_bin_small(a, c->stream));
    }
    c->g2d_clean = true;
    HIPCHK(c, hipEventRecord(c->ev_count, c->stream));
    c->pending_totals = true;
    c->did_bin = true; c->did_fwd = c->did_bwd = false;
    return GS_OK;
}

static int bin_frame(gs_ctx *c, bool special_paths);

extern "C" int gs_bin(gs_ctx *c, int32_t gx, int32_t gy) {
    if (!c) return GS_ERR_INVALID;
    if (!c->did_pre) return fail(c, GS_ERR_INVALID, "gs_bin: gs_preprocess first");
    if ((gx != 0 || gy != 0) && (gx != c->gx || gy != c->gy))
        return fail(c, GS_ERR_UNSUPPORTED, "gs_bin: blocks must equal ceil(W/16) x ceil(H/16)");
    if (bind_device(c)) return GS_ERR_HIP;
    if (int rc = settle_totals(c, nullptr, false)) return rc;               // a frame that was binned but never rendered
    return bin_frame(c, true);
}

// special_paths: small frames may take their own path (gs_bin_small.hip)
static int bin_frame(gs_ctx *c, bool special_paths) {
    const size_t n = (size_t)c->n;
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    uint32_t *perm = nullptr;
    const bool small = special_paths && c->small_bin_possible();
    c->small_bin = small; c->g2d_clean = false;
    c->perm_pending = small && c->order() != GS_ORDER_INDEX;               // the small path sorts nothing globally: renderer.sortIdxs on demand (gs_get_array)
    if (c->order() != GS_ORDER_INDEX && !small) {
        if (int rc = depth_order(c, &perm)) return rc;
    }
    c->perm_ptr = perm; c->perm_all = perm;
    int tile_bits = 1;
    while ((1LL << tile_bits) < ntiles) ++tile_bits;
    int gid_bits = 1;
    while ((1LL << gid_bits) < c->n) ++gid_bits;
    const int passes = (tile_bits + 7) / 8;
    const int lo_bits = passes <= 1 ? tile_bits : (tile_bits + 1) / 2, hi_bits = tile_bits - lo_bits;
    const int bin_path = c->cfg.bin_path == 3 ? 0 : c->cfg.bin_path;
    const bool fast = bin_path != 1 && passes <= 2 && hi_bits + gid_bits <= 32 && gs_tile_ranges_supported(c->gx, c->gy);
    // two-level path (gs_bin3.hip): lists per super-tile of 8 x 8 tiles first; its bitmap must fit in LDS
    c->sbs = gs_bin3_sb_shift(c->gx, c->gy, (c->cfg.debug_flags & GS_DEBUG_SUPER16) ? 4 : (c->cfg.debug_flags & GS_DEBUG_SUPER8) ? 3 : 0);
    const int sb = 1 << c->sbs;
    c->sgx = (c->gx + sb - 1) / sb; c->sgy = (c->gy + sb - 1) / sb;
    c->two_level = small || (fast && bin_path == 0 && gs_bin3_supported(c->sgx * c->sgy));
    c->tile_bits = tile_bits; c->gid_bits = gid_bits; c->lo_bits = lo_bits; c->hi_bits = hi_bits; c->fast_bin = fast;
    HIPCHK(c, c->ranges.ensure(sizeof(uint32_t) * 2 * (size_t)(ntiles ? ntiles : 1)));
    HIPCHK(c, c->counters.ensure(GS_COUNTER_BYTES));
    // the slab plan needs the previous frame's walked share, which the read-back below delivers: the plan of THIS frame uses
    // the share known so far (one frame of lag; only speed depends on it)
    if (small) return bin_small(c);
    const int R = plan_rounds(c);
    const int64_t n0 = c->slab_lo[1];                                       // list positions of round 0
    HIPCHK(c, c->block_sums.ensure(sizeof(uint32_t) * 3 * (n / 2048 + 2)));
    c->spec_lists = false;
    if (c->two_level) {
        {
            StageTimer t(c, GS_STAGE_COUNT_SCAN);
            if (int rc = two_level_count(c, perm, c->n, n0, nullptr, c->n > 0)) return rc;
        }
        // the previous frame's walked count (bytes 0..7 of the counter block) and this frame's totals (bytes 128..139) travel to the host:
        // stored into coherent pinned memory by the scan kernel itself (no copy command in the stream); an empty model launches nothing
        if (c->n <= 0) HIPCHK(c, hipMemcpyAsync(c->pinned + 8, c->counters.p, GS_COUNTER_BYTES, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_count, c->stream));
        c->pending_totals = true;
        // speculative launch: one round, and buffers from an earlier frame to launch against
        const size_t cap_coarse = std::min(c->cids.cap / sizeof(uint32_t), c->clr.cap / sizeof(uint16_t)), cap_fine = c->ids.cap / sizeof(uint32_t);
        if (int rc = list_cap_source(c, R, &c->cap_src)) return rc;
        if (R == 1 && cap_coarse > 0 && cap_fine > 0) {
            c->spec_lists = true; c->spec_cap_coarse = cap_coarse; c->spec_cap_fine = cap_fine;
            StageTimer t(c, GS_STAGE_TILE_SORT);
            if (int rc = two_level_lists(c, perm, c->n, n0, cap_coarse, cap_fine, c->ranges.as<uint32_t>(), c->ids.as<uint32_t>(), nullptr, nullptr, c->cap_src)) return rc;
        } else {                                                            // first frame of a ctx, or a slab frame: the host needs the totals now
            if (int rc = settle_totals(c, nullptr, true)) return rc;
            HIPCHK(c, c->ids.ensure(sizeof(uint32_t) * (size_t)(c->n_inst ? c->n_inst : 1)));
            StageTimer t(c, GS_STAGE_TILE_SORT);
            if (int rc = two_level_lists(c, perm, c->n, n0, (size_t)c->coarse_listed, (size_t)c->round_gen[0], c->ranges.as<uint32_t>(), c->ids.as<uint32_t>(), nullptr, nullptr,
                                         c->cap_src)) return rc;
        }
        c->did_bin = true; c->did_fwd = c->did_bwd = false;
        return GS_OK;
    }
    // ---- radix paths (bin_path 2 / 1; grids the two-level path refuses): the host reads the instance count before the instance passes
    c->n_coarse = 0;
    c->frame_capped = false; c->have_l2 = false; c->cap_src = nullptr;
    {
        StageTimer t(c, GS_STAGE_COUNT_SCAN);
        HIPCHK(c, c->offsets.ensure(sizeof(uint32_t) * (n + 1)));
        HIPCHK(c, gs_launch_count_scan(c->rect.as<uint16_t>(), perm, c->offsets.as<uint32_t>(), c->block_sums.as<uint32_t>(), c->n, c->stream));
    }
    // the one host read-back of the frame (the reference reads maxHits back, forward.jl:139): the instance count, the
    // generated positions of round 0 and the previous frame's walked count.  Work that does not need the count (the tile
    // ranges) is enqueued BEFORE the host waits, so the GPU stays busy while the host wakes up and launches the instance passes.
    HIPCHK(c, hipMemcpyAsync(c->pinned, c->offsets.as<uint32_t>() + n, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pinned + 1, c->offsets.as<uint32_t>() + n0, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (c->prev_counters_valid && (c->counters_grid != (((int64_t)c->gx << 32) | (int64_t)c->gy) || !c->last_walked)) c->prev_counters_valid = false;
    if (c->prev_counters_valid) {
        HIPCHK(c, gs_launch_sum_tiles(c->last_walked, c->tile_work.as<uint32_t>(), c->gx * c->gy, c->counters.as<unsigned long long>(), c->stream));
        HIPCHK(c, hipMemcpyAsync(c->pinned + 2, c->counters.p, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev_count, c->stream));
    if (fast) {
        HIPCHK(c, c->diff.ensure(sizeof(int) * gs_tile_ranges_scratch_ints(c->gx, c->gy)));
        StageTimer t(c, GS_STAGE_RANGES);
        HIPCHK(c, gs_launch_tile_ranges(c->rect.as<uint16_t>(), R > 1 ? perm : nullptr, R > 1 ? n0 : c->n, c->diff.as<int>(), c->gx, c->gy,
                                        c->ranges.as<uint32_t>(), nullptr, c->stream));
    }
    HIPCHK(c, hipEventSynchronize(c->ev_count));
    harvest_events(c, fast ? GS_STAGE_RANGES : -1);
    dsort_feedback(c);
    if (c->pinned[0] == 0xFFFFFFFFu)
        return fail(c, GS_ERR_UNSUPPORTED, "gs_bin: more than 2^32 - 2 tile instances (32-bit list offsets); reduce the scene or the image");
    if (c->prev_counters_valid && c->prev_n_inst > 0) {
        unsigned long long w = 0;
        std::memcpy(&w, c->pinned + 2, sizeof(w));
        c->walked_ratio = (double)w / (double)c->prev_n_inst;
    }
    c->prev_counters_valid = false;
    c->n_inst = (int64_t)c->pinned[0];
    c->round_gen[0] = R > 1 ? (int64_t)c->pinned[1] : c->n_inst;
    c->round_ids_off[0] = 0;
    const size_t ni1 = c->n_inst ? (size_t)c->n_inst : 1;
    HIPCHK(c, c->table.ensure(sizeof(uint32_t) * gs_sort_table_entries(c->n_inst > c->n ? c->n_inst : c->n)));
    HIPCHK(c, c->digit_total.ensure(sizeof(uint32_t) * 4 * 256));
    HIPCHK(c, c->ids.ensure(sizeof(uint32_t) * ni1));
    if (fast) {
        // ---- generate-in-pass binning on 32-bit words (gs_bin2.hip)
        const size_t nchunks = ((size_t)c->n_inst + 4095) / 4096;
        HIPCHK(c, c->cs.ensure(sizeof(uint32_t) * (nchunks + 2)));
        if (hi_bits > 0) HIPCHK(c, c->words.ensure(sizeof(uint32_t) * ni1));
        {
            StageTimer t(c, GS_STAGE_TILE_SORT);
            GsBin2Args b{};
            b.n = n0; b.n_inst = c->round_gen[0]; b.gx = c->gx; b.lo_bits = lo_bits; b.hi_bits = hi_bits; b.gid_bits = gid_bits;
            b.offsets = c->offsets.as<uint32_t>(); b.perm = perm; b.rect = c->rect.as<uint16_t>();
            b.cs = c->cs.as<uint32_t>(); b.block_hist = c->table.as<uint32_t>(); b.digit_total = c->digit_total.as<uint32_t>();
            b.buf_a = c->words.as<uint32_t>(); b.ids_out = c->ids.as<uint32_t>(); b.ballot_ranks = c->cfg.rank_mode != 0;
            HIPCHK(c, gs_bin2_build_lists(b, c->stream));
        }
    } else {
        // ---- explicit 64-bit tile|id instances, two radix passes (fallback; identical lists)
        HIPCHK(c, c->inst_a.ensure(sizeof(uint64_t) * ni1));
        HIPCHK(c, c->inst_b.ensure(sizeof(uint64_t) * ni1));
        {
            StageTimer t(c, GS_STAGE_EMIT);
            HIPCHK(c, gs_launch_emit(c->rect.as<uint16_t>(), perm, c->offsets.as<uint32_t>(), c->inst_a.as<uint64_t>(), c->n, c->gx, c->stream));
        }
        uint64_t *sorted = nullptr;
        {
            StageTimer t(c, GS_STAGE_TILE_SORT);
            int in_b = 0;
            HIPCHK(c, gs_radix_sort_u64(c->inst_a.as<uint64_t>(), c->inst_b.as<uint64_t>(), c->n_inst, 32, 32 + tile_bits,
                                        c->table.as<uint32_t>(), c->digit_total.as<uint32_t>(), &in_b, c->stream, c->cfg.rank_mode != 0));
            sorted = in_b ? c->inst_b.as<uint64_t>() : c->inst_a.as<uint64_t>();
        }
        {
            StageTimer t(c, GS_STAGE_RANGES);
            HIPCHK(c, gs_launch_ranges(sorted, c->n_inst, c->ranges.as<uint32_t>(), ntiles, c->stream));
            HIPCHK(c, gs_launch_split_ids(sorted, c->ids.as<uint32_t>(), c->n_inst, c->stream));
        }
    }
    c->did_bin = true; c->did_fwd = c->did_bwd = false;
    return GS_OK;
}
