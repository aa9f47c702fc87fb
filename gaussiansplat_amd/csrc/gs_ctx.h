// gs_ctx.h -- the renderer context behind the C ABI (include/gsplat.h) and the host-side helpers its translation units share.
// Internal: nothing here is part of the ABI.  The entry points live in
//   gs_api.hip            create / destroy, model, camera, gs_preprocess, gradients buffers, loss, SGD
//   gs_api_bin.hip        gs_bin: depth order, two-level tile lists (speculative launch, capped lists, depth slabs), radix paths
//   gs_api_composite.hip  gs_forward / gs_backward and the launch orders of the view slots
//   gs_api_comm.hip       RCCL below the boundary (gs_comm_*, gs_allreduce_grads)
//   gs_api_debug.hip      introspection and profiling hooks (gs_get_array, stage timers, tile clocks, counters)
#pragma once
#include "../../include/gsplat.h"
#include "gs_common.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;                 // grow-only with slack
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

extern std::string g_create_error;         // message of the last failed gs_create (gs_last_error(NULL))

#define GS_COUNTER_BYTES 192      // 32 B of work-counter sums (on demand) | byte 128: the binning totals | 144: list segments appended by waves | 160: debug scratch
struct gs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false, borrowed_stream = false;
    gs_config cfg{};
    std::string err;

    int64_t n = 0;
    int sh_degree = 0;
    int kind = 0;                            // 0: 3-D renderer (SplatData3D), 1: 2-D image-fitting renderer (SplatData2D)
    size_t width[5] = {3, 3, 4, 1, 3};       // floats per gaussian of the five parameter / gradient arrays
    int order() const { return kind == 1 ? (int)GS_ORDER_INDEX : cfg.order; }    // the 2-D model has no depth
    const float *means = nullptr, *scales = nullptr, *quats = nullptr, *opac = nullptr, *shs = nullptr;
    DevBuf model[5];
    GsCamera cam{};
    bool have_cam = false, did_pre = false, did_bin = false, did_fwd = false, did_bwd = false, did_bwd_composite = false;
    int gx = 0, gy = 0;

    DevBuf invcov;                           // 4 x n raw conic (introspection; the payload rows carry it scaled)
    DevBuf payload, depth_key, rect, pairs_a, pairs_b, perm, offsets, block_sums;
    DevBuf inst_a, inst_b, table, digit_total, ranges, image, trans, g2d, stage_in;
    DevBuf dbg[7];
    DevBuf ids, words, cs, diff;             // sorted gaussian ids; pass-1 words; chunk owners; 2-D difference array
    uint32_t *perm_ptr = nullptr;
    int64_t n_inst = 0;
    uint32_t *pinned = nullptr;
    const float *last_dC = nullptr;          // device dC of the last gs_backward (debug timing)

    hipEvent_t ev_count = nullptr;           // instance count landed in pinned memory
    hipEvent_t ev[GS_STAGE_COUNT][2] = {};
    bool ev_valid[GS_STAGE_COUNT] = {};      // a start/stop pair has been recorded
    bool ev_fresh[GS_STAGE_COUNT] = {};      // ... and not yet added to the accumulators
    double ev_sum[GS_STAGE_COUNT] = {};
    int64_t ev_cnt[GS_STAGE_COUNT] = {};
    DevBuf counters;                         // GS_COUNTER_BYTES: 4 x u64 {walked, evaluated} of the forward and of the backward (sum_work_counters), ...
    DevBuf tile_work, tile_clock;            // per-tile evaluated entries of the forward (the launch orders' input); debug clocks
    // ---- longest-first launch orders (gs_config.schedule 3 / 4).  After every forward ONE order kernel turns the frame's per-tile
    // work into a launch order: this frame's backward uses it, and so does the NEXT forward rendered under the same view slot.
    int view_slot = -1;                      // gs_set_view_slot: the slot of the frame being rendered (-1: none)
    // Orders are double buffered: [slot][sel] is the newest one; the order kernel of a frame writes the OTHER buffer on the side
    // stream while this frame's backward still reads the one its forward used.  Index GS_MAX_VIEW_SLOTS = frames without a slot.
    struct ViewSlot {
        DevBuf order[2];                     // launch orders, double buffered
        int sel = 0;                         // the newest one
        int64_t tiles = 0;                   // ... is valid for this grid (gx << 32 | gy; 0: no history)
        DevBuf walkbuf[2];                   // per tile: list entries the slot's forwards walked, double buffered: [wsel] = the last completed forward's
        int wsel = 0;                        // (the next frame's list caps and segment lengths), the other one is what the frame being rendered writes
        int64_t walked_grid = 0;             // ... on this grid (0: none yet)
        DevBuf &walked() { return walkbuf[wsel]; }
    };
    std::vector<ViewSlot> slots;             // GS_MAX_VIEW_SLOTS + 1 (allocated at gs_create; device buffers on first use)
    const uint32_t *last_walked = nullptr;   // per-tile walked counts of the most recent forward (the slot's array, or tile_walked)
    // ---- capped lists (gs_config.list_cap): this frame's tile lists were written only as far as the slot's history says they are walked
    bool frame_capped = false;
    int frame_parts = 1;                   // waves per tile of the frame's composite launches (gs_config.tile_parts; decided by gs_forward)
    int wave_slots = 5120;                 // waves of the composite kernels the device holds at once: CUs x 4 SIMDs x 5 (GS_FWD_MINW, GS_BWD_MINW); gs_create
    const uint32_t *cap_src = nullptr;       // the history the caps of this frame come from (null: none)
    DevBuf tile_nopen, smax, tile_ext, zero_tiles;
    GsBin3Args last_l2{};                    // the level-2 arguments of the frame's lists (gs_get_array writes the capped rest with them)
    bool have_l2 = false;
    const uint32_t *frame_order = nullptr;   // the launch order of THIS frame's composite kernels (null: tile order)
    // ---- heavy tiles: list segments of the backward (GsCompositeArgs.snap)
    DevBuf snap, snap_walked;                // the forward's snapshots of the split tiles; their walked lengths, two frame parities
    const uint32_t *snap_order = nullptr;    // the order whose split tiles this frame's forward left snapshots for (null: none)
    int frame_seg_n = 0;                     // small grids: list segments per tile of this frame's backward (0: none; the forward left snapshots for them)
    const uint32_t *seg_hist = nullptr;      // ... and the walk history their lengths come from (the slot's previous forward)
    uint32_t *pinned_split = nullptr;        // coherent pinned host words, two per view slot (one per order buffer): split tiles of that order, as its
                                             // order kernel counted them (0xFFFFFFFF: the kernel has not reported yet); null: unknown, assume some
    const uint32_t *frame_order_split = nullptr;   // ... the word of this frame's order
    int snap_parity = 0;                     // the parity of snap_walked this frame's forward writes (and its backward reads); the order kernel
                                             // behind every forward re-arms the other one and the parities swap
    // ---- side stream: the order kernel (needed by the slot's NEXT frame, not by this one) runs beside the backward composite
    hipStream_t side = nullptr;
    hipEvent_t ev_main = nullptr, ev_order = nullptr;
    bool order_pending = false;              // an order kernel is in flight on the side stream (ev_order)
    // ---- speculative binning: the lists are enqueued with the capacities of the buffers at hand while the frame's totals travel
    bool pending_totals = false;             // ev_count recorded, pinned totals not read yet
    bool spec_lists = false;                 // the lists of this frame were enqueued before the totals were known ...
    size_t spec_cap_coarse = 0, spec_cap_fine = 0;   // ... against these capacities (entries)
    int64_t n_coarse = 0;
    DevBuf tile_dead;                        // slab frames: 4 lane masks per tile (frozen pixels between rounds)
    int dbg_win_start = 0, dbg_win_len = 0;  // gs_debug_set_window: the part of the launch order the debug launches cover (len 0: all)
    int rank_probe = -1;                     // lane-order probe of the LDS atomic rank: -1 not run, 0 passed, 1 failed (ballots forced)
    // ---- binning in depth slabs (gs_config.slab_mode)
    int n_rounds = 1;                        // binning rounds of the current frame
    int64_t slab_lo[GS_MAX_ROUNDS + 1] = {}; // round r covers the list positions [slab_lo[r], slab_lo[r+1]) of the depth order
    int64_t round_gen[GS_MAX_ROUNDS] = {};   // generated instance positions of the round (>= the instances it lists)
    size_t round_ids_off[GS_MAX_ROUNDS] = {};// where the round's ids start inside `ids`
    DevBuf ranges_r[GS_MAX_ROUNDS];          // tile ranges of rounds 1.. (round 0 uses `ranges`)
    DevBuf tile_pos, tile_done, live2d, rect_r, offsets_r, live_total;
    uint32_t *perm_all = nullptr;            // the whole depth order (perm_ptr)
    // ---- two-level binning (gs_bin3.hip): lists per super-tile of 8 x 8 tiles, then per tile
    bool two_level = false;
    int sgx = 0, sgy = 0, sbs = 3;           // super-tile grid and log2 of the super-tile edge in tiles (3, or 4 on 4K-class grids)
    int64_t coarse_listed = 0;               // coarse instances of the current round
    DevBuf rect_sorted, l1_table, l1_rows, l1_partials, cids, clr, cranges, segcnt, sdone, tilecnt;
    uint32_t *bin_totals() { return counters.as<uint32_t>() + 32; }
    uint32_t *ext_count() { return counters.as<uint32_t>() + 36; }          // byte 144: list segments appended by composite waves (capped lists)
    // ---- per-tile work counters of the composite launches (walked / evaluated list entries): counters[0..3] hold their sums only
    // after sum_work_counters() (gs_get_work_counters, the radix binning paths); the two-level path sums the walked counts of the
    // previous forward inside l1_rowscan on their way to the host
    DevBuf tile_walked, tile_walked_b, tile_work_b;
    int64_t counters_grid = 0;               // the grid (gx << 32 | gy) the forward's per-tile counters were written for
    // ---- depth sort in two steps (gs_depth_sort_buckets; gs_config.depth_sort)
    DevBuf key_range;                        // two frame parities of the key-range accumulators the preprocess kernel fills
    int range_parity = 0;                    // parity of the frame being built
    bool range_valid = false;                // the 3-D preprocess of this frame filled key_range[range_parity]
    bool dsort_buckets_used = false;         // this frame's depth order came from the bucket path (its pinned stat word is live)
    int64_t dsort_classic_until = 0;         // frame id up to which the classic sort is used (an oversize bucket was reported)
    int dsort_stat_parity = 0;               // the parity gs_bin used for the bucket path's pinned stat word (gs_preprocess of the NEXT frame flips range_parity
                                             // before settle_totals of this one may run)
    uint32_t *dsort_stat() { return pinned + 100 + (dsort_stat_parity & 1); }
    // the bucket path is possible for the frame being built (same predicate in gs_preprocess, which then folds the key range, and in gs_bin)
    bool dsort_can_bucket() const {
        return cfg.depth_sort != 1 && (cfg.depth_sort == 2 || (n <= gs_depth_buckets_max_n() && frame_id > dsort_classic_until));
    }
    // ---- small frames (gs_bin_small.hip): the whole of gs_bin in one launch.  Same predicate in gs_preprocess (which then folds no key
    // range: nothing is sorted globally) and in gs_bin.  bin_path 0 only (3 = the two-level path whatever the size; tests, A/B)
    bool small_bin = false;                  // the frame being built was binned by the small path
    bool perm_pending = false;               // ... and its depth order (renderer.sortIdxs) has not been asked for yet
    bool g2d_clean = false;                  // ... and its kernel cleared the gradient rows: the frame's first composite backward needs no fill
    bool small_bin_possible() const {
        if (cfg.bin_path != 0 || cfg.depth_sort != 0 || cfg.list_cap == 2 || cfg.slab_fractions[0] > 0.0f) return false;
        if (cfg.debug_flags & (GS_DEBUG_WIDE_CURSORS | GS_DEBUG_SUPER8 | GS_DEBUG_SUPER16 | GS_DEBUG_TINY_CAPS)) return false;
        return gs_bin_small_supported(n, gx, gy);
    }
    float *bound_image = nullptr, *bound_trans = nullptr;   // gs_bind_outputs: caller-owned device buffers the forward writes directly
    float *img() { return bound_image ? bound_image : image.as<float>(); }
    float *tr() { return bound_trans ? bound_trans : trans.as<float>(); }
    int tile_bits = 0, gid_bits = 0, lo_bits = 0, hi_bits = 0;
    bool fast_bin = false;
    double walked_ratio = -1.0;              // entries walked / instances of the last completed frame (-1: none yet)
    int64_t prev_n_inst = 0;
    bool prev_counters_valid = false;        // `counters` holds the walked count of a completed forward
    int64_t frame_id = 0, ev_frame[GS_STAGE_COUNT] = {};   // a stage may run once per binning round: ev_cnt counts frames, not launches
    int64_t ev_counted[GS_STAGE_COUNT] = {};               // last frame whose pair of this stage was added to ev_cnt
    DevBuf grads_flat;                       // gs_grads_alloc
    DevBuf dpc;                              // 4 x n scratch between the two backward kernels
    DevBuf loss_maps, loss_acc, loss_in[2], loss_dc, view_cams;
    ncclComm_t comm = nullptr;
    int comm_ranks = 0;
};

// ---------------------------------------------------------------- shared helpers
inline int fail(gs_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
inline int hipfail(gs_ctx *c, hipError_t e, const char *what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return fail(c, e == hipErrorOutOfMemory ? GS_ERR_OOM : GS_ERR_HIP, m);
}
#define HIPCHK(c, call)                                              \
    do {                                                             \
        hipError_t e__ = (call);                                     \
        if (e__ != hipSuccess) return hipfail((c), e__, #call);      \
    } while (0)

// One recorded pair of a stage -> the accumulators; false when the pair has not completed yet (it stays `fresh`).
inline bool harvest_stage(gs_ctx *c, int s) {
    if (!c->ev_fresh[s]) return true;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[s][0], c->ev[s][1]) != hipSuccess) { (void)hipGetLastError(); return false; }
    c->ev_sum[s] += ms;
    if (c->ev_counted[s] != c->ev_frame[s]) { c->ev_counted[s] = c->ev_frame[s]; c->ev_cnt[s] += 1; }      // one count per frame
    c->ev_fresh[s] = false;
    return true;
}
struct StageTimer {
    gs_ctx *c; int st; bool on;
    StageTimer(gs_ctx *c_, int st_) : c(c_), st(st_), on(c_->cfg.profile_stages == 1 || c_->cfg.profile_stages == 2 + st_) {
        if (on) {
            if (c->ev_fresh[st] && !harvest_stage(c, st)) {            // about to re-record a pair nobody has read: wait for it (rare:
                (void)hipEventSynchronize(c->ev[st][1]);               // the host ran a whole frame ahead of the GPU)
                (void)harvest_stage(c, st);
            }
            c->ev_frame[st] = c->frame_id;
            (void)hipEventRecord(c->ev[st][0], c->stream);
        }
    }
    ~StageTimer() {
        if (on) { (void)hipEventRecord(c->ev[st][1], c->stream); c->ev_valid[st] = true; c->ev_fresh[st] = true; }
    }
};

// Fold every pair that has completed into the accumulators (pairs still in flight stay fresh for the next call).
inline void harvest_events(gs_ctx *c, int skip_stage = -1) {
    if (!c->cfg.profile_stages) return;
    for (int s = 0; s < GS_STAGE_COUNT; ++s)
        if (s != skip_stage) (void)harvest_stage(c, s);
}

inline int bind_device(gs_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    return GS_OK;
}

// A launch order is built only when there are more tiles than wave slots (gs_ctx::wave_slots: 256 CUs x 4 SIMDs x 5 waves on MI355X).  Below that the isolated
// kernels do gain from it (C2, 2500 tiles: forward 68 -> 61 us, backward 145 -> 122 us, tools/xcd_order.py C2 -- in tile order the
// heavy tiles of the image centre land on neighbouring SIMDs), but the frame does not: its forward is bound by cold gathers, not by
// balance, and the order kernel is one more launch in a frame that is bound by launches (C2 0.382 -> 0.400 ms, C1 0.183 -> 0.205 ms
// with it, same box; 0.393 / 0.197 with the kernel on the side stream).
inline bool lpt_schedule(const gs_ctx *c) {
    return (c->cfg.schedule == 3 || c->cfg.schedule == 4) && ((int64_t)c->gx * c->gy > c->wave_slots || (c->cfg.debug_flags & GS_DEBUG_ALWAYS_ORDER));
}
// Heavy tiles (round 5): the launch orders reserve a front region for the extra parts of split tiles (tile_lpt_order_kernel) when the
// ctx may use them at all: automatic tile_parts, the early-out on, and a grid on which the frame-wide 2 / 4 waves per tile cannot engage.
// A property of the ctx and the grid, so every order of a slot has one layout; whether a LAUNCH honours the split entries is decided
// per frame (split_ok: full lists, one binning round).
inline int lpt_front(const gs_ctx *c) {
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    return (lpt_schedule(c) && c->cfg.tile_parts == 0 && c->cfg.t_min > 0.0f && 2 * ntiles > c->wave_slots && ntiles <= GS_LPT_MAX_TILES) ? GS_LPT_FRONT : 0;
}
// the words behind an order's entries: list entries per backward segment of the split tiles (GS_SEG_SLOTS of them)
inline const uint32_t *order_seg_len(const gs_ctx *c, const uint32_t *order) { return order + lpt_front(c) + gs_lpt_order_len(c->gx, c->gy); }
inline int lpt_order_entries(const gs_ctx *c) { return lpt_front(c) + gs_lpt_order_len(c->gx, c->gy); }   // = workgroups of a launch over the order
inline int lpt_split_div(const gs_ctx *c) { return c->wave_slots * 4 / 5; }      // a tile with more work than an even share of ~4 waves per SIMD is split
// The side stream (order kernel beside the backward) costs four more runtime calls per frame: it pays when the composite kernels
// are long, and costs when the frame is bound by the host's launch rate (config C2, together with the zero fill it once carried: + 9 %).
inline bool use_side_stream(const gs_ctx *c) { return c->n >= 262144 || (c->cfg.debug_flags & GS_DEBUG_ALWAYS_ORDER); }
inline int order_index(const gs_ctx *c) { return c->view_slot >= 0 ? c->view_slot : GS_MAX_VIEW_SLOTS; }   // index into gs_ctx::slots of the frame being rendered

// ---------------------------------------------------------------- across the translation units
// gs_api_bin.hip
int settle_totals(gs_ctx *c, bool *redo, bool may_relist);
int bin_round(gs_ctx *c, int r);
int depth_order(gs_ctx *c, uint32_t **perm_out);
// gs_api_comm.hip
void comm_release(gs_ctx *c);                // destroys the ctx's RCCL communicator, if any
// gs_api_composite.hip
const uint32_t *forward_order(gs_ctx *c);
int build_frame_order(gs_ctx *c, const uint32_t *used);
