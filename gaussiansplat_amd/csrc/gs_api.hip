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
#include "../../include/gsplat.h"
#include "gs_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

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

std::string g_create_error;

// RCCL entry points, resolved from librccl.so.1 on first use (the same copy torch loaded, if any)
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err) {
        if (h) return true;
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { err = "librccl: missing symbols"; h = nullptr; return false; }
        return true;
    }
} g_rccl;

}  // namespace

#define GS_COUNTER_BYTES 192      // 128 B of work / ticket counters + the binning totals at byte 128 (one read-back for both)
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
    DevBuf inst_a, inst_b, table, digit_total, ranges, image, trans, g2d, stage_in, dbg_order;
    DevBuf dbg[7];
    DevBuf ids, words, cs, diff;             // sorted gaussian ids; pass-1 words; chunk owners; 2-D difference array
    uint32_t *perm_ptr = nullptr;
    int64_t n_inst = 0;
    uint32_t *pinned = nullptr;
    const float *last_dC = nullptr;          // device dC of the last gs_backward (debug timing)
    int variant_fwd = 0, variant_bwd = 0;

    hipEvent_t ev_count = nullptr;           // instance count landed in pinned memory
    hipEvent_t ev[GS_STAGE_COUNT][2] = {};
    bool ev_valid[GS_STAGE_COUNT] = {};      // a start/stop pair has been recorded
    bool ev_fresh[GS_STAGE_COUNT] = {};      // ... and not yet added to the accumulators
    double ev_sum[GS_STAGE_COUNT] = {};
    int64_t ev_cnt[GS_STAGE_COUNT] = {};
    DevBuf counters;                         // 128 B: 4 x u64 entries walked, evaluated by the forward; walked, evaluated by the backward;
                                             // then 8 u32 per-XCD ticket counters of the forward (byte 32) and 8 of the backward (byte 64)
    DevBuf tile_order_f, tile_order_b, tile_order_p, tile_work, tile_clock;   // experiments: queue orders (+ 9 segment bounds each); per-tile work, debug clocks
    int waves_fwd = 0, waves_bwd = 0;        // experiments: resident waves of the persistent composite grids (occupancy x CUs)
    // ---- longest-first launch orders (gs_config.schedule 3 / 4).  After every forward ONE order kernel turns the frame's per-tile
    // work into a launch order: this frame's backward uses it, and so does the NEXT forward rendered under the same view slot.
    int view_slot = -1;                      // gs_set_view_slot: the slot of the frame being rendered (-1: none)
    // Orders are double buffered: [slot][sel] is the newest one; the order kernel of a frame writes the OTHER buffer on the side
    // stream while this frame's backward still reads the one its forward used.  Index GS_MAX_VIEW_SLOTS = frames without a slot.
    DevBuf slot_order[GS_MAX_VIEW_SLOTS + 1][2];
    int slot_sel[GS_MAX_VIEW_SLOTS + 1] = {};
    int64_t slot_tiles[GS_MAX_VIEW_SLOTS + 1] = {};   // the newest order is valid for this grid (gx << 32 | gy; 0: no history)
    const uint32_t *frame_order = nullptr;   // the launch order of THIS frame's composite kernels (null: tile order)
    bool bwd_counters_zeroed = false;        // the forward kernel zeroed the backward's work counters on its way
    // ---- side stream: the order kernel (needed by the slot's NEXT frame, not by this one) runs beside the backward composite
    hipStream_t side = nullptr;
    hipEvent_t ev_main = nullptr, ev_order = nullptr;
    bool order_pending = false;              // an order kernel is in flight on the side stream (ev_order)
    int lpt_buckets = 0;                     // experiments: work classes of the order kernel (0: default)
    // ---- speculative binning: the lists are enqueued with the capacities of the buffers at hand while the frame's totals travel
    bool pending_totals = false;             // ev_count recorded, pinned totals not read yet
    bool spec_lists = false;                 // the lists of this frame were enqueued before the totals were known ...
    size_t spec_cap_coarse = 0, spec_cap_fine = 0;   // ... against these capacities (entries)
    int64_t n_coarse = 0;
    DevBuf tile_dead;                        // slab frames: 4 lane masks per tile (frozen pixels between rounds)
    int exp_bin_path = -1;                   // experiments: GS_BIN_PATH read once at gs_create
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
    int sgx = 0, sgy = 0;
    int64_t coarse_listed = 0;               // coarse instances of the current round
    DevBuf rect_sorted, l1_table, l1_rows, l1_partials, cids, clr, cranges, segcnt, sdone, tilecnt;
    uint32_t *bin_totals() { return counters.as<uint32_t>() + 32; }
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
    float *bound_image = nullptr, *bound_trans = nullptr;   // gs_bind_outputs: caller-owned device buffers the forward writes directly
    float *img() { return bound_image ? bound_image : image.as<float>(); }
    float *tr() { return bound_trans ? bound_trans : trans.as<float>(); }
    bool counters_zeroed = false;            // the forward's counters were zeroed by a binning kernel of this frame
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

namespace {

int fail(gs_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
int hipfail(gs_ctx *c, hipError_t e, const char *what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return fail(c, e == hipErrorOutOfMemory ? GS_ERR_OOM : GS_ERR_HIP, m);
}
#define HIPCHK(c, call)                                              \
    do {                                                             \
        hipError_t e__ = (call);                                     \
        if (e__ != hipSuccess) return hipfail((c), e__, #call);      \
    } while (0)

// One recorded pair of a stage -> the accumulators; false when the pair has not completed yet (it stays `fresh`).
bool harvest_stage(gs_ctx *c, int s) {
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
void harvest_events(gs_ctx *c, int skip_stage = -1) {
    if (!c->cfg.profile_stages) return;
    for (int s = 0; s < GS_STAGE_COUNT; ++s)
        if (s != skip_stage) (void)harvest_stage(c, s);
}

int bind_device(gs_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    return GS_OK;
}

// A launch order is built only when there are more tiles than wave slots (256 CUs x 4 SIMDs x 5 waves).  Below that the isolated
// kernels do gain from it (C2, 2500 tiles: forward 68 -> 61 us, backward 145 -> 122 us, tools/xcd_order.py C2 -- in tile order the
// heavy tiles of the image centre land on neighbouring SIMDs), but the frame does not: its forward is bound by cold gathers, not by
// balance, and the order kernel is one more launch in a frame that is bound by launches (C2 0.382 -> 0.400 ms, C1 0.183 -> 0.205 ms
// with it, same box; 0.393 / 0.197 with the kernel on the side stream).
bool lpt_schedule(const gs_ctx *c) {
    return (c->cfg.schedule == 3 || c->cfg.schedule == 4) && ((int64_t)c->gx * c->gy > 5120 || (c->cfg.debug_flags & GS_DEBUG_ALWAYS_ORDER));
}
// The side stream (order kernel beside the backward) costs four more runtime calls per frame: it pays when the composite kernels
// are long, and costs when the frame is bound by the host's launch rate (config C2, together with the zero fill it once carried: + 9 %).
bool use_side_stream(const gs_ctx *c) { return c->n >= 262144 || (c->cfg.debug_flags & GS_DEBUG_ALWAYS_ORDER); }

// Launch order of the frame's composite kernels (gs_config.schedule 3 / 4): what the last forward under the same view slot
// measured, else (schedule 4) what this ctx's previous slot-less forward measured; null = no history: tile order for the forward.
int order_index(const gs_ctx *c) { return c->view_slot >= 0 ? c->view_slot : GS_MAX_VIEW_SLOTS; }
const uint32_t *forward_order(gs_ctx *c) {
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    if (!lpt_schedule(c) || ntiles <= 0 || ntiles > GS_LPT_MAX_TILES) return nullptr;
    const int k = order_index(c);
    if (k == GS_MAX_VIEW_SLOTS && c->cfg.schedule != 4) return nullptr;
    if (c->slot_tiles[k] != (((int64_t)c->gx << 32) | (int64_t)c->gy)) return nullptr;          // (the order's length and groups belong to one grid)
    if (c->order_pending) {                                              // (long complete by now; an event wait on the stream costs nothing)
        if (hipStreamWaitEvent(c->stream, c->ev_order, 0) != hipSuccess) return nullptr;
        c->order_pending = false;
    }
    return c->slot_order[k][c->slot_sel[k]].as<uint32_t>();
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
    const int dst = used ? 1 - c->slot_sel[k] : c->slot_sel[k];
    DevBuf &ob = c->slot_order[k][dst];
    HIPCHK(c, ob.ensure(sizeof(uint32_t) * ((size_t)gs_lpt_order_len(c->gx, c->gy) + 16)));
    if (used && use_side_stream(c)) {
        if (c->order_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_order, 0));      // (never two in flight)
        HIPCHK(c, hipEventRecord(c->ev_main, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0));
        HIPCHK(c, gs_launch_tile_lpt_order(c->tile_work.as<uint32_t>(), 0, c->gx, c->gy, ob.as<uint32_t>(), c->side, nullptr, c->lpt_buckets));
        HIPCHK(c, hipEventRecord(c->ev_order, c->side));
        c->order_pending = true;
        c->slot_sel[k] = dst;
        // (tile_work is rewritten by the next forward of this ctx: it waits for ev_order first, see forward_order / gs_forward)
    } else {
        HIPCHK(c, gs_launch_tile_lpt_order(c->tile_work.as<uint32_t>(), 0, c->gx, c->gy, ob.as<uint32_t>(), c->stream, nullptr, c->lpt_buckets));
        c->frame_order = ob.as<uint32_t>();
    }
    c->slot_tiles[k] = ((int64_t)c->gx << 32) | (int64_t)c->gy;
    return GS_OK;
}

#ifdef GS_EXPERIMENTS
// schedule 10 / 12: persistent waves pull tiles from per-XCD ticket counters inside c->counters (zeroed by the caller),
// heaviest first (10: the forward by list length, the backward by the forward's per-tile work) or in tile order (12)
int composite_sched_queue(gs_ctx *c, GsCompositeArgs &a, int which) {
    const int ntiles = c->gx * c->gy;
    if (ntiles <= 0) return GS_OK;
    int &waves = which == 0 ? c->waves_fwd : c->waves_bwd;
    if (waves == 0) {
        waves = gs_composite_resident_waves(which, c->cfg.t_min > 0.0f, c->cfg.deterministic != 0, c->cfg.alpha_cull != 0);
        if (waves <= 0) waves = 256 * 16;
    }
    a.queue = reinterpret_cast<uint32_t *>(static_cast<char *>(c->counters.p) + (which == 0 ? 32 : 64));
    a.grid_waves = waves;
    DevBuf &ord = which == 0 ? c->tile_order_f : c->tile_order_b;            // ntiles tile ids followed by the 9 segment bounds
    HIPCHK(c, ord.ensure(sizeof(uint32_t) * ((size_t)ntiles + 16)));
    const uint32_t *src = c->cfg.schedule == 12 ? nullptr : which == 0 ? c->ranges.as<uint32_t>() : c->tile_work.as<uint32_t>();
    HIPCHK(c, gs_launch_tile_order(src, which == 0 ? 1 : 0, ntiles, ord.as<uint32_t>(), ord.as<uint32_t>() + ntiles, c->stream));
    a.tile_order = ord.as<uint32_t>(); a.order_len = ntiles;
    a.queue_seg = ord.as<uint32_t>() + ntiles;
    return GS_OK;
}
#endif

}  // namespace

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
#ifdef GS_EXPERIMENTS
    if (c0.schedule != 1 && c0.schedule != 3 && c0.schedule != 4 && c0.schedule != 10 && c0.schedule != 12) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad schedule");
#else
    if (c0.schedule != 1 && c0.schedule != 3 && c0.schedule != 4) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad schedule (0, 1, 3 or 4)");
#endif
    if (c0.slab_mode < 0 || c0.slab_mode > 1) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad slab_mode");
    if (c0.bin_path < 0 || c0.bin_path > 2) return fail(nullptr, GS_ERR_INVALID, "gs_create: bad bin_path");
    if (!(c0.slab_max_ratio >= 0.0f && c0.slab_max_ratio <= 1.0f)) return fail(nullptr, GS_ERR_INVALID, "gs_create: slab_max_ratio must be in [0, 1]");
    for (int i = 0; i < 3; ++i)
        if (!(c0.slab_fractions[i] >= 0.0f && c0.slab_fractions[i] < 1.0f)) return fail(nullptr, GS_ERR_INVALID, "gs_create: slab_fractions must be in [0, 1)");
    if (c0.debug_flags & ~(GS_DEBUG_WIDE_CURSORS | GS_DEBUG_ALWAYS_ORDER)) return fail(nullptr, GS_ERR_INVALID, "gs_create: unknown debug_flags");
    if (c0.depth_sort < 0 || c0.depth_sort > 2) return fail(nullptr, GS_ERR_INVALID, "gs_create: depth_sort must be 0, 1 or 2");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, GS_ERR_NO_DEVICE, "gs_create: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, GS_ERR_INVALID, "gs_create: device index out of range");
    gs_ctx *c = new (std::nothrow) gs_ctx();
    if (!c) return fail(nullptr, GS_ERR_OOM, "gs_create: host allocation failed");
    c->device = device;
    c->cfg = c0;
#ifdef GS_EXPERIMENTS
    // experiment switches, read ONCE here (never on the per-frame path): kernel-variant overrides for A/B runs, the binning
    // path, the work classes of the tile order kernel
    if (const char *e = std::getenv("GS_VARIANT_FWD")) c->variant_fwd = std::atoi(e);
    if (const char *e = std::getenv("GS_VARIANT_BWD")) c->variant_bwd = std::atoi(e);
    if (const char *e = std::getenv("GS_BIN_PATH")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) c->exp_bin_path = v; }
    if (const char *e = std::getenv("GS_LPT_BUCKETS")) { const int v = std::atoi(e); if (v >= 1 && v <= 32) c->lpt_buckets = v; }
#endif
    if ((e = hipSetDevice(device)) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipSetDevice"); }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { delete c; return hipfail(nullptr, e, "hipStreamCreate"); }
    c->own_stream = true;
    if ((e = hipHostMalloc((void **)&c->pinned, 512, hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return hipfail(nullptr, e, "hipHostMalloc"); }
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
    if (c->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    DevBuf *bufs[] = {&c->payload, &c->depth_key, &c->rect, &c->pairs_a, &c->pairs_b, &c->perm, &c->offsets, &c->block_sums,
                      &c->inst_a, &c->inst_b, &c->table, &c->digit_total, &c->ranges, &c->image, &c->trans, &c->g2d, &c->stage_in, &c->dbg_order,
                      &c->counters, &c->grads_flat, &c->dpc, &c->ids, &c->words, &c->cs, &c->diff,
                      &c->tile_order_f, &c->tile_order_b, &c->tile_order_p, &c->tile_work, &c->tile_clock,
                      &c->tile_pos, &c->tile_done, &c->live2d, &c->rect_r, &c->offsets_r, &c->live_total,
                      &c->rect_sorted, &c->l1_table, &c->l1_rows, &c->l1_partials, &c->cids, &c->clr, &c->cranges, &c->segcnt, &c->sdone, &c->tilecnt,
                      &c->ranges_r[0], &c->ranges_r[1], &c->ranges_r[2], &c->ranges_r[3],
                      &c->invcov, &c->loss_maps, &c->loss_acc, &c->loss_in[0], &c->loss_in[1], &c->loss_dc, &c->view_cams, &c->tile_dead, &c->key_range, &c->tile_walked, &c->tile_walked_b, &c->tile_work_b};
    for (DevBuf *b : bufs) b->release();
    for (auto &b : c->slot_order) { b[0].release(); b[1].release(); }
    for (auto &b : c->model) b.release();
    for (auto &b : c->dbg) b.release();
    for (int s = 0; s < GS_STAGE_COUNT; ++s)
        for (int k = 0; k < 2; ++k)
            if (c->ev[s][k]) (void)hipEventDestroy(c->ev[s][k]);
    if (c->ev_count) (void)hipEventDestroy(c->ev_count);
    for (hipEvent_t ev : {c->ev_main, c->ev_order}) if (ev) (void)hipEventDestroy(ev);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->pinned) (void)hipHostFree(c->pinned);
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

static int settle_totals(gs_ctx *c, bool *redo, bool may_relist);
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
    if (c->dsort_can_bucket() && c->order() != GS_ORDER_INDEX && c->n > 0) {
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
    b.rect = c->rect.as<uint16_t>(); b.perm = perm_slab; b.sdone = sdone; b.n = n_all; b.n_slab = nr; b.sgx = c->sgx; b.ns = c->sgx * c->sgy;
    b.rect_sorted = c->rect_sorted.as<uint32_t>(); b.table = c->l1_table.as<uint32_t>(); b.row_total = c->l1_rows.as<uint32_t>();
    b.partials = c->l1_partials.as<uint32_t>(); b.totals = c->bin_totals(); b.cranges = c->cranges.as<uint32_t>();
    b.cids = c->cids.as<uint32_t>(); b.clr = c->clr.as<uint16_t>();
    b.tilecnt = c->tilecnt.as<uint32_t>(); b.ntiles = c->gx * c->gy;
    b.zero_words = sdone ? nullptr : c->counters.as<uint32_t>();           // round 0 (no completed tiles yet): the forward's counters
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
        const bool same_grid = c->counters_grid == (((int64_t)c->gx << 32) | (int64_t)c->gy) && c->tile_walked.p;
        if (!same_grid) c->prev_counters_valid = false;
        b.tile_walked = c->prev_counters_valid ? c->tile_walked.as<uint32_t>() : nullptr; b.n_tile_walked = c->gx * c->gy;
    }
    HIPCHK(c, gs_bin3_l1_count(b, c->stream));
    return GS_OK;
}
static int two_level_lists(gs_ctx *c, const uint32_t *perm_slab, int64_t n_all, int64_t nr, size_t coarse, size_t fine, uint32_t *ranges, uint32_t *ids_out,
                           const uint8_t *done, const uint8_t *sdone) {
    const int ns = c->sgx * c->sgy;
    if (coarse == 0) {                                      // nothing listed: every tile range of the round is empty
        HIPCHK(c, hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)c->gx * c->gy, c->stream));
        return GS_OK;
    }
    const int64_t max_work = gs_bin3_max_work((int64_t)coarse, ns);
    HIPCHK(c, c->cids.ensure(sizeof(uint32_t) * coarse));
    HIPCHK(c, c->clr.ensure(sizeof(uint16_t) * coarse));
    HIPCHK(c, c->segcnt.ensure(sizeof(uint32_t) * 64 * (size_t)max_work));
    HIPCHK(c, gs_bin3_l1_scatter(two_level_args(c, perm_slab, n_all, nr, sdone, coarse, fine), c->stream));
    if (!sdone) c->counters_zeroed = true;
    GsBin3Args a{};
    a.cranges = c->cranges.as<uint32_t>(); a.cids = c->cids.as<uint32_t>(); a.clr = c->clr.as<uint16_t>(); a.ranges = ranges; a.tilecnt = c->tilecnt.as<uint32_t>();
    a.done = done; a.segcnt = c->segcnt.as<uint32_t>(); a.ids_out = ids_out;
    a.gx = c->gx; a.gy = c->gy; a.sgx = c->sgx; a.ns = ns; a.max_work = (int)max_work;
    a.wide = (uint64_t)fine * 4ull >= (1ull << 32) || (c->cfg.debug_flags & GS_DEBUG_WIDE_CURSORS) != 0;
    a.totals = c->bin_totals(); a.cap_coarse = (uint32_t)std::min<size_t>(coarse, 0xFFFFFFFEu); a.cap_fine = (uint32_t)std::min<size_t>(fine, 0xFFFFFFFEu);
    HIPCHK(c, gs_bin3_build_lists(a, c->stream));
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
        HIPCHK(c, gs_launch_super_done(c->tile_done.as<uint8_t>(), c->gx, c->gy, c->sgx, c->sgy, c->sdone.as<uint8_t>(), c->stream));
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
static int bin_round(gs_ctx *c, int r) {
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

static int settle_totals(gs_ctx *c, bool *redo, bool may_relist) {
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
    // a list outgrew its buffer: nothing was listed (all ranges empty).  Grow and list again with the real totals.
    HIPCHK(c, c->ids.ensure(sizeof(uint32_t) * (size_t)(c->n_inst ? c->n_inst : 1)));
    {
        StageTimer t(c, GS_STAGE_TILE_SORT);
        if (int rc = two_level_lists(c, c->perm_ptr, c->n, c->slab_lo[1], (size_t)coarse, (size_t)fine_slab, c->ranges.as<uint32_t>(), c->ids.as<uint32_t>(), nullptr, nullptr)) return rc;
    }
    if (redo) *redo = true;
    return GS_OK;
}

int gs_bin(gs_ctx *c, int32_t gx, int32_t gy) {
    if (!c) return GS_ERR_INVALID;
    if (!c->did_pre) return fail(c, GS_ERR_INVALID, "gs_bin: gs_preprocess first");
    if ((gx != 0 || gy != 0) && (gx != c->gx || gy != c->gy))
        return fail(c, GS_ERR_UNSUPPORTED, "gs_bin: blocks must equal ceil(W/16) x ceil(H/16)");
    if (bind_device(c)) return GS_ERR_HIP;
    if (int rc = settle_totals(c, nullptr, false)) return rc;               // a frame that was binned but never rendered
    const size_t n = (size_t)c->n, n1 = n ? n : 1;
    const int64_t ntiles = (int64_t)c->gx * c->gy;
    uint32_t *perm = nullptr;
    if (c->order() != GS_ORDER_INDEX) {
        StageTimer t(c, GS_STAGE_DEPTH_SORT);
        HIPCHK(c, c->pairs_a.ensure(sizeof(uint64_t) * n1));
        HIPCHK(c, c->pairs_b.ensure(sizeof(uint64_t) * n1));
        HIPCHK(c, c->perm.ensure(sizeof(uint32_t) * n1));
        HIPCHK(c, c->table.ensure(sizeof(uint32_t) * gs_sort_table_entries(c->n)));
        HIPCHK(c, c->digit_total.ensure(sizeof(uint32_t) * 4 * 256));
        int in_b = 0;
        perm = c->perm.as<uint32_t>();                  // the last pass writes the permutation itself (low word of the pairs)
        // Two steps (256 key-range buckets, then one workgroup per bucket in LDS: 4 launches) when this frame's preprocess left the
        // key range, the mean bucket is well inside a workgroup's capacity, and no oversize bucket was reported lately; else the
        // classic four LSD passes (12 launches).  Same permutation either way.
        const bool buckets = c->range_valid && c->dsort_can_bucket();
        c->dsort_buckets_used = buckets;
        if (c->range_valid && !buckets)                 // folded but not consumed (cannot happen with one predicate; kept so that a stale union never survives)
            HIPCHK(c, gs_depth_range_reset(c->key_range.as<uint32_t>() + (size_t)c->range_parity * gs_depth_range_parity_words(), c->stream, 1));
        if (buckets) {
            uint32_t *range = c->key_range.as<uint32_t>() + (size_t)c->range_parity * gs_depth_range_parity_words();
            uint32_t *other = c->key_range.as<uint32_t>() + (size_t)(c->range_parity ^ 1) * gs_depth_range_parity_words();
            c->dsort_stat_parity = c->range_parity;
            *c->dsort_stat() = 0u;                      // (no kernel of an earlier frame writes this parity's word any more: two frames back)
            HIPCHK(c, gs_depth_sort_buckets(c->depth_key.as<uint32_t>(), c->pairs_a.as<uint64_t>(), c->pairs_b.as<uint64_t>(), c->n, c->table.as<uint32_t>(),
                                            c->digit_total.as<uint32_t>(), perm, range, other, c->dsort_stat(), c->stream, c->cfg.rank_mode != 0));
        } else {
            // (depth | id) pairs are formed by the first pass from the 32-bit keys; the last pass writes only the ids
            HIPCHK(c, gs_radix_sort_u64(c->pairs_a.as<uint64_t>(), c->pairs_b.as<uint64_t>(), c->n, 32, 64, c->table.as<uint32_t>(),
                                        c->digit_total.as<uint32_t>(), &in_b, c->stream, c->cfg.rank_mode != 0, perm, c->depth_key.as<uint32_t>()));
        }
    }
    c->perm_ptr = perm; c->perm_all = perm;
    int tile_bits = 1;
    while ((1LL << tile_bits) < ntiles) ++tile_bits;
    int gid_bits = 1;
    while ((1LL << gid_bits) < c->n) ++gid_bits;
    const int passes = (tile_bits + 7) / 8;
    const int lo_bits = passes <= 1 ? tile_bits : (tile_bits + 1) / 2, hi_bits = tile_bits - lo_bits;
    const int bin_path = c->exp_bin_path >= 0 ? c->exp_bin_path : c->cfg.bin_path;        // (GS_EXPERIMENTS builds: GS_BIN_PATH, read at gs_create)
    const bool fast = bin_path != 1 && passes <= 2 && hi_bits + gid_bits <= 32 && gs_tile_ranges_supported(c->gx, c->gy);
    // two-level path (gs_bin3.hip): lists per super-tile of 8 x 8 tiles first; its bitmap must fit in LDS
    const int sb = 1 << gs_bin3_sb_shift();
    c->sgx = (c->gx + sb - 1) / sb; c->sgy = (c->gy + sb - 1) / sb;
    c->two_level = fast && bin_path == 0 && gs_bin3_supported(c->sgx * c->sgy);
    c->tile_bits = tile_bits; c->gid_bits = gid_bits; c->lo_bits = lo_bits; c->hi_bits = hi_bits; c->fast_bin = fast;
    HIPCHK(c, c->ranges.ensure(sizeof(uint32_t) * 2 * (size_t)(ntiles ? ntiles : 1)));
    HIPCHK(c, c->counters.ensure(GS_COUNTER_BYTES));
    // the slab plan needs the previous frame's walked share, which the read-back below delivers: the plan of THIS frame uses
    // the share known so far (one frame of lag; only speed depends on it)
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
        if (R == 1 && cap_coarse > 0 && cap_fine > 0) {
            c->spec_lists = true; c->spec_cap_coarse = cap_coarse; c->spec_cap_fine = cap_fine;
            StageTimer t(c, GS_STAGE_TILE_SORT);
            if (int rc = two_level_lists(c, perm, c->n, n0, cap_coarse, cap_fine, c->ranges.as<uint32_t>(), c->ids.as<uint32_t>(), nullptr, nullptr)) return rc;
        } else {                                                            // first frame of a ctx, or a slab frame: the host needs the totals now
            if (int rc = settle_totals(c, nullptr, true)) return rc;
            HIPCHK(c, c->ids.ensure(sizeof(uint32_t) * (size_t)(c->n_inst ? c->n_inst : 1)));
            StageTimer t(c, GS_STAGE_TILE_SORT);
            if (int rc = two_level_lists(c, perm, c->n, n0, (size_t)c->coarse_listed, (size_t)c->round_gen[0], c->ranges.as<uint32_t>(), c->ids.as<uint32_t>(), nullptr, nullptr)) return rc;
        }
        c->did_bin = true; c->did_fwd = c->did_bwd = false;
        return GS_OK;
    }
    // ---- radix paths (bin_path 2 / 1; grids the two-level path refuses): the host reads the instance count before the instance passes
    c->n_coarse = 0;
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
    if (c->prev_counters_valid && c->counters_grid != (((int64_t)c->gx << 32) | (int64_t)c->gy)) c->prev_counters_valid = false;
    if (c->prev_counters_valid) {
        HIPCHK(c, gs_launch_sum_tiles(c->tile_walked.as<uint32_t>(), c->tile_work.as<uint32_t>(), c->gx * c->gy, c->counters.as<unsigned long long>(), c->stream));
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

// the composite launch of round r of the frame (r = 0 unless the frame is binned in depth slabs)
static int enqueue_forward_round(gs_ctx *c, int r, const uint32_t *order) {
    const int R = c->n_rounds;
    GsCompositeArgs a{};
    a.W = c->cam.W; a.H = c->cam.H; a.gx = c->gx; a.gy = c->gy; a.t_min = c->cfg.t_min;
    a.ranges = r == 0 ? c->ranges.as<uint32_t>() : c->ranges_r[r].as<uint32_t>();
    a.ids = c->ids.as<uint32_t>() + c->round_ids_off[r]; a.payload = c->payload.as<GsPayload>();
    a.image = c->img(); a.trans = c->tr();
    a.walked = nullptr;                                                     // per tile: tile_walked / tile_work (GsCompositeArgs.walked)
    a.variant = c->variant_fwd; a.cull = c->cfg.alpha_cull != 0;
    a.resume = r > 0; a.final_round = r == R - 1;
    a.tile_work = c->tile_work.as<uint32_t>(); a.tile_walked = c->tile_walked.as<uint32_t>();
    a.tile_order = order; a.order_len = order ? gs_lpt_order_len(c->gx, c->gy) : 0;
    a.zero_words = c->counters.as<unsigned long long>() + 2;               // the backward's work counters (walked, evaluated)
    if (R > 1) { a.tile_pos = c->tile_pos.as<uint32_t>(); a.tile_done = c->tile_done.as<uint8_t>(); a.tile_dead = c->tile_dead.as<unsigned long long>(); }
#ifdef GS_EXPERIMENTS
    if (c->cfg.schedule == 10 || c->cfg.schedule == 12) {
        if (r > 0) HIPCHK(c, hipMemsetAsync(static_cast<char *>(c->counters.p) + 32, 0, 32, c->stream));      // the forward's ticket counters
        if (int rc = composite_sched_queue(c, a, 0)) return rc;
    }
#endif
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
    HIPCHK(c, c->tile_walked.ensure(sizeof(uint32_t) * (ntiles ? ntiles : 1)));
    c->counters_grid = ((int64_t)c->gx << 32) | (int64_t)c->gy;
    if (!c->counters_zeroed) HIPCHK(c, hipMemsetAsync(c->counters.p, 0, 128, c->stream));   // work counters + both sets of ticket counters
    c->counters_zeroed = false;
    const int R = c->n_rounds;
    if (R > 1) {
        HIPCHK(c, c->tile_pos.ensure(sizeof(uint32_t) * (ntiles ? ntiles : 1)));
        HIPCHK(c, c->tile_done.ensure(ntiles ? ntiles : 1));
        HIPCHK(c, c->tile_dead.ensure(sizeof(unsigned long long) * 4 * (ntiles ? ntiles : 1)));
        HIPCHK(c, hipMemsetAsync(c->tile_pos.p, 0, sizeof(uint32_t) * ntiles, c->stream));
    }
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
                HIPCHK(c, hipMemsetAsync(c->counters.p, 0, 128, c->stream));
                c->counters_zeroed = false;
                if (int rc = enqueue_forward_round(c, 0, order)) return rc;
            }
        }
    }
    if (int rc = build_frame_order(c, order)) return rc;
    c->bwd_counters_zeroed = true;                                         // by the forward kernel (GsCompositeArgs.zero_words)
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
    unsigned long long *bwd_words = c->counters.as<unsigned long long>() + 2;  // (reduced sums: sum_work_counters)
    a.variant = c->variant_bwd; a.cull = c->cfg.alpha_cull != 0;
    if (!params_only) {
        c->last_dC = dC_dev;
        // zero fill of the gradient rows (64 B per gaussian), in line: on a side stream beside the forward composite it cost more than
        // it hid (C3, same box, interleaved: 1.433 ms with the fill beside the forward, 1.412 ms in line -- its workgroups take wave slots
        // from the forward's tiles, and the two event hand-shakes per frame delay the depth sort's launches)
        HIPCHK(c, hipMemsetAsync(c->g2d.p, 0, (det ? sizeof(long long) : sizeof(float)) * GS_G2D_STRIDE * n1, c->stream));
        // launch order: the one the frame's forward used, or (no history) what the order kernel made of that forward
        if (lpt_schedule(c)) { a.tile_order = c->frame_order; a.order_len = c->frame_order ? gs_lpt_order_len(c->gx, c->gy) : 0; }
        if (!c->bwd_counters_zeroed) HIPCHK(c, hipMemsetAsync(bwd_words, 0, 16, c->stream));    // the backward's work counters
        c->bwd_counters_zeroed = false;
#ifdef GS_EXPERIMENTS
        HIPCHK(c, hipMemsetAsync(static_cast<char *>(c->counters.p) + 64, 0, 32, c->stream));   // the backward's ticket counters (schedules 10 / 12)
#endif
#ifdef GS_EXPERIMENTS
        if (c->cfg.schedule == 10 || c->cfg.schedule == 12) { if (int rc = composite_sched_queue(c, a, 1)) return rc; }
#endif
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

int gs_comm_unique_id(void *id128) {
    if (!id128) return GS_ERR_INVALID;
    std::string err;
    if (!g_rccl.load(err)) return fail(nullptr, GS_ERR_UNSUPPORTED, err);
    static_assert(sizeof(ncclUniqueId) == GS_COMM_ID_BYTES, "ncclUniqueId must be 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, GS_ERR_HIP, std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    std::memcpy(id128, &id, sizeof(id));
    return GS_OK;
}

int gs_comm_init(gs_ctx *c, int rank, int nranks, const void *id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return GS_ERR_INVALID;
    std::string err;
    if (!g_rccl.load(err)) return fail(c, GS_ERR_UNSUPPORTED, err);
    if (bind_device(c)) return GS_ERR_HIP;
    if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { c->comm = nullptr; return fail(c, GS_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error")); }
    c->comm_ranks = nranks;
    return GS_OK;
}

int gs_allreduce_grads(gs_ctx *c, const gs_grads *g) {
    if (!c || !g) return GS_ERR_INVALID;
    if (!c->comm) return fail(c, GS_ERR_INVALID, "gs_allreduce_grads: gs_comm_init first");
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)c->n;
    float *p[5] = {g->d_means, g->d_scales, g->d_quats, g->d_opacities, g->d_shs};
    const size_t w[5] = {c->width[0] * n, c->width[1] * n, c->width[2] * n, c->width[3] * n, c->width[4] * n};
    bool flat = p[0] != nullptr;
    for (int i = 0; i + 1 < 5 && flat; ++i) flat = p[i + 1] == p[i] + w[i];
    auto reduce = [&](float *buf, size_t count) -> int {
        if (!buf || !count) return GS_OK;
        const ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclFloat, ncclSum, c->comm, c->stream);
        if (r != ncclSuccess) return fail(c, GS_ERR_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        return GS_OK;
    };
    if (flat) return reduce(p[0], w[0] + w[1] + w[2] + w[3] + w[4]);            // ONE collective (59 N floats at SH3)
    for (int i = 0; i < 5; ++i) if (int rc = reduce(p[i], w[i])) return rc;
    return GS_OK;
}

int gs_comm_destroy(gs_ctx *c) {
    if (!c) return GS_ERR_INVALID;
    if (c->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(c->comm); }
    c->comm = nullptr; c->comm_ranks = 0;
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
    a.variant = variant % 100;
    a.cull = (c->cfg.alpha_cull != 0) != (variant >= 1000);                  // +1000: the other cull setting
    // variant tens digit (gs_composite.hip: apply_sched_variant): 0 the frame's own launch order (what production uses for the
    // backward and for the next forward of the slot), 1 tile order, 3 = 0 explicitly
    a.tile_order = lpt_schedule(c) ? c->frame_order : nullptr;
    a.order_len = a.tile_order ? gs_lpt_order_len(c->gx, c->gy) : 0;
    a.tile_order_band = a.tile_order;
#ifdef GS_EXPERIMENTS
    if (const char *f = std::getenv("GS_DEBUG_ORDER_FILE")) {                // experiments: a launch order made by a script (uint32 x ntiles)
        const size_t ntiles = (size_t)c->gx * c->gy;
        std::vector<uint32_t> h(ntiles);
        FILE *fp = std::fopen(f, "rb");
        if (fp && std::fread(h.data(), sizeof(uint32_t), ntiles, fp) == ntiles) {
            std::vector<uint8_t> seen(ntiles, 0);
            bool ok = true;
            for (uint32_t t : h) { if (t >= ntiles || seen[t]) { ok = false; break; } seen[t] = 1; }
            if (ok) {
                HIPCHK(c, c->dbg_order.ensure(sizeof(uint32_t) * (ntiles + 16)));
                HIPCHK(c, hipMemcpyAsync(c->dbg_order.p, h.data(), sizeof(uint32_t) * ntiles, hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                a.tile_order = c->dbg_order.as<uint32_t>(); a.tile_order_band = a.tile_order; a.order_len = (int)ntiles;
            }
        }
        if (fp) std::fclose(fp);
    }
    a.map_mode = (variant / 100) % 10;
    {   // tens digit 0 / 2 in experiment builds: the persistent queue, longest first / in tile order
        const int keep = c->cfg.schedule;
        c->cfg.schedule = 10;
        const int rc = composite_sched_queue(c, a, which);
        c->cfg.schedule = keep;
        if (rc) return rc;
        const int ntiles = c->gx * c->gy;
        HIPCHK(c, c->tile_order_p.ensure(sizeof(uint32_t) * ((size_t)ntiles + 16)));
        HIPCHK(c, gs_launch_tile_order(nullptr, 0, ntiles, c->tile_order_p.as<uint32_t>(), c->tile_order_p.as<uint32_t>() + ntiles, c->stream));
        a.tile_order_plain = c->tile_order_p.as<uint32_t>();
    }
#endif
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
    // every launch is preceded by the 8-byte reset of its ticket counter, as in a real frame (where it rides on the
    // memset of the work counters); the plain-launch variants pay it too, so the comparison stays fair
    auto launch = [&]() -> hipError_t {
        hipError_t e = hipMemsetAsync(static_cast<char *>(c->counters.p) + 32, 0, 96, c->stream);
        if (e != hipSuccess) return e;
        return which == 0 ? gs_launch_composite_fwd(a, c->stream) : gs_launch_composite_bwd(a, c->stream);
    };
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
    if (int rc = debug_composite_args(c, which, variant, a)) return rc;
    HIPCHK(c, c->tile_clock.ensure(sizeof(uint64_t) * GS_TILE_CLOCK_WORDS * (ntiles ? ntiles : 1)));
    HIPCHK(c, hipMemsetAsync(c->tile_clock.p, 0, sizeof(uint64_t) * GS_TILE_CLOCK_WORDS * ntiles, c->stream));
    a.tile_clock = c->tile_clock.as<unsigned long long>();
    for (int rep = 0; rep < 2; ++rep) {                                       // the second launch (warm) is the one kept
        HIPCHK(c, hipMemsetAsync(static_cast<char *>(c->counters.p) + 32, 0, 96, c->stream));
        HIPCHK(c, which == 0 ? gs_launch_composite_fwd(a, c->stream) : gs_launch_composite_bwd(a, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(out, c->tile_clock.p, sizeof(uint64_t) * GS_TILE_CLOCK_WORDS * ntiles, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GS_OK;
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
    HIPCHK(c, gs_launch_sum_tiles(c->tile_walked.as<uint32_t>(), c->tile_work.as<uint32_t>(), nt, w, c->stream));
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
