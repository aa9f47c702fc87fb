// gs_common.h -- shared device/host declarations of libgsplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS_TILE 16
#define GS_WAVE 64

// Per-gaussian, per-view splat payload staged through LDS by the composite kernels: ONE 64-byte row (four quads) per gaussian,
// so that the random gather of a list entry is a single memory request (48-byte rows were measured: they straddle two 64-byte
// segments for every second gaussian; tools/ab_traffic.sh at C3, same box, interleaved: 791 vs 628 MB fetched per forward launch at
// equal time).  Everything that is constant per (gaussian, view) is already in the form the inner loops use: log2 of the sigmoid,
// the conic scaled by -1/2 log2 e, and (round 3) the pixel-box edges relative to the mean, so the first three quads ARE the
// record a staging lane puts into LDS -- it modifies none of the registers the row was gathered into, which lets the gather of
// the next batch stay in flight across the whole per-entry loop (a modified component forced a copy, and with it a wait for the
// gather, before the loop).  Replaces the 26 scattered floats the reference gathers per (pixel, slot) (src/splat.jl:224-252).
#define GS_PAYLOAD_QUADS 4
#define GS_BIG 1.0e30f
struct __attribute__((aligned(16))) GsPayload {
    float mx, my;        // renderer.positions (mu')                      projection.jl:88-93
    float l2s;           // min(log2(cusigmoid(opacity)), -2.6e-7): alpha = exp2(pw + l2s) < 1 strictly   splat.jl:175-178,247
    float xlo;           // (xmin - mx) - 0.25; +BIG for an empty box: pixel-box test of splat.jl:240 as an exponent penalty
    float ka, kb, kc;    // k i0, k (i1 + i2), k i3 with k = -1/2 log2 e: pw = ka dX^2 + kb dX dY + kc dY^2   splat.jl:246
    float xhi;           // (xmax - mx) + 0.25
    float r, g, b;       // sh2color                                      splat.jl:180-193
    float ylo;           // (ymin - my) - 0.25
    float yhi;           // (ymax - my) + 0.25
    float sig;           // cusigmoid(opacity)                            splat.jl:175-178   (host read-backs only)
    uint32_t bbx;        // int16 xmin | int16 xmax << 16  (renderer.bbs)  boundingbox.jl:24-25
    uint32_t bby;        // int16 ymin | int16 ymax << 16                 boundingbox.jl:26-27
};
static_assert(sizeof(GsPayload) == 16 * GS_PAYLOAD_QUADS, "payload row size");
// box edges of a payload row from its packed pixel box: the SAME single fp32 operations the staging lanes performed in rounds 1-2
__host__ __device__ inline void gs_payload_box_edges(GsPayload &p) {
    const int xmin = (int)(short)(p.bbx & 0xFFFFu), xmax = (int)(short)(p.bbx >> 16);
    const int ymin = (int)(short)(p.bby & 0xFFFFu), ymax = (int)(short)(p.bby >> 16);
    const bool empty = xmax < xmin || ymax < ymin;      // near/far-culled or degenerate splat: lo = hi = +BIG, every pixel is "outside"
    p.xlo = empty ? GS_BIG : ((float)xmin - p.mx) - 0.25f; p.xhi = empty ? GS_BIG : ((float)xmax - p.mx) + 0.25f;
    p.ylo = empty ? GS_BIG : ((float)ymin - p.my) - 0.25f; p.yhi = empty ? GS_BIG : ((float)ymax - p.my) + 0.25f;
}
#define GS_NEG_HALF_LOG2E (-0.72134752044448170368f)
#define GS_L2S_CAP (-2.6e-7f)

struct GsCamera {
    float T[16], P[16];
    float fx, fy, near_, far_;
    float eye[3], lookAt[3];
    int32_t W, H;
};

// Optional scratch arrays of the reference renderer (export_debug).
struct GsDebugArrays {
    float *ts, *tps, *mu, *cov3d, *cov2d, *invcov, *bbs;
};

struct GsPreprocessArgs {
    int64_t n;
    int sh_degree;
    int order;
    int gx, gy;
    const float *means, *scales, *quats, *opac, *shs;
    GsPayload *payload;
    float *invcov;        // 4 x n: renderer.invCov2ds (cov2d.jl:30-45), the raw conic; not read by the composite kernels (their rows carry it scaled)
    uint32_t *depth_key;
    uint16_t *rect;       // 4 x n : x0 x1 y0 y1 (1-based inclusive, x0 == 0 -> no tile)
    GsDebugArrays dbg;
    uint32_t *key_range;  // may be null: 64 minima + 64 maxima of the depth keys, GS_KEY_RANGE_STRIDE words apart (every wave folds its
                          // extremes into slot blockIdx % 64): the key range the two-step depth sort cuts into buckets (gs_sort.hip)
};
#define GS_KEY_RANGE_SLOTS 64
#define GS_KEY_RANGE_STRIDE 64
// launchers (each enqueues on `stream`, returns hipGetLastError())
hipError_t gs_launch_preprocess(const GsPreprocessArgs &a, const GsCamera &cam, hipStream_t stream);

// 2-D image-fitting renderer (gs_preprocess2d.hip): SplatData2D, reference src/splat.jl:20-26
struct GsPreprocess2DArgs {
    int64_t n;
    int W, H, gx, gy;
    const float *means, *scales, *rots, *opac, *colors;    // 2n, 2n, n, n, 3n
    GsPayload *payload;
    float *invcov;                                         // 4 x n raw conic
    uint32_t *depth_key;
    uint16_t *rect;
    GsDebugArrays dbg;                                     // mu, cov2d, invcov, bbs used
};
struct GsPreprocess2DBwdArgs {
    int64_t n;
    int W, H;
    const float *scales, *rots, *opac;
    const float *g2d;
    const long long *g2d_fixed;
    float *d_means, *d_scales, *d_rots, *d_opac, *d_colors;   // accumulate (+=) or overwrite; may be null
    int overwrite;
};
hipError_t gs_launch_preprocess2d(const GsPreprocess2DArgs &a, hipStream_t stream);
hipError_t gs_launch_preprocess2d_bwd(const GsPreprocess2DBwdArgs &a, hipStream_t stream);

struct GsSortScratch {     // sized by gs_sort_scratch_bytes
    uint32_t *block_hist;  // [256][nblocks]
    uint32_t *digit_total; // [256]
};
size_t gs_sort_table_entries(int64_t n_max);
// Stable LSD radix sort of n 64-bit keys on bits [bit_lo, bit_hi); result ends in *out_is_b.  final_low32 != null: the
// last pass writes only the low 32 bits of every key (the id of a (key | id) pair) to final_low32, not the 64-bit keys.
hipError_t gs_radix_sort_u64(uint64_t *a, uint64_t *b, int64_t n, int bit_lo, int bit_hi,
                             uint32_t *block_hist, uint32_t *digit_total, int *result_in_b,
                             hipStream_t stream, bool ballot_ranks = false, uint32_t *final_low32 = nullptr,
                             const uint32_t *keys32 = nullptr);

// Depth sort in two steps (256 key-range buckets, then one workgroup per bucket inside LDS): four launches instead of twelve.
// range_acc: this frame's key-range accumulators (GsPreprocessArgs.key_range), reset_acc: the other parity's (re-armed for the
// next frame); host_stat: coherent pinned word that receives the size of a bucket too large for the fast path (else untouched).
size_t gs_depth_range_words();
size_t gs_depth_range_parity_words();
int64_t gs_depth_buckets_max_n();
hipError_t gs_depth_range_reset(uint32_t *acc, hipStream_t s, int nparity = 2);   // acc: first word of the first parity to re-arm
hipError_t gs_depth_sort_buckets(const uint32_t *keys32, uint64_t *pairs_a, uint64_t *pairs_b, int64_t n, uint32_t *block_hist,
                                 uint32_t *digit_total, uint32_t *perm, const uint32_t *range_acc, uint32_t *reset_acc,
                                 uint32_t *host_stat, hipStream_t stream, bool ballot_ranks);

// checks on the current device that one ds_add_rtn_u32 hands same-address lanes their pre-values in ascending lane order
hipError_t gs_probe_lds_atomic_order(hipStream_t s, int *mismatches);
hipError_t gs_launch_depth_pairs(const uint32_t *depth_key, uint64_t *pairs, int64_t n, hipStream_t s);
hipError_t gs_launch_unpack_perm(const uint64_t *pairs, uint32_t *perm, int64_t n, hipStream_t s);
// counts[s] = tiles of gaussian perm[s] (perm may be null = identity), then exclusive scan
// into offsets[0..n]; offsets[n] = total instances.
hipError_t gs_launch_count_scan(const uint16_t *rect, const uint32_t *perm, uint32_t *offsets,
                                uint32_t *block_sums, int64_t n, hipStream_t s);
// later rounds of a slab frame: 2-D prefix sums of the live (not completed) tiles, and the count/scan pass that drops the
// gaussians whose rectangle holds no live tile (rect_out, indexed by gaussian id, receives the rectangles of the round)
hipError_t gs_launch_live_prefix(const uint8_t *done, int gx, int gy, uint32_t *live2d, uint32_t *rowp_scratch, hipStream_t s);
hipError_t gs_launch_count_scan_live(const uint16_t *rect, const uint32_t *perm, const uint32_t *live2d, int gx, uint16_t *rect_out,
                                     uint32_t *offsets, uint32_t *block_sums, int64_t n, hipStream_t s);
hipError_t gs_launch_emit(const uint16_t *rect, const uint32_t *perm, const uint32_t *offsets,
                          uint64_t *inst, int64_t n, int gx, hipStream_t s);
hipError_t gs_launch_ranges(const uint64_t *inst, int64_t n_inst, uint32_t *ranges, int64_t n_tiles,
                            hipStream_t s);
hipError_t gs_launch_split_ids(const uint64_t *inst, uint32_t *ids, int64_t n_inst, hipStream_t s);
// digit totals + exclusive scan of a [256][nblocks] per-block histogram table (one radix pass)
hipError_t gs_launch_radix_scan(uint32_t *block_hist, int nblocks, uint32_t *digit_total, hipStream_t s);

// low-traffic binning (gs_bin2.hip)
size_t gs_tile_ranges_scratch_ints(int gx, int gy);
bool gs_tile_ranges_supported(int gx, int gy);          // difference array must fit in LDS
// rect is indexed by gaussian id; perm (may be null = identity) maps the n list positions of the slab to gaussian ids;
// done (may be null): tiles with done[t] != 0 get an empty range
hipError_t gs_launch_tile_ranges(const uint16_t *rect, const uint32_t *perm, int64_t n, int *scratch, int gx, int gy, uint32_t *ranges,
                                 const uint8_t *done, hipStream_t s);
struct GsBin2Args {
    int64_t n, n_inst;
    int gx;
    int lo_bits, hi_bits, gid_bits;   // tile id = hi:lo ; hi_bits + gid_bits <= 32
    const uint32_t *offsets, *perm;
    const uint16_t *rect;
    uint32_t *cs;                     // nchunks + 1
    uint32_t *block_hist, *digit_total;
    uint32_t *buf_a;                  // n_inst words (pass-1 output)
    uint32_t *ids_out;                // n_inst gaussian ids in (tile, list order)
    bool ballot_ranks;                // portable ballot ranking instead of the LDS-atomic rank
    const uint8_t *done;              // later rounds of a slab frame: tiles completed earlier take no instances (null: none)
    uint32_t *live_total;             // with `done`: device word receiving the number of instances actually listed; the second
                                      // pass then reads its key count from it (n_inst is only the upper bound used for grids)
};
hipError_t gs_bin2_build_lists(const GsBin2Args &b, hipStream_t s);

// two-level binning (gs_bin3.hip): lists per super-tile of 8 x 8 tiles (level 1), then the tile lists as filtered copies of
// the super-tile lists (level 2)
struct GsBin3L1 {
    const uint16_t *rect;      // fine tile rectangles by gaussian id
    const uint32_t *perm;      // list position -> gaussian id of the round's slab (null: identity)
    const uint8_t *sdone;      // per super-tile: completed in an earlier round (null: none)
    int64_t n, n_slab;         // positions whose instances are summed (totals[2]) / positions listed by this round
    int sgx, ns;
    int sbs;                   // log2 of the super-tile edge in tiles: 3 (8 x 8 tiles) or 4 (16 x 16: grids whose 8 x 8 super-tiles are too many)
    uint32_t *rect_sorted;     // 2 x n_slab words: rectangles in list order
    uint32_t *table;           // gs_bin3_table_words(n_slab, ns)
    uint32_t *row_total;       // ns
    uint32_t *partials;        // gs_bin3_partial_words(n, ns)
    uint32_t *totals;          // 3: coarse instances listed, fine instances of the slab (upper bound of what the round lists), of all n
    uint32_t *cranges;         // 2 x ns
    uint32_t *cids;            // totals[0] gaussian ids in (super-tile, list order)
    uint16_t *clr;             // totals[0] rectangles clipped to the super-tile
    uint32_t *tilecnt;         // ntiles words: zeroed by gs_bin3_l1_count, accumulated and scanned by gs_bin3_build_lists
    int ntiles;
    uint32_t *zero_words;      // gs_bin3_l1_scatter zeroes these 32 words (may be null)
    uint32_t cap_coarse, cap_fine; // entries cids / clr and ids_out can hold (lists are dropped, not overrun, when the totals exceed them)
    // read-back without a copy command: when host_totals is set (coherent pinned host memory) the scan kernel stores the three totals
    // there as well, and the two words at walked_src (device: the previous frame's walked count) to host_walked
    uint32_t *host_totals, *host_walked;
    const uint32_t *walked_src;
    const uint32_t *tile_walked;           // != null: host_walked receives the 64-bit sum of these n_tile_walked per-tile counts instead
    int n_tile_walked;
};
struct GsBin3Args {
    const uint32_t *cranges;   // 2 x ns: [start, end) of every super-tile's list inside cids
    const uint32_t *cids;      // gaussian ids in (super-tile, list order)
    const uint16_t *clr;       // the entries' rectangles clipped to their super-tile (lx0 | lx1 << sbs | ly0 << 2 sbs | ly1 << 3 sbs)
    uint32_t *ranges;          // 2 x tiles: the tile ranges of this round (written here: exclusive scan of the tile counts)
    uint32_t *tilecnt;         // tiles: hits per tile, zero on entry
    const uint8_t *done;       // per tile: completed in an earlier round, takes no entries (null: none)
    uint32_t *segcnt;          // [max_work][4^sbs] hits per (segment, local tile)
    uint32_t *ids_out;         // gaussian ids in (tile, list order)
    int gx, gy, sgx, ns;
    int sbs;                   // log2 of the super-tile edge in tiles (3 or 4)
    int max_work;              // upper bound of the number of (super-tile, segment) work items
    int wide;                  // the lists reach beyond 4 GB from ids_out: 64-bit store addresses
    const uint32_t *totals;    // the round's totals (GsBin3L1.totals) and the capacities they are checked against
    uint32_t cap_coarse, cap_fine;
    // ---- capped lists (gs_config.list_cap; DESIGN.md "lists nobody walks").  cap_src != null: per tile, the list entries the view
    // slot's previous forward walked before its early-out.  A tile's list is then WRITTEN only up to the end of the first segment
    // (L2_SEG entries of its super-tile's list) at which it holds GS_LIST_CAP(walked) entries; the ranges still reserve the full
    // list, and a composite wave that reaches the written end with live pixels writes the next segment's entries itself
    // (gs_composite.hip: extend_tile_list) -- same entries, same order, so image and gradients do not change.
    const uint32_t *cap_src;   // [tiles] or null (no caps: every list written in full)
    uint32_t *tile_nopen;      // [tiles] segments of its super-tile's list the tile takes entries from
    uint32_t *smax;            // [ns]    the largest tile_nopen of the super-tile (later segments have no work at all)
    uint2 *tile_ext;           // [tiles] {written length of the list (entries from its range start), coarse index where the unwritten rest starts or GS_CONT_NONE}
    uint32_t *ext_count;       // one word zeroed by the cap pass: list segments appended by composite waves (GsCompositeArgs.ext_count)
};
#define GS_CONT_NONE 0xFFFFFFFFu
// entries kept for a tile whose slot history walked w: a quarter more, and never less than two batches
__host__ __device__ inline uint32_t gs_list_cap(uint32_t w) { const uint32_t c = w + (w >> 2) + 128u; return c < w ? 0xFFFFFFFFu : c; }
int gs_bin3_seg();             // L2_SEG: coarse entries per level-2 work item
int gs_bin3_sb_shift(int gx, int gy, int force);       // 3 or 4 (force: 3 / 4 = that edge whatever the grid, tests; else by the grid)
bool gs_bin3_supported(int ns);
int64_t gs_bin3_max_work(int64_t coarse_instances, int ns);
size_t gs_bin3_table_words(int64_t n_slab, int ns);
size_t gs_bin3_partial_words(int64_t n, int ns);
hipError_t gs_bin3_l1_count(const GsBin3L1 &b, hipStream_t s);
hipError_t gs_bin3_l1_scatter(const GsBin3L1 &b, hipStream_t s);
hipError_t gs_bin3_build_lists(const GsBin3Args &a, hipStream_t s);
hipError_t gs_bin3_write_lists(const GsBin3Args &a, hipStream_t s);      // the write pass alone (counts and ranges of the frame still valid)
hipError_t gs_launch_super_done(const uint8_t *done, int gx, int gy, int sgx, int sgy, int sbs, uint8_t *sdone, hipStream_t s);

// small frames (gs_bin_small.hip): the whole of gs_bin in one launch, one workgroup per tile
#define GS_BIN_SMALL_MAX_N 16384
#define GS_BIN_SMALL_MAX_TILES 1024
#define GS_BIN_SMALL_MAX_PAIRS (4ll << 20)      // gaussians x tiles: every tile tests every gaussian, and the ids buffer is sized for all of them
struct GsBinSmallArgs {
    const uint32_t *depth_key; // [n] keys of the depth order; null: index order (nothing is sorted)
    const uint2 *rect;         // [n] tile rectangles by gaussian id (x0 | x1 << 16, y0 | y1 << 16; 1-based inclusive; x0 == 0: none)
    int n, gx, gy, ntiles;
    uint32_t *ranges;          // out [2 x ntiles] [start, end) of every tile's list
    uint32_t *ids;             // out the lists (at most n x ntiles entries)
    uint32_t *totals;          // device, 3 words as GsBin3L1.totals: {0, listed, listed}
    uint32_t *host_totals, *host_walked;   // coherent pinned host memory, as GsBin3L1
    const uint32_t *walked_src, *tile_walked;
    int n_tile_walked;
    uint4 *zero; size_t zero_words16;      // the gradient rows of the frame's backward (GsCompositeArgs.g2d): cleared here, which spares the frame a fill launch
};
bool gs_bin_small_supported(int64_t n, int gx, int gy);
hipError_t gs_bin_small(const GsBinSmallArgs &a, hipStream_t s);

#define GS_TILE_CLOCK_WORDS 15
#define GS_MAX_ROUNDS 4   // binning rounds (depth slabs) of one frame
#define GS_G2D_STRIDE 16   // floats (or fixed-point words) per gaussian row of the composite backward's sums: ten used, padded to ONE
                           // 64-byte sector so that the 9-lane atomic of a (tile, splat) entry is a single memory-side request
struct GsCompositeArgs {
    int W, H, gx, gy;
    float t_min;
    const uint32_t *ranges;    // 2 x tiles
    const uint32_t *ids;       // gaussian id per sorted instance
    const GsPayload *payload;
    float *image;              // W*H*3 planar
    float *trans;              // W*H
    // backward only
    const float *dC;           // W*H*3
    float *g2d;                // GS_G2D_STRIDE x n (atomic accumulate): drgb3 | S0 Sx Sy Sxx Sxy - Syy (raw moments, see below)
    long long *g2d_fixed;      // deterministic mode: the same sums as fixed point (integer atomics commute)
    unsigned long long *walked; // debug launches only (gs_debug_time_composite + 10000): [0] list entries walked (staged), [1] entries
                                // evaluated after the no-op cull, ONE ATOMIC EACH PER TILE on two shared words.  Production counts per
                                // tile instead (tile_walked, tile_work; summed on demand): 2 x tiles same-address atomics drain at the
                                // memory side one after the other at the END of the kernel -- + 28 us on the 69 us forward of C2
                                // (2500 tiles), + 15 us at C3 (tools/abtest.py variants 10010 / 10030)
    uint32_t *tile_walked;     // list entries walked (staged) per tile; may be null
    int cull;                  // 1: drop (tile, splat) entries that are provably no-ops while staging (gs_config.alpha_cull)
    int variant;               // debug launches (gs_debug_time_composite, gs_debug_tile_clock): tens digit 1 = tile order instead of tile_order
    const uint32_t *tile_order; // plain launch: block b composites tile_order[b] (0xFFFFFFFF: none); null: block b = tile b
    int order_len;              // entries of tile_order = blocks of the plain launch (0: gx * gy)
    uint32_t *tile_work;       // evaluated entries per tile (forward: the backward's exact work measure and the next launch order); may be null
    unsigned long long *zero_words; // forward: two 64-bit words zeroed by block 0 (the backward's work counters: saves a memset command); may be null
    unsigned long long *tile_clock; // debug: GS_TILE_CLOCK_WORDS per tile {start, end (s_memrealtime, 100 MHz), HW_ID | XCC_ID << 32, walked << 32 | evaluated,
                                    // shader cycles (s_memtime) inside the per-entry loops, shader cycles outside them (staging, waiting for the gathers),
                                    // strip slots executed << 32 | slots if the live pixels were packed 64 to a slot, strips with a live pixel << 32 | live pixels
                                    // (the last four summed over the evaluated entries); forward only: six words = the evaluated entries by the
                                    // slots K = 1 .. 4 they would run on if live pixels were packed {anywhere, by whole rows, inside their column}}
    // frames binned in depth slabs (several rounds of binning + forward)
    int nseg;                  // backward: number of list segments per tile (= rounds of the frame, >= 1)
    const uint32_t *seg_ranges[GS_MAX_ROUNDS];   // backward: per round 2 x tiles [start, end) into seg_ids[round]
    const uint32_t *seg_ids[GS_MAX_ROUNDS];
    uint32_t *tile_pos;        // forward: per tile, list position reached by the earlier rounds (read, then += this round's segment); may be null
    uint8_t *tile_done;        // forward: per tile, 1 = every pixel frozen (written each round; read when `resume`); may be null
    unsigned long long *tile_dead; // forward, slab frames: 4 lane masks per tile, bit l of word p = pixel slot p of lane l is frozen
    int resume;                // forward: continue from the pixel state the previous round left in image / trans
    int final_round;           // forward: last round of the frame: transmittance is written plain (no sign flag)
    // capped lists (GsBin3Args.cap_src): tile_ext != null -> a tile's written list has tile_ext[t].x entries; the forward extends it from
    // the super-tile's coarse list (cranges / cids / clr) when it gets there with live pixels and stores the new end; the backward
    // only reads tile_ext[t].x (it stops where the forward stopped, or at the end the forward left)
    uint2 *tile_ext;
    const uint32_t *cranges, *cids;
    const uint16_t *clr;
    uint32_t *ids_w;           // == ids, writable
    int sgx, sbs;              // super-tile grid width, log2 of the super-tile edge in tiles
    uint32_t *ext_count;       // segments appended by the forward's waves (one atomic per extension: the rare path), may be null
    // small grids: 1, 2 or 4 waves (workgroups) per tile, each owning 4, 2 or 1 of the tile's four 16 x 4 pixel strips and walking the
    // tile's list on its own (gs_config.tile_parts; single-round frames with the early-out and full lists only).  Block b of the launch
    // is part b / len of tile (order[]) b % len, len = the launch length of one part.
    int parts;
    // heavy tiles (tile_lpt_order_kernel, front region): the order's entries may name a part of a tile.  split_ok = 0: this launch composites
    // whole tiles (capped lists, slab rounds, no early-out): a split tile's first part stands for the tile, its other parts do nothing.
    int split_ok;
    // heavy tiles, backward (round 5): a split tile's backward runs as up to GS_SEG_MAX SEGMENTS OF ITS LIST, every segment a whole-tile wave
    // of its own.  The forward's parts leave, at every seg_len[slot] entries, a snapshot (C, T) of their pixels -- NaN in T: frozen by
    // then -- in snap[slot][boundary][4][256], and the walked length in snap_walked[slot]; a backward segment starts from the snapshot in
    // front of it: T as the forward had it, S = (C_final - C_snapshot) . dC.  slot = 8 * (position in the XCD's list) + XCD, positions below
    // front / 24.  seg_len (null: no segments) lives behind the order's entries; front = entries of the order's front region.
    float *snap;
    const uint32_t *seg_len;
    uint32_t *snap_walked;
    int front;
    uint32_t *bw_walked, *bw_work;   // forward: the backward's per-tile counters, zeroed for segmented tiles (its segments ADD to them)
    // ... and on SMALL grids (fewer tiles than wave slots, no launch order) EVERY tile's backward runs as seg_n = 2 / 4 list segments: snapshot
    // slot = tile, (seg_n - 1) snapshots per tile, segment length from the view slot's previous walk of the tile (seg_hist[tile]:
    // gs_seg_len_all), block b of the backward = segment b / len of tile b % len.  seg_hist = null: not this mode.
    const uint32_t *seg_hist;
    int seg_n;
    int clock_by_block;        // debug (tile_clock): records indexed by workgroup instead of by tile (launches with split tiles)
};
// the written entries of capped lists, summed over the tiles: out[0] = sum ext[t].x
hipError_t gs_launch_sum_listed(const uint2 *ext, int n, unsigned long long *out, hipStream_t s);
// longest-first launch order of the tiles for a plain launch (gs_composite.hip: groups of 8 x 8 tiles dealt to the XCDs by work,
// GS_LPT_BUCKETS work classes inside an XCD's list; one workgroup).  order: gs_lpt_order_len(gx, gy) entries, holes = 0xFFFFFFFF.
// zero14 (may be null): fourteen 64-bit words zeroed on the way (the backward's work and ticket counters: saves a memset command)
#define GS_LPT_MAX_TILES 35000
int gs_lpt_order_len(int gx, int gy);
// out[0] = shader cycles (s_memtime), out[1] = 100 MHz ticks (s_memrealtime) of one wave over ~20 us: the chip's clock right now
hipError_t gs_launch_clock_probe(unsigned long long *out, hipStream_t s);
// out[0] = sum of a[0 .. n), out[1] = sum of b[0 .. n) (64 bit): the per-tile work counters of a composite launch, on demand
hipError_t gs_launch_sum_tiles(const uint32_t *a, const uint32_t *b, int n, unsigned long long *out, hipStream_t s);
// front (a multiple of 24; 0: none): entries reserved at the start of `order` for the extra parts of split tiles (tiles with at least
// sum(work) / split_div work: two parts, from twice that: four; the front / 24 heaviest tiles of every XCD are eligible, three entries
// each); the ordinary entries start at order[front]
#define GS_LPT_FRONT 2304
#define GS_SEG_MAX 8                 // list segments of a heavy tile's backward (GS_SEG_MAX - 1 snapshots per tile)
#define GS_SEG_MIN_LEN 1024          // ... none shorter than this many entries
#define GS_SEG_ALL_MIN_LEN 128       // ... on small grids, where every tile is segmented (two batches)
#define GS_SEG_SLOTS (GS_LPT_FRONT / 3)                          // tiles that may be split: 8 XCDs x GS_LPT_FRONT / 24
#define GS_SEG_SNAP_FLOATS ((GS_SEG_MAX - 1) * 4 * 256)          // floats of one tile's snapshots
// walked (with front > 0; may be null): per-tile walked list entries of the forward whose work is ranked -> seg_len[GS_SEG_SLOTS] behind the
// order's entries (order + front + gs_lpt_order_len): list entries per backward segment of a split tile (0: not split / not segmented);
// zero_words (may be null): GS_SEG_SLOTS words zeroed on the way (the walked lengths of the NEXT frame's snapshots)
hipError_t gs_launch_tile_lpt_order(const uint32_t *work_or_ranges, int ranges_mode, int gx, int gy, uint32_t *order, hipStream_t s,
                                    unsigned long long *zero14 = nullptr, int buckets = 0, int front = 0, int split_div = 1,
                                    const uint32_t *walked = nullptr, uint32_t *zero_words = nullptr, uint32_t *host_nsplit = nullptr);
// host_nsplit (may be null): coherent pinned host word that receives the number of split tiles of the order
// small grids: entries per list segment of a tile that walked w entries last time (0: one segment, no snapshots)
#ifndef GS_SEG_ALL_NUM
#define GS_SEG_ALL_NUM 5             // first-segment share of the slot's previous walk with two segments: 5 / 8 (measured at C2, composite backward:
#define GS_SEG_ALL_DEN 8             // 3/8 0.131, 1/2 0.125, 5/8 0.121, 11/16 0.134, 3/4 0.141 ms -- the history is the walk of the tile's FIRST pixel part, a lower bound)
#endif
#ifndef GS_SEG_ALL_SHORT
#define GS_SEG_ALL_SHORT 1           // a walk of up to three batches (C1: 50 ... 150 entries, the evaluated ones at the front) is cut after the FIRST batch; 0: by the rule below, which leaves most of them whole (A/B)
#endif
__host__ __device__ inline uint32_t gs_seg_len_all(uint32_t w, int seg_n) {
    if (GS_SEG_ALL_SHORT && w >= 80u && w <= (seg_n == 2 ? 192u : 64u * (uint32_t)seg_n + 64u)) return 64u;   // (at least sixteen entries behind the first cut)
    const uint32_t per = seg_n == 2 ? (w * GS_SEG_ALL_NUM + GS_SEG_ALL_DEN - 1u) / GS_SEG_ALL_DEN : (w + (uint32_t)seg_n - 1u) / (uint32_t)seg_n, sl = ((per + 63u) & ~63u) < (uint32_t)GS_SEG_ALL_MIN_LEN ? (uint32_t)GS_SEG_ALL_MIN_LEN : ((per + 63u) & ~63u);
    return w > sl ? sl : 0u;
}
int gs_seg_units(int front);
int gs_composite_grid_blocks(const GsCompositeArgs &a, int bwd);      // workgroups of the launch these arguments describe
hipError_t gs_launch_composite_fwd(const GsCompositeArgs &a, hipStream_t s);
hipError_t gs_launch_composite_bwd(const GsCompositeArgs &a, hipStream_t s);

struct GsPreprocessBwdArgs {
    int64_t n;
    int sh_degree;
    const float *means, *scales, *quats, *opac, *shs;
    const float *g2d;
    const long long *g2d_fixed;   // non-null: read the 2-D gradients from the fixed-point buffer
    float *d_means, *d_scales, *d_quats, *d_opac, *d_shs;   // accumulate (+=) or overwrite; may be null
    float *dpc;           // scratch 4 x n: d L / d tps[1:3] through the colour
    int overwrite;        // 1: store instead of accumulate
    float sgd_scale;      // != 0 (accumulate mode only): target = fma(sgd_scale, gradient, target) -- with the parameter arrays as
                          // targets and sgd_scale = -lr this IS the SGD step, fused (gs_backward_sgd)
};
// phases: bit 0 the SH / colour kernel (d_shs, dpc), bit 1 the geometry chain (reads dpc); 3 = both, in that order
hipError_t gs_launch_preprocess_bwd(const GsPreprocessBwdArgs &a, const GsCamera &cam, hipStream_t s, int phases = 3);

// colour-factored gradient exchange (gs_preprocess_bwd.hip)
hipError_t gs_launch_pack_drgb(const float *g2d, const long long *g2d_fixed, float *out, int64_t n, hipStream_t s);
hipError_t gs_launch_sh_from_views(int64_t n, int sh_degree, const float *means, int nviews, const float *cams, const float *drgb,
                                   float *d_shs, int overwrite, hipStream_t s);

// The composite backward accumulates, per gaussian, the colour gradient and the RAW moments of dd = dL/d(log alpha)
// about the splat's 2-D mean: row = [dr dg db | S0 Sx Sy Sxx Sxy - Syy].  With the view's sig and conic M (column
// major i0 i1 i2 i3, mc = (i1 + i2)/2):  dL/dsig = -S0/sig,  dL/dmu = -(i0 Sx + mc Sy, mc Sx + i3 Sy),
// dL/dM = 1/2 [Sxx Sxy; Sxy Syy].  In place: row becomes [dr dg db dsig dmx dmy d00 d01 d10 d11].
template <typename R>
__host__ __device__ inline void gs_g2d_to_grads(R (&g2)[10], R sig, R i0, R mc, R i3) {   // g2: the ten used words of a row
#pragma clang fp contract(off)   // (called from two kernels that must produce the same bits: gs_geom_bwd_body.inc)
    const R S0 = g2[3], Sx = g2[4], Sy = g2[5], Sxx = g2[6], Sxy = g2[7], Syy = g2[9];
    g2[3] = sig > R(0) ? -S0 / sig : R(0);
    g2[4] = -(i0 * Sx + mc * Sy);
    g2[5] = -(mc * Sx + i3 * Sy);
    g2[6] = R(0.5) * Sxx; g2[7] = R(0.5) * Sxy; g2[8] = R(0.5) * Sxy; g2[9] = R(0.5) * Syy;
}

// deterministic mode: fixed-point scale of g2d component c.  2^-40 for the colour gradient and the moments of order
// 0 and 1 (range +-8.4e6), 2^-28 for the second-order moments Sxx, Sxy, Syy (components 6, 7, 9): they carry dX^2 in
// pixels^2 and would saturate at 8.4e6 for very large footprints; range +-3.4e10, resolution 3.7e-9.
#define GS_FIXED_SCALE 1099511627776.0f            // 2^40
#define GS_FIXED_INV (1.0 / 1099511627776.0)
#define GS_FIXED_SCALE2 268435456.0f               // 2^28
#define GS_FIXED_INV2 (1.0 / 268435456.0)
__host__ __device__ inline double gs_fixed_inv(int comp) { return comp >= 6 ? GS_FIXED_INV2 : GS_FIXED_INV; }

// loss + SGD (gs_loss.hip)
// loss accumulators: GS_LOSS_SLOTS pairs {sum |x - y|, sum ssim}, GS_LOSS_SLOT_STRIDE doubles apart (summed by the host)
#define GS_LOSS_SLOTS 64
#define GS_LOSS_SLOT_STRIDE 8
hipError_t gs_loss_run(int W, int H, int C, const float *img, const float *gt, float *maps, double *acc, float *dC, float lam,
                       const float *win121, hipStream_t s);
hipError_t gs_launch_sgd(float *p, const float *g, float lr, size_t n, hipStream_t s);
