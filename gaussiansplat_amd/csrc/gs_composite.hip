// gs_composite.hip -- per-tile alpha composite (forward) and its adjoint for gfx950.
//
// Forward replaces splatDraw (reference src/splat.jl:195-269): every pixel of a 16x16 tile
// walks the tile's splat list front to back, alpha = sigmoid(o) * exp(-1/2 d' Sigma^-1 d),
// C += rgb*alpha*T, T *= 1-alpha, for list entries whose pixel box contains the pixel
// (splat.jl:240); no alpha clamp / 1/255 cut (reference has none).  The reference gathers
// 26 floats per (pixel, slot) from global memory and evaluates SH per pixel; here the per-view
// payload (three quads of a 48- or 64-byte row) is gathered once per (tile, splat) with coalesced id loads, staged in LDS
// and broadcast to the lanes.
//
// Mapping (wave64-first, not a 16x16 CUDA block): ONE wave per tile, lane l owns the four
// pixels (x = l & 15, y = (l >> 4) + 4p).  The tile-wide transmittance vote is a single
// 64-bit ballot and there is no workgroup barrier on the critical path.  Backward replaces
// splatGrads (splat.jl:271-396), which is not a valid adjoint of the 3-D forward:
// it is the derived adjoint, walking the list in the SAME order with the suffix colour obtained
// as D - prefix (D = C_final . dC), so T is recomputed exactly as in the forward and never divided
// back.
//
// Scheduling.  Tiles differ a lot in work (the early-out point, the share of no-op entries, ragged image edges) and a wave
// lives as long as its tile, so the launch ORDER decides how ragged the end of the kernel is: a plain launch over a
// longest-first permutation of the tiles (tile_lpt_order_kernel below; groups of 8 x 8 tiles dealt to the XCDs so that the
// tiles listing a gaussian share one L2).  Persistent waves on ticket queues were measured slower (profiles/HISTORY.md).
//
// Per-splat gradient sums (backward).  Nine sums over the tile's 256 pixels are needed per (tile, splat).  Each lane
// first adds its four pixels, then the 64 x 9 partials are transposed through LDS: nine conflict-free ds_write_b32 rows
// [component][lane], then lane (c, s) = (l >> 2, l & 3), l < 36, reads the sixteen partials of quarter s of component
// c with four ds_read_b128 and adds them; two quad-permute DPP adds finish, and nine lanes issue ONE atomic
// instruction covering the gaussian's row: 17 adds per entry on the VALU -- the unit these kernels are bound by.
//
// While staging a 64-entry batch each lane also bounds the largest alpha its entry can reach on this
// tile (rect_can_contribute); entries that are no-ops in fp32 are dropped and the batch is compacted in
// LDS, so the per-pixel loops only see entries that matter (gs_config.alpha_cull; 56 % of the walked
// entries at C3).  The lists themselves and the batch boundaries of the early-out rule are untouched.
//
// Both kernels are VALU bound, not HBM bound; see DESIGN.md section 5 for the roofline accounting and the measured
// instruction costs that shaped the inner loops.
#include "gs_common.h"

// Only the fmaf() calls written below fuse: with the compiler free to contract, T - alpha*T came out as one fma in one
// template instantiation and as mul + sub in another, so "cull on" and "cull off" differed in the last bit of T.
#pragma clang fp contract(off)

#ifndef GS_BWD_MINW
#define GS_BWD_MINW 6               // __launch_bounds__ waves/SIMD of the backward: 80 VGPRs, which every production instantiation meets without a spill since
                                    // the live rectangle comes from the live masks on the scalar unit (LiveRectAcc: 91 -> 79 VGPRs; round 2: 96 at five waves)
#endif
#ifndef GS_FWD_MINW
#define GS_FWD_MINW 5               // __launch_bounds__ waves/SIMD of the forward (89 VGPRs)
#endif
#ifndef GS_LIVE_RECT
#define GS_LIVE_RECT 1              // no-op test against the rectangle of the pixels still taking entries (0: against the whole tile; A/B builds)
#endif
#ifndef GS_FWD_PACK
#define GS_FWD_PACK 1               // forward: pack the live pixels into one or two slots once they fit (0: A/B builds)
#endif
#ifndef GS_FWD_NO_PREFETCH
#define GS_FWD_NO_PREFETCH 0        // 1: measurement build -- no payload rows / ids gathered ahead of the early-out decision (exposes the gather latency)
#endif
#ifndef GS_BWD_PAIR
#define GS_BWD_PAIR 1               // small grids: the backward keeps two entries in flight per wave (0: A/B builds)
#endif
#ifndef GS_FWD_UNROLL
#define GS_FWD_UNROLL 2             // entries interleaved in the forward's per-entry loop
#endif
constexpr int kFwdUnroll = GS_FWD_UNROLL;
#define CB 64                       // splats staged per batch
#ifndef L2_SEG
#define L2_SEG 2048                 // gs_bin3.hip: coarse entries per level-2 work item (gs_bin3_seg() on the host side)
#endif
__device__ __forceinline__ int gs_bin3_seg_const() { return L2_SEG; }
#define NEG_HALF_LOG2E (-0.72134752044448170368f)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// A loop-invariant constant the compiler must keep in a VGPR: measured on gfx950 (tools/valu_ubench3.hip) a
// v_fma_f32 with an SGPR source issues at 4.4 cycles per wave64, with VGPR sources only at 2.3.
__device__ __forceinline__ float vgpr_const(float x) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// The tile of this workgroup: its own index, or -- plain launch over a launch order (gs_config.schedule 3 / 4) -- order[blockIdx]
// (holes of the order: GS_LPT_NONE).  -1: nothing to do.  nparts / part: the waves that share the tile and which of them this is --
// the same for every tile (a.parts: small grids), or per tile, carried by the order entry: tile | part << 28 | log2(nparts) << 30
// (tile_lpt_order_kernel splits the tiles whose work stands far above the rest: a trained scene's heavy tail).
#define GS_ORDER_TILE_MASK 0x0FFFFFFFu
__device__ __forceinline__ int tile_of_block(const GsCompositeArgs &a, int ntiles, int &part, int &nparts, int first_block = 0) {
    const int len = (a.tile_order && a.order_len > 0) ? a.order_len : ((ntiles + 7) / 8) * 8;   // composite_grid(): blocks of one part
    int b = (int)blockIdx.x - first_block;
    part = 0; nparts = a.parts > 1 ? a.parts : 1;
    if (a.parts > 1) { part = b / len; b -= part * len; if (part >= a.parts) return -1; }   // the parts of a tile: same XCD (len % 8 == 0)
    if (a.tile_order) {
        if (b >= (a.order_len > 0 ? a.order_len : ntiles)) return -1;
        uint32_t t = a.tile_order[b];
        if (t == 0xFFFFFFFFu) return -1;
        if (a.parts <= 1 && (t >> 28)) {                                  // a split tile's entry
            part = (int)((t >> 28) & 3u); nparts = 1 << (t >> 30);
            if (!a.split_ok) { if (part) return -1; nparts = 1; }         // this launch composites whole tiles only: part 0 stands for the tile
        }
        t &= GS_ORDER_TILE_MASK;
        return t < (uint32_t)ntiles ? (int)t : -1;
    }
    return b < ntiles ? b : -1;
}
// the pixel strips (slots) part `part` of a tile's `nparts` owns: all four, a pair, or one
__device__ __forceinline__ uint32_t strips_of_part(int nparts, int part) {
    return nparts == 4 ? (1u << part) : nparts == 2 ? (3u << (2 * part)) : 0xFu;
}

__device__ __forceinline__ unsigned long long wave_hw_id() {
    // HW_REG_HW_ID (4): wave, simd, cu, sh, se ids; HW_REG_XCC_ID (20): the XCD
    const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    return (unsigned long long)hw | ((unsigned long long)xcc << 32);
}

// ---------------------------------------------------------------- staging
// The pixel-box test is arithmetic: v_cmp / v_cndmask cost ~4 cycles each on
// gfx950 (fma: 2), so the box is applied as an exponent penalty
//     pw' = pw - BIG * |d - med3(d, lo, hi)|        (d = pixel - mu on that axis)
// which is exactly 0 inside the box (med3 returns d itself) and drives exp2 to 0 outside.
// lo/hi = box edge - mu -/+ 0.25 are lane independent and staged once per (tile, splat).
// A (tile, splat) entry whose largest alpha over the tile's pixels is below 2^-27 is a no-op in the
// reference's own fp32 arithmetic: T*(1-alpha) == T exactly (alpha < 2^-25 already rounds 1-alpha to 1)
// and rgb*alpha*T is below 7.5e-9*|rgb|, under half an ulp of any accumulated colour above 1e-7.  Such
// entries are dropped while staging (gs_config.alpha_cull, default on; lists stay the reference's).
#define GS_ALPHA_CULL_LOG2 (-27.0f)
// Lane-independent terms of one splat, formed once per (tile, splat) by the staging lane from the payload row:
// q0 = {mu_x, mu_y, log2 sig, x_lo}, q1 = {k i0, k (i1+i2), k i3, x_hi}, q2 = {r, g, b, y_lo}, y_hi   (k = -1/2 log2 e)
// Can any pixel of the rectangle [rx0,rx1] x [ry0,ry1] (relative to mu, already clipped to the splat's pixel box)
// reach alpha >= 2^GS_ALPHA_CULL_LOG2?  f(dx,dy) = A dx^2 + B dx dy + C dy^2 (log2 units, concave) is maximised over
// the rectangle at the centre if it is inside, else on an edge facing the centre, where the 1-D maximiser is clamped
// to the edge (tests/test_cull_bound.py checks this restated in NumPy against brute force).  Anything not provably
// concave and finite counts as contributing.
__device__ __forceinline__ bool rect_can_contribute(float A, float B, float C, float hBrA, float hBrC, bool concave, float l2s,
                                                    float rx0, float rx1, float ry0, float ry1) {
    const float cx = __builtin_amdgcn_fmed3f(0.0f, rx0, rx1), cy = __builtin_amdgcn_fmed3f(0.0f, ry0, ry1);
    const float dy1 = __builtin_amdgcn_fmed3f(-(hBrC * cx), ry0, ry1);
    const float dx2 = __builtin_amdgcn_fmed3f(-(hBrA * cy), rx0, rx1);
    const float f1 = fmaf(A * cx, cx, dy1 * fmaf(B, cx, C * dy1));
    const float f2 = fmaf(C * cy, cy, dx2 * fmaf(B, cy, A * dx2));
    const float fm = (cx != 0.0f && cy != 0.0f) ? fmaxf(f1, f2) : (cx != 0.0f ? f1 : f2);
    const bool nopix = rx0 > rx1 || ry0 > ry1;                          // no pixel of the rectangle inside the box
    return !(nopix || (concave && fm + l2s < GS_ALPHA_CULL_LOG2));
}

// keep = false: the whole (tile, splat) entry is a no-op (gs_config.alpha_cull).
// STRIPS (backward): additionally, bit p of `strips` = the 16 x 4 pixel strip p of the tile (rows 4p .. 4p+3: the pixels the
// lanes hold in slot p) can reach alpha >= 2^-27 somewhere.  The backward skips the dead strips of an entry with wave-uniform
// branches: at C3 24 % of the strips of the evaluated entries are dead.  `keep` is the same tile-level test in both kernels, so
// forward and backward evaluate the same entries.
// (qx0 .. qx1) x (qy0 .. qy1): the pixels the entry is tested against -- the tile, or a rectangle inside it that holds every pixel
// still taking entries
template <bool STRIPS>
__device__ __forceinline__ void stage_record(const float4 &n0, const float4 &n1, const float4 &n3, const int qx0, const int qx1, const int qy0, const int qy1,
                                             const int ty0, bool &keep, uint32_t &strips) {
    // payload quads (gs_common.h): n0 = {mu_x, mu_y, log2 sig (capped below 0), x_lo}, n1 = {k i0, k (i1+i2), k i3, x_hi}, n2 = {r, g, b, y_lo},
    // n3 = {y_hi, sig, box x, box y}: n0..n2 and n3.x go to LDS as they are; this is only the no-op test.
    // log2(sig) arrives capped three ulps below 0 (gs_preprocess.hip), so alpha = exp2(pw + l2s) < 1 strictly (pw <= 0: the conic
    // is PSD) even when v_exp_f32 returns a value one ulp high; only matters when sigmoid(o) > 1 - 1.8e-7 (o > 15.5): relative change 1.8e-7
    const uint32_t bbx = __float_as_uint(n3.z), bby = __float_as_uint(n3.w);
    const int xmin = (int)(short)(bbx & 0xFFFFu), xmax = (int)(short)(bbx >> 16);
    const int ymin = (int)(short)(bby & 0xFFFFu), ymax = (int)(short)(bby >> 16);
    const bool empty = xmax < xmin || ymax < ymin;
    const float l2s = n0.z;
    const float A = n1.x, B = n1.y, C = n1.z;
    const bool concave = A < 0.0f && C < 0.0f && 4.0f * A * C - B * B > 0.0f;
    const float rx0 = (float)max(qx0, xmin) - n0.x, rx1 = (float)min(qx1, xmax) - n0.x;
    const float ry0 = (float)max(qy0, ymin) - n0.y, ry1 = (float)min(qy1, ymax) - n0.y;
    const float hBrA = 0.5f * B * fast_rcp(A), hBrC = 0.5f * B * fast_rcp(C);
    keep = !empty && rect_can_contribute(A, B, C, hBrA, hBrC, concave, l2s, rx0, rx1, ry0, ry1);
    strips = 0xFu;
    if (STRIPS) {
        strips = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float sy0 = (float)max(max(ty0 + 4 * p, qy0), ymin) - n0.y, sy1 = (float)min(min(ty0 + 4 * p + 3, qy1), ymax) - n0.y;
            if (keep && rect_can_contribute(A, B, C, hBrA, hBrC, concave, l2s, rx0, rx1, sy0, sy1)) strips |= 1u << p;
        }
    }
}

// The gathered row of the NEXT batch must stay untouched in the registers it was loaded into until the per-entry loop of the current
// batch is over: any earlier "use" -- even a register copy the allocator inserts to split a live range -- makes the compiler wait for
// the gather before the loop, i.e. exposes one memory round trip per batch.  Passing the sixteen components through an empty asm
// AFTER the loop makes that the first use.
__device__ __forceinline__ void first_use_here(float4 &a, float4 &b, float4 &c, float4 &d) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w),
                      "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
}

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d));
    return __builtin_amdgcn_readfirstlane(v);
}

// The rectangle of the live pixels of a tile whose lanes own FIXED pixels (lane -> column lane & 15, row (lane >> 4) + 4 p of slot p): from the four
// live masks, on the scalar unit -- the same numbers as four wave-wide min / max reductions (24 cross-lane moves) at a batch boundary where pixels froze.
// tx0 / ty0: the tile's first pixel column / row (1-based).  No live pixel: (2^20, -1, 2^20, -1), as the reductions over nothing give.
#ifndef GS_RECT_FROM_MASKS
#define GS_RECT_FROM_MASKS 1
#endif
struct LiveRectAcc {
    uint64_t any = 0; uint32_t rows = 0;
    __device__ __forceinline__ void slot(const int p, const uint64_t m) {   // one slot's live mask at a time (few scalar registers alive)
        any |= m;
        uint64_t t = m | (m >> 8); t |= t >> 4; t |= t >> 2; t |= t >> 1;     // bit 16 r = row r of the slot has a live pixel
        t &= 0x0001000100010001ull;
        rows |= (uint32_t)((t | (t >> 15) | (t >> 30) | (t >> 45)) & 0xFull) << (4 * p);
    }
    __device__ __forceinline__ void rect(const int tx0, const int ty0, int &qx0, int &qx1, int &qy0, int &qy1) const {
        const uint32_t cols = (uint32_t)((any | (any >> 16) | (any >> 32) | (any >> 48)) & 0xFFFFull);
        if (cols == 0u) { qx0 = 1 << 20; qx1 = -1; qy0 = 1 << 20; qy1 = -1; return; }
        qx0 = tx0 + (int)__builtin_ctz(cols); qx1 = tx0 + 31 - (int)__builtin_clz(cols);
        qy0 = ty0 + (int)__builtin_ctz(rows); qy1 = ty0 + 31 - (int)__builtin_clz(rows);
    }
};

// kept-entry slot of this lane inside the wave's keep mask
__device__ __forceinline__ int slot_of(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// one staged entry as the per-pixel loops read it back from LDS (13 dwords, wave-uniform addresses: broadcasts)
struct Entry { float4 q0, q1, q2; float yhi; };
__device__ __forceinline__ Entry load_entry(const float4 *sp, const float *syhi, int k) {
    Entry e;
    e.q0 = sp[3 * k]; e.q1 = sp[3 * k + 1]; e.q2 = sp[3 * k + 2]; e.yhi = syhi[k];
    return e;
}

// ---------------------------------------------------------------- capped lists: the wave writes the rest of its tile's list itself
// gs_bin wrote the tile's list only up to the end of a segment of its super-tile's coarse list (GsBin3Args.cap_src: what the view
// slot's previous forward walked, and a quarter more).  A forward wave that gets to the written end with pixels still taking
// entries appends the hits of the NEXT segment -- the same test and the same order as l2_write_kernel (gs_bin3.hip), into the
// entries the tile's range reserves for them -- and goes on; cont moves to the following segment (GS_CONT_NONE after the last).
// Returns the new written end.  Rare by construction (a camera jump under a reused slot, a tile that saturated later than last
// time); the walk, the batch boundaries and therefore image, transmittance and gradients are those of the full list.
__device__ __forceinline__ uint32_t extend_tile_list(const GsCompositeArgs &a, const int tile, uint32_t &cont, uint32_t s1) {
    const int lane = threadIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int sbs = a.sbs;
    const uint32_t sm = (1u << sbs) - 1u;
    const int S = (ty >> sbs) * a.sgx + (tx >> sbs);
    const uint32_t lx = (uint32_t)tx & sm, ly = (uint32_t)ty & sm;
    const uint32_t c1 = a.cranges[2 * S + 1];
    const uint32_t seg = (uint32_t)gs_bin3_seg_const();
    const uint32_t e_end = min(cont + seg, c1);
    for (uint32_t e0 = cont; e0 < e_end; e0 += GS_WAVE) {
        const uint32_t e = e0 + (uint32_t)lane;
        bool hit = false;
        uint32_t id = 0;
        if (e < e_end) {
            const uint32_t lr = a.clr[e];
            hit = (lr & sm) <= lx && lx <= ((lr >> sbs) & sm) && ((lr >> (2 * sbs)) & sm) <= ly && ly <= ((lr >> (3 * sbs)) & sm);
            id = a.cids[e];
        }
        const uint64_t bal = __ballot(hit);
        if (hit) a.ids_w[s1 + (uint32_t)slot_of(bal)] = id;
        s1 += (uint32_t)__popcll(bal);
    }
    cont = e_end < c1 ? e_end : GS_CONT_NONE;
    if (lane == 0 && a.ext_count) atomicAdd(a.ext_count, 1u);
    __threadfence();                                        // the wave reads these ids back: stores complete, this CU's L1 lines dropped
    return s1;
}

// ---------------------------------------------------------------- forward
template <bool EARLY, bool CULL, bool CLK, bool SLAB, bool SNAP = false>
// SNAP: the launch order has split tiles whose backward runs as list segments: their forward waves leave snapshots (GsCompositeArgs.snap).
// An instantiation of its own, chosen by the host from the order kernel's count of split tiles: the snapshot code costs the ordinary
// kernel six spilled VGPRs at five waves per SIMD, and the BASELINE scenes split nothing.
// SLAB: the frame is binned in depth slabs (several rounds; resume / tile_pos / tile_done / tile_dead): its own instantiation, the
// single-round kernel carries none of that state (with it the compiler spilled: 96 VGPRs + 28 bytes of scratch against 90).
// CLK: per-tile debug clocks (gs_debug_tile_clock); a separate instantiation so that the production kernel carries none of it.
// (Skipping the dead 16 x 4 strips of an entry with wave-uniform branches, as the backward does, was measured on the forward
// too: 0.352 vs 0.357 ms at C3 -- its per-strip work is a 12-instruction dependent chain, the branches cost what they save.)
//
// PACKING THE LIVE PIXELS (round 4; single-round frames with the early-out).  A frozen pixel takes no further entry, but its lane
// slot is executed as long as the tile walks: measured at C3 (profiles/r04c_tile_tail_C3.json), 21 % of the evaluated entries are
// composited while 64 or fewer of the tile's 256 pixels are live, another 11 % with 128 or fewer.  At a batch boundary where the live
// pixels fit two slots (or one) the wave writes the frozen pixels' final colour and transmittance to memory, moves the live ones --
// {x, y, C, T}, through the staging buffer in LDS, in slot-then-lane order -- into slots 0 .. K-1 and runs the rest of the list with
// K = 2 or 1 slots per entry instead of 4.  A packed slot carries its own x (the shared per-lane column is gone), so a slot costs
// 18 VALU instructions instead of 12 + 6 shared: K = 2 is 36 against 54, K = 1 is 18.  Every pixel sees the same entries in the same
// order with the same arithmetic, so the image and the transmittance are bit-identical to the unpacked walk.
__device__ __forceinline__ void forward_tile(const GsCompositeArgs &a, const int tile, const int part, const int nparts, float4 *sp, float *syhi, const float nbig,
                                             const int snap_slot = -1, const uint32_t seg_len = 0) {
    const int lane = threadIdx.x;
    const int px = (tile % a.gx) * GS_TILE + (lane & 15) + 1;
    const int py0 = (tile / a.gx) * GS_TILE + (lane >> 4) + 1;
    const float fx = (float)px;
    const uint32_t s0 = a.ranges[2 * tile];
    uint32_t s1 = a.ranges[2 * tile + 1];
    uint32_t cont = GS_CONT_NONE;                                  // capped lists: where the unwritten rest of the list starts in the coarse list
    const bool capped = !SLAB && EARLY && a.tile_ext != nullptr;
    if (capped) { const uint2 ex = a.tile_ext[tile]; s1 = s0 + ex.x; cont = ex.y; }
    const int ty0 = py0 - (lane >> 4);                             // first pixel row of the tile (1-based)
    unsigned long long clk0 = 0;
    if (CLK) clk0 = __builtin_amdgcn_s_memrealtime();

    // Slab frames (binning in depth slabs): the tile's list arrives in several rounds.  A later round resumes the
    // pixel state the previous one left in image / trans and in tile_dead (four 64-bit lane masks per tile: pixel slot p of
    // lane l is frozen -- a flag of its own, because a LIVE pixel's T may be negative when a conic is not positive definite
    // and then looks like nothing else than a negative T), continues the list position (tile_pos) so that the 64-entry batch
    // boundaries -- where the early-out rule freezes pixels -- stay those of the whole list, and a tile whose pixels are all
    // frozen is marked done and takes no further instances.
    if (SLAB && a.resume && a.tile_done[tile]) return;
    const uint32_t gp0 = SLAB && a.tile_pos ? a.tile_pos[tile] : 0u;
    const uint32_t plane = (uint32_t)a.W * (uint32_t)a.H;           // pixel indices fit 32 bits: W, H <= 32767 (gs_set_camera), 3 W H < 2^32
    // (a frozen pixel's transmittance is final: it goes to memory when the pixel freezes, not at the end of the tile)
    float Cr[4], Cg[4], Cb[4], T[4], fy[4];
    bool dead[4];
    uint32_t walked = 0, evaluated = 0;
    constexpr bool PACK = GS_FWD_PACK && EARLY && !SLAB;
    const uint32_t own = (EARLY && !SLAB) ? strips_of_part(nparts, part) : 0xFu;   // several waves per tile: the strips this wave composites
    bool first_pack = PACK && nparts > 1;                              // ... packed into K = 2 / 1 slots at the first batch
    int K = 4;                                                          // slots per entry: 4 = the tile's pixels in place; 2 / 1 = live pixels packed
    // heavy tiles (GsCompositeArgs.snap): at every seg_len entries this wave leaves (C, T) of its pixels for the backward's list segments
    // (three scalars of state -- the slot, the segment length, the next boundary -- everything else is recomputed where it is used: this
    // kernel has no SGPR to spare)
    uint32_t next_snap = (SNAP && snap_slot >= 0) ? seg_len : 0xFFFFFFFFu;
    // packed: slots 2 and 3 hold no pixel, and fy[2], fy[3] hold the x of the pixels in slots 0 and 1 (0 = the slot is empty)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        Cr[p] = Cg[p] = Cb[p] = 0.0f;
        fy[p] = (float)(py0 + 4 * p);
        const bool in = (px <= a.W && py0 + 4 * p <= a.H) && ((own >> p) & 1u);   // (a strip of another part: no pixel of this wave)
        T[p] = in ? 1.0f : 0.0f;
        dead[p] = !in;
        if (SLAB && a.resume && in) {
            const uint32_t o = (uint32_t)(px - 1) + (uint32_t)a.W * (uint32_t)(py0 + 4 * p - 1);
            Cr[p] = a.image[o]; Cg[p] = a.image[o + plane]; Cb[p] = a.image[o + 2u * plane];
            dead[p] = ((a.tile_dead[4 * (size_t)tile + p] >> lane) & 1ull) != 0ull;
            T[p] = dead[p] ? 0.0f : a.trans[o];                           // (a frozen pixel's transmittance stays where the round that froze it put it)
        }
    }
    // The no-op test of an entry (stage_record) runs against the rectangle that holds every pixel STILL TAKING ENTRIES, not against the
    // whole tile: an entry that can only reach frozen pixels (or pixels outside a ragged image edge) is as much a no-op as one that
    // reaches none.  Refreshed at the batch boundaries where pixels froze; forward and backward freeze the same pixels at the same
    // boundaries, so both evaluate the same entries (C3: 6.8 % fewer than with the tile's rectangle, profiles/r04e_tile_tail_C3.json).
    int qx0, qx1, qy0, qy1;
    auto live_rect = [&]() {
        // (the backward takes this rectangle from the four live masks on the scalar unit, LiveRectAcc; here the same code costs the kernel its last
        // scalar registers -- 106, three to seven VGPRs spilled -- so the forward keeps the four wave reductions)
        int lx0 = 1 << 20, lx1 = -1, ly0 = 1 << 20, ly1 = -1;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!dead[p]) {
                const int x = (int)((!PACK || K == 4) ? fx : fy[2 + (p & 1)]), y = (int)fy[p];
                lx0 = min(lx0, x); lx1 = max(lx1, x); ly0 = min(ly0, y); ly1 = max(ly1, y);
            }
        }
        qx0 = wave_min_i32(lx0); qx1 = wave_max_i32(lx1); qy0 = wave_min_i32(ly0); qy1 = wave_max_i32(ly1);
        if (!GS_LIVE_RECT) { qx0 = (tile % a.gx) * GS_TILE + 1; qx1 = qx0 + GS_TILE - 1; qy0 = ty0; qy1 = ty0 + GS_TILE - 1; }
    };
    {   // at the start every pixel inside the image is live: the tile's rectangle clipped to the image (no reduction needed), unless a
        // slab round resumes with frozen pixels
        qx0 = (tile % a.gx) * GS_TILE + 1; qx1 = min(qx0 + GS_TILE - 1, a.W);
        qy0 = ty0 + 4 * (int)__builtin_ctz(own); qy1 = min(ty0 + 4 * (31 - (int)__builtin_clz(own)) + 3, a.H);
        if (SLAB && a.resume) live_rect();
    }
    const float4 *pay4 = reinterpret_cast<const float4 *>(a.payload);
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0, n3 = n0;
    uint32_t pos = s0 + lane;
    // Gathers run ahead of their use: the payload rows one batch (held in n0..n2 while the previous batch is composited), the
    // ids they are addressed by TWO batches (id2), so that the row loads of the next batch are issued from a register instead of
    // behind a second dependent round trip to memory.  A tile whose batches keep few entries (light tiles, the ones that run
    // when the chip is emptying) is bound by exactly this chain: one exposed load latency per batch instead of two.
    uint32_t id2 = 0;
    uint32_t gp = gp0;                                                  // list position of `base` in the tile's whole list
    uint32_t base = s0;
    unsigned long long t_loop = 0, t_stage = 0, t_mark = 0;             // debug clocks (a.tile_clock): shader cycles inside / outside the per-entry loops
    // debug (a.tile_clock): per-entry strip slots executed, the slots live pixels compacted to 64 per slot would need, the strips with
    // any live pixel, and live pixels, each summed over the evaluated entries (frozen-pixel work inside live strips)
    unsigned long long clk_exec = 0, clk_ideal = 0, clk_alive = 0, clk_pix = 0;
    uint32_t clk_live = 0, clk_strips = 0;
    // ... and, for three ways of packing the live pixels into fewer 64-lane slots, the evaluated entries by the slots K = 1 .. 4 they
    // would run on: any pixel anywhere (K = ceil(live / 64)); whole pixel ROWS moved (a lane keeps its column: K = ceil(rows with a
    // live pixel / 4)); pixels moved inside their COLUMN (K = ceil(fullest column / 4))
    uint32_t clk_hist[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int clk_k[3] = {4, 4, 4};
    uint32_t clk_bbkeep = 0;                                            // ... and the entries the no-op test would keep against the whole tile
    if (CLK) t_mark = __builtin_amdgcn_s_memtime();
    for (;;) {                                                          // (capped lists: once more per segment the wave appends itself)
    bool stopped = false;
    {
        const uint32_t pos2 = pos + min((uint32_t)CB - (gp & (CB - 1)), s1 - base);     // first position of the second batch + lane
        if (pos < s1) { const size_t g = a.ids[pos]; n0 = pay4[4 * g]; n1 = pay4[4 * g + 1]; n2 = pay4[4 * g + 2]; n3 = pay4[4 * g + 3]; }
        if (pos2 < s1) id2 = a.ids[pos2];
    }
    for (; base < s1;) {
        const uint32_t phase = gp & (CB - 1);
        const int cnt = (int)min((uint32_t)CB - phase, s1 - base);      // batches end at multiples of CB of the WHOLE list
        if (EARLY && phase == 0) {
            bool live = false, froze = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (!dead[p] && T[p] < a.t_min) {                        // frozen from here on: its transmittance is final
                    const float x = (!PACK || K == 4) ? fx : fy[2 + (p & 1)];
                    if (a.trans) a.trans[(uint32_t)((int)x - 1) + (uint32_t)a.W * (uint32_t)((int)fy[p] - 1)] = T[p];
                    if (SNAP && snap_slot >= 0) {                        // ... and every later snapshot says so (NaN: no segment takes this pixel up again)
                        const int nsn = a.seg_hist ? a.seg_n - 1 : GS_SEG_MAX - 1;      // snapshots per tile
                        float *st = a.snap + (size_t)snap_slot * (size_t)(nsn * 4 * 256) + 768 + ((int)fy[p] - ty0) * GS_TILE + ((int)x - (px - (lane & 15)));
                        for (int kk = (int)(next_snap / seg_len) - 1; kk < nsn; ++kk) st[kk * (4 * 256)] = __int_as_float(0x7FC00000);
                    }
                    dead[p] = true; T[p] = 0.0f; froze = true;
                }
                live = live || !dead[p];
            }
            if (__ballot(live) == 0ull) { stopped = true; break; }
            const bool anyfroze = __ballot(froze) != 0ull;                // (nothing froze: rectangle and live count are what they were)
            if (anyfroze) live_rect();
            if (PACK && K > 1 && (anyfroze || first_pack)) {
                first_pack = false;
                uint64_t lm[4];
                uint32_t nlive = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p) { lm[p] = __ballot(!dead[p]); nlive += (uint32_t)__popcll(lm[p]); }
                const int Kn = (int)((nlive + 63u) >> 6);
                if (Kn <= 2 && Kn < K) {
                    // (1) the pixels that are dropped are final: their colour goes to memory now (their transmittance went when they froze)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float x = K == 4 ? fx : fy[2 + (p & 1)];
                        const bool mine = K == 4 ? (px <= a.W && py0 + 4 * p <= a.H && ((own >> p) & 1u)) : (p < 2 && x > 0.5f);
                        if (mine && dead[p] && a.image) {
                            const uint32_t o = (uint32_t)((int)x - 1) + (uint32_t)a.W * (uint32_t)((int)fy[p] - 1);
                            a.image[o] = Cr[p]; a.image[o + plane] = Cg[p]; a.image[o + 2u * plane] = Cb[p];
                        }

                    }
                    // (2) live pixels -> ranks in slot-then-lane order -> six planes of 128 floats in the staging buffer -> slots 0 .. Kn-1
                    float *buf = reinterpret_cast<float *>(sp);
                    __syncthreads();
                    uint32_t rb = 0;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        if (!dead[p]) {
                            const uint32_t r = rb + (uint32_t)slot_of(lm[p]);
                            buf[r] = K == 4 ? fx : fy[2 + (p & 1)]; buf[128 + r] = fy[p];
                            buf[256 + r] = Cr[p]; buf[384 + r] = Cg[p]; buf[512 + r] = Cb[p]; buf[640 + r] = T[p];
                        }
                        rb += (uint32_t)__popcll(lm[p]);
                    }
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t idx = (uint32_t)(q * 64 + lane);
                        const bool has = idx < nlive;
                        fy[2 + q] = has ? buf[idx] : 0.0f;
                        fy[q] = has ? buf[128 + idx] : 0.0f;
                        Cr[q] = has ? buf[256 + idx] : 0.0f; Cg[q] = has ? buf[384 + idx] : 0.0f; Cb[q] = has ? buf[512 + idx] : 0.0f;
                        T[q] = has ? buf[640 + idx] : 0.0f;
                        dead[q] = !has; dead[2 + q] = true;
                        T[2 + q] = 0.0f;
                    }
                    K = Kn;
                }
            }
        }
        if (SNAP && EARLY && !SLAB && gp == next_snap) {                         // (wave-uniform; a batch boundary: seg_len is a multiple of CB)
            const int ksnap = (int)(next_snap / seg_len) - 1;            // snapshots written so far = index of this one
            const int nsn = a.seg_hist ? a.seg_n - 1 : GS_SEG_MAX - 1;
            if (ksnap < nsn) {
                float *sn = a.snap + (size_t)snap_slot * (size_t)(nsn * 4 * 256) + ksnap * (4 * 256);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (!dead[p]) {
                        const float x = (!PACK || K == 4) ? fx : fy[2 + (p & 1)];
                        const int idx = ((int)fy[p] - ty0) * GS_TILE + ((int)x - (px - (lane & 15)));
                        sn[idx] = Cr[p]; sn[256 + idx] = Cg[p]; sn[512 + idx] = Cb[p]; sn[768 + idx] = T[p];
                    }
                }
            }
            next_snap += seg_len;
        }
        if (CLK && (phase == 0 || base == s0)) {                         // debug: live pixels / strips with a live pixel in this batch
            clk_live = 0; clk_strips = 0;
            uint32_t rows = 0, colcnt[16];
            for (int cidx = 0; cidx < 16; ++cidx) colcnt[cidx] = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint64_t m = __ballot(!dead[p]); clk_live += (uint32_t)__popcll(m); clk_strips += m ? 1u : 0u;
                for (int r4 = 0; r4 < 4; ++r4) rows += ((m >> (16 * r4)) & 0xFFFFull) ? 1u : 0u;
                for (int cidx = 0; cidx < 16; ++cidx) colcnt[cidx] += (uint32_t)__popcll(m & (0x0001000100010001ull << cidx));
            }
            uint32_t cmax = 0;
            for (int cidx = 0; cidx < 16; ++cidx) cmax = max(cmax, colcnt[cidx]);
            clk_k[0] = (int)((clk_live + 63u) >> 6); clk_k[1] = (int)((rows + 3u) >> 2); clk_k[2] = (int)((cmax + 3u) >> 2);
        }
#if GS_FWD_NO_PREFETCH                                                  // measurement build (tools/fwd_traffic_split.sh): the rows of a batch are gathered
        if (base > s0 && pos < s1) {                                    // only once the tile is known to go on: nothing is fetched for nobody
            const size_t g = a.ids[pos]; n0 = pay4[4 * g]; n1 = pay4[4 * g + 1]; n2 = pay4[4 * g + 2]; n3 = pay4[4 * g + 3];
        }
#endif
        uint32_t strips;
        bool keep;
        stage_record<false>(n0, n1, n3, qx0, qx1, qy0, qy1, ty0, keep, strips);
        int slot = lane, nk = cnt;
        if (!CULL) keep = true;
        if (CULL) {                                                     // compact the batch to the entries that can matter
            keep = keep && lane < cnt;
            const uint64_t m = __ballot(keep);
            slot = slot_of(m); nk = __popcll(m);
        }
        if (CLK && CULL) {                                              // debug: what the tile's own rectangle would have kept
            bool k2; uint32_t st2;
            stage_record<false>(n0, n1, n3, px - (lane & 15), px - (lane & 15) + GS_TILE - 1, ty0, ty0 + GS_TILE - 1, ty0, k2, st2);
            clk_bbkeep += (uint32_t)__popcll(__ballot(k2 && lane < cnt));
        }
        __syncthreads();                                                // one wave: orders LDS reads/writes only
        if (keep) {
            sp[3 * slot] = n0; sp[3 * slot + 1] = n1; sp[3 * slot + 2] = n2;
            syhi[slot] = n3.x;
        }
        __syncthreads();
        base += (uint32_t)cnt; gp += (uint32_t)cnt;
        pos = base + lane;
#if !GS_FWD_NO_PREFETCH
        {
            if (pos < s1) { const size_t g = id2; n0 = pay4[4 * g]; n1 = pay4[4 * g + 1]; n2 = pay4[4 * g + 2]; n3 = pay4[4 * g + 3]; }
            const uint32_t pos2 = pos + min((uint32_t)CB, s1 - base);                   // (batches after the first start at multiples of CB)
            if (base < s1 && pos2 < s1) id2 = a.ids[pos2];
        }
#endif
        if (CLK) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_stage += t - t_mark; t_mark = t; }
// one pixel slot p of one entry (A0, B0: the column terms of the slot's x)
#define GS_FWD_PIXEL(e, A0, B0, p) do {                                               \
            const float dY = fy[p] - (e).q0.y;                                              \
            const float ey = dY - __builtin_amdgcn_fmed3f(dY, (e).q2.w, (e).yhi);           \
            const float pw = fmaf(dY, fmaf((e).q1.z, dY, (B0)), (A0));                      \
            const float w = fast_exp2(fmaf(nbig, fabsf(ey), pw)) * T[p];                    \
            Cr[p] = fmaf((e).q2.x, w, Cr[p]);                                               \
            Cg[p] = fmaf((e).q2.y, w, Cg[p]);                                               \
            Cb[p] = fmaf((e).q2.z, w, Cb[p]);                                               \
            T[p] = T[p] - w;                                                                \
        } while (0)
// the column terms of an entry for a pixel column x
#define GS_FWD_COLUMN(e, XCOL, A0, B0)                                                        \
            const float dX##A0 = (XCOL) - (e).q0.x;                                           \
            const float ex##A0 = dX##A0 - __builtin_amdgcn_fmed3f(dX##A0, (e).q0.w, (e).q1.w);   /* 0 inside the box columns */ \
            const float A0 = fmaf(nbig, fabsf(ex##A0), fmaf((e).q1.x * dX##A0, dX##A0, (e).q0.z)); /* k i0 dX^2 + log2 sig - penalty */ \
            const float B0 = (e).q1.y * dX##A0
        if (!PACK || K == 4) {
#pragma clang loop unroll_count(kFwdUnroll)
            for (int k = 0; k < nk; ++k) {
                const Entry e = load_entry(sp, syhi, k);
                GS_FWD_COLUMN(e, fx, A0, B0);
                GS_FWD_PIXEL(e, A0, B0, 0); GS_FWD_PIXEL(e, A0, B0, 1); GS_FWD_PIXEL(e, A0, B0, 2); GS_FWD_PIXEL(e, A0, B0, 3);
            }
        } else if (K == 2) {                                            // packed: every slot has its own column
#pragma clang loop unroll_count(kFwdUnroll)
            for (int k = 0; k < nk; ++k) {
                const Entry e = load_entry(sp, syhi, k);
                GS_FWD_COLUMN(e, fy[2], A0, B0);
                GS_FWD_COLUMN(e, fy[3], A1, B1);
                GS_FWD_PIXEL(e, A0, B0, 0); GS_FWD_PIXEL(e, A1, B1, 1);
            }
        } else {
#pragma clang loop unroll_count(4)
            for (int k = 0; k < nk; ++k) {
                const Entry e = load_entry(sp, syhi, k);
                GS_FWD_COLUMN(e, fy[2], A0, B0);
                GS_FWD_PIXEL(e, A0, B0, 0);
            }
        }
#undef GS_FWD_PIXEL
#undef GS_FWD_COLUMN
        walked += (uint32_t)cnt; evaluated += (uint32_t)nk;
        first_use_here(n0, n1, n2, n3);
        if (CLK) {
            const unsigned long long t = __builtin_amdgcn_s_memtime(); t_loop += t - t_mark; t_mark = t;
            clk_exec += (unsigned long long)K * (uint32_t)nk; clk_ideal += (unsigned long long)nk * ((clk_live + 63u) >> 6);
            clk_alive += (unsigned long long)nk * clk_strips; clk_pix += (unsigned long long)nk * clk_live;
            for (int w = 0; w < 3; ++w) if (clk_k[w] >= 1 && clk_k[w] <= 4) clk_hist[w][clk_k[w] - 1] += (uint32_t)nk;
        }
    }
    if (!capped || stopped || cont == GS_CONT_NONE) break;
    // the written list is used up, its pixels still take entries and the super-tile's list goes on: append the next segment's hits
    // (whether the pixels are frozen at the next batch boundary is decided by the loop above, at the same position as in the full list)
    s1 = extend_tile_list(a, tile, cont, s1);
    pos = base + lane;
    }
    if (capped && lane == 0) a.tile_ext[tile] = make_uint2(s1 - s0, cont);   // (unchanged unless the list was extended) the backward's list length
    bool anylive = false;
#pragma unroll
    for (int p = 0; p < 4; ++p) anylive = anylive || !dead[p];
    const bool all_dead = __ballot(anylive) == 0ull;                    // complete: no pixel takes anything further
    if (SLAB && a.tile_dead && !a.final_round) {                        // the frozen flags travel to the next round beside the pixels
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned long long m = __ballot(dead[p]);
            if (lane == 0) a.tile_dead[4 * (size_t)tile + p] = m;
        }
    }
    if (SNAP && snap_slot >= 0 && lane == 0) {
        if (a.snap_walked) atomicMax(a.snap_walked + snap_slot, gp0 + walked);   // how far the tile's list was walked: the longest walk of its parts
        if (part == 0) { if (a.bw_walked) a.bw_walked[tile] = 0u; if (a.bw_work) a.bw_work[tile] = 0u; }   // (the backward's segments add theirs)
    }
    if (lane == 0 && part == 0) {                                       // (tile_parts > 1: the counters of a tile are those of its first part)
        if (a.tile_work) a.tile_work[tile] = SLAB && a.resume ? a.tile_work[tile] + evaluated : evaluated;
        if (a.tile_walked) a.tile_walked[tile] = SLAB && a.resume ? a.tile_walked[tile] + walked : walked;
        if (SLAB && a.tile_done) a.tile_done[tile] = all_dead ? 1 : 0;
        if (SLAB && a.tile_pos) a.tile_pos[tile] = gp0 + (s1 - s0);
    }
    if (PACK && K < 4) {                                                // packed: the pixels still held, wherever they belong
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (fy[2 + q] > 0.5f) {
                const uint32_t o = (uint32_t)((int)fy[2 + q] - 1) + (uint32_t)a.W * (uint32_t)((int)fy[q] - 1);
                if (a.image) { a.image[o] = Cr[q]; a.image[o + plane] = Cg[q]; a.image[o + 2u * plane] = Cb[q]; }
                if (a.trans && !dead[q]) a.trans[o] = T[q];
            }
        }
    } else if (px <= a.W) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int py = py0 + 4 * p;
            if (py <= a.H && ((own >> p) & 1u)) {
                const uint32_t o = (uint32_t)(px - 1) + (uint32_t)a.W * (uint32_t)(py - 1);
                if (a.image) { a.image[o] = Cr[p]; a.image[o + plane] = Cg[p]; a.image[o + 2u * plane] = Cb[p]; }
                if (a.trans && !(EARLY && dead[p])) a.trans[o] = T[p];
            }
        }
    }
    if (CLK && a.tile_clock && lane == 0) {
        unsigned long long *c = a.tile_clock + GS_TILE_CLOCK_WORDS * (size_t)(a.clock_by_block ? (int)blockIdx.x : tile);
        c[0] = clk0; c[1] = __builtin_amdgcn_s_memrealtime(); c[2] = wave_hw_id();
        c[3] = ((unsigned long long)walked << 32) | evaluated;
        c[4] = t_loop; c[5] = t_stage;
        c[6] = (clk_exec << 32) | (clk_ideal & 0xFFFFFFFFull); c[7] = (clk_alive << 32) | (clk_pix & 0xFFFFFFFFull);
        c[14] = clk_bbkeep;
        for (int w = 0; w < 3; ++w) {
            c[8 + 2 * w] = ((unsigned long long)clk_hist[w][0] << 32) | clk_hist[w][1];
            c[9 + 2 * w] = ((unsigned long long)clk_hist[w][2] << 32) | clk_hist[w][3];
        }
    }
}

template <bool EARLY, int MINW, bool CULL, bool CLK = false, bool SLAB = false, bool SNAP = false>
__global__ __launch_bounds__(64, MINW) void composite_fwd_kernel(GsCompositeArgs a) {
    __shared__ float4 sp[CB * 3];
    __shared__ float syhi[CB];
    const int ntiles = a.gx * a.gy;
    const float nbig = vgpr_const(-GS_BIG);
    int part, nparts;
    const int tile = tile_of_block(a, ntiles, part, nparts);
    if (tile < 0) return;
    int snap_slot = -1;
    uint32_t sl = 0;
    if (SNAP && EARLY && !SLAB && a.snap && a.seg_hist) {                  // small grid: every tile is segmented, slot = tile
        sl = gs_seg_len_all(a.seg_hist[tile], a.seg_n);
        if (sl) snap_slot = tile;
    } else if (SNAP && EARLY && !SLAB && a.snap && nparts > 1 && a.parts <= 1) {  // a split tile of the order: its slot = 8 x position in the XCD's list + XCD
        const int b = (int)blockIdx.x, slot = (b < a.front ? ((b >> 3) / 3) : ((b - a.front) >> 3)) * 8 + (b & 7);
        sl = a.seg_len[slot];
        if (sl) snap_slot = slot;
    }
    forward_tile<EARLY, CULL, CLK, SLAB, SNAP>(a, tile, part, nparts, sp, syhi, nbig, snap_slot, sl);
}

// ---------------------------------------------------------------- wave64 reduction of the nine per-splat sums
// The transposed reduction through LDS (header comment).  RS floats per component row: 64 lanes + 4 of padding, so that the
// sixteen-float quarters read with ds_read_b128 by lanes (c, s) fall on distinct banks (start bank (4 c + 16 s) mod 64 inside
// every 16-lane service group).  (Round 1's reduce-scatter tree on ds_swizzle / ds_bpermute and a software-pipelined form of
// this one were measured slower and are gone: profiles/HISTORY.md.)
#define RS 68
#define RED_FLOATS (9 * RS)

// g2d row of a gaussian: [dr dg db | S0 Sx Sy Sxx Sxy (unused) Syy] -- the colour gradient and the raw moments
// S.. = sum over pixels of dd * {1, dX, dY, dX^2, dX dY, dY^2}, dd = d L / d(log alpha); gs_g2d_to_grads (gs_common.h)
// turns them into d{sig, mu, conic} once per gaussian.
template <bool DET>
__device__ __forceinline__ void add_to_row(const GsCompositeArgs &a, uint32_t gid, int ocomp, float v) {
    if (DET) {     // 2^-40 (2^-28 for the second moments) fixed point, saturating; integer atomics are order independent
        const float sc = fminf(fmaxf(v * (ocomp >= 6 ? GS_FIXED_SCALE2 : GS_FIXED_SCALE), -9.0e18f), 9.0e18f);
        char *rowp = reinterpret_cast<char *>(a.g2d_fixed) + (size_t)gid * (8 * GS_G2D_STRIDE);
        atomicAdd(reinterpret_cast<unsigned long long *>(rowp + 8u * (uint32_t)ocomp), (unsigned long long)__float2ll_rn(sc));
    } else {       // row base in SGPRs, per-lane byte offset in one VGPR
        char *rowp = reinterpret_cast<char *>(a.g2d) + (size_t)gid * (4 * GS_G2D_STRIDE);
        atomicAdd(reinterpret_cast<float *>(rowp + 4u * (uint32_t)ocomp), v);
    }
}

// quad-permute DPP moves (no LDS round trip at the end of the chain)
__device__ __forceinline__ float dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); }

// ---------------------------------------------------------------- backward
// Same staging and arithmetic pixel-box penalty as the forward; per-batch freeze of saturated pixels
// (T = S = 0 makes every later contribution exactly zero); alpha < 1 strictly (stage_record), so
// 1/(1-alpha) needs no guard; d sig = -(1/sig) * sum(dd) needs no accumulator of its own.
// DET: the per-(tile, splat) sums are added as fixed-point integers (64-bit integer atomics are
// order independent, so the gradients are bitwise reproducible run to run); otherwise float atomics.

// per-pixel arithmetic of one entry: updates T, S; returns the lane's nine partial sums
// v = {dr, dg, db, S0, Sx, Sy, Sxx, Sxy, Syy}.  live: wave-uniform 4-bit mask of the strips (pixel slots) the entry can touch.
__device__ __forceinline__ void backward_entry(const Entry &e, const float fx, const float (&fy)[4], const float nbig,
                                               const float (&dCr)[4], const float (&dCg)[4], const float (&dCb)[4],
                                               float (&T)[4], float (&S)[4], float (&v)[9], bool &any, const uint32_t live) {
    const float dX = fx - e.q0.x;
    const float ex = dX - __builtin_amdgcn_fmed3f(dX, e.q0.w, e.q1.w);
    const float A0 = fmaf(nbig, fabsf(ex), fmaf(e.q1.x * dX, dX, e.q0.z));
    const float B0 = e.q1.y * dX;
    float ar = 0.0f, ag = 0.0f, ab = 0.0f, q0s = 0.0f, q1s = 0.0f, q2s = 0.0f, asum = 0.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (!((live >> p) & 1u)) continue;                        // wave-uniform branch: a dead strip costs nothing
        const float dY = fy[p] - e.q0.y;
        const float ey = dY - __builtin_amdgcn_fmed3f(dY, e.q2.w, e.yhi);
        const float xp = fmaf(nbig, fabsf(ey), fmaf(dY, fmaf(e.q1.z, dY, B0), A0));
        const float al = fast_exp2(xp);
        asum += al;
        const float w = al * T[p];
        const float cdot = fmaf(e.q2.x, dCr[p], fmaf(e.q2.y, dCg[p], e.q2.z * dCb[p]));
        S[p] = fmaf(-cdot, w, S[p]);
        const float inv = fast_rcp(1.0f - al);
        const float dalpha = fmaf(T[p], cdot, -(S[p] * inv));
        const float dd = -(al * dalpha);
        const float ddy = dd * dY;
        ar = fmaf(w, dCr[p], ar); ag = fmaf(w, dCg[p], ag); ab = fmaf(w, dCb[p], ab);
        q0s += dd; q1s += ddy; q2s = fmaf(ddy, dY, q2s);
        T[p] = T[p] - w;
    }
    any = asum != 0.0f;
    // raw moments of dd = d L / d(log alpha) about the splat's mean; the factors that are constant per gaussian
    // (1/sig, the conic, 1/2) are applied once per gaussian by the parameter kernels, not once per (tile, splat)
    const float qx = dX * q0s;
    v[0] = ar; v[1] = ag; v[2] = ab; v[3] = q0s;
    v[4] = qx; v[5] = q1s;
    v[6] = dX * qx; v[7] = dX * q1s; v[8] = q2s;
}

template <bool EARLY, bool DET, bool CULL, bool CLK, bool PAIR = false>
// PAIR (small grids: a wave is alone on its SIMD and the per-entry chain -- payload read, exp2, nine LDS writes, four LDS reads, adds, atomic -- is
// pure latency, 0.33 us per evaluated entry): two staged entries in flight, reduced through two LDS buffers (`red` holds 2 x RED_FLOATS then).
// Same sums per entry, same values; only the order in which the atomics of neighbouring entries are issued differs.
// snap_in != null / seg_start, seg_end: this wave differentiates list entries [seg_start, seg_end) of the tile only (a heavy tile's backward
// runs as segments of its list, GsCompositeArgs.snap), starting from the (C, T) the forward left at seg_start (snap_in: [4][256]; null:
// the list's start).  seg_tile: the tile's counters are the sum over its segments.
__device__ __forceinline__ void backward_tile(const GsCompositeArgs &a, const int tile, const int part, const int nparts, float4 *sp, float *syhi, uint32_t *sid,
                                              uint32_t *sstrip, float *red, const float nbig, const float *snap_in = nullptr, const uint32_t seg_start = 0,
                                              const uint32_t seg_end = 0xFFFFFFFFu, const bool seg_tile = false) {
    const int lane = threadIdx.x;
    const int px = (tile % a.gx) * GS_TILE + (lane & 15) + 1;
    const int py0 = (tile / a.gx) * GS_TILE + (lane >> 4) + 1;
    const float fx = (float)px;
    const size_t plane = (size_t)a.W * a.H;
    const int ty0 = py0 - (lane >> 4);
    unsigned long long clk0 = 0;
    if (CLK) clk0 = __builtin_amdgcn_s_memrealtime();
    // transposed reduction: lane (c, s) = (rl >> 2, rl & 3) sums quarter s of component c; lanes >= 36 mirror lanes 0..27
    // (same addresses: broadcasts, no bank conflicts), their sums are not used
    const int rl = lane < 36 ? lane : lane - 36;
    const float *rrow = red + (rl >> 2) * RS + (rl & 3) * 16;
    float *wrow = red + lane;
    const int ocomp_t = (lane < 36 && (lane & 3) == 0) ? ((lane >> 2) < 8 ? (lane >> 2) : 9) : -1;

    float dCr[4], dCg[4], dCb[4], T[4], S[4], fy[4];
    bool dead[4];
    uint32_t walked = 0, evaluated = 0;
    const uint32_t own = EARLY ? strips_of_part(nparts, part) : 0xFu;     // several waves per tile (frames with the early-out): the strips this wave differentiates
    // (Packing the live pixels into one slot once 64 or fewer are left -- what the forward does -- was built for this kernel too and
    // measured equal to slower, same box: the per-splat reduction, which packing does not shorten, is too large a share of an entry, and
    // the packed loop cost the kernel 18 spilled registers: profiles/r04g_ab_backward_packing.log.)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int py = py0 + 4 * p;
        const bool in = (px <= a.W && py <= a.H) && ((own >> p) & 1u);
        const size_t o = in ? (size_t)(px - 1) + (size_t)a.W * (py - 1) : 0;
        fy[p] = (float)py;
        dCr[p] = in ? a.dC[o] : 0.0f;
        dCg[p] = in ? a.dC[o + plane] : 0.0f;
        dCb[p] = in ? a.dC[o + 2 * plane] : 0.0f;
        T[p] = in ? 1.0f : 0.0f;
        S[p] = in ? (a.image[o] * dCr[p] + a.image[o + plane] * dCg[p] + a.image[o + 2 * plane] * dCb[p]) : 0.0f;
        dead[p] = !in;
        if (snap_in && in) {                                               // a later segment of a heavy tile's list: the forward's state at its start
            const int idx = (lane >> 4) * GS_TILE + 64 * p + (lane & 15);   // pixel (x, y) of the tile: 16 (y - ty0) + (x - tx0)
            const float tin = snap_in[768 + idx];
            if (tin != tin) { T[p] = 0.0f; S[p] = 0.0f; dead[p] = true; }   // frozen before this segment
            else {
                T[p] = tin;
                S[p] = (a.image[o] - snap_in[idx]) * dCr[p] + (a.image[o + plane] - snap_in[256 + idx]) * dCg[p] + (a.image[o + 2 * plane] - snap_in[512 + idx]) * dCb[p];
            }
        }
    }
    // the rectangle of the pixels still taking entries: the no-op test runs against it, exactly as in the forward (same pixels frozen at
    // the same batch boundaries, so the same entries are evaluated)
    int qx0, qx1, qy0, qy1;
    auto live_rect = [&]() {
        if (GS_RECT_FROM_MASKS && GS_LIVE_RECT) {
            LiveRectAcc acc;
#pragma unroll
            for (int p = 0; p < 4; ++p) acc.slot(p, __ballot(!dead[p]));
            acc.rect(px - (lane & 15), ty0, qx0, qx1, qy0, qy1);
            return;
        }
        int lx0 = 1 << 20, lx1 = -1, ly0 = 1 << 20, ly1 = -1;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!dead[p]) { const int x = (int)fx, y = (int)fy[p]; lx0 = min(lx0, x); lx1 = max(lx1, x); ly0 = min(ly0, y); ly1 = max(ly1, y); }
        }
        qx0 = wave_min_i32(lx0); qx1 = wave_max_i32(lx1); qy0 = wave_min_i32(ly0); qy1 = wave_max_i32(ly1);
        if (!GS_LIVE_RECT) { qx0 = (tile % a.gx) * GS_TILE + 1; qx1 = qx0 + GS_TILE - 1; qy0 = ty0; qy1 = ty0 + GS_TILE - 1; }
    };
    qx0 = (tile % a.gx) * GS_TILE + 1; qx1 = min(qx0 + GS_TILE - 1, a.W);   // every pixel of the part's strips inside the image is live
    qy0 = ty0 + 4 * (int)__builtin_ctz(own); qy1 = min(ty0 + 4 * (31 - (int)__builtin_clz(own)) + 3, a.H);
    if (snap_in) live_rect();                                               // (... except those the snapshot says were frozen by then)
    const float4 *pay4 = reinterpret_cast<const float4 *>(a.payload);
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0, n3 = n0;
    uint32_t nid = 0;
    // the sixteen partials a lane reads back of the entry being reduced
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    uint32_t pend_gid = 0;
    auto finish = [&]() {                                                // sums of the entry -> one atomic
        float s = ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w));
        s += ((r2.x + r2.y) + (r2.z + r2.w)) + ((r3.x + r3.y) + (r3.z + r3.w));
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        if (ocomp_t >= 0) add_to_row<DET>(a, pend_gid, ocomp_t, s);
    };
    // The tile's list is the concatenation of its segments, one per binning round of the frame (one segment unless the
    // frame was binned in depth slabs); batches end at multiples of CB of the WHOLE list, as in the forward.
    uint32_t gp = seg_start;                                              // (a multiple of CB)
    uint32_t alive = own;                                                 // strips with a live pixel (refreshed at every batch boundary)
    bool stop = false;
    unsigned long long t_loop = 0, t_stage = 0, t_mark = 0;               // debug clocks (a.tile_clock)
    unsigned long long clk_exec = 0, clk_ideal = 0, clk_alive = 0, clk_pix = 0;   // (see the forward)
    uint32_t clk_live = 0, clk_strips = 0;
    if (CLK) t_mark = __builtin_amdgcn_s_memtime();
    for (int sg = 0; sg < a.nseg && !stop; ++sg) {
    const uint32_t *ids = a.seg_ids[sg];
    const uint32_t r0s = a.seg_ranges[sg][2 * tile];
    // capped lists: the list ends where gs_bin -- or the forward, if it had to go further -- stopped writing it (>= what the forward walked)
    const uint32_t r1s = a.tile_ext ? r0s + a.tile_ext[tile].x : a.seg_ranges[sg][2 * tile + 1];
    // a heavy tile's list segment (else the whole list): [s0, s1)
    const uint32_t s0 = min(r1s, r0s + seg_start), s1 = seg_end < r1s - r0s ? r0s + seg_end : r1s;
    uint32_t pos = s0 + lane;
    uint32_t id2 = 0;                                                     // ids run two batches ahead, payload rows one (see the forward)
    {
        const uint32_t pos2 = pos + min((uint32_t)CB - (gp & (CB - 1)), s1 - s0);
        if (pos < s1) { nid = ids[pos]; n0 = pay4[4 * (size_t)nid]; n1 = pay4[4 * (size_t)nid + 1]; n2 = pay4[4 * (size_t)nid + 2]; n3 = pay4[4 * (size_t)nid + 3]; }
        if (pos2 < s1) id2 = ids[pos2];
    }
    for (uint32_t base = s0; base < s1;) {
        const uint32_t phase = gp & (CB - 1);
        const int cnt = (int)min((uint32_t)CB - phase, s1 - base);
        if (EARLY && phase == 0) {
            bool live = false, froze = false;
            alive = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (!dead[p] && T[p] < a.t_min) { dead[p] = true; T[p] = 0.0f; S[p] = 0.0f; froze = true; }
                live = live || !dead[p];
                if (__ballot(!dead[p]) != 0ull) alive |= 1u << p;           // strip p still has a pixel that takes entries
            }
            if (alive == 0u) { stop = true; break; }
            if (__ballot(froze) != 0ull) live_rect();
        }
        if (CLK && (phase == 0 || gp == 0)) {
            clk_live = 0; clk_strips = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) { const uint64_t m = __ballot(!dead[p]); clk_live += (uint32_t)__popcll(m); clk_strips += m ? 1u : 0u; }
        }
        uint32_t strips;
        bool keep;
        stage_record<CULL>(n0, n1, n3, qx0, qx1, qy0, qy1, ty0, keep, strips);
        if (CULL && EARLY) strips &= alive;                                 // a strip of frozen pixels (T = S = 0) adds exact zeros
        int slot = lane, nk = cnt;
        if (!CULL) keep = true;
        uint64_t mq[4] = {~0ull, ~0ull, ~0ull, ~0ull};                    // bit k: strip p of the k-th staged entry is live
        if (CULL) {
            keep = keep && lane < cnt;
            const uint64_t m = __ballot(keep);
            slot = slot_of(m); nk = __popcll(m);
        }
        __syncthreads();
        if (keep) {
            sp[3 * slot] = n0; sp[3 * slot + 1] = n1; sp[3 * slot + 2] = n2;
            syhi[slot] = n3.x;
            sid[slot] = nid;                                             // gaussian id of the staged entry
            if (CULL) sstrip[slot] = strips;
        }
        __syncthreads();
        if (CULL) {
            const uint32_t mine = lane < nk ? sstrip[lane] : 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p) mq[p] = __ballot((mine >> p) & 1u);
        }
        auto live_of = [&](int k) -> uint32_t {                          // wave-uniform strip mask of staged entry k
            return (uint32_t)((mq[0] >> k) & 1ull) | ((uint32_t)((mq[1] >> k) & 1ull) << 1) | ((uint32_t)((mq[2] >> k) & 1ull) << 2)
                   | ((uint32_t)((mq[3] >> k) & 1ull) << 3);
        };
        base += (uint32_t)cnt; gp += (uint32_t)cnt;
        pos = base + lane;
        if (pos < s1) { nid = id2; n0 = pay4[4 * (size_t)nid]; n1 = pay4[4 * (size_t)nid + 1]; n2 = pay4[4 * (size_t)nid + 2]; n3 = pay4[4 * (size_t)nid + 3]; }
        {
            const uint32_t pos2 = pos + min((uint32_t)CB, s1 - base);
            if (base < s1 && pos2 < s1) id2 = ids[pos2];
        }
        if (CLK) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_stage += t - t_mark; t_mark = t; }
        int k = 0;
        if (PAIR) {
            float *wrow2 = wrow + RED_FLOATS;
            const float4 *rr = reinterpret_cast<const float4 *>(rrow), *rr2 = reinterpret_cast<const float4 *>(rrow + RED_FLOATS);
            for (; k + 1 < nk; k += 2) {
                const Entry ea = load_entry(sp, syhi, k), eb = load_entry(sp, syhi, k + 1);
                const uint32_t gida = (uint32_t)__builtin_amdgcn_readfirstlane((int)sid[k]), gidb = (uint32_t)__builtin_amdgcn_readfirstlane((int)sid[k + 1]);
                float va[9], vb[9];
                bool anya = true, anyb = true;
                backward_entry(ea, fx, fy, nbig, dCr, dCg, dCb, T, S, va, anya, live_of(k));
                backward_entry(eb, fx, fy, nbig, dCr, dCg, dCb, T, S, vb, anyb, live_of(k + 1));
#pragma unroll
                for (int c = 0; c < 9; ++c) { wrow[c * RS] = va[c]; wrow2[c * RS] = vb[c]; }
                const float4 a0 = rr[0], a1 = rr[1], a2 = rr[2], a3 = rr[3], b0 = rr2[0], b1 = rr2[1], b2 = rr2[2], b3 = rr2[3];
                r0 = a0; r1 = a1; r2 = a2; r3 = a3; pend_gid = gida; finish();
                r0 = b0; r1 = b1; r2 = b2; r3 = b3; pend_gid = gidb; finish();
            }
        }
        for (; k < nk; ++k) {
            const Entry e = load_entry(sp, syhi, k);
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readfirstlane((int)sid[k]);
            float v[9];
            bool any = true;
            backward_entry(e, fx, fy, nbig, dCr, dCg, dCb, T, S, v, any, live_of(k));
            if (!CULL && __ballot(any) == 0ull) continue;                // nobody in the tile touched it
#pragma unroll
            for (int c = 0; c < 9; ++c) wrow[c * RS] = v[c];             // LDS serves one wave's accesses in order: the reads below see these
            const float4 *rr = reinterpret_cast<const float4 *>(rrow);
            r0 = rr[0]; r1 = rr[1]; r2 = rr[2]; r3 = rr[3];
            pend_gid = gid;
            finish();
        }
        walked += (uint32_t)cnt; evaluated += (uint32_t)nk;
        first_use_here(n0, n1, n2, n3);
        if (CLK) {
            const unsigned long long t = __builtin_amdgcn_s_memtime(); t_loop += t - t_mark; t_mark = t;
            const uint64_t below = nk >= 64 ? ~0ull : ((1ull << nk) - 1ull);
            clk_exec += CULL ? (unsigned long long)(__popcll(mq[0] & below) + __popcll(mq[1] & below) + __popcll(mq[2] & below) + __popcll(mq[3] & below)) : 4ull * (uint32_t)nk;
            clk_ideal += (unsigned long long)nk * ((clk_live + 63u) >> 6);
            clk_alive += (unsigned long long)nk * clk_strips; clk_pix += (unsigned long long)nk * clk_live;
        }
    }
    }
    if (lane == 0 && part == 0) {                                         // (tile_parts > 1: the counters of a tile are those of its first part)
        if (a.walked) { atomicAdd(a.walked, (unsigned long long)walked); atomicAdd(a.walked + 1, (unsigned long long)evaluated); }
        if (seg_tile) {                                                   // a heavy tile's segments add up (the forward zeroed the words)
            if (a.tile_walked) atomicAdd(a.tile_walked + tile, walked);
            if (a.tile_work) atomicAdd(a.tile_work + tile, evaluated);
        } else {
            if (a.tile_walked) a.tile_walked[tile] = walked;
            if (a.tile_work) a.tile_work[tile] = evaluated;
        }
    }
    if (CLK && a.tile_clock && lane == 0) {
        unsigned long long *c = a.tile_clock + GS_TILE_CLOCK_WORDS * (size_t)(a.clock_by_block ? (int)blockIdx.x : tile);
        c[0] = clk0; c[1] = __builtin_amdgcn_s_memrealtime(); c[2] = wave_hw_id();
        c[3] = ((unsigned long long)walked << 32) | evaluated;
        c[4] = t_loop; c[5] = t_stage;
        c[6] = (clk_exec << 32) | (clk_ideal & 0xFFFFFFFFull); c[7] = (clk_alive << 32) | (clk_pix & 0xFFFFFFFFull);
    }
}

template <bool EARLY, int MINW, bool DET, bool CULL, bool CLK = false, bool PAIR = false>
__global__ __launch_bounds__(64, MINW) void composite_bwd_kernel(GsCompositeArgs a) {
    __shared__ float4 sp[CB * 3];
    __shared__ float syhi[CB];
    __shared__ uint32_t sid[CB];
    __shared__ uint32_t sstrip[CB];
    __shared__ __attribute__((aligned(16))) float red[PAIR ? 2 * RED_FLOATS : RED_FLOATS];
    const int ntiles = a.gx * a.gy;
    const float nbig = vgpr_const(-GS_BIG);
    int part = 0, nparts = 1, tile;
    const float *snap_in = nullptr;
    uint32_t seg_start = 0, seg_end = 0xFFFFFFFFu;
    bool seg_tile = false;
    const int segunits = (EARLY && a.snap && !a.seg_hist) ? 8 * (a.front / 24) * (GS_SEG_MAX - 1) : 0;   // workgroups in front of the order: segments 1 .. of the heavy tiles
    if (EARLY && a.snap && a.seg_hist) {                                      // small grid: block b = unit b / len of tile b % len; unit = segment x parts + pixel part
        const int len = ((ntiles + 7) / 8) * 8, unit = (int)blockIdx.x / len;
        tile = (int)blockIdx.x - unit * len;
        nparts = a.parts > 1 ? a.parts : 1; part = unit % nparts;
        const int k = unit / nparts;
        if (tile >= ntiles || k >= a.seg_n) return;
        const uint32_t sl = gs_seg_len_all(a.seg_hist[tile], a.seg_n);
        if (k > 0) {
            if (sl == 0u) return;                                             // this tile runs as one segment
            snap_in = a.snap + (size_t)tile * (size_t)((a.seg_n - 1) * 4 * 256) + (size_t)(k - 1) * (4 * 256);
            seg_start = (uint32_t)k * sl;
        }
        if (sl) { seg_end = k == a.seg_n - 1 ? 0xFFFFFFFFu : seg_start + sl; seg_tile = true; }
    } else if ((int)blockIdx.x < segunits) {
        const int b = (int)blockIdx.x, x = b & 7, q = b >> 3, pos = q / (GS_SEG_MAX - 1), k = q - pos * (GS_SEG_MAX - 1) + 1, slot = 8 * pos + x;
        const uint32_t sl = a.seg_len[slot], t = a.tile_order[a.front + 8 * pos + x];
        if (sl == 0u || t == 0xFFFFFFFFu || (t >> 30) == 0u) return;          // the tile at this position is not split, or walks one segment
        const uint32_t wk = a.snap_walked[slot];
        if ((uint32_t)k * sl >= wk) return;                                   // the forward never got here
        tile = (int)(t & GS_ORDER_TILE_MASK);
        if (tile >= ntiles) return;
        snap_in = a.snap + (size_t)slot * GS_SEG_SNAP_FLOATS + (size_t)(k - 1) * (4 * 256);
        seg_start = (uint32_t)k * sl; seg_end = k == GS_SEG_MAX - 1 ? 0xFFFFFFFFu : seg_start + sl; seg_tile = true;
    } else {
        tile = tile_of_block(a, ntiles, part, nparts, segunits);
        if (tile < 0) return;
        if (segunits && nparts > 1) {                                         // a split tile: with segments its first entry is segment 0, its other parts do nothing
            const int b = (int)blockIdx.x - segunits, slot = (b < a.front ? ((b >> 3) / 3) : ((b - a.front) >> 3)) * 8 + (b & 7);
            const uint32_t sl = a.seg_len[slot];
            if (sl) {
                if (part) return;
                nparts = 1; seg_end = sl; seg_tile = true;
            }
        }
    }
    backward_tile<EARLY, DET, CULL, CLK, PAIR>(a, tile, part, nparts, sp, syhi, sid, sstrip, red, nbig, snap_in, seg_start, seg_end, seg_tile);
}

// Longest-first order for a PLAIN launch (gs_config.schedule 3 / 4).  The dispatcher hands workgroups out in blockIdx order,
// round-robin over the XCDs: block b runs on XCD b % 8.  order[8 j + x] is therefore the j-th tile of XCD x, and the kernel
// decides two things.
// (1) WHICH tiles an XCD gets.  Round 2 / early round 3 kept what the plain tile order has (XCD = tile % 8): neighbours in x on
//     eight different XCDs, so a gaussian's payload row was fetched by nearly every XCD whose L2 it then filled.  Now the tiles are
//     dealt in GROUPS of 8 x 8 (lpt_group_side): the groups are ranked by their work and dealt to the XCDs in snake order (rank k ->
//     XCD k % 8, every second round reversed), which balances the XCDs' work to better than 1 % on the synthetic scenes
//     (tools/xcd_order.py) and gives every XCD spatially compact sets of tiles.  Measured at C3 (rocprofv3 FETCH_SIZE, per launch):
//     forward 338 -> 202 MB, backward 378 -> 225 MB requested from the fabric -- backward traffic 1.02 x its algorithmic bytes.
// (2) IN WHICH ORDER an XCD starts its tiles: heavier first, so the workgroups that start last are the lightest ones and the
//     kernel ends without a tail (C3 backward round 2: 0.84 -> 0.74 ms).  A STABLE counting sort on NB work classes (work / max in
//     NB steps); inside a class the tiles stay in (group, row, column) order, so neighbours still start together.  Only the START
//     order of the light tail matters for the balance -- with 5 waves per SIMD the first 5120 of C3's 8160 tiles start at once
//     whatever their order -- so a few classes lose nothing.
// Ranks come from an LDS bitmap [XCD][class][slot]: slot = 64 * (ordinal of the tile's group on its XCD) + position in the group;
// atomic OR (order free), word prefix, popcount below the own bit -- the construction of the level-1 binning (gs_bin3.hip).
// An XCD's list may be shorter than the longest one: the holes of `order` hold GS_LPT_NONE and their workgroups exit at once;
// order has gs_lpt_order_len(gx, gy) entries and that is the launch's grid.  One workgroup; work = src[t], or the list length
// (ranges_mode).
#ifndef GS_LPT_BUCKETS
#define GS_LPT_BUCKETS 32
#endif
#define GS_LPT_NONE 0xFFFFFFFFu
// group side: 8 tiles (a gaussian of the C3 scene covers 5-6 tiles across), smaller on small grids so that every XCD still gets
// sixteen or more groups to balance with (config C2, 50 x 50 tiles: 4; a 16 x 16 grid: single tiles, i.e. longest-first over the XCDs)
static inline int lpt_groups(int gx, int gy, int gs) { return ((gx + gs - 1) / gs) * ((gy + gs - 1) / gs); }
static inline int lpt_group_side(int gx, int gy) {
    for (int gs = 8; gs > 1; gs >>= 1) if (lpt_groups(gx, gy, gs) >= 128) return gs;
    return 1;
}
int gs_lpt_order_len(int gx, int gy) { const int gs = lpt_group_side(gx, gy); return 8 * gs * gs * ((lpt_groups(gx, gy, gs) + 7) / 8); }

// (3) SPLIT TILES (round 5; front > 0).  A trained scene's work is heavy-tailed: a tile whose list is ten times the median is one wave64
//     running alone long after the rest of the chip has drained (measured on the clustered scene of tools/clustered_probe.py: 55 % of the
//     SIMD time idle).  A tile whose work is at least (sum of all work) / split_div -- about what a wave slot gets when the work is spread
//     evenly -- is composited by two waves, from twice that by four, each owning two or one of the tile's four 16 x 4 pixel strips (the
//     machinery of gs_config.tile_parts).  Its first part keeps the tile's place in the order (entry = tile | log2(parts) << 30); the other
//     parts go to the FRONT region order[0 .. front) -- they are among the heaviest units of the launch and must start first.  Only the
//     front / 24 heaviest tiles of every XCD's list are eligible (the heaviest must never be the one left whole): the tile at position
//     pos of XCD x owns the entries order[8 (3 pos + j) + x], j = 0 .. 2.  The ordinary entries follow from order[front] on; unused
//     entries of the front region hold GS_LPT_NONE (an empty workgroup).  front = 0: no tile is split.
__global__ __launch_bounds__(1024) void tile_lpt_order_kernel(const uint32_t *__restrict__ src, int ranges_mode, int ntiles, int gx, int ng, int gs, int nb,
                                                               uint32_t *__restrict__ order, unsigned long long *__restrict__ zero14, int front, int split_div,
                                                               const uint32_t *__restrict__ walked, uint32_t *__restrict__ zero_words,
                                                               uint32_t *__restrict__ host_nsplit) {
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t wmax, wtot, nsplit;
    __shared__ uint32_t rowtot[8 * 32], rowstart[8 * 32], xcount[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int GT = gs * gs;                                              // tiles of a full group (gs = 8, 4, 2 or 1: a power of two)
    const int ordmax = (ng + 7) >> 3;                                    // groups per XCD (upper bound)
    const int per = ordmax * GT;                                         // slots per XCD
    const int W = (per + 31) >> 5;                                       // bitmap words per row
    const int rows = 8 * nb;                                             // row = XCD * nb + class
    const int ngx = (gx + gs - 1) / gs;
    uint32_t *bm = lds;                                                  // [rows][W]
    uint32_t *gsum = bm + rows * W;                                      // [ng] work of the group
    uint16_t *pre = reinterpret_cast<uint16_t *>(gsum + ng);             // [rows][W] set bits of the row below word w
    uint16_t *ginfo = pre + rows * W;                                    // [ng] XCD | ordinal << 3
    uint8_t *cls = reinterpret_cast<uint8_t *>(ginfo + ng + (ng & 1) + ((rows * W) & 1));   // [ntiles]
    for (int i = tid; i < rows * W + ng; i += 1024) bm[i] = 0;          // bitmap and group sums
    for (int i = tid; i < front; i += 1024) order[i] = GS_LPT_NONE;     // the front region: filled by the split tiles at the very end
    uint32_t *seg_len = order + front + 8 * per;                        // [front / 3] list entries per backward segment of the split tiles (0: none)
    if (front > 0) for (int i = tid; i < front / 3; i += 1024) { seg_len[i] = 0; if (zero_words) zero_words[i] = 0; }
    if (tid == 0) { wmax = 1; nsplit = 0; }
    if (zero14 && tid < 14) zero14[tid] = 0ull;
    __syncthreads();
    // the tiles' work is read twice (maximum + group sums, then classes), eight independent loads in flight per thread each time:
    // the second read comes from L2, and a copy in LDS would not fit beside the bitmap for 4K-class grids
    auto load8 = [&](int t0, uint32_t (&v)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * 1024 + tid;
            v[k] = 0;
            if (t < ntiles) v[k] = ranges_mode ? src[2 * t + 1] - src[2 * t] : src[t];
        }
    };
    auto group_of = [&](int t, int &local) {
        const int ty = t / gx, tx = t - ty * gx;
        local = (ty & (gs - 1)) * gs + (tx & (gs - 1));
        return (ty / gs) * ngx + tx / gs;
    };
    uint32_t m = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 8 * 1024) {
        uint32_t v[8];
        load8(t0, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * 1024 + tid;
            m = max(m, v[k]);
            if (t < ntiles && v[k]) { int local; atomicAdd(&gsum[group_of(t, local)], v[k]); }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if (lane == 0) atomicMax(&wmax, m);
    __syncthreads();
    if (wv == 0) {                                                       // the frame's total work (the group sums are complete)
        uint32_t sw = 0;
        for (int g = lane; g < ng; g += 64) sw += gsum[g];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sw += (uint32_t)__shfl_xor((int)sw, d);
        if (lane == 0) wtot = sw;
    }
    // rank of every group by work (ties by index), dealt in snake order: round r = rank / 8 gives each XCD one group
    for (int g = tid; g < ng; g += 1024) {
        const uint32_t mine = gsum[g];
        int rank = 0;
        for (int h = 0; h < ng; ++h) { const uint32_t o = gsum[h]; rank += (o > mine || (o == mine && h < g)) ? 1 : 0; }
        const int r = rank >> 3, x = (r & 1) ? 7 - (rank & 7) : (rank & 7);
        ginfo[g] = (uint16_t)(x | (r << 3));
    }
    __syncthreads();
    const float scale = (float)nb / (float)wmax;
    for (int t0 = 0; t0 < ntiles; t0 += 8 * 1024) {
        uint32_t v[8];
        load8(t0, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * 1024 + tid;
            if (t < ntiles) {
                const int c = min(nb - 1, max(0, nb - 1 - (int)((float)v[k] * scale)));        // class 0 = heaviest
                cls[t] = (uint8_t)c;
                int local;
                const uint32_t gi = ginfo[group_of(t, local)];
                const int i = (int)(gi >> 3) * GT + local;
                atomicOr(&bm[((int)(gi & 7u) * nb + c) * W + (i >> 5)], 1u << (i & 31));
            }
        }
    }
    __syncthreads();
    for (int r = wv; r < rows; r += 16) {                                // one wave per row: exclusive scan of the word popcounts
        uint32_t carry = 0;
        for (int w0 = 0; w0 < W; w0 += 64) {
            const int w = w0 + lane;
            const uint32_t c = w < W ? (uint32_t)__popc(bm[r * W + w]) : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
            if (w < W) pre[r * W + w] = (uint16_t)(carry + incl - c);
            carry += (uint32_t)__shfl((int)incl, 63);
        }
        if (lane == 0) rowtot[r] = carry;
    }
    __syncthreads();
    if (tid < 8) {                                                       // class starts inside each XCD's list, and its length
        uint32_t run = 0;
        for (int c = 0; c < nb; ++c) { rowstart[tid * nb + c] = run; run += rowtot[tid * nb + c]; }
        xcount[tid] = run;
    }
    __syncthreads();
    uint32_t *main_order = order + front;
    for (int i = tid; i < 8 * per; i += 1024)                            // holes behind the shorter lists
        if ((uint32_t)(i >> 3) >= xcount[i & 7]) main_order[i] = GS_LPT_NONE;
    const uint32_t thr = front > 0 ? max(wtot / (uint32_t)max(split_div, 1), 256u) : 0xFFFFFFFFu, eligible = (uint32_t)front / 24u;
    for (int t = tid; t < ntiles; t += 1024) {
        int local;
        const uint32_t gi = ginfo[group_of(t, local)];
        const int x = (int)(gi & 7u), r = x * nb + cls[t], i = (int)(gi >> 3) * GT + local;
        const uint32_t pos = rowstart[r] + pre[r * W + (i >> 5)] + (uint32_t)__popc(bm[r * W + (i >> 5)] & ((1u << (i & 31)) - 1u));
        uint32_t entry = (uint32_t)t;
        if (pos < eligible) {                                            // among the heaviest of its XCD (front = 0: nobody)
            const uint32_t w = ranges_mode ? src[2 * t + 1] - src[2 * t] : src[t];
            if (w >= thr) {
                const uint32_t pc = w >= 2u * thr ? 2u : 1u, extra = (1u << pc) - 1u;
                for (uint32_t j = 0; j < extra; ++j) order[8u * (3u * pos + j) + (uint32_t)x] = (uint32_t)t | ((j + 1u) << 28) | (pc << 30);
                entry |= pc << 30;
                atomicAdd(&nsplit, 1u);
                if (walked) {                                            // the backward's list segments: GS_SEG_MAX at most, none shorter than GS_SEG_MIN_LEN
                    const uint32_t wk = walked[t], per_seg = (wk + GS_SEG_MAX - 1) / GS_SEG_MAX;
                    const uint32_t sl = max((per_seg + 63u) & ~63u, (uint32_t)GS_SEG_MIN_LEN);
                    seg_len[8u * pos + (uint32_t)x] = wk > sl ? sl : 0u;   // (a walk of one segment: nothing to split)
                }
            }
        }
        main_order[8u * pos + (uint32_t)x] = entry;
    }
    if (host_nsplit) {                                                   // coherent pinned host word: how many tiles of this order are split (the host picks
        __syncthreads();                                                 // the composite kernels' instantiations by it; it holds 0xFFFFFFFF until this store lands)
        if (tid == 0) *host_nsplit = nsplit;
    }
}

hipError_t gs_launch_tile_lpt_order(const uint32_t *work_or_ranges, int ranges_mode, int gx, int gy, uint32_t *order, hipStream_t s,
                                    unsigned long long *zero14, int buckets, int front, int split_div, const uint32_t *walked, uint32_t *zero_words, uint32_t *host_nsplit) {
    const int ntiles = gx * gy;
    if (ntiles <= 0) return hipSuccess;
    if (ntiles > GS_LPT_MAX_TILES) return hipErrorInvalidValue;          // beyond 8K-class images: the callers keep launch order
    int nb = buckets > 0 && buckets <= 32 ? buckets : GS_LPT_BUCKETS;
    const int gs = lpt_group_side(gx, gy), ng = lpt_groups(gx, gy, gs), W = (gs_lpt_order_len(gx, gy) / 8 + 31) / 32;
    auto lds_of = [&](int b) { return (size_t)8 * b * W * 4 + (size_t)ng * 4 + ((size_t)8 * b * W + ng + 2) * 2 + (size_t)ntiles + 16; };
    while (nb > 2 && lds_of(nb) > 150 * 1024) nb >>= 1;                  // very large grids: fewer work classes
    const size_t lds = lds_of(nb);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tile_lpt_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (front < 0 || (front % 24) || ntiles > (int)GS_ORDER_TILE_MASK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tile_lpt_order_kernel, dim3(1), dim3(1024), lds, s, work_or_ranges, ranges_mode, ntiles, gx, ng, gs, nb, order, zero14, front, split_div, walked, zero_words, host_nsplit);
    return hipGetLastError();
}

// ---------------------------------------------------------------- small helpers and launchers
// the per-tile work counters of a composite launch, summed when somebody asks (gs_get_work_counters, the radix binning paths)
__global__ __launch_bounds__(1024) void sum_tiles_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, int n,
                                                          unsigned long long *__restrict__ out) {
    __shared__ unsigned long long sm[2][16];
    unsigned long long sa = 0, sb = 0;
    for (int i = threadIdx.x; i < n; i += 1024) { if (a) sa += a[i]; if (b) sb += b[i]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sa += __shfl_down(sa, d); sb += __shfl_down(sb, d); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = sa; sm[1][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x < 2) { unsigned long long t = 0; for (int k = 0; k < 16; ++k) t += sm[threadIdx.x][k]; out[threadIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void sum_listed_kernel(const uint2 *__restrict__ ext, int n, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long sm[16];
    unsigned long long sa = 0;
    for (int i = threadIdx.x; i < n; i += 1024) sa += ext[i].x;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sa += __shfl_down(sa, d);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sa;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int k = 0; k < 16; ++k) t += sm[k]; out[0] = t; }
}
hipError_t gs_launch_sum_listed(const uint2 *ext, int n, unsigned long long *out, hipStream_t s) {
    hipLaunchKernelGGL(sum_listed_kernel, dim3(1), dim3(1024), 0, s, ext, n, out);
    return hipGetLastError();
}
hipError_t gs_launch_sum_tiles(const uint32_t *a, const uint32_t *b, int n, unsigned long long *out, hipStream_t s) {
    hipLaunchKernelGGL(sum_tiles_kernel, dim3(1), dim3(1024), 0, s, a, b, n, out);
    return hipGetLastError();
}

__global__ void clock_probe_kernel(unsigned long long *__restrict__ out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    for (int i = 0; i < (1 << 20) && r1 - r0 < 2000ull; ++i) r1 = __builtin_amdgcn_s_memrealtime();      // 20 us, bounded
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
hipError_t gs_launch_clock_probe(unsigned long long *out, hipStream_t s) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, out);
    return hipGetLastError();
}

// workgroups in front of the order's in a backward launch: list segments 1 .. GS_SEG_MAX - 1 of the tiles that may be split (composite_bwd_kernel)
int gs_seg_units(int front) { return 8 * (front / 24) * (GS_SEG_MAX - 1); }
static dim3 composite_grid(const GsCompositeArgs &a, int ntiles, bool bwd = false) {   // (tile_of_block computes the same length of one part)
    const int len = (a.tile_order && a.order_len > 0) ? a.order_len : ((ntiles + 7) / 8) * 8;
    if (bwd && a.snap && a.seg_hist && a.t_min > 0.0f) return dim3((unsigned)(len * a.seg_n * (a.parts > 1 ? a.parts : 1)));   // small grid: segments x pixel parts per tile
    return dim3((unsigned)(len * (a.parts > 1 ? a.parts : 1) + ((bwd && a.snap && a.t_min > 0.0f) ? gs_seg_units(a.front) : 0)));
}

int gs_composite_grid_blocks(const GsCompositeArgs &a, int bwd) { return (int)composite_grid(a, a.gx * a.gy, bwd != 0).x; }   // (debug clocks by workgroup: rows of the record)

// variant (debug launches, gs_debug_time_composite / gs_debug_tile_clock): tens digit 1 = tile order instead of the frame's launch order
static GsCompositeArgs apply_sched_variant(const GsCompositeArgs &a0) {
    GsCompositeArgs a = a0;
    if ((a.variant / 10) % 10 == 1) { a.tile_order = nullptr; a.order_len = 0; a.split_ok = 0; }
    return a;
}

hipError_t gs_launch_composite_fwd(const GsCompositeArgs &a0, hipStream_t s) {
    const GsCompositeArgs a = apply_sched_variant(a0);
    const int ntiles = a.gx * a.gy;
    if (ntiles <= 0) return hipSuccess;
    const dim3 grid = composite_grid(a, ntiles), block(64);
    const bool early = a.t_min > 0.0f;
    if (a.parts > 1 && ((a.parts != 2 && a.parts != 4) || !early || a.tile_ext || a.tile_pos || (a.tile_clock && !a.clock_by_block))) return hipErrorInvalidValue;
    if (a.split_ok && (!early || a.tile_ext || a.tile_pos || a.parts > 1)) return hipErrorInvalidValue;   // split entries: frames with the early-out, full lists, one round
    if (a.snap && !a.seg_hist && (!a.split_ok || !a.seg_len || !a.snap_walked || !a.tile_order || a.front <= 0)) return hipErrorInvalidValue;
    if (a.snap && a.seg_hist && (a.tile_order || a.seg_n < 2 || a.seg_n > GS_SEG_MAX || !early || a.tile_ext || a.tile_pos)) return hipErrorInvalidValue;
    if (a.tile_ext && (gs_bin3_seg() != L2_SEG || !early || !a.cranges || !a.cids || !a.clr || !a.ids_w || a.tile_pos)) return hipErrorInvalidValue;
    if (a.tile_pos) {                                                     // a round of a slab frame (t_min > 0 by construction: plan_rounds)
        if (!early || a.tile_clock) return hipErrorInvalidValue;
        // built for FOUR waves per SIMD: the resume / tile_pos / tile_done / tile_dead state needs ~120 VGPRs, and at five (96) it spilled 12-14 of them
        if (a.cull) hipLaunchKernelGGL((composite_fwd_kernel<true, 4, true, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((composite_fwd_kernel<true, 4, false, false, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.tile_clock) {                                                   // debug clocks: instantiations of their own (alpha_cull on only)
        if (!a.cull) return hipErrorInvalidValue;
        if (early) hipLaunchKernelGGL((composite_fwd_kernel<true, GS_FWD_MINW, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((composite_fwd_kernel<false, GS_FWD_MINW, true, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (early && a.snap) {                                                // the order has split tiles: their waves leave snapshots
        if (a.cull) hipLaunchKernelGGL((composite_fwd_kernel<true, GS_FWD_MINW, true, false, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((composite_fwd_kernel<true, GS_FWD_MINW, false, false, false, true>), grid, block, 0, s, a);
    } else if (early) {
        if (a.cull) hipLaunchKernelGGL((composite_fwd_kernel<true, GS_FWD_MINW, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((composite_fwd_kernel<true, GS_FWD_MINW, false>), grid, block, 0, s, a);
    } else {
        if (a.cull) hipLaunchKernelGGL((composite_fwd_kernel<false, GS_FWD_MINW, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((composite_fwd_kernel<false, GS_FWD_MINW, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t gs_launch_composite_bwd(const GsCompositeArgs &a0, hipStream_t s) {
    const GsCompositeArgs a = apply_sched_variant(a0);
    const int ntiles = a.gx * a.gy;
    if (ntiles <= 0) return hipSuccess;
    const bool early = a.t_min > 0.0f;
    const dim3 grid = composite_grid(a, ntiles, true), block(64);
    if (a.snap && !a.seg_hist && (!a.split_ok || !a.seg_len || !a.snap_walked || !a.tile_order || a.front <= 0)) return hipErrorInvalidValue;
    if (a.snap && a.seg_hist && (a.tile_order || a.seg_n < 2 || a.seg_n > GS_SEG_MAX || !early || a.tile_ext || a.nseg > 1)) return hipErrorInvalidValue;
    if (a.parts > 1 && ((a.parts != 2 && a.parts != 4) || !early || a.tile_ext || a.nseg > 1 || (a.tile_clock && !a.clock_by_block))) return hipErrorInvalidValue;
    if (a.split_ok && (!early || a.tile_ext || a.nseg > 1 || a.parts > 1)) return hipErrorInvalidValue;
#define GS_B2(E, D) do { if (a.cull) hipLaunchKernelGGL((composite_bwd_kernel<E, GS_BWD_MINW, D, true>), grid, block, 0, s, a); \
                         else hipLaunchKernelGGL((composite_bwd_kernel<E, GS_BWD_MINW, D, false>), grid, block, 0, s, a); } while (0)
#define GS_B(E) do { if (a.g2d_fixed) GS_B2(E, true); else GS_B2(E, false); } while (0)
    if (a.tile_clock) {                                                   // debug clocks: instantiations of their own (alpha_cull on, float atomics only)
        if (!a.cull || a.g2d_fixed) return hipErrorInvalidValue;
        if (early) hipLaunchKernelGGL((composite_bwd_kernel<true, 5, false, true, true>), grid, block, 0, s, a);        // (the clocks cost registers: five waves)
        else hipLaunchKernelGGL((composite_bwd_kernel<false, 5, false, true, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    // small grids (several waves per tile: every wave alone on its SIMD): two entries in flight per wave, registers to spare (two waves per SIMD)
    if (GS_BWD_PAIR && early && a.cull && !a.tile_order && (a.parts > 1 || a.seg_hist)) {
        if (a.g2d_fixed) hipLaunchKernelGGL((composite_bwd_kernel<true, 2, true, true, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((composite_bwd_kernel<true, 2, false, true, false, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (early) GS_B(true); else GS_B(false);
#undef GS_B
#undef GS_B2
    return hipGetLastError();
}
