// gs_bin3.hip -- two-level tile binning for gfx950 (default path of gs_bin from round 2 on).
//
// Same result as the reference's compactIdxs (src/forward.jl:118-161: hitBinning, scan!, compactHits) and as the
// radix paths of gs_sort.hip / gs_bin2.hip -- per-tile lists in (tile, list order), bit-identical -- but no pass ever
// sorts the I tile-instances (30 M at 1 M gaussians / 1080p), and nothing depends on the order in which the LDS unit
// serves the lanes of an atomic:
//
//   level 1  the gaussians are listed per SUPER-TILE of 8 x 8 tiles (128 x 128 pixels): 2.2 M coarse instances instead
//            of 30 M.  A workgroup owns G consecutive list positions (depth order) and builds an LDS bitmap
//            [super-tile][position] with atomic ORs (order-free).  The row popcounts are the workgroup's histogram
//            (l1_hist); after an exclusive scan of the table along the workgroups (l1_rowscan) the same bitmap gives
//            every coarse instance its stable rank -- the popcount of its row below its own bit -- and l1_scatter writes
//            the gaussian id and the instance's rectangle clipped to the super-tile (12 bits) to their final place.
//   level 2  a tile's list is the sub-sequence of its super-tile's list whose rectangles cover the tile, in the same
//            order.  A workgroup takes a segment of L2_SEG entries of one super-tile list; each of its eight waves owns
//            one tile ROW of the super-tile: it compacts the segment's entries to those touching its row through a
//            small LDS ring and, per full batch of 64 such entries, one ballot per tile gives the count and every hit
//            lane's rank; the hit lanes store their gaussian id behind the tile's cursor -- ascending positions, so the
//            order is the list order by construction (no key, no digit, no rank table).  The cursor of (segment, tile)
//            starts at the tile's range start + the hits of the earlier segments of the same super-tile, which a count
//            pass (a 9 x 9 difference array per segment) provides.
//
// HBM traffic per instance: the 4-byte id written once (+ 6 B per COARSE instance written and read twice), against
// 16 B per instance for gs_bin2.hip.  The tile ranges are the exclusive scan of the level-2 hit counts.
#include "gs_common.h"

// Super-tile edge: 8 tiles (SBS = 3), or 16 tiles (SBS = 4) on grids whose 8 x 8 super-tiles would be more than GS_BIN3_NS8_MAX:
// level 1 works per (chunk of positions, super-tile) and its cost per position grows with the number of super-tiles (4K: 510 of
// 8 x 8 -- l1_scatter 444 us at C5 -- against 135 of 16 x 16, the count of a 1080p frame), while level 2 only tests twice the rows.
#define GS_BIN3_NS8_MAX 256
#define L1_THREADS 256
#ifndef L2_SEG
#define L2_SEG 2048          // entries of a super-tile list per workgroup
#endif
#define L2_THREADS 256

int gs_bin3_sb_shift(int gx, int gy, int force) {
    if (force == 3 || force == 4) return force;
    return ((gx + 7) / 8) * ((gy + 7) / 8) > GS_BIN3_NS8_MAX ? 4 : 3;
}
int gs_bin3_seg() { return L2_SEG; }
int64_t gs_bin3_max_work(int64_t coarse_instances, int ns) { return coarse_instances / L2_SEG + ns; }
// list positions per level-1 workgroup: the bitmap (ns x G bits), its word prefix (ns x G/32 u16), 2 ns starts and the
// staging buffer (24 G bytes) share LDS
static size_t l1_lds_bytes(int ns, int g) { return (size_t)ns * (g / 8 + g / 16 + 8) + 24 * (size_t)g; }
int gs_bin3_group(int ns) {
    int g = 512;                                          // measured at C3 (135 super-tiles): 1024 -> 67 us, 512 -> 60 us, 256 -> 64 us for level 1
    while (g > 256 && l1_lds_bytes(ns, g) > 72 * 1024) g >>= 1;
    return g;
}
bool gs_bin3_supported(int ns) { return l1_lds_bytes(ns, 256) <= 140 * 1024 && ns < (1 << 16); }     // (l1_scatter packs S << 16 | clipped rectangle)

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < GS_WAVE; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *sm, int nwaves) {
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) v += __shfl_down(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int i = 0; i < nwaves; ++i) t += sm[i];
    return t;
}

// ---------------------------------------------------------------- level 1
struct L1Args {
    const uint16_t *rect;        // fine tile rectangles by gaussian id (1-based inclusive, x0 == 0: none)
    const uint32_t *perm;        // list position -> gaussian id (null: identity)
    const uint8_t *sdone;        // per super-tile: completed in an earlier round, takes no entries (null: none)
    int64_t n, n_slab;           // positions summed for the frame's instance count / positions listed by this round
    int sgx, ns, G, nwg;         // super-tile grid, positions per workgroup, workgroups covering n_slab
    int sbs;                     // log2 of the super-tile edge in tiles (3 or 4)
    uint2 *rect_sorted;          // [n_slab] rectangles in list order (written by l1_hist, read by l1_scatter)
    uint32_t *table;             // [ns][nwg] coarse instances per (super-tile, workgroup) -> exclusive scan along nwg
    uint32_t *row_total;         // [ns]
    uint32_t *partials;          // [3][nwg_all]: per workgroup coarse instances, fine instances of the slab, of all positions
    uint32_t *totals;            // [3] the sums of `partials` (0xFFFFFFFF: does not fit 32 bits)
    uint32_t *cranges;           // [ns][2]
    uint32_t *cids;              // coarse lists: gaussian ids in (super-tile, list order)
    uint16_t *clr;               //               rectangle clipped to the super-tile: lx0 | lx1 << sbs | ly0 << 2 sbs | ly1 << 3 sbs
    int nwg_all;
    uint32_t *tilecnt;           // [ntiles] zeroed here for the level-2 count pass
    int ntiles;
    uint32_t *zero_words;        // 32 words zeroed by l1_scatter (the forward's work counters: saves a memset command); may be null
    uint32_t cap_coarse, cap_fine; // entries the coarse lists (cids, clr) / the tile lists (ids) can hold: a frame whose totals exceed
                                 // them writes no list at all (every range empty) and the host redoes it with larger buffers
    uint32_t *host_totals, *host_walked;   // coherent pinned host memory (may be null): the totals and the two words at walked_src, stored
    const uint32_t *walked_src;            // by l1_rowscan itself -- a copy command in the stream is a blit kernel of 4 us plus its gaps
    const uint32_t *tile_walked;           // != null: host_walked receives the 64-bit SUM of these n_tile_walked per-tile counts (the previous
    int n_tile_walked;                     // forward's walked list entries) instead of the two words at walked_src
};
// The lists are enqueued BEFORE the host has seen the frame's totals (speculative launch, gs_api.hip): every kernel that
// writes them checks the totals against the capacities of the buffers it was given.
__device__ __forceinline__ bool lists_overflow(const uint32_t *totals, uint32_t cap_coarse, uint32_t cap_fine) {
    return totals[0] > cap_coarse || totals[1] > cap_fine;
}

// Chunk of G list positions a level-1 workgroup works on.  Every workgroup touches one table word and one short run of the
// coarse lists PER SUPER-TILE, and the words / runs of consecutive chunks are neighbours in memory -- but workgroup b runs on
// XCD b % 8, whose L2 is its own: with chunk = b a 128-byte line was filled a few bytes at a time from eight L2s (C5, 510
// super-tiles x 9.8 k workgroups: l1_hist fetched 600 MB and wrote 224 MB for 140 MB algorithmic, l1_scatter wrote 422 MB for
// 120 MB).  So XCD x gets the contiguous chunks [x per, (x + 1) per), in launch order: a line is filled, and re-read, inside ONE
// L2.  The grid is 8 per blocks; -1 = no chunk (exit before any barrier).  Which workgroup handles a chunk changes no result.
__device__ __forceinline__ int l1_chunk(const int b, const int nchunks) {
    const int per = (nchunks + 7) >> 3, j = b >> 3, c = (b & 7) * per + j;
    return (j < per && c < nchunks) ? c : -1;
}
static inline unsigned l1_grid(int nchunks) { return 8u * (unsigned)((nchunks + 7) >> 3); }

// One list position per thread (blockDim = G): sets the position's bits in the LDS bitmap bm[ns][G / 32].
template <bool FROM_SORTED>
__device__ __forceinline__ void l1_bitmap(const L1Args &a, uint32_t *bm, int64_t base, uint2 &rr, uint32_t &coarse, uint32_t &fine_slab,
                                          uint32_t &fine_all) {
    const int b = threadIdx.x, wpr = a.G >> 5;
    const int64_t s = base + b;
    coarse = fine_slab = fine_all = 0;
    rr = make_uint2(0u, 0u);
    if (s < a.n) {
        if (FROM_SORTED) { if (s < a.n_slab) rr = a.rect_sorted[s]; }
        else rr = reinterpret_cast<const uint2 *>(a.rect)[a.perm ? (int64_t)a.perm[s] : s];
    }
    const uint32_t x0 = rr.x & 0xFFFFu, x1 = rr.x >> 16, y0 = rr.y & 0xFFFFu, y1 = rr.y >> 16;
    if (x0 == 0u) return;
    const uint32_t area = (x1 - x0 + 1u) * (y1 - y0 + 1u);
    fine_all = area;
    if (s >= a.n_slab) return;
    fine_slab = area;
    const int cx0 = (int)(x0 - 1u) >> a.sbs, cx1 = (int)(x1 - 1u) >> a.sbs, cy0 = (int)(y0 - 1u) >> a.sbs, cy1 = (int)(y1 - 1u) >> a.sbs;
    const uint32_t bit = 1u << (b & 31);
    uint32_t *col = bm + (b >> 5);
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) {
            const int S = cy * a.sgx + cx;
            if (a.sdone && a.sdone[S]) continue;
            atomicOr(&col[S * wpr], bit);
            ++coarse;
        }
}

// row popcounts of the bitmap with all threads: wpr / 4 lanes share a row (four words each), rows_per_pass rows at a time.
// pre (may be null): set bits of the row below each word; rowcnt[S] = set bits of the row
__device__ __forceinline__ void l1_row_counts(const uint32_t *bm, int ns, int wpr, uint16_t *pre, uint32_t *rowcnt) {
    const int tid = threadIdx.x, lpr = wpr >> 2;            // lanes per row: 8 (G = 1024), 4, 2
    const int rows_per_pass = blockDim.x / lpr;
    const int sub = tid % lpr;
    for (int S0 = 0; S0 < ns; S0 += rows_per_pass) {
        const int S = S0 + tid / lpr;
        uint32_t c[4] = {0, 0, 0, 0};
        if (S < ns) {
            const uint4 w4 = *reinterpret_cast<const uint4 *>(bm + S * wpr + sub * 4);
            c[0] = (uint32_t)__popc(w4.x); c[1] = (uint32_t)__popc(w4.y); c[2] = (uint32_t)__popc(w4.z); c[3] = (uint32_t)__popc(w4.w);
        }
        const uint32_t tot = c[0] + c[1] + c[2] + c[3];
        uint32_t incl = tot;
        for (int d = 1; d < lpr; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (sub >= d) incl += u; }
        if (S < ns) {
            uint32_t e = incl - tot;
            if (pre) {
                uint16_t *p = pre + S * wpr + sub * 4;
                p[0] = (uint16_t)e; p[1] = (uint16_t)(e + c[0]); p[2] = (uint16_t)(e + c[0] + c[1]); p[3] = (uint16_t)(e + c[0] + c[1] + c[2]);
            }
            if (sub == lpr - 1) rowcnt[S] = incl;
        }
    }
}

__global__ __launch_bounds__(1024) void l1_hist_kernel(L1Args a) {
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t sm[16];
    uint32_t *bm = lds;
    uint32_t *rowcnt = lds + a.ns * (a.G >> 5);
    const int tid = threadIdx.x, wpr = a.G >> 5, nt = blockDim.x;
    const int chunk = l1_chunk((int)blockIdx.x, a.nwg_all);
    if (chunk < 0) return;
    const int64_t base = (int64_t)chunk * a.G;
    const bool listed = base < a.n_slab;                    // workgroups beyond the slab only add up the frame's instance count
    if (listed) {
        for (int i = tid; i < a.ns * wpr; i += nt) bm[i] = 0;
        __syncthreads();
    }
    uint2 rr;
    uint32_t coarse, fs, fa;
    l1_bitmap<false>(a, bm, base, rr, coarse, fs, fa);
    if (listed) {
        if (base + tid < a.n_slab) a.rect_sorted[base + tid] = rr;
        __syncthreads();
        l1_row_counts(bm, a.ns, wpr, nullptr, rowcnt);
        __syncthreads();
        for (int S = tid; S < a.ns; S += nt) a.table[(size_t)S * a.nwg + chunk] = rowcnt[S];
    }
    coarse = block_sum_u32(coarse, sm, nt >> 6);
    fs = block_sum_u32(fs, sm, nt >> 6);
    fa = block_sum_u32(fa, sm, nt >> 6);
    if (tid == 0) { a.partials[chunk] = coarse; a.partials[a.nwg_all + chunk] = fs; a.partials[2 * (size_t)a.nwg_all + chunk] = fa; }
}

// blocks 0 .. ns-1: exclusive scan of table row S along the workgroups + the row total; blocks ns .. ns+2: the three totals;
// further blocks: zero the per-tile hit counters of the level-2 count pass
__global__ __launch_bounds__(256) void l1_rowscan_kernel(L1Args a) {
    __shared__ uint32_t sm[4];
    __shared__ unsigned long long wide[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if ((int)blockIdx.x >= a.ns + 3) {
        const int i = ((int)blockIdx.x - a.ns - 3) * 256 + tid;
        if (i < a.ntiles) a.tilecnt[i] = 0;
        return;
    }
    if ((int)blockIdx.x >= a.ns) {
        const int which = (int)blockIdx.x - a.ns;
        const uint32_t *p = a.partials + (size_t)which * a.nwg_all;
        unsigned long long t = 0, wk = 0;
        for (int i = tid; i < a.nwg_all; i += 256) t += p[i];
        if (which == 0 && a.host_walked && a.tile_walked) {                 // 16-byte loads, four in flight per thread: this block must not
            const uint4 *v4 = reinterpret_cast<const uint4 *>(a.tile_walked);  // become the kernel's critical path on 4K-class grids
            const int n4 = a.n_tile_walked >> 2;
#pragma unroll 4
            for (int i = tid; i < n4; i += 256) { const uint4 v = v4[i]; wk += (unsigned long long)v.x + v.y + v.z + v.w; }
            for (int i = 4 * n4 + tid; i < a.n_tile_walked; i += 256) wk += a.tile_walked[i];
        }
#pragma unroll
        for (int d = GS_WAVE / 2; d > 0; d >>= 1) { t += __shfl_down(t, d); wk += __shfl_down(wk, d); }
        if (lane == 0) { wide[wv] = t; wide[4 + wv] = wk; }
        __syncthreads();
        if (tid == 0) {
            const unsigned long long tot = wide[0] + wide[1] + wide[2] + wide[3];
            const uint32_t t32 = tot >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;
            a.totals[which] = t32;
            if (a.host_totals) {
                a.host_totals[which] = t32;
                if (which == 0 && a.host_walked) {
                    if (a.tile_walked) {
                        const unsigned long long w = wide[4] + wide[5] + wide[6] + wide[7];
                        a.host_walked[0] = (uint32_t)w; a.host_walked[1] = (uint32_t)(w >> 32);
                    } else { a.host_walked[0] = a.walked_src[0]; a.host_walked[1] = a.walked_src[1]; }
                }
            }
        }
        return;
    }
    uint32_t *row = a.table + (size_t)blockIdx.x * a.nwg;
    uint32_t carry = 0;
    for (int base = 0; base < a.nwg; base += 256) {
        const int i = base + tid;
        const uint32_t v = i < a.nwg ? row[i] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v, lane);
        __syncthreads();
        if (lane == 63) sm[wv] = incl;
        __syncthreads();
        uint32_t woff = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k < wv) woff += sm[k]; all += sm[k]; }
        if (i < a.nwg) row[i] = carry + woff + incl - v;
        carry += all;
    }
    if (tid == 0) a.row_total[blockIdx.x] = carry;
}

// staging capacity of l1_scatter (coarse instances per workgroup written as runs; the rare surplus is written directly)
#define L1_CAP(G) (3 * (G))
__global__ __launch_bounds__(1024) void l1_scatter_kernel(L1Args a) {
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t sm[32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wpr = a.G >> 5, nt = blockDim.x, nw = nt >> 6;
    const uint32_t cap = (uint32_t)L1_CAP(a.G);
    uint32_t *bm = lds;                                                 // [ns][wpr]
    uint32_t *gstart = bm + a.ns * wpr;                                 // [ns] first position of (S, this workgroup) in the coarse lists
    uint32_t *lstart = gstart + a.ns;                                   // [ns] first staging slot of S
    uint32_t *st_id = lstart + a.ns;                                    // [cap] staged gaussian ids, grouped by super-tile
    uint32_t *st_ls = st_id + cap;                                      // [cap] clipped rectangle | S << 16
    uint16_t *pre = reinterpret_cast<uint16_t *>(st_ls + cap);          // [ns][wpr] set bits of the row below word w
    const int chunk = l1_chunk((int)blockIdx.x, a.nwg);
    const int64_t base = (int64_t)chunk * a.G;
    if (a.zero_words && blockIdx.x == 0 && tid < 32) a.zero_words[tid] = 0;
    if (chunk < 0 || lists_overflow(a.totals, a.cap_coarse, a.cap_fine)) return;     // uniform over the workgroup
    // issued first, used after the bitmap phase: the list-start operands of the first scan pass (all of them when ns <= G)
    const uint32_t v_first = tid < a.ns ? a.row_total[tid] : 0u;
    const uint32_t tb_first = tid < a.ns ? a.table[(size_t)tid * a.nwg + chunk] : 0u;
    for (int i = tid; i < a.ns * wpr; i += nt) bm[i] = 0;
    __syncthreads();
    uint2 rr;
    uint32_t coarse, fs, fa;
    l1_bitmap<true>(a, bm, base, rr, coarse, fs, fa);
    const int64_t s = base + tid;
    const uint32_t gid = (rr.x & 0xFFFFu) && s < a.n_slab ? (a.perm ? a.perm[s] : (uint32_t)s) : 0u;
    __syncthreads();
    l1_row_counts(bm, a.ns, wpr, pre, lstart);
    __syncthreads();
    // two exclusive scans over the super-tiles: the list starts (row totals; every workgroup redoes it: ns words) and the
    // staging starts (this workgroup's row counts)
    uint32_t carry = 0, lcarry = 0;
    for (int b0 = 0; b0 < a.ns; b0 += nt) {
        const int S = b0 + tid;
        const uint32_t lc = S < a.ns ? lstart[S] : 0u;
        uint32_t v = v_first, tb = tb_first;
        if (b0 > 0) { v = S < a.ns ? a.row_total[S] : 0u; tb = S < a.ns ? a.table[(size_t)S * a.nwg + chunk] : 0u; }
        const uint32_t incl = wave_incl_scan_u32(v, lane), lincl = wave_incl_scan_u32(lc, lane);
        __syncthreads();
        if (lane == 63) { sm[wv] = incl; sm[16 + wv] = lincl; }
        __syncthreads();
        uint32_t woff = 0, all = 0, lwoff = 0, lall = 0;
        for (int k = 0; k < nw; ++k) {
            if (k < wv) { woff += sm[k]; lwoff += sm[16 + k]; }
            all += sm[k]; lall += sm[16 + k];
        }
        if (S < a.ns) {
            const uint32_t st = carry + woff + incl - v;
            gstart[S] = st + tb;
            lstart[S] = lcarry + lwoff + lincl - lc;
            if (chunk == 0) { a.cranges[2 * S] = st; a.cranges[2 * S + 1] = st + v; }
        }
        carry += all; lcarry += lall;
    }
    __syncthreads();
    const uint32_t x0 = rr.x & 0xFFFFu;
    if (x0 != 0u && s < a.n_slab) {
        const int fx0 = (int)x0 - 1, fx1 = (int)(rr.x >> 16) - 1, fy0 = (int)(rr.y & 0xFFFFu) - 1, fy1 = (int)(rr.y >> 16) - 1;
        const int sbs = a.sbs, SB = 1 << sbs;
        const int cx0 = fx0 >> sbs, cx1 = fx1 >> sbs, cy0 = fy0 >> sbs, cy1 = fy1 >> sbs;
        const uint32_t below = (1u << (tid & 31)) - 1u;
        const int w = tid >> 5;
        for (int cy = cy0; cy <= cy1; ++cy) {
            const int oy = cy << sbs;
            const uint32_t lry = (uint32_t)(((max(fy0, oy) - oy) << (2 * sbs)) | ((min(fy1, oy + SB - 1) - oy) << (3 * sbs)));
            for (int cx = cx0; cx <= cx1; ++cx) {
                const int S = cy * a.sgx + cx;
                if (a.sdone && a.sdone[S]) continue;
                const uint32_t rank = pre[S * wpr + w] + (uint32_t)__popc(bm[S * wpr + w] & below);
                const int ox = cx << sbs;
                const uint32_t lr = lry | (uint32_t)((max(fx0, ox) - ox) | ((min(fx1, ox + SB - 1) - ox) << sbs));
                const uint32_t lp = lstart[S] + rank;
                if (lp < cap) { st_id[lp] = gid; st_ls[lp] = lr | ((uint32_t)S << 16); }
                else { const uint32_t pos = gstart[S] + rank; a.cids[pos] = gid; a.clr[pos] = (uint16_t)lr; }
            }
        }
    }
    __syncthreads();
    const uint32_t staged = min(lcarry, cap);
    for (uint32_t j = tid; j < staged; j += nt) {                       // runs: consecutive slots of a super-tile are consecutive list positions
        const uint32_t ls = st_ls[j], S = ls >> 16;
        const uint32_t pos = gstart[S] + (j - lstart[S]);
        a.cids[pos] = st_id[j];
        a.clr[pos] = (uint16_t)(ls & 0xFFFFu);
    }
}

// ---------------------------------------------------------------- level 2
// Work item blockIdx.x -> (super-tile S, entries [e0, e1) of its list, w0 = first work item of S).  Every workgroup
// rebuilds the prefix of the per-super-tile segment counts from the coarse ranges (ns <= a few thousand words from L2).
template <int NT>
__device__ bool find_work(const GsBin3Args &a, uint32_t *sh, int &S, uint32_t &e0, uint32_t &e1, uint32_t &w0) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t w = blockIdx.x;
    constexpr int R = NT / GS_WAVE;                        // sh[0 .. R): the waves' sums; sh[R .. R + 4): the work item found
    if (tid == 0) { sh[R] = 0xFFFFFFFFu; sh[R + 1] = 0; }
    uint32_t carry = 0;
    for (int base = 0; base < a.ns; base += NT) {
        const int s = base + tid;
        uint32_t nseg = 0, c0 = 0, c1 = 0;
        if (s < a.ns) { const uint2 cr = reinterpret_cast<const uint2 *>(a.cranges)[s]; c0 = cr.x; c1 = cr.y; nseg = (c1 - c0 + (L2_SEG - 1)) / L2_SEG; }
        const uint32_t incl = wave_incl_scan_u32(nseg, lane);
        __syncthreads();                                   // sh[0 .. R) free again (and the initial sh[R] visible)
        if (lane == 63) sh[wv] = incl;
        __syncthreads();
        uint32_t woff = 0, all = 0;
#pragma unroll
        for (int k = 0; k < NT / GS_WAVE; ++k) { if (k < wv) woff += sh[k]; all += sh[k]; }
        const uint32_t excl = carry + woff + incl - nseg;
        if (nseg && excl <= w && w < excl + nseg) { sh[R] = (uint32_t)s; sh[R + 1] = excl; sh[R + 2] = c0; sh[R + 3] = c1; }
        carry += all;
        if (carry > w) break;                              // uniform: carry is the same in every thread
    }
    __syncthreads();
    if (sh[R] == 0xFFFFFFFFu) return false;
    S = (int)sh[R]; w0 = sh[R + 1];
    const uint32_t c0 = sh[R + 2], c1 = sh[R + 3];
    e0 = c0 + (w - w0) * L2_SEG;
    e1 = min(c1, e0 + L2_SEG);
    return true;
}

// Hits per (segment, local tile): every entry adds the four corners of its clipped rectangle to a 9 x 9 difference array;
// 16 private copies (lane & 15) keep the lanes of one LDS atomic instruction off each other's cells (rectangles that
// cover the whole super-tile all hit the same four corners).
#define CNT_COPIES 16
template <int SBS>
__global__ __launch_bounds__(L2_THREADS) void l2_count_kernel(GsBin3Args a) {
    constexpr int SB = 1 << SBS, CNT_CELLS = (SB + 1) * (SB + 1);
    static_assert(SB * SB <= L2_THREADS, "one thread per local tile");
    __shared__ int diff[CNT_COPIES * CNT_CELLS];
    __shared__ uint32_t sh[12];
    int S; uint32_t e0, e1, w0;
    if (lists_overflow(a.totals, a.cap_coarse, a.cap_fine)) return;
    if (!find_work<L2_THREADS>(a, sh, S, e0, e1, w0)) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < CNT_COPIES * CNT_CELLS; i += L2_THREADS) diff[i] = 0;
    __syncthreads();
    int *my = diff + (tid & (CNT_COPIES - 1)) * CNT_CELLS;
    for (uint32_t e = e0 + tid; e < e1; e += L2_THREADS) {
        const uint32_t lr = a.clr[e];
        const int lx0 = lr & (SB - 1), lx1 = (lr >> SBS) & (SB - 1), ly0 = (lr >> (2 * SBS)) & (SB - 1), ly1 = (lr >> (3 * SBS)) & (SB - 1);
        atomicAdd(&my[ly0 * (SB + 1) + lx0], 1);
        atomicAdd(&my[ly0 * (SB + 1) + lx1 + 1], -1);
        atomicAdd(&my[(ly1 + 1) * (SB + 1) + lx0], -1);
        atomicAdd(&my[(ly1 + 1) * (SB + 1) + lx1 + 1], 1);
    }
    __syncthreads();
    for (int cell = tid; cell < CNT_CELLS; cell += L2_THREADS) {
        int s = 0;
#pragma unroll
        for (int c = 0; c < CNT_COPIES; ++c) s += diff[c * CNT_CELLS + cell];
        diff[cell] = s;                                    // copy 0 now holds the sum (each cell is read and written by its own thread)
    }
    __syncthreads();
    if (tid < SB * SB) {
        const int ty = tid >> SBS, tx = tid & (SB - 1);
        int s = 0;
        for (int y = 0; y <= ty; ++y)
            for (int x = 0; x <= tx; ++x) s += diff[y * (SB + 1) + x];
        a.segcnt[(size_t)blockIdx.x * (SB * SB) + tid] = (uint32_t)s;
        const int gtx = (S % a.sgx) * SB + tx, gty = (S / a.sgx) * SB + ty;
        if (s && gtx < a.gx && gty < a.gy) atomicAdd(&a.tilecnt[gty * a.gx + gtx], (uint32_t)s);   // the tile's list length (integer adds commute)
    }
}

// Capped lists (GsBin3Args.cap_src): one wave per super-tile, lane = local tile.  A tile takes the entries of the segments of its
// super-tile's list in order until it holds gs_list_cap(walked by the slot's previous forward) of them; tile_nopen = how many
// segments that is, tile_ext = {entries those segments give it = the written LENGTH of its list, coarse index of the first segment it
// does NOT take or GS_CONT_NONE when it takes them all}, smax = the largest tile_nopen of the super-tile.  The lists themselves
// (positions, order, ranges) are those of the uncapped path: only fewer of their entries are written.
template <int SBS>
__global__ __launch_bounds__(GS_WAVE) void l2_cap_kernel(GsBin3Args a) {
    constexpr int SB = 1 << SBS;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.ext_count) *a.ext_count = 0u;      // (also when nothing is listed: gs_get_list_stats reads it)
    if (lists_overflow(a.totals, a.cap_coarse, a.cap_fine)) return;
    const int S = blockIdx.x, lane = threadIdx.x;
    uint32_t w0 = 0;                                           // first work item of S: segments of the super-tiles before it
    for (int b = 0; b < S; b += GS_WAVE) {
        const int s = b + lane;
        if (s < S) { const uint2 cr = reinterpret_cast<const uint2 *>(a.cranges)[s]; w0 += (cr.y - cr.x + (L2_SEG - 1)) / L2_SEG; }
    }
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) w0 += (uint32_t)__shfl_xor((int)w0, d);
    const uint2 cr = reinterpret_cast<const uint2 *>(a.cranges)[S];
    const uint32_t nseg = (cr.y - cr.x + (L2_SEG - 1)) / L2_SEG;
    uint32_t m = 0;
    for (int lt = lane; lt < SB * SB; lt += GS_WAVE) {                  // local tiles: one per lane (8 x 8) or four (16 x 16)
        const int tx = (S % a.sgx) * SB + (lt & (SB - 1)), ty = (S / a.sgx) * SB + (lt >> SBS);
        const bool in = tx < a.gx && ty < a.gy;
        const int t = ty * a.gx + tx;
        const uint32_t cap = in ? gs_list_cap(a.cap_src[t]) : 0u;
        uint32_t run = 0, k = 0;
        for (; k < nseg && run < cap; ++k) run += a.segcnt[(size_t)(w0 + k) * (SB * SB) + lt];       // 256 coalesced bytes per segment and wave
        m = max(m, in ? k : 0u);
        if (in) {
            a.tile_nopen[t] = k;
            a.tile_ext[t] = make_uint2(run, k < nseg ? cr.x + k * L2_SEG : GS_CONT_NONE);
        }
    }
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if (lane == 0) a.smax[S] = m;
}

// tile ranges = exclusive scan of the tile counts in tile order (one workgroup); completed tiles get an empty range
__global__ __launch_bounds__(1024) void l2_ranges_kernel(const uint32_t *__restrict__ tilecnt, int ntiles, const uint8_t *__restrict__ done,
                                                          uint32_t *__restrict__ ranges, const uint32_t *__restrict__ totals, uint32_t cap_coarse,
                                                          uint32_t cap_fine, uint2 *__restrict__ ext) {      // ext (capped lists): reset when nothing is listed
    __shared__ uint32_t sm[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (lists_overflow(totals, cap_coarse, cap_fine)) {                  // nothing was listed: every range empty
        for (int t = tid; t < ntiles; t += 1024) { ranges[2 * t] = 0; ranges[2 * t + 1] = 0; if (ext) ext[t] = make_uint2(0u, GS_CONT_NONE); }
        return;
    }
    uint32_t carry = 0;
    for (int b0 = 0; b0 < ntiles; b0 += 1024 * 4) {
        uint32_t v[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = b0 + tid * 4 + k;
            v[k] = t < ntiles && !(done && done[t]) ? tilecnt[t] : 0u;
            tot += v[k];
        }
        const uint32_t incl = wave_incl_scan_u32(tot, lane);
        __syncthreads();
        if (lane == 63) sm[wv] = incl;
        __syncthreads();
        uint32_t woff = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { if (k < wv) woff += sm[k]; all += sm[k]; }
        uint32_t off = carry + woff + incl - tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = b0 + tid * 4 + k;
            if (t < ntiles) { ranges[2 * t] = off; ranges[2 * t + 1] = off + v[k]; }
            off += v[k];
        }
        carry += all;
    }
}

// One wave per tile ROW of the super-tile (SB waves, SB tiles each).  A wave first compacts the segment's entries to those
// that touch its row (about half at C3) through a small LDS ring, and runs the per-tile ballots only on full batches of
// 64 compacted entries: SB^2 x entries tests become ~ SB x entries (row filter) + SB x 0.47 x entries x SB (tile tests).
#define RING 128
// cur[k]: byte offset of tile k's cursor from `out` (WIDE: entry index, for lists beyond 4 GB)
template <bool WIDE, int SB>
__device__ __forceinline__ void emit_batch(uint32_t id, uint32_t mrow, uint32_t (&cur)[SB], uint32_t *__restrict__ out) {
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const bool hit = (mrow & (1u << k)) != 0u;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (WIDE) { if (hit) out[(size_t)cur[k] + rank] = id; cur[k] += (uint32_t)__popcll(bal); }
        else {
            if (hit) *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(out) + (cur[k] + (rank << 2))) = id;
            cur[k] += (uint32_t)__popcll(bal) << 2;
        }
    }
}
template <bool WIDE, int SBS>
__global__ __launch_bounds__(GS_WAVE << SBS) void l2_write_kernel(GsBin3Args a) {
    constexpr int SB = 1 << SBS, NT = GS_WAVE * SB;
    __shared__ uint32_t sid[L2_SEG];
    __shared__ uint16_t scol[L2_SEG];                       // column bits of the entry inside the super-tile
    __shared__ uint16_t srow[L2_SEG];                       // row bits
    __shared__ uint32_t ring_id[SB][RING], ring_m[SB][RING];
    __shared__ uint32_t sh[NT / GS_WAVE + 4];               // (find_work: the waves' sums + the work item)
    __shared__ uint32_t sdead[SB * SB / 32];
    int S; uint32_t e0, e1, w0;
    if (lists_overflow(a.totals, a.cap_coarse, a.cap_fine)) return;
    if (!find_work<NT>(a, sh, S, e0, e1, w0)) return;
    const uint32_t kseg = blockIdx.x - w0;                  // this work item's segment of the super-tile's list
    if (a.tile_nopen && kseg >= a.smax[S]) return;          // capped lists: no tile of the super-tile takes entries this deep
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ox = (S % a.sgx) * SB, oy = (S / a.sgx) * SB;
    const int cnt = (int)(e1 - e0);
    constexpr int ITEMS = L2_SEG / NT;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = tid + k * NT;
        if (i < cnt) {
            const uint32_t lr = a.clr[e0 + i];
            const int lx0 = lr & (SB - 1), lx1 = (lr >> SBS) & (SB - 1), ly0 = (lr >> (2 * SBS)) & (SB - 1), ly1 = (lr >> (3 * SBS)) & (SB - 1);
            sid[i] = a.cids[e0 + i];
            scol[i] = (uint16_t)((2u << lx1) - (1u << lx0));
            srow[i] = (uint16_t)((2u << ly1) - (1u << ly0));
        }
    }
    // cursors of the wave's tiles (row wv of the super-tile): range start + the hits of the earlier segments
    uint32_t base = 0;
    if (lane < SB) {
        const int t = SB * wv + lane;
        const int tx = ox + lane, ty = oy + wv;
        if (tx < a.gx && ty < a.gy) {
            base = a.ranges[2 * (ty * a.gx + tx)];
#pragma unroll 8
            for (uint32_t w = w0; w < blockIdx.x; ++w) base += a.segcnt[(size_t)w * (SB * SB) + t];
        }
    }
    if (wv < SB * SB / GS_WAVE) {                          // tiles outside the grid or completed in an earlier round take nothing
        const int lt = wv * GS_WAVE + lane;                // local tile
        const int tx = ox + (lt & (SB - 1)), ty = oy + (lt >> SBS);
        const bool dead = tx >= a.gx || ty >= a.gy || (a.done && a.done[ty * a.gx + tx]) ||
                          (a.tile_nopen && kseg >= a.tile_nopen[ty * a.gx + tx]);       // (|| short-circuits: tiles outside the grid are not looked up)
        const unsigned long long b = __ballot(dead);
        if (lane == 0) { sdead[2 * wv] = (uint32_t)b; sdead[2 * wv + 1] = (uint32_t)(b >> 32); }
    }
    uint32_t cur[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) cur[k] = __builtin_amdgcn_readlane(WIDE ? base : base << 2, k);
    __syncthreads();
    // live tiles of this wave's row: bits wv * SB .. wv * SB + SB - 1 of the dead mask
    const uint32_t liverow = ~(sdead[(wv * SB) >> 5] >> ((wv * SB) & 31)) & ((1u << SB) - 1u);
    uint32_t *__restrict__ out = a.ids_out;
    uint32_t *rid = ring_id[wv], *rm = ring_m[wv];
    uint32_t fill = 0;
    if (liverow)
    for (int b = 0; b < cnt; b += GS_WAVE) {
        const int i = b + lane;
        uint32_t mrow = 0, id = 0;
        if (i < cnt && ((srow[i] >> wv) & 1u)) { mrow = scol[i] & liverow; id = sid[i]; }
        const bool touch = mrow != 0u;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(touch);
        if (bal == 0ull) continue;
        const uint32_t pos = fill + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (touch) { rid[pos] = id; rm[pos] = mrow; }
        fill += (uint32_t)__popcll(bal);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (fill >= GS_WAVE) {                              // uniform
            const uint32_t bid = rid[lane], bm = rm[lane];
            const uint32_t tid2 = rid[GS_WAVE + lane], tm2 = rm[GS_WAVE + lane];
            emit_batch<WIDE, SB>(bid, bm, cur, out);
            fill -= GS_WAVE;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if ((uint32_t)lane < fill) { rid[lane] = tid2; rm[lane] = tm2; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (fill) {
        uint32_t bid = 0, bm = 0;
        if ((uint32_t)lane < fill) { bid = rid[lane]; bm = rm[lane]; }
        emit_batch<WIDE, SB>(bid, bm, cur, out);
    }
}

// sdone[S] = 1 when every tile of super-tile S that lies inside the grid completed in an earlier round
template <int SBS>
__global__ __launch_bounds__(L2_THREADS) void super_done_kernel(const uint8_t *__restrict__ done, int gx, int gy, int sgx, int ns, uint8_t *__restrict__ sdone) {
    constexpr int SB = 1 << SBS;
    const int lane = threadIdx.x & 63, S = blockIdx.x * (L2_THREADS / GS_WAVE) + (threadIdx.x >> 6);
    if (S >= ns) return;
    bool all = true;
    for (int lt = lane; lt < SB * SB; lt += GS_WAVE) {
        const int tx = (S % sgx) * SB + (lt & (SB - 1)), ty = (S / sgx) * SB + (lt >> SBS);
        all = all && (tx >= gx || ty >= gy || done[ty * gx + tx]);
    }
    const unsigned long long b = __ballot(all);
    if (lane == 0) sdone[S] = b == ~0ull ? 1 : 0;
}
hipError_t gs_launch_super_done(const uint8_t *done, int gx, int gy, int sgx, int sgy, int sbs, uint8_t *sdone, hipStream_t s) {
    const int ns = sgx * sgy, per = L2_THREADS / GS_WAVE;
    if (sbs == 4) hipLaunchKernelGGL(super_done_kernel<4>, dim3((ns + per - 1) / per), dim3(L2_THREADS), 0, s, done, gx, gy, sgx, ns, sdone);
    else hipLaunchKernelGGL(super_done_kernel<3>, dim3((ns + per - 1) / per), dim3(L2_THREADS), 0, s, done, gx, gy, sgx, ns, sdone);
    return hipGetLastError();
}

// ---------------------------------------------------------------- drivers
static L1Args l1_args(const GsBin3L1 &b) {
    L1Args a{};
    a.rect = b.rect; a.perm = b.perm; a.sdone = b.sdone; a.n = b.n; a.n_slab = b.n_slab; a.sgx = b.sgx; a.ns = b.ns;
    a.sbs = b.sbs == 4 ? 4 : 3;
    a.G = gs_bin3_group(b.ns);
    a.nwg = (int)((b.n_slab + a.G - 1) / a.G); a.nwg_all = (int)((b.n + a.G - 1) / a.G);
    a.rect_sorted = reinterpret_cast<uint2 *>(b.rect_sorted); a.table = b.table; a.row_total = b.row_total;
    a.partials = b.partials; a.totals = b.totals; a.cranges = b.cranges; a.cids = b.cids; a.clr = b.clr;
    a.tilecnt = b.tilecnt; a.ntiles = b.ntiles; a.zero_words = b.zero_words;
    a.cap_coarse = b.cap_coarse; a.cap_fine = b.cap_fine;
    a.host_totals = b.host_totals; a.host_walked = b.host_walked; a.walked_src = b.walked_src;
    a.tile_walked = b.tile_walked; a.n_tile_walked = b.n_tile_walked;
    return a;
}
size_t gs_bin3_table_words(int64_t n_slab, int ns) { const int G = gs_bin3_group(ns); return (size_t)ns * (size_t)((n_slab + G - 1) / G + 1); }
size_t gs_bin3_partial_words(int64_t n, int ns) { const int G = gs_bin3_group(ns); return 3 * (size_t)((n + G - 1) / G + 1); }

// histogram + scan: after this the three totals (coarse instances of the slab, fine instances of the slab, of all n) are on the device
hipError_t gs_bin3_l1_count(const GsBin3L1 &b, hipStream_t s) {
    const L1Args a = l1_args(b);
    if (a.nwg_all <= 0) return hipMemsetAsync(b.totals, 0, 3 * sizeof(uint32_t), s);
    const size_t lds = sizeof(uint32_t) * ((size_t)a.ns * (a.G >> 5) + a.ns);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(l1_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(l1_hist_kernel, dim3(l1_grid(a.nwg_all)), dim3(a.G), lds, s, a);
    hipLaunchKernelGGL(l1_rowscan_kernel, dim3(a.ns + 3 + (a.ntiles + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}
// coarse lists (cids, clr, cranges); gs_bin3_l1_count of the same arguments first
hipError_t gs_bin3_l1_scatter(const GsBin3L1 &b, hipStream_t s) {
    const L1Args a = l1_args(b);
    if (a.nwg <= 0) return hipMemsetAsync(b.cranges, 0, 2 * sizeof(uint32_t) * (size_t)b.ns, s);
    const size_t lds = sizeof(uint32_t) * ((size_t)a.ns * (a.G >> 5) + 2 * (size_t)a.ns + 2 * (size_t)L1_CAP(a.G)) + sizeof(uint16_t) * (size_t)a.ns * (a.G >> 5);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(l1_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(l1_scatter_kernel, dim3(l1_grid(a.nwg)), dim3(a.G), lds, s, a);
    return hipGetLastError();
}

hipError_t gs_bin3_build_lists(const GsBin3Args &a, hipStream_t s) {
    if (a.max_work <= 0) return hipSuccess;
    GsBin3Args b = a;
    if (!b.cap_src || !b.tile_nopen || !b.smax || !b.tile_ext || b.done) { b.cap_src = nullptr; b.tile_nopen = nullptr; b.smax = nullptr; b.tile_ext = nullptr; }
    if (b.sbs == 4) hipLaunchKernelGGL(l2_count_kernel<4>, dim3(b.max_work), dim3(L2_THREADS), 0, s, b);
    else hipLaunchKernelGGL(l2_count_kernel<3>, dim3(b.max_work), dim3(L2_THREADS), 0, s, b);
    if (b.cap_src) {
        if (b.sbs == 4) hipLaunchKernelGGL(l2_cap_kernel<4>, dim3(b.ns), dim3(GS_WAVE), 0, s, b);
        else hipLaunchKernelGGL(l2_cap_kernel<3>, dim3(b.ns), dim3(GS_WAVE), 0, s, b);
    }
    hipLaunchKernelGGL(l2_ranges_kernel, dim3(1), dim3(1024), 0, s, b.tilecnt, b.gx * b.gy, b.done, b.ranges, b.totals, b.cap_coarse, b.cap_fine, b.tile_ext);
    return gs_bin3_write_lists(b, s);
}
hipError_t gs_bin3_write_lists(const GsBin3Args &a, hipStream_t s) {
    if (a.max_work <= 0) return hipSuccess;
    if (a.sbs == 4) {
        if (a.wide) hipLaunchKernelGGL((l2_write_kernel<true, 4>), dim3(a.max_work), dim3(GS_WAVE << 4), 0, s, a);
        else hipLaunchKernelGGL((l2_write_kernel<false, 4>), dim3(a.max_work), dim3(GS_WAVE << 4), 0, s, a);
    } else {
        if (a.wide) hipLaunchKernelGGL((l2_write_kernel<true, 3>), dim3(a.max_work), dim3(GS_WAVE << 3), 0, s, a);
        else hipLaunchKernelGGL((l2_write_kernel<false, 3>), dim3(a.max_work), dim3(GS_WAVE << 3), 0, s, a);
    }
    return hipGetLastError();
}
