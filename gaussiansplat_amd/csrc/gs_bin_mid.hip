// gs_bin_mid.hip -- gs_bin (compactIdxs, reference src/forward.jl:103 and 118-161) for MID-SIZE frames in TWO launches.
//
// A frame of a hundred thousand gaussians on a 50 x 50 tile grid (BASELINE C2) takes the general path ten dependent launches (four for the
// depth order, three + three for the two-level lists): 77 us of a 0.28 ms frame, most of it between kernels.  The small-frame path
// (gs_bin_small.hip: every tile looks at every gaussian) does not scale to it, but its two ideas do:
//   * nothing is sorted globally: a tile's list is the set of gaussians whose rectangle covers it (hitBinning, forward.jl:118-131) in
//     the order of their own (depth key, id) pairs, which is the order CUDA.sortperm(forward.jl:103) puts them in;
//   * where a tile's list starts (scan!, forward.jl:145-150) needs no scan over tiles: with D the 2-D difference array of the
//     rectangles (four +-1 per gaussian; its 2-D prefix sum is the hit count per tile), the number of hits of all tiles before tile t
//     is sum over (y, x) of D[y][x] x #{tiles t' < t right of and below (y, x)} -- a closed-form weight per entry, a few thousand
//     entries, added up by every tile's workgroup for itself.
// bin_mid_l1_kernel (one thread per gaussian): the four difference atomics, and the gaussian's id + rectangle clipped to the super-tile
//     into the candidate list of every super-tile (8 x 8 tiles) it touches -- slots handed out per workgroup through LDS, one global
//     atomic per (workgroup, super-tile); the order inside a list does not matter.
// bin_mid_tiles_kernel (one workgroup of four waves per tile): tests the candidates of its super-tile, gathers the depth keys of the
//     hits, ranks them in LDS -- up to 512 by counting, up to 4096 by a bitonic network on the 64-bit (key, id) pairs -- and writes its
//     range and ids (compactHits, compact.jl:3-21); the last tile's workgroup stores the frame's totals into pinned host memory.
// Pairs are distinct, so the lists are the stable (key, id) order of the radix paths, bit for bit (tests/test_gpu_bin_mid.py).
// What does not fit -- a super-tile with more candidates than its region holds, a tile with more than 4096 hits, lists beyond the ids
// buffer -- is reported to the host with the totals: the tile's range stays empty, the host bins the frame again with the general path
// and keeps to it for 64 frames (gs_api_bin.hip: settle_totals).
#include "gs_common.h"

#define BM_NT 256
#define BM_NW (BM_NT / GS_WAVE)
#define BM_COUNT_MAX (2 * BM_NT)                                                  // hits ranked by counting (two per thread)
static_assert(GS_BIN_MID_MAX_SUPER <= BM_NT, "one super-tile per thread in the level-1 kernel");

__global__ __launch_bounds__(BM_NT) void bin_mid_l1_kernel(GsBinMidArgs a) {
    __shared__ uint32_t lcnt[GS_BIN_MID_MAX_SUPER], lbase[GS_BIN_MID_MAX_SUPER], ltake[GS_BIN_MID_MAX_SUPER];
    const int tid = threadIdx.x;
    const int g = (int)blockIdx.x * BM_NT + tid;
    for (int i = tid; i < a.ns; i += BM_NT) { lcnt[i] = 0u; ltake[i] = 0u; }
    __syncthreads();
    uint2 rc = make_uint2(0u, 0u);
    if (g < a.n) rc = a.rect[g];
    const uint32_t x0 = rc.x & 0xFFFFu, x1 = rc.x >> 16, y0 = rc.y & 0xFFFFu, y1 = rc.y >> 16;   // 1-based inclusive; x0 == 0: no tile
    uint32_t sx0 = 1, sx1 = 0, sy0 = 1, sy1 = 0;
    if (x0 != 0u) {
        const int w = a.gx + 1;
        atomicAdd(&a.diff_cur[(y0 - 1u) * w + (x0 - 1u)], 1);
        atomicAdd(&a.diff_cur[(y0 - 1u) * w + x1], -1);
        atomicAdd(&a.diff_cur[y1 * w + (x0 - 1u)], -1);
        atomicAdd(&a.diff_cur[y1 * w + x1], 1);
        sx0 = (x0 - 1u) >> 3; sx1 = (x1 - 1u) >> 3; sy0 = (y0 - 1u) >> 3; sy1 = (y1 - 1u) >> 3;
        for (uint32_t sy = sy0; sy <= sy1; ++sy)
            for (uint32_t sx = sx0; sx <= sx1; ++sx) atomicAdd(&lcnt[sy * (uint32_t)a.sgx + sx], 1u);
    }
    __syncthreads();
    for (int i = tid; i < a.ns; i += BM_NT) lbase[i] = lcnt[i] ? atomicAdd(&a.scount_cur[i], lcnt[i]) : 0u;
    __syncthreads();
    if (x0 != 0u)
        for (uint32_t sy = sy0; sy <= sy1; ++sy)
            for (uint32_t sx = sx0; sx <= sx1; ++sx) {
                const uint32_t S = sy * (uint32_t)a.sgx + sx;
                const uint32_t slot = lbase[S] + atomicAdd(&ltake[S], 1u);
                if (slot < a.cap_s) {                                             // (beyond the region: the tiles kernel sees the count and reports it)
                    const uint32_t lx0 = max(x0 - 1u, 8u * sx) - 8u * sx, lx1 = min(x1 - 1u, 8u * sx + 7u) - 8u * sx;
                    const uint32_t ly0 = max(y0 - 1u, 8u * sy) - 8u * sy, ly1 = min(y1 - 1u, 8u * sy + 7u) - 8u * sy;
                    a.cand[(size_t)S * a.cap_s + slot] = make_uint2((uint32_t)g, lx0 | (lx1 << 3) | (ly0 << 6) | (ly1 << 9));
                }
            }
}

__global__ __launch_bounds__(BM_NT) void bin_mid_tiles_kernel(GsBinMidArgs a) {
    __shared__ unsigned long long hits[GS_BIN_MID_TILE_CAP];                      // the tile's (key << 32 | id) pairs
    __shared__ long long wsum[BM_NW];
    __shared__ unsigned long long wide[BM_NW];
    __shared__ uint32_t lds_h;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, t = blockIdx.x;
    const int tx = t % a.gx, ty = t / a.gx;                                        // 0-based here
    const uint32_t S = (uint32_t)(ty >> 3) * (uint32_t)a.sgx + (uint32_t)(tx >> 3), lx = (uint32_t)tx & 7u, ly = (uint32_t)ty & 7u;
    const int nd = (a.gx + 1) * (a.gy + 1);
    // ---- the NEXT frame's difference array and candidate counts (the other parity): every workgroup clears a slice
    for (int i = t * BM_NT + tid; i < nd + a.ns; i += a.ntiles * BM_NT) {
        if (i < nd) a.diff_next[i] = 0; else a.scount_next[i - nd] = 0u;
    }
    if (tid == 0) lds_h = 0u;
    const uint32_t listed = a.scount_cur[S];
    const uint32_t nc = min(listed, a.cap_s);
    bool overflow = listed > a.cap_s;
    // ---- where my list starts: hits of all tiles before mine = sum of D[y][x] x #{t' < t : y' >= y, x' >= x}
    long long acc = 0;
    for (int i = tid; i < nd; i += BM_NT) {
        const int d = a.diff_cur[i];
        if (d != 0) {
            const int y = i / (a.gx + 1), x = i - y * (a.gx + 1);
            const long long wgt = (long long)max(0, ty - y) * max(0, a.gx - x) + (ty >= y ? max(0, tx - x) : 0);
            acc += (long long)d * wgt;
        }
    }
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    if (lane == 0) wsum[q] = acc;
    __syncthreads();
    long long start64 = 0;
#pragma unroll
    for (int k = 0; k < BM_NW; ++k) start64 += wsum[k];
    const uint32_t start = (uint32_t)start64;
    // ---- my super-tile's candidates: the ones whose clipped rectangle holds my tile, in any order
    const uint2 *cand = a.cand + (size_t)S * a.cap_s;
    for (uint32_t base = 0; base < nc; base += BM_NT) {
        const uint32_t i = base + (uint32_t)tid;
        bool in = false;
        uint32_t id = 0;
        if (i < nc) {
            const uint2 c = cand[i];
            id = c.x;
            in = (c.y & 7u) <= lx && lx <= ((c.y >> 3) & 7u) && ((c.y >> 6) & 7u) <= ly && ly <= ((c.y >> 9) & 7u);
        }
        const unsigned long long m = __ballot(in);
        if (m == 0ull) continue;                                                  // (wave-uniform)
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&lds_h, (uint32_t)__popcll(m));
        pos = (uint32_t)__shfl((int)pos, 0) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (in && pos < GS_BIN_MID_TILE_CAP) hits[pos] = id;
    }
    __syncthreads();
    const uint32_t h = lds_h;                                                     // (exact even when it exceeds what was kept)
    if (h > GS_BIN_MID_TILE_CAP || (unsigned long long)start + h > a.cap_fine) overflow = true;
    if (overflow) {
        if (tid == 0) { reinterpret_cast<uint2 *>(a.ranges)[t] = make_uint2(start, start); *a.host_overflow = 1u; }
    } else {
        if (tid == 0) reinterpret_cast<uint2 *>(a.ranges)[t] = make_uint2(start, start + h);
        for (uint32_t i = tid; i < h; i += BM_NT) { const uint32_t id = (uint32_t)hits[i]; hits[i] = ((unsigned long long)(a.depth_key ? a.depth_key[id] : 0u) << 32) | id; }
        __syncthreads();
        if (h <= BM_COUNT_MAX) {
            // ---- every hit counts the pairs below its own (the same LDS word for the whole wave: a broadcast)
            const unsigned long long me0 = (uint32_t)tid < h ? hits[tid] : ~0ull, me1 = (uint32_t)tid + BM_NT < h ? hits[tid + BM_NT] : ~0ull;
            uint32_t b0 = 0, b1 = 0;
            const uint32_t lim = (uint32_t)(q * GS_WAVE) < h ? h : 0u;            // (a wave without hits skips the loop)
            for (uint32_t j = 0; j < lim; ++j) { const unsigned long long p = hits[j]; b0 += p < me0 ? 1u : 0u; b1 += p < me1 ? 1u : 0u; }
            if ((uint32_t)tid < h) a.ids[start + b0] = (uint32_t)me0;
            if ((uint32_t)tid + BM_NT < h) a.ids[start + b1] = (uint32_t)me1;
        } else {
            // ---- bitonic network over the next power of two (padding sorts behind everything)
            uint32_t p2 = 1;
            while (p2 < h) p2 <<= 1;
            for (uint32_t i = h + tid; i < p2; i += BM_NT) hits[i] = ~0ull;
            __syncthreads();
            for (uint32_t k = 2; k <= p2; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = tid; i < (p2 >> 1); i += BM_NT) {
                        const uint32_t lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;      // the i-th pair of this step
                        const unsigned long long x = hits[lo], y = hits[hi];
                        const bool up = (lo & k) == 0u;
                        if ((x > y) == up) { hits[lo] = y; hits[hi] = x; }
                    }
                    __syncthreads();
                }
            for (uint32_t i = tid; i < h; i += BM_NT) a.ids[start + i] = (uint32_t)hits[i];
        }
    }
    if (t != a.ntiles - 1) return;
    // ---- the last tile: the frame's totals {candidates kept (informational), listed, all} and the previous forward's walked entries
    unsigned long long wk = 0;
    if (a.host_walked && a.tile_walked)
        for (int i = tid; i < a.n_tile_walked; i += BM_NT) wk += a.tile_walked[i];
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) wk += __shfl_down(wk, d);
    if (lane == 0) wide[q] = wk;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long tot = (unsigned long long)start64 + h;
        const uint32_t total = tot >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;
        a.totals[0] = 0u; a.totals[1] = total; a.totals[2] = total;
        if (a.host_walked) {
            if (a.tile_walked) { unsigned long long s = 0; for (int k = 0; k < BM_NW; ++k) s += wide[k]; a.host_walked[0] = (uint32_t)s; a.host_walked[1] = (uint32_t)(s >> 32); }
            else { a.host_walked[0] = a.walked_src[0]; a.host_walked[1] = a.walked_src[1]; }
        }
        if (a.host_totals) { a.host_totals[0] = 0u; a.host_totals[1] = total; a.host_totals[2] = total; }
    }
}

bool gs_bin_mid_supported(int64_t n, int gx, int gy) {
    const int64_t nt = (int64_t)gx * gy, ns = (int64_t)((gx + 7) / 8) * ((gy + 7) / 8);
    return n >= 1 && n <= GS_BIN_MID_MAX_N && nt >= 1 && nt <= GS_BIN_MID_MAX_TILES && ns <= GS_BIN_MID_MAX_SUPER;
}

hipError_t gs_bin_mid(const GsBinMidArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(bin_mid_l1_kernel, dim3((unsigned)((a.n + BM_NT - 1) / BM_NT)), dim3(BM_NT), 0, s, a);
    hipLaunchKernelGGL(bin_mid_tiles_kernel, dim3((unsigned)a.ntiles), dim3(BM_NT), 0, s, a);
    return hipGetLastError();
}
