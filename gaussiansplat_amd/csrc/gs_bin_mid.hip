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
// bin_mid_l1_kernel (up to 32 workgroups of 1024): the four difference updates per gaussian, accumulated in LDS; and the gaussian's id +
//     rectangle clipped to the super-tile into the candidate list of every super-tile (8 x 8 tiles) it touches -- slots handed out per
//     workgroup through LDS, one global atomic per (workgroup, super-tile); the order inside a list does not matter.
// bin_mid_tiles_kernel (one workgroup of four waves per tile): tests the candidates of its super-tile, gathers the depth keys of the
//     hits, ranks them in LDS -- up to 64 by counting, up to 4096 by a bitonic network on the 64-bit (key, id) pairs -- and writes its
//     range and ids (compactHits, compact.jl:3-21); the last tile's workgroup stores the frame's totals into pinned host memory.
// Pairs are distinct, so the lists are the stable (key, id) order of the radix paths, bit for bit (tests/test_gpu_bin_mid.py).
// What does not fit -- a super-tile with more candidates than its region holds, a tile with more than 4096 hits, lists beyond the ids
// buffer -- is reported to the host with the totals: the tile's range stays empty, the host bins the frame again with the general path
// and keeps to it for 64 frames (gs_api_bin.hip: settle_totals).
#include "gs_common.h"
#include <algorithm>

#define BM_NT 256
#define BM_NW (BM_NT / GS_WAVE)
#define BM_COUNT_MAX GS_WAVE                                                      // hits ranked by counting (one per lane of the first wave); more: bitonic network

// A few large workgroups: the difference array and the candidate counts are accumulated in LDS and reach global memory as one atomic per
// (workgroup, cell) -- same-address global atomics drain at ~ 64 ns each on this chip (4 M of them took 257 us, gs_bin2.hip), so the
// 400 k corner updates of a C2 frame must not be global ones.
#define BM_L1_NT 1024
#define BM_L1_MAX_WG 32
__global__ __launch_bounds__(BM_L1_NT) void bin_mid_l1_kernel(GsBinMidArgs a) {
    extern __shared__ int ldiff[];                                                // (gy + 1) x (gx + 1)
    __shared__ uint32_t lcnt[GS_BIN_MID_MAX_SUPER], lbase[GS_BIN_MID_MAX_SUPER], ltake[GS_BIN_MID_MAX_SUPER];
    const int tid = threadIdx.x, w = a.gx + 1, cells = w * (a.gy + 1);
    for (int i = tid; i < cells; i += BM_L1_NT) ldiff[i] = 0;
    for (int i = tid; i < a.ns; i += BM_L1_NT) { lcnt[i] = 0u; ltake[i] = 0u; }
    __syncthreads();
    const int per = (a.n + (int)gridDim.x - 1) / (int)gridDim.x, g0 = (int)blockIdx.x * per, g1 = min(a.n, g0 + per);
    for (int g = g0 + tid; g < g1; g += BM_L1_NT) {
        const uint2 rc = a.rect[g];
        const uint32_t x0 = rc.x & 0xFFFFu, x1 = rc.x >> 16, y0 = rc.y & 0xFFFFu, y1 = rc.y >> 16;   // 1-based inclusive; x0 == 0: no tile
        if (x0 == 0u) continue;
        atomicAdd(&ldiff[(y0 - 1u) * w + (x0 - 1u)], 1);
        atomicAdd(&ldiff[(y0 - 1u) * w + x1], -1);
        atomicAdd(&ldiff[y1 * w + (x0 - 1u)], -1);
        atomicAdd(&ldiff[y1 * w + x1], 1);
        const uint32_t sx0 = (x0 - 1u) >> 3, sx1 = (x1 - 1u) >> 3, sy0 = (y0 - 1u) >> 3, sy1 = (y1 - 1u) >> 3;
        for (uint32_t sy = sy0; sy <= sy1; ++sy)
            for (uint32_t sx = sx0; sx <= sx1; ++sx) atomicAdd(&lcnt[sy * (uint32_t)a.sgx + sx], 1u);
    }
    __syncthreads();
    for (int i = tid; i < a.ns; i += BM_L1_NT) lbase[i] = lcnt[i] ? atomicAdd(&a.scount_cur[i], lcnt[i]) : 0u;
    for (int i = tid; i < cells; i += BM_L1_NT) { const int d = ldiff[i]; if (d != 0) atomicAdd(&a.diff_cur[i], d); }
    __syncthreads();
    for (int g = g0 + tid; g < g1; g += BM_L1_NT) {
        const uint2 rc = a.rect[g];
        const uint32_t x0 = rc.x & 0xFFFFu, x1 = rc.x >> 16, y0 = rc.y & 0xFFFFu, y1 = rc.y >> 16;
        if (x0 == 0u) continue;
        const uint32_t sx0 = (x0 - 1u) >> 3, sx1 = (x1 - 1u) >> 3, sy0 = (y0 - 1u) >> 3, sy1 = (y1 - 1u) >> 3;
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
    for (int x = lane; x <= a.gx; x += GS_WAVE)                                   // a lane per column, a wave per row (rows below mine weigh nothing), eight rows in flight
        for (int y0 = q; y0 <= ty; y0 += 8 * BM_NW) {
            int d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = a.diff_cur[min(y0 + u * BM_NW, ty) * (a.gx + 1) + x];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int y = y0 + u * BM_NW;
                if (y <= ty) acc += (long long)d[u] * ((long long)(ty - y) * max(0, a.gx - x) + max(0, tx - x));
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
    for (uint32_t base = 0; base < nc; base += 8 * BM_NT) {                       // eight loads in flight per thread (the list was written by another XCD: every load is a trip to memory)
        uint2 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = cand[min(base + (uint32_t)(u * BM_NT + tid), nc - 1u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = base + (uint32_t)(u * BM_NT + tid);
            const bool in = i < nc && (c[u].y & 7u) <= lx && lx <= ((c[u].y >> 3) & 7u) && ((c[u].y >> 6) & 7u) <= ly && ly <= ((c[u].y >> 9) & 7u);
            const unsigned long long m = __ballot(in);
            if (m == 0ull) continue;                                              // (wave-uniform)
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&lds_h, (uint32_t)__popcll(m));
            pos = (uint32_t)__shfl((int)pos, 0) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (in && pos < GS_BIN_MID_TILE_CAP) hits[pos] = c[u].x;
        }
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
            if ((uint32_t)tid < h) {
                const unsigned long long me = hits[tid];
                uint32_t below = 0;
                for (uint32_t j = 0; j < h; ++j) below += hits[j] < me ? 1u : 0u;
                a.ids[start + below] = (uint32_t)me;
            }
        } else {
            // ---- bitonic network over the next power of two (padding sorts behind everything)
            uint32_t p2 = 1;
            while (p2 < h) p2 <<= 1;
            for (uint32_t i = h + tid; i < p2; i += BM_NT) hits[i] = ~0ull;
            __syncthreads();
            // A compare-exchange with distance j <= 64 stays inside a block of 128 pairs = the 64 comparators of ONE wave: those steps need
            // no workgroup barrier (a wave's LDS operations are performed in order), only the steps with j >= 128 do -- 6 barriers instead of
            // 45 for 512 pairs, and the network is bound by exactly that latency.
            volatile unsigned long long *vh = hits;
            auto step = [&](uint32_t i, uint32_t j, uint32_t k) {
                const uint32_t lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;          // the i-th pair of this step
                const unsigned long long x = vh[lo], y = vh[hi];
                if ((x > y) == ((lo & k) == 0u)) { vh[lo] = y; vh[hi] = x; }
            };
            const uint32_t nblk = p2 >> 7;                                         // (h > 64: at least one block)
            for (uint32_t blk = q; blk < nblk; blk += BM_NW)
                for (uint32_t k = 2; k <= 128u; k <<= 1)
                    for (uint32_t j = k >> 1; j > 0; j >>= 1) { step(blk * GS_WAVE + lane, j, k); __builtin_amdgcn_wave_barrier(); }
            __syncthreads();
            for (uint32_t k = 256; k <= p2; k <<= 1) {
                for (uint32_t j = k >> 1; j >= 128u; j >>= 1) {
                    for (uint32_t i = tid; i < (p2 >> 1); i += BM_NT) step(i, j, k);
                    __syncthreads();
                }
                for (uint32_t blk = q; blk < nblk; blk += BM_NW)
                    for (uint32_t j = 64; j > 0; j >>= 1) { step(blk * GS_WAVE + lane, j, k); __builtin_amdgcn_wave_barrier(); }
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
    const size_t lds = sizeof(int) * (size_t)(a.gx + 1) * (a.gy + 1);
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bin_mid_l1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    const int wgs = std::max(1, std::min(BM_L1_MAX_WG, (a.n + 4095) / 4096));
    hipLaunchKernelGGL(bin_mid_l1_kernel, dim3((unsigned)wgs), dim3(BM_L1_NT), lds, s, a);
    hipLaunchKernelGGL(bin_mid_tiles_kernel, dim3((unsigned)a.ntiles), dim3(BM_NT), 0, s, a);
    return hipGetLastError();
}
