// gs_bin_small.hip -- gs_bin (compactIdxs, reference src/forward.jl:103 and 118-161) for SMALL frames in ONE launch.
//
// A frame of ten thousand gaussians on a 16 x 16 tile grid (BASELINE C1) spends its time between kernels, not inside them: the
// general path takes ten dependent launches (4 for the depth order, 3 + 3 for the two-level lists) of ~5 us each around
// microseconds of work.  Up to GS_BIN_SMALL_MAX_N gaussians and GS_BIN_SMALL_MAX_TILES tiles every tile can afford to look at every
// gaussian, nothing has to be sorted globally and no workgroup has to hear from another:
//   * a tile's list is the set of gaussians whose rectangle covers the tile (hitBinning, forward.jl:118-131), in the order
//     CUDA.sortperm(forward.jl:103) would give them -- which is the order of their own (depth key, id) pairs;
//   * the list starts where the lists of the tiles before it end (scan!, forward.jl:145-150), and that number has a closed form per
//     gaussian: the tiles of its rectangle with a smaller index are its full rows above the tile's row and the part of the
//     tile's own row left of the tile.
// bin_small_kernel, one workgroup of sixteen waves per tile: every wave loads a sixteenth of the model's rectangles (all before the
// first is looked at), adds up the closed form (-> the start of the tile's list) and tests them against the tile, 64 per ballot; the hits
// -- (depth key << 32 | id) -- are compacted into LDS in index order and ranked: up to 256 hits by counting (every hit counts the pairs
// below its own, LDS broadcasts), more by a bitonic network on the 64-bit pairs.  Pairs are distinct, so either way the result is the
// stable (key, id) order of the radix paths, bit for bit; in index order nothing is sorted.  ids and the range are written
// (compactHits, compact.jl:3-21); the last tile's workgroup stores the frame's totals into pinned host memory.
// Same ranges and ids as every other path (tests/test_gpu_bin_small.py).  renderer.sortIdxs is not needed by anything downstream and
// is computed on demand (gs_get_array).
#include "gs_common.h"

#define BL_NT 1024
#define BL_NW (BL_NT / GS_WAVE)
#define BL_MAX_ROUNDS (GS_BIN_SMALL_MAX_N / BL_NT)
#define BL_COUNT_MAX 256                                                          // hits ranked by counting (one per thread of the first four waves)
static_assert(BL_MAX_ROUNDS * BL_NT == GS_BIN_SMALL_MAX_N, "GS_BIN_SMALL_MAX_N: a multiple of the workgroup");

static int bl_pow2ceil(int n) { int p = 1; while (p < n) p <<= 1; return p; }
static size_t bl_lds_bytes(int n) { return sizeof(unsigned long long) * (size_t)bl_pow2ceil(n > 0 ? n : 1); }

__global__ __launch_bounds__(BL_NT) void bin_small_kernel(GsBinSmallArgs a) {
    extern __shared__ unsigned long long hits[];                                  // [pow2ceil(n)] the tile's (key << 32 | id) pairs
    __shared__ uint32_t cnt[BL_NW], pre[BL_NW];
    __shared__ unsigned long long wide[BL_NW];
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, t = blockIdx.x;
    const bool sort = a.depth_key != nullptr;
    const uint32_t tx = (uint32_t)(t % a.gx) + 1u, ty = (uint32_t)(t / a.gx) + 1u;        // 1-based, as the rectangles
    // ---- every rectangle this wave is responsible for, loaded before anything is waited for (clamped index, no branch)
    const int rounds = (a.n + BL_NT - 1) / BL_NT, first = q * rounds * GS_WAVE;
    uint2 rc[BL_MAX_ROUNDS];
#pragma unroll
    for (int r = 0; r < BL_MAX_ROUNDS; ++r) rc[r] = a.rect[min(first + min(r, rounds - 1) * GS_WAVE + lane, a.n - 1)];
    for (size_t i = (size_t)t * BL_NT + tid; i < a.zero_words16; i += (size_t)a.ntiles * BL_NT) a.zero[i] = make_uint4(0u, 0u, 0u, 0u);
    unsigned long long hit[BL_MAX_ROUNDS];
    uint32_t mine = 0, before = 0;
#pragma unroll
    for (int r = 0; r < BL_MAX_ROUNDS; ++r) {
        const int s = first + r * GS_WAVE + lane;
        const uint32_t x0 = rc[r].x & 0xFFFFu, x1 = rc[r].x >> 16, y0 = rc[r].y & 0xFFFFu, y1 = rc[r].y >> 16;
        const bool valid = r < rounds && s < a.n && x0 != 0u;
        hit[r] = __ballot(valid && x0 <= tx && tx <= x1 && y0 <= ty && ty <= y1);
        mine += (uint32_t)__popcll(hit[r]);
        // tiles of this rectangle with an index below t: its rows above row ty in full, and of row ty the columns left of tx
        const uint32_t yb = min(y1, ty - 1u), xb = min(x1, tx - 1u);
        uint32_t c = yb >= y0 ? (yb - y0 + 1u) * (x1 - x0 + 1u) : 0u;
        if (y0 <= ty && ty <= y1 && xb >= x0) c += xb - x0 + 1u;
        before += valid ? c : 0u;
    }
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) before += (uint32_t)__shfl_down((int)before, d);
    if (lane == 0) { cnt[q] = mine; pre[q] = before; }
    __syncthreads();
    uint32_t off = 0, h = 0, start = 0;                                           // (at most n x tiles <= 4 M entries: 32 bits)
#pragma unroll
    for (int k = 0; k < BL_NW; ++k) { if (k < q) off += cnt[k]; h += cnt[k]; start += pre[k]; }
    // ---- my hits into LDS, in index order (waves, rounds and lanes ascending == ids ascending)
#pragma unroll
    for (int r = 0; r < BL_MAX_ROUNDS; ++r) {
        const unsigned long long m = hit[r];
        if (m == 0ull) continue;                                                  // (wave-uniform)
        if ((m >> lane) & 1ull) {
            const uint32_t id = (uint32_t)(first + r * GS_WAVE + lane);
            hits[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(sort ? a.depth_key[id] : 0u) << 32) | id;
        }
        off += (uint32_t)__popcll(m);
    }
    if (tid == 0) reinterpret_cast<uint2 *>(a.ranges)[t] = make_uint2(start, start + h);
    if (!sort || h <= 1u) {
        __syncthreads();
        for (uint32_t i = tid; i < h; i += BL_NT) a.ids[start + i] = (uint32_t)hits[i];
    } else if (h <= BL_COUNT_MAX) {
        // ---- few hits: every hit counts the pairs below its own (the same LDS word for the whole wave: a broadcast)
        __syncthreads();
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
        for (uint32_t i = h + tid; i < p2; i += BL_NT) hits[i] = ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= p2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < (p2 >> 1); i += BL_NT) {
                    const uint32_t lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;      // the i-th pair of this step
                    const unsigned long long x = hits[lo], y = hits[hi];
                    const bool up = (lo & k) == 0u;
                    if ((x > y) == up) { hits[lo] = y; hits[hi] = x; }
                }
                __syncthreads();
            }
        for (uint32_t i = tid; i < h; i += BL_NT) a.ids[start + i] = (uint32_t)hits[i];
    }
    if (t != a.ntiles - 1) return;
    // ---- the last tile: the frame's totals {coarse instances (none on this path), listed, all}, and the previous forward's walked
    // entries summed on their way to the host
    unsigned long long wk = 0;
    if (a.host_walked && a.tile_walked)
        for (int i = tid; i < a.n_tile_walked; i += BL_NT) wk += a.tile_walked[i];
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) wk += __shfl_down(wk, d);
    if (lane == 0) wide[q] = wk;
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = start + h;
        a.totals[0] = 0u; a.totals[1] = total; a.totals[2] = total;
        if (a.host_walked) {
            if (a.tile_walked) { unsigned long long s = 0; for (int k = 0; k < BL_NW; ++k) s += wide[k]; a.host_walked[0] = (uint32_t)s; a.host_walked[1] = (uint32_t)(s >> 32); }
            else { a.host_walked[0] = a.walked_src[0]; a.host_walked[1] = a.walked_src[1]; }
        }
        if (a.host_totals) { a.host_totals[0] = 0u; a.host_totals[1] = total; a.host_totals[2] = total; }
    }
}

bool gs_bin_small_supported(int64_t n, int gx, int gy) {
    const int64_t nt = (int64_t)gx * gy;
    return n >= 1 && n <= GS_BIN_SMALL_MAX_N && nt >= 1 && nt <= GS_BIN_SMALL_MAX_TILES && n * nt <= GS_BIN_SMALL_MAX_PAIRS;
}

hipError_t gs_bin_small(const GsBinSmallArgs &a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bin_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bl_lds_bytes(GS_BIN_SMALL_MAX_N));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(bin_small_kernel, dim3((unsigned)a.ntiles), dim3(BL_NT), bl_lds_bytes(a.n), s, a);
    return hipGetLastError();
}
