// gs_bin2.hip -- low-traffic tile binning for gfx950 (default path of gs_bin).
//
// Same result as the reference's compactIdxs (src/forward.jl:118-161: hitBinning, scan!, compactHits)
// with the per-tile lists in (tile, list order) -- bit-identical to the 64-bit key sort of
// gs_sort.hip -- but the I tile-instances (30 M at 1 M gaussians / 1080p) never exist as 64-bit keys:
//
//   pass 1  "generate + scatter": a workgroup owns 4096 consecutive instance positions of the
//           virtual emission order (gaussians in list order, each followed by the tiles of its
//           rectangle).  It rebuilds position -> (gaussian, tile) from the exclusive scan of the
//           per-gaussian tile counts (load-balanced expand in LDS: owner marks + max-scan), ranks
//           the keys by the LOW tile digit with wave64 ballots and writes 32-bit words
//           [high tile digit | gaussian id].  Nothing is read from an instance array.
//   pass 2  one ordinary stable radix pass over those 32-bit words on the high digit; the output
//           keeps only the gaussian id (the tile is implied by the tile ranges).
//   ranges  per-tile counts come from a 2-D difference array (4 integer atomics per gaussian,
//           binning.jl:26-31 rectangles) + prefix sums, not from scanning the instance list.
//
// HBM traffic per instance: 4 B written + 4 B read (hist) + 4 B read + 4 B written = 16 B, against
// 8 + 2 x 24 + 8 = 64 B for emit + two 64-bit passes + ranges.  Integer/byte work: coalesced 4-byte
// streams, LDS histograms, LDS-staged scatter; no MFMA.
#include "gs_common.h"

#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_CHUNK (RS_THREADS * RS_ITEMS)
#define RS_RADIX 256
#define RS_WAVES (RS_THREADS / GS_WAVE)

// ---------------------------------------------------------------- tile counts -> ranges
// Per-tile list lengths = number of rectangles covering each tile.  Each rectangle is four +-1
// corners of a 2-D difference array; DIFF_BLOCKS workgroups accumulate their share of the
// gaussians in a private LDS copy (LDS atomics), the copies are summed, and one workgroup turns
// the array into counts (2-D prefix sums in LDS) and the counts into [start,end) ranges.
// (Global atomics on the ~8 K shared cells serialise: 4 M adds took 257 us on MI355X.)
#define DIFF_BLOCKS 128
__global__ __launch_bounds__(256) void rect_diff_kernel(const uint16_t *__restrict__ rect, const uint32_t *__restrict__ perm, int64_t n,
                                                         int *__restrict__ partial, int pitch, int cells) {
    extern __shared__ int ldiff[];
    for (int i = threadIdx.x; i < cells; i += 256) ldiff[i] = 0;
    __syncthreads();
    const int64_t per = (n + DIFF_BLOCKS - 1) / DIFF_BLOCKS;
    const int64_t g0 = (int64_t)blockIdx.x * per, g1 = min(n, g0 + per);
    for (int64_t s = g0 + threadIdx.x; s < g1; s += 256) {
        const int64_t g = perm ? (int64_t)perm[s] : s;               // a slab of the list: positions -> gaussian ids
        const uint2 r = reinterpret_cast<const uint2 *>(rect)[g];
        const int x0 = (int)(r.x & 0xFFFFu), x1 = (int)(r.x >> 16), y0 = (int)(r.y & 0xFFFFu), y1 = (int)(r.y >> 16);
        if (x0 == 0) continue;
        // inclusive 1-based rectangle [x0,x1] x [y0,y1]
        atomicAdd(&ldiff[(y0 - 1) * pitch + (x0 - 1)], 1);
        atomicAdd(&ldiff[(y0 - 1) * pitch + x1], -1);
        atomicAdd(&ldiff[y1 * pitch + (x0 - 1)], -1);
        atomicAdd(&ldiff[y1 * pitch + x1], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += 256) partial[(size_t)blockIdx.x * cells + i] = ldiff[i];
}

__global__ void diff_sum_kernel(const int *__restrict__ partial, int *__restrict__ diff, int cells) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    int s = 0;
#pragma unroll 8
    for (int b = 0; b < DIFF_BLOCKS; ++b) s += partial[(size_t)b * cells + i];
    diff[i] = s;
}

// one workgroup: 2-D prefix sums of the difference array (in LDS), then the exclusive scan over tile ids
__global__ __launch_bounds__(1024) void ranges_from_diff_kernel(const int *__restrict__ diff, int pitch, int gx, int gy,
                                                                 uint32_t *__restrict__ ranges, const uint8_t *__restrict__ done) {
    extern __shared__ int ld[];
    __shared__ uint32_t sm[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cells = pitch * (gy + 1);
    for (int i = tid; i < cells; i += 1024) ld[i] = diff[i];
    __syncthreads();
    for (int y = tid; y < gy; y += 1024) {                      // along x
        int acc = 0;
        for (int x = 0; x < gx; ++x) { acc += ld[y * pitch + x]; ld[y * pitch + x] = acc; }
    }
    __syncthreads();
    for (int x = tid; x < gx; x += 1024) {                      // along y -> ld[y][x] = #gaussians covering the tile
        int acc = 0;
        for (int y = 0; y < gy; ++y) { acc += ld[y * pitch + x]; ld[y * pitch + x] = acc; }
    }
    if (tid == 0) carry = 0;
    __syncthreads();
    const int ntiles = gx * gy;
    for (int b0 = 0; b0 < ntiles; b0 += 1024) {
        const int t = b0 + tid;
        uint32_t v = t < ntiles ? (uint32_t)ld[(t / gx) * pitch + (t % gx)] : 0u;
        if (done && t < ntiles && done[t]) v = 0u;                     // completed tiles take no instances in this round
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < GS_WAVE; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += sm[k];
        const uint32_t c = carry;
        if (t < ntiles) { ranges[2 * t] = c + woff + incl - v; ranges[2 * t + 1] = c + woff + incl; }
        __syncthreads();
        if (tid == 1023) carry = c + woff + incl;
        __syncthreads();
    }
}

size_t gs_tile_ranges_scratch_ints(int gx, int gy) { return (size_t)(gx + 1) * (gy + 1) * (DIFF_BLOCKS + 1); }
bool gs_tile_ranges_supported(int gx, int gy) { return (size_t)(gx + 1) * (gy + 1) * sizeof(int) <= 150 * 1024; }

hipError_t gs_launch_tile_ranges(const uint16_t *rect, const uint32_t *perm, int64_t n, int *scratch, int gx, int gy, uint32_t *ranges,
                                 const uint8_t *done, hipStream_t s) {
    const int pitch = gx + 1, cells = pitch * (gy + 1);
    int *partial = scratch, *diff = scratch + (size_t)cells * DIFF_BLOCKS;
    const size_t lds = sizeof(int) * (size_t)cells;
    if (lds > 48 * 1024) {                        // 4K-class grids: opt in to a large dynamic LDS segment
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rect_diff_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(ranges_from_diff_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(rect_diff_kernel, dim3(DIFF_BLOCKS), dim3(256), lds, s, rect, perm, n, partial, pitch, cells);
    hipLaunchKernelGGL(diff_sum_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, partial, diff, cells);
    hipLaunchKernelGGL(ranges_from_diff_kernel, dim3(1), dim3(1024), lds, s, diff, pitch, gx, gy, ranges, done);
    return hipGetLastError();
}

// ---------------------------------------------------------------- chunk owners
// cs[b] = list position s of the gaussian owning instance position min(4096 b, I-1):
// offsets[s] <= q < offsets[s+1] (upper_bound - 1; zero-count gaussians are skipped over).
__global__ void chunk_owner_kernel(const uint32_t *__restrict__ offsets, int64_t n, int64_t n_inst, uint32_t *__restrict__ cs, int nchunks) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nchunks) return;
    int64_t q = (int64_t)b * RS_CHUNK;
    if (q > n_inst - 1) q = n_inst - 1;
    int64_t lo = 0, hi = n;                        // first index in [0, n] with offsets[idx] > q
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)offsets[mid] > q) hi = mid; else lo = mid + 1;
    }
    cs[b] = (uint32_t)(lo - 1);
}

// ---------------------------------------------------------------- load-balanced expand
struct ExpandArgs {
    const uint32_t *offsets;       // exclusive scan of tile counts in list order, n+1 entries
    const uint32_t *perm;          // list position -> gaussian id (null: identity)
    const uint16_t *rect;          // per gaussian x0 x1 y0 y1
    const uint32_t *cs;            // chunk owners, nchunks+1 entries
    int64_t n_inst;
    int gx, nchunks;
    int lo_bits, gid_bits;
    const uint8_t *done;           // per tile: 1 = the tile completed in an earlier round, its instances are dropped (null: none)
};

#define EXP_RCAP 1024     // gaussians of a chunk whose records are staged in LDS (else: global loads)

// own[p] (p < cnt) := 1 + (list position of the owner of instance position base+p) - s_lo ;
// rec[r] = {offset, gaussian id, x0 | y0 << 16, rectangle width} for the chunk's gaussians when they fit
__device__ void expand_owners(uint32_t *own, uint4 *rec, const ExpandArgs &a, int64_t base, int cnt, uint32_t s_lo, uint32_t s_end,
                              uint32_t *sm) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool staged = (s_end - s_lo) < EXP_RCAP;
    for (int i = tid; i < RS_CHUNK; i += RS_THREADS) own[i] = 0;
    __syncthreads();
    for (uint32_t r = tid; r <= s_end - s_lo; r += RS_THREADS) {
        const uint32_t c0 = a.offsets[s_lo + r], c1 = a.offsets[s_lo + r + 1];
        if (c1 > c0) {
            const int64_t q = ((int64_t)c0 > base ? (int64_t)c0 : base) - base;   // first position inside the chunk
            if (q < cnt) {
                own[q] = r + 1;
                if (staged) {
                    const uint32_t gid = a.perm ? a.perm[s_lo + r] : s_lo + r;
                    const uint2 rc = reinterpret_cast<const uint2 *>(a.rect)[gid];
                    const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.y & 0xFFFFu;
                    rec[r] = make_uint4(c0, gid, x0 | (y0 << 16), (rc.x >> 16) - x0 + 1u);
                }
            }
        }
    }
    __syncthreads();
    // inclusive max-scan; thread t owns entries [16t, 16t+16)
    uint32_t v[RS_ITEMS], run = 0;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) { v[i] = own[tid * RS_ITEMS + i]; run = max(run, v[i]); v[i] = run; }
    uint32_t incl = run;
#pragma unroll
    for (int d = 1; d < GS_WAVE; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (lane >= d) incl = max(incl, u); }
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    uint32_t pre = __shfl_up(incl, 1);
    if (lane == 0) pre = 0;
    for (int k = 0; k < w; ++k) pre = max(pre, sm[k]);
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) own[tid * RS_ITEMS + i] = max(v[i], pre);
    __syncthreads();
}

__device__ __forceinline__ void expand_item(const ExpandArgs &a, const uint32_t *own, const uint4 *rec, bool staged, int64_t base, int li,
                                            uint32_t s_lo, uint32_t &tile, uint32_t &gid) {
    const uint32_t r = own[li] - 1u;
    uint32_t off, xy, wdt;
    if (staged) { const uint4 q = rec[r]; off = q.x; gid = q.y; xy = q.z; wdt = q.w; }
    else {
        const uint32_t s = s_lo + r;
        off = a.offsets[s];
        gid = a.perm ? a.perm[s] : s;
        const uint2 rc = reinterpret_cast<const uint2 *>(a.rect)[gid];
        const uint32_t x0 = rc.x & 0xFFFFu;
        xy = x0 | ((rc.y & 0xFFFFu) << 16); wdt = (rc.x >> 16) - x0 + 1u;
    }
    const uint32_t j = (uint32_t)(base + li - (int64_t)off);
    // j / wdt without an integer divide: j < 2^16 * 2^16 is far beyond need (rect area <= 65536 tiles),
    // the float quotient is off by at most one and corrected
    uint32_t q = (uint32_t)((float)j * __builtin_amdgcn_rcpf((float)wdt));
    int rem = (int)(j - q * wdt);
    if (rem < 0) { --q; rem += (int)wdt; } else if (rem >= (int)wdt) { ++q; rem -= (int)wdt; }
    const uint32_t ty = (xy >> 16) + q, tx = (xy & 0xFFFFu) + (uint32_t)rem;
    tile = (ty - 1u) * (uint32_t)a.gx + (tx - 1u);
}

__global__ __launch_bounds__(RS_THREADS) void gen_hist_kernel(ExpandArgs a, uint32_t *__restrict__ block_hist) {
    __shared__ uint32_t own[RS_CHUNK];
    __shared__ uint4 rec[EXP_RCAP];
    __shared__ uint32_t h[RS_RADIX];
    __shared__ uint32_t sm[RS_WAVES];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const int cnt = (int)min((int64_t)RS_CHUNK, a.n_inst - base);
    const uint32_t s_lo = a.cs[blockIdx.x];
    uint32_t s_end = a.cs[blockIdx.x + 1];
    h[tid] = 0;
    // cs[b+1] owns position 4096(b+1) (or I-1): the last position of THIS chunk is owned by it or an earlier one
    expand_owners(own, rec, a, base, cnt, s_lo, s_end, sm);
    const bool staged = (s_end - s_lo) < EXP_RCAP;
    const uint32_t mask = (1u << a.lo_bits) - 1u;
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int li = i * RS_THREADS + tid;
        if (li < cnt) {
            uint32_t tile, gid;
            expand_item(a, own, rec, staged, base, li, s_lo, tile, gid);
            if (!a.done || !a.done[tile]) atomicAdd(&h[tile & mask], 1u);
        }
    }
    __syncthreads();
    block_hist[(size_t)tid * a.nchunks + blockIdx.x] = h[tid];
}

// stable ranking shared by both scatter kernels: wave w owns items [w*1024, (w+1)*1024) in 16
// wave-striped rounds; order = (wave, round, lane)
// Two ways to get a key's stable rank among the wave's earlier same-digit keys:
//  * ballots (portable): 8 wave64 ballots build the set of same-digit lanes, a running LDS counter adds
//    the earlier rounds;
//  * LDS atomic (default): ONE ds_add_rtn_u32 on the wave's counter row.  When several lanes of one wave
//    instruction hit the same LDS address, gfx950 hands out the pre-values in ascending lane order
//    (tools/lds_atomic_order.hip: 0 mismatches in 1.7e8 lane-ops) -- exactly the stable rank.  Measured
//    behaviour, not an architectural guarantee: gs_config.rank_mode = 1 selects the ballot form, and the
//    GPU tests compare every list bit-for-bit against the oracle with both.
__device__ __forceinline__ void rank_round_atomic(uint32_t dg, bool valid, uint32_t *wc, uint32_t &rank) {
    rank = 0;
    if (valid) rank = atomicAdd(&wc[dg], 1u);
}

__device__ __forceinline__ void rank_round(uint32_t dg, bool valid, int lane, volatile uint32_t *wc, uint32_t &rank) {
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((dg >> b) & 1u);
        peers &= ((dg >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = wc[dg];
    rank = before + (uint32_t)__popcll(peers & lt_mask);
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lt_mask) == 0ull) wc[dg] = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
}

// per-wave exclusive offsets (wcnt), chunk digit prefix (lpre); call with all 256 threads
// returns the number of ranked keys of the chunk (== the chunk's key count unless instances were dropped)
__device__ __forceinline__ uint32_t digit_prefixes(uint32_t (*wcnt)[RS_RADIX], uint32_t *lpre, uint32_t *sm) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) { const uint32_t c = wcnt[k][tid]; wcnt[k][tid] = tot; tot += c; }
    uint32_t incl = tot;
#pragma unroll
    for (int d = 1; d < GS_WAVE; d <<= 1) { const uint32_t u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    uint32_t woff = 0, all = 0;
    for (int k = 0; k < RS_WAVES; ++k) { if (k < w) woff += sm[k]; all += sm[k]; }
    lpre[tid] = woff + incl - tot;
    __syncthreads();
    return all;
}

template <bool ATOMIC_RANK>
__global__ __launch_bounds__(RS_THREADS) void gen_scatter_kernel(ExpandArgs a, const uint32_t *__restrict__ block_hist,
                                                                  uint32_t *__restrict__ out) {
    __shared__ uint32_t own[RS_CHUNK];                   // owners, then reused as the reorder buffer
    __shared__ uint4 rec[EXP_RCAP];                      // gaussian records, then reused for the digit bytes
    uint8_t *sdg = reinterpret_cast<uint8_t *>(rec);
    __shared__ uint32_t wcnt[RS_WAVES][RS_RADIX];
    __shared__ uint32_t lpre[RS_RADIX];
    __shared__ uint32_t gbase[RS_RADIX];
    __shared__ uint32_t sm[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const int cnt = (int)min((int64_t)RS_CHUNK, a.n_inst - base);
    const uint32_t s_lo = a.cs[blockIdx.x], s_end = a.cs[blockIdx.x + 1];
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) wcnt[k][tid] = 0;
    gbase[tid] = block_hist[(size_t)tid * a.nchunks + blockIdx.x];
    expand_owners(own, rec, a, base, cnt, s_lo, s_end, sm);
    const bool staged = (s_end - s_lo) < EXP_RCAP;
    const uint32_t mask = (1u << a.lo_bits) - 1u;
    uint32_t val[RS_ITEMS], rank[RS_ITEMS], dgs[RS_ITEMS];
    uint32_t vmask = 0;                                  // bit r: item r is a live instance (inside the chunk, tile not completed)
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int li = w * (GS_WAVE * RS_ITEMS) + r * GS_WAVE + lane;
        bool valid = li < cnt;
        uint32_t tile = 0, gid = 0;
        if (valid) expand_item(a, own, rec, staged, base, li, s_lo, tile, gid);
        if (valid && a.done && a.done[tile]) valid = false;
        if (valid) vmask |= 1u << r;
        dgs[r] = valid ? (tile & mask) : (RS_RADIX - 1);
        val[r] = ((tile >> a.lo_bits) << a.gid_bits) | gid;
        if (ATOMIC_RANK) rank_round_atomic(dgs[r], valid, wcnt[w], rank[r]);
        else rank_round(dgs[r], valid, lane, wcnt[w], rank[r]);
    }
    __syncthreads();                                     // every wave is done reading own[] and rec[]
    const int nlive = (int)digit_prefixes(wcnt, lpre, sm);
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        if (vmask & (1u << r)) {
            const uint32_t p = lpre[dgs[r]] + wcnt[w][dgs[r]] + rank[r];
            own[p] = val[r];
            sdg[p] = (uint8_t)dgs[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int li = r * RS_THREADS + tid;
        if (li < nlive) {
            const uint32_t dg = sdg[li];
            out[(size_t)gbase[dg] + (uint32_t)(li - (int)lpre[dg])] = own[li];
        }
    }
}

// ---------------------------------------------------------------- 32-bit radix pass (high tile digit)
// n_dev != null: the key count is read from the device (a round whose live-instance count is only known there); the
// launch then covers an upper bound and the surplus workgroups contribute empty histograms / write nothing
__global__ __launch_bounds__(RS_THREADS) void rs32_hist_kernel(const uint32_t *__restrict__ keys, int64_t n, int shift, uint32_t mask,
                                                                uint32_t *__restrict__ block_hist, int nblocks, const uint32_t *__restrict__ n_dev) {
    __shared__ uint32_t h[RS_RADIX];
    if (n_dev) n = (int64_t)*n_dev;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    block_hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <bool ATOMIC_RANK>
__global__ __launch_bounds__(RS_THREADS) void rs32_scatter_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int64_t n,
                                                                   int shift, uint32_t mask, uint32_t out_mask,
                                                                   const uint32_t *__restrict__ block_hist, int nblocks,
                                                                   const uint32_t *__restrict__ n_dev) {
    __shared__ uint32_t skeys[RS_CHUNK];
    if (n_dev) n = (int64_t)*n_dev;
    __shared__ uint32_t wcnt[RS_WAVES][RS_RADIX];
    __shared__ uint32_t lpre[RS_RADIX];
    __shared__ uint32_t gbase[RS_RADIX];
    __shared__ uint32_t sm[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const int cnt = (int)max((int64_t)0, min((int64_t)RS_CHUNK, n - base));
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) wcnt[k][tid] = 0;
    gbase[tid] = block_hist[(size_t)tid * nblocks + blockIdx.x];
    __syncthreads();
    uint32_t key[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int li = w * (GS_WAVE * RS_ITEMS) + r * GS_WAVE + lane;
        const bool valid = li < cnt;
        key[r] = valid ? in[base + li] : 0xFFFFFFFFu;
        const uint32_t dg = valid ? ((key[r] >> shift) & mask) : (RS_RADIX - 1);
        if (ATOMIC_RANK) rank_round_atomic(dg, valid, wcnt[w], rank[r]);
        else rank_round(dg, valid, lane, wcnt[w], rank[r]);
    }
    __syncthreads();
    digit_prefixes(wcnt, lpre, sm);
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int li = w * (GS_WAVE * RS_ITEMS) + r * GS_WAVE + lane;
        if (li < cnt) {
            const uint32_t dg = (key[r] >> shift) & mask;
            skeys[lpre[dg] + wcnt[w][dg] + rank[r]] = key[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int li = r * RS_THREADS + tid;
        if (li < cnt) {
            const uint32_t k = skeys[li];
            const uint32_t dg = (k >> shift) & mask;
            out[(size_t)gbase[dg] + (uint32_t)(li - (int)lpre[dg])] = k & out_mask;
        }
    }
}

// live instances of a round = sum of the first pass's digit totals (instances of completed tiles were never counted)
__global__ void sum_digit_totals_kernel(const uint32_t *__restrict__ digit_total, uint32_t *__restrict__ out) {
    uint32_t v = 0;
    for (int i = threadIdx.x; i < RS_RADIX; i += 64) v += digit_total[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if (threadIdx.x == 0) *out = v;
}

// ---------------------------------------------------------------- driver
hipError_t gs_bin2_build_lists(const GsBin2Args &b, hipStream_t s) {
    if (b.n_inst <= 0) return hipSuccess;
    const int nchunks = (int)((b.n_inst + RS_CHUNK - 1) / RS_CHUNK);
    hipLaunchKernelGGL(chunk_owner_kernel, dim3((nchunks + 1 + 255) / 256), dim3(256), 0, s, b.offsets, b.n, b.n_inst, b.cs, nchunks);
    ExpandArgs a{};
    a.offsets = b.offsets; a.perm = b.perm; a.rect = b.rect; a.cs = b.cs; a.n_inst = b.n_inst; a.gx = b.gx; a.nchunks = nchunks;
    a.lo_bits = b.lo_bits; a.gid_bits = b.gid_bits; a.done = b.done;
    hipLaunchKernelGGL(gen_hist_kernel, dim3(nchunks), dim3(RS_THREADS), 0, s, a, b.block_hist);
    hipError_t e = gs_launch_radix_scan(b.block_hist, nchunks, b.digit_total, s);
    if (e != hipSuccess) return e;
    if (b.live_total) hipLaunchKernelGGL(sum_digit_totals_kernel, dim3(1), dim3(64), 0, s, b.digit_total, b.live_total);
    uint32_t *first_out = b.hi_bits > 0 ? b.buf_a : b.ids_out;
    if (b.ballot_ranks) hipLaunchKernelGGL(gen_scatter_kernel<false>, dim3(nchunks), dim3(RS_THREADS), 0, s, a, b.block_hist, first_out);
    else hipLaunchKernelGGL(gen_scatter_kernel<true>, dim3(nchunks), dim3(RS_THREADS), 0, s, a, b.block_hist, first_out);
    if (b.hi_bits > 0) {
        const uint32_t hmask = (1u << b.hi_bits) - 1u, gmask = b.gid_bits >= 32 ? 0xFFFFFFFFu : ((1u << b.gid_bits) - 1u);
        hipLaunchKernelGGL(rs32_hist_kernel, dim3(nchunks), dim3(RS_THREADS), 0, s, b.buf_a, b.n_inst, b.gid_bits, hmask, b.block_hist, nchunks,
                           (const uint32_t *)b.live_total);
        e = gs_launch_radix_scan(b.block_hist, nchunks, b.digit_total, s);
        if (e != hipSuccess) return e;
        if (b.ballot_ranks)
            hipLaunchKernelGGL(rs32_scatter_kernel<false>, dim3(nchunks), dim3(RS_THREADS), 0, s, b.buf_a, b.ids_out, b.n_inst, b.gid_bits, hmask,
                               gmask, b.block_hist, nchunks, (const uint32_t *)b.live_total);
        else
            hipLaunchKernelGGL(rs32_scatter_kernel<true>, dim3(nchunks), dim3(RS_THREADS), 0, s, b.buf_a, b.ids_out, b.n_inst, b.gid_bits, hmask,
                               gmask, b.block_hist, nchunks, (const uint32_t *)b.live_total);
    }
    return hipGetLastError();
}
