// gs_sort.hip -- tile binning for gfx950: instance count/scan/emit, stable LSD radix sort,
// tile ranges.
//
// Replaces the reference's dense tiles x N machinery -- hitBinning (src/binning.jl:3-35), the
// UInt16 CUDA.scan! over the gaussian axis and CUDA.maximum (src/forward.jl:137-141),
// compactHits (src/compact.jl:3-21, one 1024-thread block PER GAUSSIAN) and
// CUDA.sortperm(-tps[3,:]) (src/forward.jl:103) -- by sparse (tile | depth) keys.
//
// The 64-bit key tile<<32 | depth is sorted as a factorised LSD radix sort: the 32 depth bits
// are identical for every tile-instance of a gaussian, so those four digit passes run on the
// N gaussians BEFORE expansion (depth<<32 | id pairs); instances are then emitted in that
// order and only the tile bits (two 8-bit digits) are sorted over the I instances.  Every
// pass is stable, so the result is bit-identical to a stable sort of the full 64-bit keys
// with ties in gaussian-index order.  All of this is HBM-bound integer work: coalesced 8-byte
// streams, LDS histograms and an LDS-staged scatter; no MFMA.
#include "gs_common.h"

#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_CHUNK (RS_THREADS * RS_ITEMS)   // 4096 keys per workgroup
#define RS_RADIX 256
#define RS_WAVES (RS_THREADS / GS_WAVE)

size_t gs_sort_table_entries(int64_t n_max) {
    const int64_t nb = (n_max + RS_CHUNK - 1) / RS_CHUNK;
    return (size_t)(nb > 0 ? nb : 1) * RS_RADIX;
}

// ---------------------------------------------------------------- radix pass: histogram
// keys32 != null (first pass of the depth sort): the keys are the 32-bit depth keys themselves; the pair (key << 32 | index)
// is never stored before the first scatter.  digit_total != null: the 256 digit totals are accumulated here with one atomic
// per (workgroup, digit) -- for the small tables of the depth sort that saves the separate totals launch.
// ---- depth-bucket digit (gs_depth_sort_buckets): the first pass of the two-step depth sort splits the keys into 256 buckets of
// equal DEPTH width over the frame's depth range: b = floor((z(key) - zmin) * 256 / (zmax - zmin)), z(key) the float the key
// encodes (the key is the order-preserving image of +-tps[3]).  fp32 subtraction, multiplication by a positive constant and the
// float -> int conversion are all monotone, so b is monotone in the key and (bucket, key, id) order == (key, id) order.  (Equal
// widths in KEY space -- the float's bit pattern -- were tried first: the C3 depths 26 .. 34 straddle an exponent boundary, the
// far quarter of the depth range got an eighth of the buckets, and its buckets held 7 k of a workgroup's 8 k capacity.)
// The range arrives in DS_SLOTS minima and DS_SLOTS maxima (the preprocess kernel folds every wave's extremes of the FINITE
// depths into slot blockIdx % DS_SLOTS; slots DS_STRIDE words apart so that the atomics spread over the memory channels).
// Keys outside the range (non-finite depths; a stale range) are clamped into the end buckets: correct whatever the range is.
#define DS_BUCKETS 256
#define DS_SLOTS GS_KEY_RANGE_SLOTS
#define DS_STRIDE GS_KEY_RANGE_STRIDE
struct DsMap { uint32_t kmin, kmax; float zmin, scale; };
__device__ __forceinline__ float ds_key_value(uint32_t key) {              // inverse of the depth-key map of gs_preprocess.hip
    return __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}
__device__ __forceinline__ DsMap ds_load_map(const uint32_t *__restrict__ acc, uint32_t *sh2) {
    const int tid = threadIdx.x;                                            // blockDim >= 128
    if (tid < 2 * DS_SLOTS) {
        uint32_t v = acc[(size_t)tid * DS_STRIDE];
#pragma unroll
        for (int d = GS_WAVE / 2; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = tid < DS_SLOTS ? min(v, o) : max(v, o); }
        if ((tid & 63) == 0) sh2[tid >> 6] = v;
    }
    __syncthreads();
    DsMap m;
    const uint32_t lo = sh2[0], hi = sh2[1];
    m.kmin = lo <= hi ? lo : 0u;                                            // (no finite key at all: kmin = kmax = 0, key 0 -> bucket 0, every other key -> bucket 255: one bucket, sorted correctly the slow way)
    m.kmax = lo <= hi ? hi : 0u;
    m.zmin = ds_key_value(m.kmin);
    const float span = ds_key_value(m.kmax) - m.zmin;                       // may overflow to +Inf: scale 0, one bucket
    m.scale = (span > 0.0f && span < 3.0e38f) ? 255.99f / span : 0.0f;
    return m;
}
__device__ __forceinline__ uint32_t ds_bucket(const DsMap &m, uint32_t key) {
    if (key <= m.kmin) return 0u;
    if (key >= m.kmax) return DS_BUCKETS - 1;
    const float t = (ds_key_value(key) - m.zmin) * m.scale;                 // >= 0: kmin < key < kmax
    return min((uint32_t)t, (uint32_t)(DS_BUCKETS - 1));
}

template <int NT, bool BUCKETS = false>
__global__ __launch_bounds__(NT) void rs_hist_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ keys32, int64_t n,
                                                      int shift, uint32_t mask, uint32_t *__restrict__ block_hist, int nblocks,
                                                      uint32_t *__restrict__ digit_total, const uint32_t *__restrict__ range_acc = nullptr) {
    __shared__ uint32_t h[RS_RADIX];
    __shared__ uint32_t sh2[2];
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    uint32_t k32[RS_CHUNK / NT];                         // bucket pass: the keys are requested before the range is folded (one exposed
    if (BUCKETS) {                                       // round trip instead of two: these kernels are latency, not bandwidth)
#pragma unroll
        for (int i = 0; i < RS_CHUNK / NT; ++i) { const int64_t idx = base + (int64_t)i * NT + threadIdx.x; k32[i] = idx < n ? keys32[idx] : 0u; }
    }
    DsMap map{};
    if (BUCKETS) map = ds_load_map(range_acc, sh2);
    if (threadIdx.x < RS_RADIX) h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_CHUNK / NT; ++i) {
        const int64_t idx = base + (int64_t)i * NT + threadIdx.x;
        if (idx < n) {
            const uint32_t dg = BUCKETS ? ds_bucket(map, k32[i])
                                        : keys32 ? ((keys32[idx] >> (shift - 32)) & mask) : ((uint32_t)(keys[idx] >> shift) & mask);
            atomicAdd(&h[dg], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < RS_RADIX) {
        const uint32_t c = h[threadIdx.x];
        block_hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = c;   // [digit][block]
        if (digit_total && c) atomicAdd(&digit_total[threadIdx.x], c);
    }
}

// ---------------------------------------------------------------- radix pass: scan
// One workgroup per digit: exclusive scan of its row of per-block counts, offset by the total
// of all smaller digits (each workgroup re-reduces the rows below it -- nblocks*256 dwords from
// L2, negligible next to the key traffic).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < GS_WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ uint32_t block_reduce_u32(uint32_t v, uint32_t *sm) {
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) v += __shfl_down(v, d);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
    return t;
}

__global__ __launch_bounds__(RS_THREADS) void rs_digit_totals_kernel(const uint32_t *__restrict__ block_hist, int nblocks,
                                                                      uint32_t *__restrict__ digit_total) {
    __shared__ uint32_t sm[RS_WAVES];
    const uint32_t *row = block_hist + (size_t)blockIdx.x * nblocks;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < nblocks; i += RS_THREADS) s += row[i];
    s = block_reduce_u32(s, sm);
    if (threadIdx.x == 0) digit_total[blockIdx.x] = s;
}

// row_total_out != null ("relative" mode of the small sorts): digit_total is not read, the row is scanned from zero and its
// total is stored; the scatter kernel adds the totals of the smaller digits itself (no totals pass, no atomics)
__global__ __launch_bounds__(RS_THREADS) void rs_scan_kernel(uint32_t *__restrict__ block_hist, int nblocks,
                                                              const uint32_t *__restrict__ digit_total, uint32_t *__restrict__ row_total_out) {
    // one workgroup per digit row.  The row is walked in tiles of 256 x 8 consecutive counters: thread t owns the
    // eight counters [8t, 8t+8) of the tile, so the wave's loads and stores are contiguous 2 KiB runs (the earlier
    // thread-owns-a-slice layout read with a stride of nblocks/256 and ran at 0.7 TB/s on 124 K-block tables)
    constexpr int PER = 8, TILE = RS_THREADS * PER;
    __shared__ uint32_t sm[RS_WAVES];
    __shared__ uint32_t carry_s;
    const int d = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t below = 0;
    if (!row_total_out) {
        below = (threadIdx.x < d) ? digit_total[threadIdx.x] : 0u;
        below = block_reduce_u32(below, sm);
        __syncthreads();
    }
    uint32_t *row = block_hist + (size_t)d * nblocks;
    uint32_t carry = below;
    for (int base = 0; base < nblocks; base += TILE) {
        const int i0 = base + threadIdx.x * PER;
        uint32_t v[PER], s = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { v[k] = (i0 + k < nblocks) ? row[i0 + k] : 0u; s += v[k]; }
        const uint32_t incl = wave_incl_scan(s, lane);
        __syncthreads();                                   // sm / carry_s of the previous tile are consumed
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        uint32_t run = carry + incl - s;
        for (int k = 0; k < w; ++k) run += sm[k];
#pragma unroll
        for (int k = 0; k < PER; ++k) { if (i0 + k < nblocks) row[i0 + k] = run; run += v[k]; }
        if (threadIdx.x == RS_THREADS - 1) carry_s = run;  // exclusive prefix of the next tile
        __syncthreads();
        carry = carry_s;
    }
    if (row_total_out && threadIdx.x == 0) row_total_out[d] = carry;
}

// ---------------------------------------------------------------- radix pass: stable scatter
// Wave w owns keys [w*1024, (w+1)*1024) of the chunk in 16 rounds of 64 (wave-striped), so
// the stable rank order is (wave, round, lane).  Ranks come from wave64 ballots (8 per round:
// the set of lanes holding the same digit) and a per-wave running LDS counter; keys are then
// placed digit-contiguously in LDS and written out in runs.
// out32 != null (last pass of a (key | id) pair sort): only the low word -- the id -- is written, as 32 bits.
// NT threads per 4096-key chunk: 256 (sixteen rounds per wave) for the big instance sorts, 1024 (four rounds, sixteen waves)
// for the depth sort, whose 244 workgroups at 1 M gaussians would otherwise leave one wave per SIMD to hide every latency.
template <int NT, bool ATOMIC_RANK, bool BUCKETS = false>
__global__ __launch_bounds__(NT) void rs_scatter_kernel(const uint64_t *__restrict__ in, const uint32_t *__restrict__ in32,
                                                         uint64_t *__restrict__ out,
                                                         int64_t n, int shift, uint32_t mask,
                                                         const uint32_t *__restrict__ block_hist, int nblocks,
                                                         uint32_t *__restrict__ out32, const uint32_t *__restrict__ row_total,
                                                         const uint32_t *__restrict__ range_acc = nullptr) {
    constexpr int NW = NT / GS_WAVE, ITEMS = RS_CHUNK / NT;
    __shared__ uint64_t skeys[RS_CHUNK];                 // 32 KiB
    __shared__ uint32_t wcnt[NW][RS_RADIX];              // running count per (wave, digit)
    __shared__ uint32_t lpre[RS_RADIX];                  // exclusive prefix over digits in this chunk
    __shared__ uint32_t gbase[RS_RADIX];
    __shared__ uint32_t sm[RS_RADIX / GS_WAVE];
    __shared__ uint32_t sh2[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const int64_t remain = n - base;
    const int cnt = remain < RS_CHUNK ? (int)remain : RS_CHUNK;
    uint64_t key[ITEMS];                                 // requested first: the prologue below (range, table entry, totals) hides their latency
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int li = w * (GS_WAVE * ITEMS) + r * GS_WAVE + lane;      // index inside the chunk
        const bool valid = li < cnt;
        key[r] = !valid ? ~0ull : in32 ? (((uint64_t)in32[base + li] << 32) | (uint32_t)(base + li)) : in[base + li];
    }
    DsMap map{};
    if (BUCKETS) map = ds_load_map(range_acc, sh2);
    auto digit_of = [&](uint64_t k) -> uint32_t { return BUCKETS ? ds_bucket(map, (uint32_t)(k >> 32)) : ((uint32_t)(k >> shift) & mask); };
    for (int i = tid; i < NW * RS_RADIX; i += NT) (&wcnt[0][0])[i] = 0;
    {   // table entry (+ in relative mode the totals of the smaller digits: exclusive scan of the 256 row totals)
        uint32_t g = 0, t = 0;
        if (tid < RS_RADIX) { g = block_hist[(size_t)tid * nblocks + blockIdx.x]; if (row_total) t = row_total[tid]; }
        if (row_total) {
            const uint32_t incl = wave_incl_scan(t, lane);
            if (tid < RS_RADIX && lane == 63) sm[w] = incl;
            __syncthreads();
            if (tid < RS_RADIX) { for (int k = 0; k < w; ++k) g += sm[k]; g += incl - t; }
            __syncthreads();
        }
        if (tid < RS_RADIX) gbase[tid] = g;
    }
    __syncthreads();

    uint32_t rank[ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int li = w * (GS_WAVE * ITEMS) + r * GS_WAVE + lane;
        const bool valid = li < cnt;
        const uint32_t dg = valid ? digit_of(key[r]) : (RS_RADIX - 1);
        if (ATOMIC_RANK) {                                              // see gs_bin2.hip rank_round_atomic
            rank[r] = 0;
            if (valid) rank[r] = atomicAdd(&wcnt[w][dg], 1u);
        } else {
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bal = __ballot((dg >> b) & 1u);
                peers &= ((dg >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t before = wcnt[w][dg];                        // same-digit keys of earlier rounds
            rank[r] = before + (uint32_t)__popcll(peers & lt_mask);
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & lt_mask) == 0ull) wcnt[w][dg] = before + (uint32_t)__popcll(peers);   // group leader
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // thread `tid` == digit: per-wave exclusive offsets and the chunk's digit prefix
    uint32_t tot = 0;
    if (tid < RS_RADIX) {
#pragma unroll
        for (int k = 0; k < NW; ++k) { const uint32_t c = wcnt[k][tid]; wcnt[k][tid] = tot; tot += c; }
    }
    const uint32_t incl = wave_incl_scan(tot, lane);
    if (tid < RS_RADIX && lane == 63) sm[w] = incl;
    __syncthreads();
    if (tid < RS_RADIX) {
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += sm[k];
        lpre[tid] = woff + incl - tot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int li = w * (GS_WAVE * ITEMS) + r * GS_WAVE + lane;
        if (li < cnt) {
            const uint32_t dg = digit_of(key[r]);
            skeys[lpre[dg] + wcnt[w][dg] + rank[r]] = key[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int li = r * NT + tid;
        if (li < cnt) {
            const uint64_t k = skeys[li];
            const uint32_t dg = digit_of(k);
            const size_t o = (size_t)gbase[dg] + (uint32_t)(li - (int)lpre[dg]);
            if (out32) out32[o] = (uint32_t)k; else out[o] = k;
        }
    }
}

// ---------------------------------------------------------------- lane-order probe of the LDS atomic rank
// 256 workgroups x 32 rounds: every wave draws digits from a hash (all-distinct, few-valued, constant and 32-valued
// patterns, ~6 % inactive lanes), takes atomicAdd-return on an LDS counter row and compares the value with the ballot
// rank (number of lower active lanes with the same digit).  Any mismatch makes gs_create fall back to ballot ranks.
__global__ __launch_bounds__(256) void lds_atomic_order_probe_kernel(unsigned *__restrict__ bad) {
    __shared__ unsigned cnt[4][RS_RADIX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned nbad = 0;
    for (int r = 0; r < 32; ++r) {
        for (int i = lane; i < RS_RADIX; i += 64) cnt[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
        unsigned h = (unsigned)(((blockIdx.x * 4 + w) * 32 + r) * 64 + lane) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const int mode = (r + w) & 3;
        const unsigned d = mode == 0 ? (h >> 24) : mode == 1 ? ((h >> 24) & 7u) : mode == 2 ? 5u : ((h >> 24) & 31u);
        const bool active = ((h >> 8) & 15u) != 0u;
        unsigned got = 0;
        if (active) got = atomicAdd(&cnt[w][d], 1u);
        unsigned long long peers = __ballot(active);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const unsigned long long bal = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? bal : ~bal; }
        const unsigned want = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        if (active && got != want) ++nbad;
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
}

hipError_t gs_probe_lds_atomic_order(hipStream_t s, int *mismatches) {
    unsigned *d = nullptr, h = 0;
    hipError_t e = hipMalloc(&d, sizeof(unsigned));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d, 0, sizeof(unsigned), s);
    if (e == hipSuccess) { hipLaunchKernelGGL(lds_atomic_order_probe_kernel, dim3(256), dim3(256), 0, s, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    *mismatches = (int)h;
    return e;
}

hipError_t gs_launch_radix_scan(uint32_t *block_hist, int nblocks, uint32_t *digit_total, hipStream_t stream) {
    hipLaunchKernelGGL(rs_digit_totals_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, stream, block_hist, nblocks, digit_total);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, stream, block_hist, nblocks, digit_total, (uint32_t *)nullptr);
    return hipGetLastError();
}

// keys32 != null: the first pass reads the 32-bit keys and forms (key << 32 | index) on the fly (`a` is then only scratch).
// digit_total must hold 4 x 256 words.
hipError_t gs_radix_sort_u64(uint64_t *a, uint64_t *b, int64_t n, int bit_lo, int bit_hi,
                             uint32_t *block_hist, uint32_t *digit_total, int *result_in_b, hipStream_t stream, bool ballot_ranks,
                             uint32_t *final_low32, const uint32_t *keys32) {
    *result_in_b = 0;
    if (n <= 0) return hipSuccess;
    const int nblocks = (int)((n + RS_CHUNK - 1) / RS_CHUNK);
    uint64_t *src = a, *dst = b;
    // equal-width digits of at most 8 bits: fewer bins per pass = longer contiguous runs in the scatter
    const int total = bit_hi - bit_lo;
    const int passes = (total + 7) / 8;
    const int width = (total + passes - 1) / passes;
    // small sorts (the depth sort: at most 2048 chunks) run 1024 threads per chunk -- same chunks and table, four times the
    // waves in flight -- and need no totals pass: the scan stores each digit row's total (digit_total + 256 pass) and the
    // scatter adds up the totals of the smaller digits itself.  3 launches per pass, no atomics.
    const bool small = nblocks <= 2048 && passes <= 4;
    int pass = 0;
    for (int shift = bit_lo; shift < bit_hi; shift += width, ++pass) {
        const int bits = (bit_hi - shift) < width ? (bit_hi - shift) : width;
        const uint32_t mask = (1u << bits) - 1u;
        const uint32_t *k32 = pass == 0 ? keys32 : nullptr;
        uint32_t *o32 = (final_low32 && shift + width >= bit_hi) ? final_low32 : nullptr;      // last pass: ids only
        if (small) {
            uint32_t *tot = digit_total + pass * RS_RADIX;
            hipLaunchKernelGGL(rs_hist_kernel<1024>, dim3(nblocks), dim3(1024), 0, stream, src, k32, n, shift, mask, block_hist, nblocks, (uint32_t *)nullptr);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, stream, block_hist, nblocks, (const uint32_t *)nullptr, tot);
            if (ballot_ranks) hipLaunchKernelGGL((rs_scatter_kernel<1024, false>), dim3(nblocks), dim3(1024), 0, stream, src, k32, dst, n, shift, mask, block_hist, nblocks, o32, tot);
            else hipLaunchKernelGGL((rs_scatter_kernel<1024, true>), dim3(nblocks), dim3(1024), 0, stream, src, k32, dst, n, shift, mask, block_hist, nblocks, o32, tot);
        } else {
            hipLaunchKernelGGL(rs_hist_kernel<RS_THREADS>, dim3(nblocks), dim3(RS_THREADS), 0, stream, src, k32, n, shift, mask, block_hist, nblocks, (uint32_t *)nullptr);
            hipLaunchKernelGGL(rs_digit_totals_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, stream, block_hist, nblocks, digit_total);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, stream, block_hist, nblocks, digit_total, (uint32_t *)nullptr);
            if (ballot_ranks) hipLaunchKernelGGL((rs_scatter_kernel<RS_THREADS, false>), dim3(nblocks), dim3(RS_THREADS), 0, stream, src, k32, dst, n, shift, mask, block_hist, nblocks, o32, (const uint32_t *)nullptr);
            else hipLaunchKernelGGL((rs_scatter_kernel<RS_THREADS, true>), dim3(nblocks), dim3(RS_THREADS), 0, stream, src, k32, dst, n, shift, mask, block_hist, nblocks, o32, (const uint32_t *)nullptr);
        }
        uint64_t *t = src; src = dst; dst = t;
        *result_in_b ^= 1;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- depth sort in two steps: 256 key-range buckets, then LDS
// Replaces the twelve dependent launches of the four-pass LSD sort of the (depth | id) pairs (CUDA.sortperm, src/forward.jl:103)
// by four: histogram, scan and stable scatter of ONE pass whose digit is the key's bucket (ds_bucket: 256 buckets of equal key
// width over the frame's key range), then ds_local_kernel: one workgroup per bucket finishes the order inside LDS.  A bucket's
// elements arrive in index order (the scatter is stable), so a stable LSD sort on (key - the bucket's smallest key) -- as many
// 8-bit digits as the bucket's key spread needs: two at C3 -- gives (key, id) order.  Same permutation as the classic path, bit
// for bit (tests/test_gpu_dsort.py).  A bucket that does not fit the registers of its workgroup (8192 elements: a pathological
// depth distribution, e.g. a wall of gaussians at one depth plus an outlier) is sorted by the same workgroup through global
// memory, chunk by chunk (slow but correct), and reported to the host (host_stat), which returns to the classic path.
#define DS_ITEMS 8
#define DS_PATHOLOGICAL 8
#define DS_CAP_OF(NT) ((NT) * DS_ITEMS)

// Element `li` of the (at most NT * 8) elements a workgroup holds in registers: wave w owns the `per` consecutive elements from
// w * per on, round r of the wave the 64 from r * 64 on (per = a multiple of 64 sized to the element count, so that every wave
// has work and a short bucket costs few rounds).  (wave, round, lane) ascending == li ascending: the stable order.
#define DS_LI(w, r, lane, per) ((w) * (per) + (r) * GS_WAVE + (lane))

// One stable counting-sort step on digit (sub >> shift) & 255 of the elements held in registers (valid if li < cnt; `rounds`
// rounds per wave): afterwards skey / sid hold them digit-contiguously, lpre[d] is the first slot of digit d and ltot[d] their
// number.  Ranks from wave ballots (as rs_scatter_kernel).
template <int NT>
__device__ __forceinline__ void ds_stage(const uint32_t (&sub)[DS_ITEMS], const uint32_t (&id)[DS_ITEMS], const int cnt, const int per, const int rounds,
                                         const int shift, uint32_t *skey, uint32_t *sid, uint32_t (*wcnt)[RS_RADIX], uint32_t *lpre, uint32_t *ltot,
                                         uint32_t *sm) {
    constexpr int NW = NT / GS_WAVE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < NW * RS_RADIX; i += NT) (&wcnt[0][0])[i] = 0;
    __syncthreads();                                                    // (also: every reader of skey / sid of the previous step is done)
    uint32_t rank[DS_ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < DS_ITEMS; ++r) {
        rank[r] = 0;
        if (r < rounds && DS_LI(w, r, 0, per) < cnt) {                  // wave-uniform
            const int li = DS_LI(w, r, lane, per);
            const bool valid = li < cnt;
            const uint32_t dg = valid ? ((sub[r] >> shift) & 255u) : 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bal = __ballot((dg >> b) & 1u);
                peers &= ((dg >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t before = wcnt[w][dg];
            rank[r] = before + (uint32_t)__popcll(peers & lt_mask);
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & lt_mask) == 0ull) wcnt[w][dg] = before + (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    uint32_t tot = 0;
    if (tid < RS_RADIX) {
#pragma unroll
        for (int k = 0; k < NW; ++k) { const uint32_t c = wcnt[k][tid]; wcnt[k][tid] = tot; tot += c; }
    }
    const uint32_t incl = wave_incl_scan(tot, lane);
    if (tid < RS_RADIX && lane == 63) sm[w] = incl;
    __syncthreads();
    if (tid < RS_RADIX) {
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += sm[k];
        lpre[tid] = woff + incl - tot;
        ltot[tid] = tot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < DS_ITEMS; ++r) {
        const int li = DS_LI(w, r, lane, per);
        if (r < rounds && li < cnt) {
            const uint32_t dg = (sub[r] >> shift) & 255u;
            const uint32_t o = lpre[dg] + wcnt[w][dg] + rank[r];
            skey[o] = sub[r]; sid[o] = id[r];
        }
    }
    __syncthreads();
}

template <int NT>
__device__ __forceinline__ void ds_block_minmax(uint32_t &mn, uint32_t &mx, uint32_t *sm) {     // sm: 2 * NW words
    constexpr int NW = NT / GS_WAVE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, d)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, d)); }
    __syncthreads();
    if (lane == 0) { sm[w] = mn; sm[NW + w] = mx; }
    __syncthreads();
    mn = sm[0]; mx = sm[NW];
#pragma unroll
    for (int k = 1; k < NW; ++k) { mn = min(mn, sm[k]); mx = max(mx, sm[NW + k]); }
    __syncthreads();
}

// pairs: the bucket-ordered (key << 32 | id) pairs; scratch: as many words again (only touched by oversize buckets);
// bucket_total: the 256 bucket sizes (row totals of the scan); reset_acc: the range accumulators of the OTHER frame parity,
// re-armed here for the next preprocess (nothing reads them during this frame); host_stat: coherent pinned word, receives the
// size of an oversize bucket (the host zeroes it before the launch).  NT: 1024 threads (buckets of up to 8192), 256 for small
// models (2048: the same four waves do everything, fewer to synchronise).
template <int NT>
__global__ __launch_bounds__(NT) void ds_local_kernel(uint64_t *__restrict__ pairs, uint64_t *__restrict__ scratch,
                                                       const uint32_t *__restrict__ bucket_total, uint32_t *__restrict__ perm,
                                                       uint32_t *__restrict__ reset_acc, uint32_t *__restrict__ host_stat) {
    constexpr int NW = NT / GS_WAVE, CAP = DS_CAP_OF(NT);
    __shared__ uint32_t skey[CAP];
    __shared__ uint32_t sid[CAP];
    __shared__ uint32_t wcnt[NW][RS_RADIX];
    __shared__ uint32_t lpre[RS_RADIX], ltot[RS_RADIX], gbase[RS_RADIX];
    __shared__ uint32_t sm[2 * NW + 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
    if (b == 0 && reset_acc) for (int i = tid; i < 2 * DS_SLOTS; i += NT) reset_acc[(size_t)i * DS_STRIDE] = i < DS_SLOTS ? 0xFFFFFFFFu : 0u;
    // my bucket: [start, start + cnt) = exclusive prefix of the 256 totals (thread t < 256 holds total t)
    uint32_t start = 0, cnt = 0;
    {
        const uint32_t t = tid < RS_RADIX ? bucket_total[tid] : 0u;
        const uint32_t incl = wave_incl_scan(t, lane);
        if (tid < RS_RADIX && lane == 63) sm[w] = incl;
        __syncthreads();
        if (tid == b) { uint32_t woff = 0; for (int k = 0; k < w; ++k) woff += sm[k]; sm[8] = woff + incl - t; sm[9] = t; }
        __syncthreads();
        start = sm[8]; cnt = sm[9];
        __syncthreads();
    }
    if (cnt == 0) return;
    uint32_t sub[DS_ITEMS], id[DS_ITEMS];
    if (cnt <= (uint32_t)CAP) {
        // ---- the bucket lives in registers / LDS from here to the final store
        const int per = (int)((cnt + NT - 1) / NT) * GS_WAVE, rounds = per / GS_WAVE;
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
        for (int r = 0; r < DS_ITEMS; ++r) {
            const int li = DS_LI(w, r, lane, per);
            sub[r] = 0; id[r] = 0;
            if (r < rounds && li < (int)cnt) { const uint64_t p = pairs[(size_t)start + li]; sub[r] = (uint32_t)(p >> 32); id[r] = (uint32_t)p; mn = min(mn, sub[r]); mx = max(mx, sub[r]); }
        }
        ds_block_minmax<NT>(mn, mx, sm);
        const uint32_t spread = mx - mn;
        const int bits = spread ? 32 - __builtin_clz(spread) : 0, passes = (bits + 7) >> 3;
        if (passes == 0) {                                              // one key value: index order is the order
#pragma unroll
            for (int r = 0; r < DS_ITEMS; ++r) { const int li = DS_LI(w, r, lane, per); if (r < rounds && li < (int)cnt) perm[(size_t)start + li] = id[r]; }
            return;
        }
#pragma unroll
        for (int r = 0; r < DS_ITEMS; ++r) sub[r] -= mn;
        for (int p = 0; p < passes; ++p) {
            ds_stage<NT>(sub, id, (int)cnt, per, rounds, 8 * p, skey, sid, wcnt, lpre, ltot, sm);
            if (p + 1 < passes) {
#pragma unroll
                for (int r = 0; r < DS_ITEMS; ++r) {
                    const int li = DS_LI(w, r, lane, per);
                    if (r < rounds && li < (int)cnt) { sub[r] = skey[li]; id[r] = sid[li]; }
                }
            }
        }
        for (int li = tid; li < (int)cnt; li += NT) perm[(size_t)start + li] = sid[li];
        return;
    }
    // ---- oversize bucket: LSD passes through global memory, CAP elements at a time (pairs <-> scratch).  A few chunks per bucket are
    // the normal case for models beyond 1.3 M gaussians (C5: 19.5 k pairs per bucket, three chunks); only a bucket of more than
    // DS_PATHOLOGICAL chunks is reported to the host (one workgroup would sort a large share of the model alone)
    if (tid == 0 && host_stat && cnt > (uint32_t)(DS_PATHOLOGICAL * CAP)) *host_stat = cnt;
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    for (uint32_t i = tid; i < cnt; i += NT) { const uint32_t k = (uint32_t)(pairs[(size_t)start + i] >> 32); mn = min(mn, k); mx = max(mx, k); }
    ds_block_minmax<NT>(mn, mx, sm);
    const uint32_t spread = mx - mn;
    const int bits = spread ? 32 - __builtin_clz(spread) : 0, passes = (bits + 7) >> 3;
    if (passes == 0) {
        for (uint32_t i = tid; i < cnt; i += NT) perm[(size_t)start + i] = (uint32_t)pairs[(size_t)start + i];
        return;
    }
    uint64_t *src = pairs + start, *dst = scratch + start;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        if (tid < RS_RADIX) gbase[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += NT) atomicAdd(&gbase[(((uint32_t)(src[i] >> 32) - mn) >> shift) & 255u], 1u);
        __syncthreads();
        {                                                               // exclusive prefix over the digits
            const uint32_t t = tid < RS_RADIX ? gbase[tid] : 0u;
            const uint32_t incl = wave_incl_scan(t, lane);
            if (tid < RS_RADIX && lane == 63) sm[w] = incl;
            __syncthreads();
            if (tid < RS_RADIX) { uint32_t woff = 0; for (int k = 0; k < w; ++k) woff += sm[k]; gbase[tid] = woff + incl - t; }
            __syncthreads();
        }
        for (uint32_t c0 = 0; c0 < cnt; c0 += CAP) {
            const int nc = (int)min((uint32_t)CAP, cnt - c0);
            const int per = ((nc + NT - 1) / NT) * GS_WAVE, rounds = per / GS_WAVE;
#pragma unroll
            for (int r = 0; r < DS_ITEMS; ++r) {
                const int li = DS_LI(w, r, lane, per);
                sub[r] = 0; id[r] = 0;
                if (r < rounds && li < nc) { const uint64_t pr = src[(size_t)c0 + li]; sub[r] = (uint32_t)(pr >> 32) - mn; id[r] = (uint32_t)pr; }
            }
            ds_stage<NT>(sub, id, nc, per, rounds, shift, skey, sid, wcnt, lpre, ltot, sm);
            for (int li = tid; li < nc; li += NT) {
                const uint32_t k = skey[li], dg = (k >> shift) & 255u;
                const size_t o = (size_t)gbase[dg] + (uint32_t)(li - (int)lpre[dg]);
                if (p + 1 < passes) dst[o] = ((uint64_t)(k + mn) << 32) | sid[li];
                else perm[(size_t)start + o] = sid[li];
            }
            __syncthreads();
            if (tid < RS_RADIX) gbase[tid] += ltot[tid];
            __syncthreads();
        }
        uint64_t *t = src; src = dst; dst = t;
    }
}

__global__ void ds_reset_range_kernel(uint32_t *__restrict__ acc, int nparity) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nparity * 2 * DS_SLOTS) acc[(size_t)i * DS_STRIDE] = (i % (2 * DS_SLOTS)) < DS_SLOTS ? 0xFFFFFFFFu : 0u;
}
size_t gs_depth_range_words() { return (size_t)2 * 2 * DS_SLOTS * DS_STRIDE; }           // two frame parities
size_t gs_depth_range_parity_words() { return (size_t)2 * DS_SLOTS * DS_STRIDE; }
int64_t gs_depth_buckets_max_n() { return (int64_t)DS_BUCKETS * DS_CAP_OF(1024) * 4; }     // mean bucket of four chunks (8.4 M gaussians)
hipError_t gs_depth_range_reset(uint32_t *acc, hipStream_t s, int nparity) {
    hipLaunchKernelGGL(ds_reset_range_kernel, dim3(1), dim3(256), 0, s, acc, nparity);
    return hipGetLastError();
}

// perm = ids in (key, id) order of the n 32-bit keys.  pairs_a / pairs_b: n words each of 64 bits; table / digit_total as for
// gs_radix_sort_u64; range_acc: this frame's key-range accumulators (filled by the preprocess kernel), reset_acc: the other parity's.
hipError_t gs_depth_sort_buckets(const uint32_t *keys32, uint64_t *pairs_a, uint64_t *pairs_b, int64_t n, uint32_t *block_hist,
                                 uint32_t *digit_total, uint32_t *perm, const uint32_t *range_acc, uint32_t *reset_acc,
                                 uint32_t *host_stat, hipStream_t stream, bool ballot_ranks) {
    if (n <= 0) return hipSuccess;
    const int nblocks = (int)((n + RS_CHUNK - 1) / RS_CHUNK);
    hipLaunchKernelGGL((rs_hist_kernel<1024, true>), dim3(nblocks), dim3(1024), 0, stream, (const uint64_t *)nullptr, keys32, n, 0, 0u, block_hist, nblocks,
                       (uint32_t *)nullptr, range_acc);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, stream, block_hist, nblocks, (const uint32_t *)nullptr, digit_total);
    if (ballot_ranks) hipLaunchKernelGGL((rs_scatter_kernel<1024, false, true>), dim3(nblocks), dim3(1024), 0, stream, (const uint64_t *)nullptr, keys32, pairs_b, n, 0, 0u,
                                         block_hist, nblocks, (uint32_t *)nullptr, digit_total, range_acc);
    else hipLaunchKernelGGL((rs_scatter_kernel<1024, true, true>), dim3(nblocks), dim3(1024), 0, stream, (const uint64_t *)nullptr, keys32, pairs_b, n, 0, 0u,
                            block_hist, nblocks, (uint32_t *)nullptr, digit_total, range_acc);
    // small models: four waves per bucket (capacity 2048: eight times the mean bucket at 64 K gaussians)
    if (n <= 65536) hipLaunchKernelGGL(ds_local_kernel<256>, dim3(DS_BUCKETS), dim3(256), 0, stream, pairs_b, pairs_a, digit_total, perm, reset_acc, host_stat);
    else hipLaunchKernelGGL(ds_local_kernel<1024>, dim3(DS_BUCKETS), dim3(1024), 0, stream, pairs_b, pairs_a, digit_total, perm, reset_acc, host_stat);
    return hipGetLastError();
}

// ---------------------------------------------------------------- depth pairs
__global__ void depth_pairs_kernel(const uint32_t *__restrict__ key, uint64_t *__restrict__ pairs, int64_t n) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) pairs[g] = ((uint64_t)key[g] << 32) | (uint32_t)g;
}
__global__ void unpack_perm_kernel(const uint64_t *__restrict__ pairs, uint32_t *__restrict__ perm, int64_t n) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) perm[g] = (uint32_t)pairs[g];
}
hipError_t gs_launch_depth_pairs(const uint32_t *depth_key, uint64_t *pairs, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(depth_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, depth_key, pairs, n);
    return hipGetLastError();
}
hipError_t gs_launch_unpack_perm(const uint64_t *pairs, uint32_t *perm, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pairs, perm, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- instance count + exclusive scan
#define SC_THREADS 256
#define SC_ITEMS 8
#define SC_CHUNK (SC_THREADS * SC_ITEMS)

__device__ __forceinline__ uint32_t rect_area(const uint16_t *__restrict__ rect, int64_t g) {
    const uint2 r = reinterpret_cast<const uint2 *>(rect)[g];
    const uint32_t x0 = r.x & 0xFFFFu, x1 = r.x >> 16, y0 = r.y & 0xFFFFu, y1 = r.y >> 16;
    return x0 == 0u ? 0u : (x1 - x0 + 1u) * (y1 - y0 + 1u);
}

// pass 1: per-chunk totals
// The gathered counts (rect[perm[idx]] is a random 8-byte read per gaussian) are parked in offsets[idx], so that pass 3
// streams them back instead of gathering a second time.
__global__ __launch_bounds__(SC_THREADS) void count_reduce_kernel(const uint16_t *__restrict__ rect, const uint32_t *__restrict__ perm,
                                                                   int64_t n, uint32_t *__restrict__ block_sums, uint32_t *__restrict__ counts) {
    __shared__ uint32_t sm[SC_THREADS / GS_WAVE];
    const int64_t base = (int64_t)blockIdx.x * SC_CHUNK;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * SC_THREADS + threadIdx.x;
        if (idx < n) { const uint32_t c = rect_area(rect, perm ? (int64_t)perm[idx] : idx); counts[idx] = c; s += c; }
    }
    s = block_reduce_u32(s, sm);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s;
}
// pass 2: one workgroup scans the chunk totals in place (exclusive) and stores the grand total
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(uint32_t *__restrict__ block_sums, int nb, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t sm[16];
    __shared__ uint32_t carry;
    __shared__ unsigned long long wide[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    unsigned long long t64 = 0;                          // the same total in 64 bits: offsets are 32-bit, so it must fit
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < nb) ? block_sums[i] : 0u;
        t64 += v;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += sm[k];
        const uint32_t c = carry;
        if (i < nb) block_sums[i] = c + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + incl;
        __syncthreads();
    }
#pragma unroll
    for (int d = GS_WAVE / 2; d > 0; d >>= 1) t64 += __shfl_down(t64, d);
    if (lane == 0) wide[w] = t64;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int k = 0; k < 16; ++k) tot += wide[k];
        *total_out = tot >= 0xFFFFFFFFull ? 0xFFFFFFFFu : carry;       // sentinel: too many tile instances (gs_bin refuses)
    }
}
// pass 3: per-chunk exclusive scan with the chunk base.  Thread t owns SC_ITEMS consecutive
// gaussians so the scan order is the list order.
__global__ __launch_bounds__(SC_THREADS) void count_scan_kernel(int64_t n, const uint32_t *__restrict__ block_sums,
                                                                 uint32_t *__restrict__ offsets) {
    __shared__ uint32_t sm[SC_THREADS / GS_WAVE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SC_CHUNK + (int64_t)threadIdx.x * SC_ITEMS;
    uint32_t c[SC_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        const int64_t idx = base + i;
        c[i] = (idx < n) ? offsets[idx] : 0u;                            // the count parked by pass 1
        s += c[i];
    }
    const uint32_t incl = wave_incl_scan(s, lane);
    if (lane == 63) sm[w] = incl;
    __syncthreads();
    uint32_t off = block_sums[blockIdx.x] + incl - s;
    for (int k = 0; k < w; ++k) off += sm[k];
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        const int64_t idx = base + i;
        if (idx < n) offsets[idx] = off;
        off += c[i];
    }
}

// ---------------------------------------------------------------- later rounds of a slab frame
// live2d[y * (gx + 1) + x] = number of NOT completed tiles with tx < x and ty < y (0-based): 2-D prefix sums with a zero
// border, from the per-tile done flags the forward of the previous round wrote.  Two small kernels: one wave per tile row
// (shuffle scan along x), then one thread per column adding the rows up (independent loads, coalesced across the columns).
__global__ __launch_bounds__(256) void live_rows_kernel(const uint8_t *__restrict__ done, int gx, int gy, uint32_t *__restrict__ rowp) {
    const int lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= gy) return;
    const int pitch = gx + 1;
    uint32_t carry = 0;
    if (lane == 0) rowp[(y + 1) * pitch] = 0;
    for (int x0 = 0; x0 < gx; x0 += 64) {
        const int x = x0 + lane;
        const uint32_t v = (x < gx && !done[y * gx + x]) ? 1u : 0u;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (x < gx) rowp[(y + 1) * pitch + x + 1] = carry + incl;
        carry += __shfl(incl, 63);
    }
}
__global__ __launch_bounds__(256) void live_cols_kernel(const uint32_t *__restrict__ rowp, int gx, int gy, uint32_t *__restrict__ live2d) {
    const int x = blockIdx.x * 256 + threadIdx.x, pitch = gx + 1;
    if (x > gx) return;
    live2d[x] = 0;
    uint32_t acc = 0;
#pragma unroll 8
    for (int y = 1; y <= gy; ++y) { acc += rowp[y * pitch + x]; live2d[y * pitch + x] = acc; }
}
hipError_t gs_launch_live_prefix(const uint8_t *done, int gx, int gy, uint32_t *live2d, uint32_t *rowp_scratch, hipStream_t s) {
    hipLaunchKernelGGL(live_rows_kernel, dim3((gy + 3) / 4), dim3(256), 0, s, done, gx, gy, rowp_scratch);
    hipLaunchKernelGGL(live_cols_kernel, dim3((gx + 1 + 255) / 256), dim3(256), 0, s, rowp_scratch, gx, gy, live2d);
    return hipGetLastError();
}

// count pass of a later round: a gaussian whose tile rectangle holds no live tile drops out of the round (its entry of
// rect_out becomes the empty rectangle); the others are shrunk to the bounding box of the live tiles inside their
// rectangle (four binary searches on the 2-D prefix sums), so that the generate passes enumerate as few completed tiles as
// possible -- the instances of the completed tiles that remain inside the box are dropped there (gs_bin2.hip, ExpandArgs.done).
__device__ __forceinline__ uint32_t live_count(const uint32_t *__restrict__ L, int pitch, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {
    return L[y1 * pitch + x1] - L[(y0 - 1) * pitch + x1] - L[y1 * pitch + (x0 - 1)] + L[(y0 - 1) * pitch + (x0 - 1)];   // 1-based inclusive
}
__global__ __launch_bounds__(SC_THREADS) void count_reduce_live_kernel(const uint16_t *__restrict__ rect, const uint32_t *__restrict__ perm,
                                                                        int64_t n, const uint32_t *__restrict__ live2d, int gx,
                                                                        uint16_t *__restrict__ rect_out, uint32_t *__restrict__ block_sums,
                                                                        uint32_t *__restrict__ counts) {
    __shared__ uint32_t sm[SC_THREADS / GS_WAVE];
    const int64_t base = (int64_t)blockIdx.x * SC_CHUNK;
    const int pitch = gx + 1;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        const int64_t idx = base + (int64_t)i * SC_THREADS + threadIdx.x;
        if (idx < n) {
            const int64_t g = perm ? (int64_t)perm[idx] : idx;
            uint2 r = reinterpret_cast<const uint2 *>(rect)[g];
            uint32_t x0 = r.x & 0xFFFFu, x1 = r.x >> 16, y0 = r.y & 0xFFFFu, y1 = r.y >> 16;
            uint32_t c = 0;
            if (x0 != 0u && live_count(live2d, pitch, x0, x1, y0, y1) != 0u) {
                uint32_t lo = y0, hi = y1;                               // first live row
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (live_count(live2d, pitch, x0, x1, y0, mid)) hi = mid; else lo = mid + 1; }
                y0 = lo; hi = y1;                                        // last live row
                while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (live_count(live2d, pitch, x0, x1, mid, y1)) lo = mid; else hi = mid - 1; }
                y1 = lo;
                lo = x0; hi = x1;                                        // first / last live column inside those rows
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (live_count(live2d, pitch, x0, mid, y0, y1)) hi = mid; else lo = mid + 1; }
                x0 = lo; hi = x1;
                while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (live_count(live2d, pitch, mid, x1, y0, y1)) lo = mid; else hi = mid - 1; }
                x1 = lo;
                c = (x1 - x0 + 1u) * (y1 - y0 + 1u);
                r = make_uint2(x0 | (x1 << 16), y0 | (y1 << 16));
            } else r = make_uint2(0u, 0u);
            reinterpret_cast<uint2 *>(rect_out)[g] = r;
            counts[idx] = c; s += c;
        }
    }
    s = block_reduce_u32(s, sm);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s;
}
hipError_t gs_launch_count_scan_live(const uint16_t *rect, const uint32_t *perm, const uint32_t *live2d, int gx, uint16_t *rect_out,
                                     uint32_t *offsets, uint32_t *block_sums, int64_t n, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(offsets, 0, sizeof(uint32_t), s);
    const int nb = (int)((n + SC_CHUNK - 1) / SC_CHUNK);
    hipLaunchKernelGGL(count_reduce_live_kernel, dim3(nb), dim3(SC_THREADS), 0, s, rect, perm, n, live2d, gx, rect_out, block_sums, offsets);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb, offsets + n);
    hipLaunchKernelGGL(count_scan_kernel, dim3(nb), dim3(SC_THREADS), 0, s, n, block_sums, offsets);
    return hipGetLastError();
}

hipError_t gs_launch_count_scan(const uint16_t *rect, const uint32_t *perm, uint32_t *offsets, uint32_t *block_sums,
                                int64_t n, hipStream_t s) {
    if (n <= 0) return hipMemsetAsync(offsets, 0, sizeof(uint32_t), s);
    const int nb = (int)((n + SC_CHUNK - 1) / SC_CHUNK);
    hipLaunchKernelGGL(count_reduce_kernel, dim3(nb), dim3(SC_THREADS), 0, s, rect, perm, n, block_sums, offsets);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb, offsets + n);
    hipLaunchKernelGGL(count_scan_kernel, dim3(nb), dim3(SC_THREADS), 0, s, n, block_sums, offsets);
    return hipGetLastError();
}

// ---------------------------------------------------------------- instance emission
// One wave per 64 list positions; the wave walks its 64 gaussians and writes each one's
// tile instances with all lanes (coalesced 8-byte stores), so a gaussian covering hundreds of
// tiles does not serialise on one lane.  Key = tile<<32 | gaussian id, tile = (ty-1)*gx+(tx-1)
// (SURVEY 8a A6).
__global__ __launch_bounds__(256) void emit_kernel(const uint16_t *__restrict__ rect, const uint32_t *__restrict__ perm,
                                                    const uint32_t *__restrict__ offsets, uint64_t *__restrict__ inst,
                                                    int64_t n, int gx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t s = wave * GS_WAVE + lane;
    uint32_t g = 0, off = 0, x0 = 0, y0 = 0, wdt = 0, cnt = 0;
    if (s < n) {
        g = perm ? perm[s] : (uint32_t)s;
        off = offsets[s];
        const uint2 r = reinterpret_cast<const uint2 *>(rect)[g];
        x0 = r.x & 0xFFFFu; y0 = r.y & 0xFFFFu;
        if (x0 != 0u) { wdt = (r.x >> 16) - x0 + 1u; cnt = wdt * ((r.y >> 16) - y0 + 1u); }
    }
    unsigned long long todo = __ballot(cnt != 0u);
    while (todo) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t kg = __shfl(g, k), koff = __shfl(off, k), kx0 = __shfl(x0, k), ky0 = __shfl(y0, k);
        const uint32_t kw = __shfl(wdt, k), kc = __shfl(cnt, k);
        for (uint32_t j = lane; j < kc; j += GS_WAVE) {
            const uint32_t ty = ky0 + j / kw, tx = kx0 + j % kw;
            const uint32_t tile = (ty - 1u) * (uint32_t)gx + (tx - 1u);
            inst[(size_t)koff + j] = ((uint64_t)tile << 32) | kg;
        }
    }
}
hipError_t gs_launch_emit(const uint16_t *rect, const uint32_t *perm, const uint32_t *offsets, uint64_t *inst, int64_t n,
                          int gx, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rect, perm, offsets, inst, n, gx);
    return hipGetLastError();
}

// ---------------------------------------------------------------- tile ranges
__global__ void ranges_kernel(const uint64_t *__restrict__ inst, int64_t n_inst, uint32_t *__restrict__ ranges) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_inst) return;
    const uint32_t t = (uint32_t)(inst[p] >> 32);
    if (p == 0 || (uint32_t)(inst[p - 1] >> 32) != t) ranges[2 * (size_t)t] = (uint32_t)p;
    if (p == n_inst - 1 || (uint32_t)(inst[p + 1] >> 32) != t) ranges[2 * (size_t)t + 1] = (uint32_t)(p + 1);
}
hipError_t gs_launch_ranges(const uint64_t *inst, int64_t n_inst, uint32_t *ranges, int64_t n_tiles, hipStream_t s) {
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)n_tiles, s);
    if (e != hipSuccess || n_inst <= 0) return e;
    hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)((n_inst + 255) / 256)), dim3(256), 0, s, inst, n_inst, ranges);
    return hipGetLastError();
}

__global__ void split_ids_kernel(const uint64_t *__restrict__ inst, uint32_t *__restrict__ ids, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) ids[p] = (uint32_t)inst[p];
}
hipError_t gs_launch_split_ids(const uint64_t *inst, uint32_t *ids, int64_t n_inst, hipStream_t s) {
    if (n_inst <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_ids_kernel, dim3((unsigned)((n_inst + 255) / 256)), dim3(256), 0, s, inst, ids, n_inst);
    return hipGetLastError();
}
