// gs_composite.hip -- per-tile alpha composite (forward) and its adjoint for gfx950.
//
// Forward replaces splatDraw (reference src/splat.jl:195-269): every pixel of a 16x16 tile
// walks the tile's splat list front to back, alpha = sigmoid(o) * exp(-1/2 d' Sigma^-1 d),
// C += rgb*alpha*T, T *= 1-alpha, for list entries whose pixel box contains the pixel
// (splat.jl:240); no alpha clamp / 1/255 cut (reference has none).  The reference gathers
// 26 floats per (pixel, slot) from global memory and evaluates SH per pixel; here the per-view
// payload (48 B) is gathered once per (tile, splat) with coalesced id loads, staged in LDS
// and broadcast to the lanes.
//
// Mapping (wave64-first, not a 16x16 CUDA block): ONE wave per tile, lane l owns the four
// pixels (x = l & 15, y = (l >> 4) + 4p).  The tile-wide transmittance vote is a single
// 64-bit ballot and there is no workgroup barrier on the critical path.  Backward replaces
// splatGrads (splat.jl:271-396), which is not a valid adjoint of the 3-D forward:
// it is the derived adjoint, walking the list in the SAME order with the suffix colour obtained
// as D - prefix (D = C_final . dC), so T is recomputed exactly as in the forward and never divided
// back; the nine per-splat sums are reduced by a reduce-scatter tree on the LDS crossbar and written
// with one 9-lane atomic per (tile, splat).
//
// While staging a 64-entry batch each lane also bounds the largest alpha its entry can reach on this
// tile (rect_can_contribute); entries that are no-ops in fp32 are dropped and the batch is compacted in
// LDS, so the per-pixel loops only see entries that matter (gs_config.alpha_cull; 44 % of the walked
// entries at C3).  The lists themselves and the batch boundaries of the early-out rule are untouched.
//
// Both kernels are VALU bound (50 / 150 VALU wave-instructions per evaluated entry, VALU pipe ~90 % busy;
// profiles/), not HBM bound; see DESIGN.md section 5 for the roofline accounting and the measured
// instruction costs that shaped the inner loops.
#include "gs_common.h"

#define CB 64                       // splats staged per batch
#define NEG_HALF_LOG2E (-0.72134752044448170368f)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// A loop-invariant constant the compiler must keep in a VGPR: measured on gfx950 (tools/valu_ubench3.hip) a
// v_fma_f32 with an SGPR source issues at 4.4 cycles per wave64, with VGPR sources only at 2.3.
__device__ __forceinline__ float vgpr_const(float x) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Workgroup -> tile map.  Measured on MI355X at C3 (tools/abtest.py, variants +100/+200): the plain
// order (neighbouring tiles run at the same time on DIFFERENT XCDs and share their splat payloads
// through the Infinity Cache) is 5 % faster than giving each XCD a contiguous band of tiles or
// whole tile rows -- L2 affinity buys less here than it costs in balance.  Speed only, never correctness.
__device__ __forceinline__ int tile_of_block(int b, int ntiles, int gx, int mode) {
    if (mode == 0) return b;                       // plain order (default)
    if (mode == 2) {                               // tile rows dealt round-robin to the XCDs
        const int xcd = b & 7, idx = b >> 3;
        return ((idx / gx) * 8 + xcd) * gx + idx % gx;
    }
    const int per = (ntiles + 7) >> 3;             // mode 1: one contiguous band of tiles per XCD
    return (b & 7) * per + (b >> 3);
}

// ---------------------------------------------------------------- forward
// The pixel-box test is arithmetic: v_cmp / v_cndmask cost ~4 cycles each on
// gfx950 (fma: 2), so the box is applied as an exponent penalty
//     pw' = pw - BIG * |d - med3(d, lo, hi)|        (d = pixel - mu on that axis)
// which is exactly 0 inside the box (med3 returns d itself) and drives exp2 to 0 outside.
// lo/hi = box edge - mu -/+ 0.25 are lane independent and staged once per (tile, splat).
#define GS_BIG 1.0e30f
// A (tile, splat) entry whose largest alpha over the tile's pixels is below 2^-27 is a no-op in the
// reference's own fp32 arithmetic: T*(1-alpha) == T exactly (alpha < 2^-25 already rounds 1-alpha to 1)
// and rgb*alpha*T is below 7.5e-9*|rgb|, under half an ulp of any accumulated colour above 1e-7.  Such
// entries are dropped while staging (gs_config.alpha_cull, default on; lists stay the reference's).
#define GS_ALPHA_CULL_LOG2 (-27.0f)
// Lane-independent terms of one splat, computed once per (tile, splat) by the staging lane:
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
__device__ __forceinline__ float stage_record(float4 &q0, float4 &q1, float4 &q2, const float4 &n0, const float4 &n1, const float4 &n2,
                                              const int tx0, const int ty0, bool &keep) {
    const uint32_t bbx = __float_as_uint(n0.w), bby = __float_as_uint(n2.w);
    const int xmin = (int)(short)(bbx & 0xFFFFu), xmax = (int)(short)(bbx >> 16);
    const int ymin = (int)(short)(bby & 0xFFFFu), ymax = (int)(short)(bby >> 16);
    const bool empty = xmax < xmin || ymax < ymin;
    // log2(sig), capped one ulp below 0 so alpha = exp2(pw + l2s) < 1 strictly (pw <= 0: the conic is PSD); only
    // matters when sigmoid(o) rounds to exactly 1.0f (o > 16.6): relative change 6e-8
    const float l2s = fminf(__builtin_amdgcn_logf(n0.z), -8.6e-8f);
    // an empty box (near/far-culled splat) gets lo = hi = +BIG: every pixel is "outside"
    const float xlo = empty ? GS_BIG : ((float)xmin - n0.x) - 0.25f, xhi = empty ? GS_BIG : ((float)xmax - n0.x) + 0.25f;
    const float ylo = empty ? GS_BIG : ((float)ymin - n0.y) - 0.25f, yhi = empty ? GS_BIG : ((float)ymax - n0.y) + 0.25f;
    q0 = make_float4(n0.x, n0.y, l2s, xlo);
    q1 = make_float4(NEG_HALF_LOG2E * n1.x, NEG_HALF_LOG2E * (n1.y + n1.z), NEG_HALF_LOG2E * n1.w, xhi);
    q2 = make_float4(n2.x, n2.y, n2.z, ylo);
    {
        const float A = q1.x, B = q1.y, C = q1.z;
        const bool concave = A < 0.0f && C < 0.0f && 4.0f * A * C - B * B > 0.0f;
        const float rx0 = (float)max(tx0, xmin) - n0.x, rx1 = (float)min(tx0 + GS_TILE - 1, xmax) - n0.x;
        const float ry0 = (float)max(ty0, ymin) - n0.y, ry1 = (float)min(ty0 + GS_TILE - 1, ymax) - n0.y;
        keep = !empty && rect_can_contribute(A, B, C, 0.5f * B * fast_rcp(A), 0.5f * B * fast_rcp(C), concave, l2s, rx0, rx1, ry0, ry1);
    }
    return yhi;
}

// kept-entry slot of this lane inside the wave's keep mask
__device__ __forceinline__ int slot_of(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <bool EARLY, int UNROLL, int MINW, bool CULL>
__global__ __launch_bounds__(64, MINW) void composite_fwd_kernel(GsCompositeArgs a) {
    __shared__ float4 sp[CB * 3];
    __shared__ float syhi[CB];
    const int ntiles = a.gx * a.gy;
    const int tile = tile_of_block(blockIdx.x, ntiles, a.gx, a.map_mode);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int px = (tile % a.gx) * GS_TILE + (lane & 15) + 1;
    const int py0 = (tile / a.gx) * GS_TILE + (lane >> 4) + 1;
    const float fx = (float)px;
    const float nbig = vgpr_const(-GS_BIG);
    const uint32_t s0 = a.ranges[2 * tile], s1 = a.ranges[2 * tile + 1];

    const int tx0 = px - (lane & 15), ty0 = py0 - (lane >> 4);     // first pixel of the tile (1-based)

    float Cr[4], Cg[4], Cb[4], T[4], Tdead[4], fy[4];
    bool dead[4];
    uint32_t walked = 0, evaluated = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        Cr[p] = Cg[p] = Cb[p] = 0.0f;
        fy[p] = (float)(py0 + 4 * p);
        const bool in = (px <= a.W && py0 + 4 * p <= a.H);
        T[p] = in ? 1.0f : 0.0f;
        Tdead[p] = 0.0f; dead[p] = !in;
    }
    const float4 *pay4 = reinterpret_cast<const float4 *>(a.payload);
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    uint32_t pos = s0 + lane;
    if (pos < s1) { const size_t g = a.ids[pos]; n0 = pay4[3 * g]; n1 = pay4[3 * g + 1]; n2 = pay4[3 * g + 2]; }
    for (uint32_t base = s0; base < s1; base += CB) {
        const int cnt = (int)min((uint32_t)CB, s1 - base);
        if (EARLY) {
            bool live = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (!dead[p] && T[p] < a.t_min) { dead[p] = true; Tdead[p] = T[p]; T[p] = 0.0f; }
                live = live || !dead[p];
            }
            if (__ballot(live) == 0ull) break;
        }
        float4 q0, q1, q2;
        bool keep;
        const float yhi_l = stage_record(q0, q1, q2, n0, n1, n2, tx0, ty0, keep);
        int slot = lane, nk = cnt;
        if (CULL) {                                                     // compact the batch to the entries that can matter
            keep = keep && lane < cnt;
            const uint64_t m = __ballot(keep);
            slot = slot_of(m); nk = __popcll(m);
        } else keep = true;
        __syncthreads();                                                // one wave: orders LDS reads/writes only
        if (keep) {
            sp[3 * slot] = q0; sp[3 * slot + 1] = q1; sp[3 * slot + 2] = q2;
            syhi[slot] = yhi_l;
        }
        __syncthreads();
        pos = base + CB + lane;
        if (pos < s1) { const size_t g = a.ids[pos]; n0 = pay4[3 * g]; n1 = pay4[3 * g + 1]; n2 = pay4[3 * g + 2]; }
#pragma unroll UNROLL
        for (int k = 0; k < nk; ++k) {
            const float4 q0k = sp[3 * k], q1k = sp[3 * k + 1], q2k = sp[3 * k + 2];
            const float yhi = syhi[k];
            const float dX = fx - q0k.x;
            const float ex = dX - __builtin_amdgcn_fmed3f(dX, q0k.w, q1k.w);          // 0 inside the box columns
            const float A0 = fmaf(nbig, fabsf(ex), fmaf(q1k.x * dX, dX, q0k.z));  // k i0 dX^2 + log2 sig - penalty
            const float B0 = q1k.y * dX;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float dY = fy[p] - q0k.y;
                const float ey = dY - __builtin_amdgcn_fmed3f(dY, q2k.w, yhi);
                const float pw = fmaf(dY, fmaf(q1k.z, dY, B0), A0);
                const float w = fast_exp2(fmaf(nbig, fabsf(ey), pw)) * T[p];
                Cr[p] = fmaf(q2k.x, w, Cr[p]);
                Cg[p] = fmaf(q2k.y, w, Cg[p]);
                Cb[p] = fmaf(q2k.z, w, Cb[p]);
                T[p] = T[p] - w;
            }
        }
        walked += (uint32_t)cnt; evaluated += (uint32_t)nk;
    }
    if (lane == 0 && a.walked) { atomicAdd(a.walked, (unsigned long long)walked); atomicAdd(a.walked + 1, (unsigned long long)evaluated); }
    if (px <= a.W) {
        const size_t plane = (size_t)a.W * a.H;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int py = py0 + 4 * p;
            if (py <= a.H) {
                const size_t o = (size_t)(px - 1) + (size_t)a.W * (py - 1);
                if (a.image) { a.image[o] = Cr[p]; a.image[o + plane] = Cg[p]; a.image[o + 2 * plane] = Cb[p]; }
                if (a.trans) a.trans[o] = (EARLY && dead[p]) ? Tdead[p] : T[p];
            }
        }
    }
}

// ---------------------------------------------------------------- wave64 reductions
// Nine per-splat sums are needed per (tile, splat).  Eight go through a reduce-scatter tree over the six lane bits:
// a fold pairs two registers, so one add sums BOTH across one lane bit and leaves value A in the lower lanes and
// value B in the upper ones -- 4 + 2 + 1 folds bring eight registers down to one whose 8-lane groups each hold one
// value, three butterfly steps finish inside the groups.  The ninth sum is a plain six-step butterfly.
//
// Every exchange runs on the LDS crossbar (ds_swizzle inside 32 lanes, ds_bpermute across the halves), because the
// VALU is the unit these kernels are bound by: a fold costs it two selects and one add (10.7 cycles per wave64) and
// a butterfly step one add (2.3), against 14.3 for a v_permlane32/16_swap fold and 4.9 for a DPP add (measured,
// tools/valu_ubench2.hip, valu_ubench3.hip).  RED = 1 keeps the lane-swap / DPP form for re-measurement; at C3 the
// LDS form is 6 % faster (tools/abtest.py).
typedef unsigned int gs_u2 __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __int_as_float(t);
}
template <int PATTERN>
__device__ __forceinline__ float swz_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PATTERN));
}
// lower lanes of the bit <- a(l) + a(partner) ; upper lanes <- b(partner) + b(l)
template <int PATTERN>
__device__ __forceinline__ float fold_swz(float a, float b, bool upper) {
    const float send = upper ? a : b, keep = upper ? b : a;
    return keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), PATTERN));
}
__device__ __forceinline__ float fold32_lds(float a, float b, bool upper, int xaddr) {
    const float send = upper ? a : b, keep = upper ? b : a;
    return keep + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(send)));
}
// every lane of the 8-lane group (16r .. 16r+7) ends with the wave total of (v0, v2, v1, v3)[r], of group
// (16r+8 .. 16r+15) with that of (v4, v6, v5, v7)[r]
__device__ __forceinline__ float reduce8_lds(const float (&v)[8], int lane, int xaddr) {
    const bool u32 = (lane & 32) != 0, u16 = (lane & 16) != 0, u8 = (lane & 8) != 0;
    const float b0 = fold32_lds(v[0], v[1], u32, xaddr), b1 = fold32_lds(v[2], v[3], u32, xaddr);
    const float b2 = fold32_lds(v[4], v[5], u32, xaddr), b3 = fold32_lds(v[6], v[7], u32, xaddr);
    const float c0 = fold_swz<0x401F>(b0, b1, u16), c1 = fold_swz<0x401F>(b2, b3, u16);       // xor 16
    float d = fold_swz<0x201F>(c0, c1, u8);                                                  // xor 8
    d = swz_add<0x101F>(d); d = swz_add<0x081F>(d); d = swz_add<0x041F>(d);                  // xor 4, 2, 1
    return d;
}
__device__ __forceinline__ float wave_sum_lds(float v, int xaddr) {
    v = swz_add<0x041F>(v); v = swz_add<0x081F>(v); v = swz_add<0x101F>(v); v = swz_add<0x201F>(v); v = swz_add<0x401F>(v);
    return v + __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(v)));
}

// The same tree on the VALU's own cross-lane paths: v_permlane32_swap / v_permlane16_swap folds, one bank-masked
// DPP fold (lanes of a disabled bank keep their value), row_shr adds.  Totals land in lanes 16r+7 and 16r+15.
__device__ __forceinline__ float fold32(float a, float b) {
    const gs_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold16(float a, float b) {
    const gs_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// Inline asm: the hazard recogniser does not see the DPP reads, hence the leading s_nop (VALU write -> DPP read).
__device__ __forceinline__ float fold8(float a, float b) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3"
                 : "+v"(b) : "v"(a));
    return b;
}
__device__ __forceinline__ float reduce8_swap(const float (&v)[8]) {
    const float b0 = fold32(v[0], v[1]), b1 = fold32(v[2], v[3]), b2 = fold32(v[4], v[5]), b3 = fold32(v[6], v[7]);
    float d = fold8(fold16(b0, b1), fold16(b2, b3));
    d = dpp_add<0x114, 0xF, 0xF>(d);      // row_shr:4
    d = dpp_add<0x112, 0xF, 0xF>(d);      // row_shr:2
    d = dpp_add<0x111, 0xF, 0xF>(d);      // row_shr:1
    return d;
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xF, 0xF>(v);      // row_shr:1
    v = dpp_add<0x112, 0xF, 0xF>(v);      // row_shr:2
    v = dpp_add<0x114, 0xF, 0xF>(v);      // row_shr:4
    v = dpp_add<0x118, 0xF, 0xF>(v);      // row_shr:8   -> lane 15 of each row = row sum
    v = dpp_add<0x142, 0xA, 0xF>(v);      // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC, 0xF>(v);      // row_bcast:31 into rows 2,3 -> lane 63 = total
    return v;
}

// g2d row of a gaussian: [dr dg db | S0 Sx Sy Sxx Sxy (unused) Syy] -- the colour gradient and the raw moments
// S.. = sum over pixels of dd * {1, dX, dY, dX^2, dX dY, dY^2}, dd = d L / d(log alpha); gs_g2d_to_grads (gs_common.h)
// turns them into d{sig, mu, conic} once per gaussian.  Nine lanes of the wave issue the one atomic: lane 16r+7 adds
// (dr, db, dg, S0)[r], lane 16r+15 adds (Sx, Sxx, Sy, Sxy)[r], lane 62 Syy.
__device__ __forceinline__ int out_component_tree(int lane) {
    const int row = lane >> 4, pos = lane & 15;
    if (pos == 7) return row == 0 ? 0 : row == 1 ? 2 : row == 2 ? 1 : 3;
    if (pos == 15) return row == 0 ? 4 : row == 1 ? 6 : row == 2 ? 5 : 7;
    if (lane == 62) return 9;
    return -1;
}

// ---------------------------------------------------------------- backward
// Same staging and arithmetic pixel-box penalty as the forward; per-batch freeze of saturated pixels
// (T = S = 0 makes every later contribution exactly zero); alpha < 1 strictly (stage_record), so
// 1/(1-alpha) needs no guard; d sig = -(1/sig) * sum(dd) needs no accumulator of its own.
// DET: the per-(tile, splat) sums are added as 2^-40 fixed-point integers (64-bit integer atomics are
// order independent, so the gradients are bitwise reproducible run to run); otherwise float atomics.
template <bool EARLY, int MINW, bool DET, int RED, bool CULL>      // RED: 0 reduction tree on the LDS crossbar, 1 on lane swaps + DPP
__global__ __launch_bounds__(64, MINW) void composite_bwd_kernel(GsCompositeArgs a) {
    __shared__ float4 sp[CB * 3];
    __shared__ float syhi[CB];
    __shared__ uint32_t sid[CB];
    const int ntiles = a.gx * a.gy;
    const int tile = tile_of_block(blockIdx.x, ntiles, a.gx, a.map_mode);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int px = (tile % a.gx) * GS_TILE + (lane & 15) + 1;
    const int py0 = (tile / a.gx) * GS_TILE + (lane >> 4) + 1;
    const float fx = (float)px;
    const float nbig = vgpr_const(-GS_BIG);
    const uint32_t s0 = a.ranges[2 * tile], s1 = a.ranges[2 * tile + 1];
    const size_t plane = (size_t)a.W * a.H;
    const int ocomp = out_component_tree(lane);
    const int xaddr = (lane ^ 32) << 2;                                  // ds_bpermute address of the partner lane
    const uint32_t ooff = ocomp >= 0 ? 4u * (uint32_t)ocomp : 0u;        // byte offset inside the gaussian's g2d row

    const int tx0 = px - (lane & 15), ty0 = py0 - (lane >> 4);

    float dCr[4], dCg[4], dCb[4], T[4], S[4], fy[4];
    bool dead[4];
    uint32_t walked = 0, evaluated = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int py = py0 + 4 * p;
        const bool in = (px <= a.W && py <= a.H);
        const size_t o = in ? (size_t)(px - 1) + (size_t)a.W * (py - 1) : 0;
        fy[p] = (float)py;
        dCr[p] = in ? a.dC[o] : 0.0f;
        dCg[p] = in ? a.dC[o + plane] : 0.0f;
        dCb[p] = in ? a.dC[o + 2 * plane] : 0.0f;
        T[p] = in ? 1.0f : 0.0f;
        S[p] = in ? (a.image[o] * dCr[p] + a.image[o + plane] * dCg[p] + a.image[o + 2 * plane] * dCb[p]) : 0.0f;
        dead[p] = !in;
    }
    const float4 *pay4 = reinterpret_cast<const float4 *>(a.payload);
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    uint32_t nid = 0;
    uint32_t pos = s0 + lane;
    if (pos < s1) { nid = a.ids[pos]; n0 = pay4[3 * (size_t)nid]; n1 = pay4[3 * (size_t)nid + 1]; n2 = pay4[3 * (size_t)nid + 2]; }
    for (uint32_t base = s0; base < s1; base += CB) {
        const int cnt = (int)min((uint32_t)CB, s1 - base);
        if (EARLY) {
            bool live = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (!dead[p] && T[p] < a.t_min) { dead[p] = true; T[p] = 0.0f; S[p] = 0.0f; }
                live = live || !dead[p];
            }
            if (__ballot(live) == 0ull) break;
        }
        float4 q0, q1, q2;
        bool keep;
        const float yhi_l = stage_record(q0, q1, q2, n0, n1, n2, tx0, ty0, keep);
        int slot = lane, nk = cnt;
        if (CULL) {
            keep = keep && lane < cnt;
            const uint64_t m = __ballot(keep);
            slot = slot_of(m); nk = __popcll(m);
        } else keep = true;
        __syncthreads();
        if (keep) {
            sp[3 * slot] = q0; sp[3 * slot + 1] = q1; sp[3 * slot + 2] = q2;
            syhi[slot] = yhi_l;
            sid[slot] = nid;                                             // gaussian id of the staged entry
        }
        __syncthreads();
        pos = base + CB + lane;
        if (pos < s1) { nid = a.ids[pos]; n0 = pay4[3 * (size_t)nid]; n1 = pay4[3 * (size_t)nid + 1]; n2 = pay4[3 * (size_t)nid + 2]; }
        for (int k = 0; k < nk; ++k) {
            const float4 q0k = sp[3 * k], q1k = sp[3 * k + 1], q2k = sp[3 * k + 2];
            const float yhi = syhi[k];
            const float dX = fx - q0k.x;
            const float ex = dX - __builtin_amdgcn_fmed3f(dX, q0k.w, q1k.w);
            const float A0 = fmaf(nbig, fabsf(ex), fmaf(q1k.x * dX, dX, q0k.z));
            const float B0 = q1k.y * dX;
            float al[4], dY[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                dY[p] = fy[p] - q0k.y;
                const float ey = dY[p] - __builtin_amdgcn_fmed3f(dY[p], q2k.w, yhi);
                al[p] = fast_exp2(fmaf(nbig, fabsf(ey), fmaf(dY[p], fmaf(q1k.z, dY[p], B0), A0)));
            }
            if (!CULL && __ballot(((al[0] + al[1]) + (al[2] + al[3])) != 0.0f) == 0ull) continue;   // nobody in the tile touched it
            float ar, ag, ab, q0s, q1s, q2s;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float w = al[p] * T[p];
                const float cdot = fmaf(q2k.x, dCr[p], fmaf(q2k.y, dCg[p], q2k.z * dCb[p]));
                S[p] = fmaf(-cdot, w, S[p]);
                const float inv = fast_rcp(1.0f - al[p]);
                const float dalpha = fmaf(T[p], cdot, -(S[p] * inv));
                const float dd = -(al[p] * dalpha);
                const float ddy = dd * dY[p];
                if (p == 0) {                                             // plain products: no fma against a zero
                    ar = w * dCr[0]; ag = w * dCg[0]; ab = w * dCb[0];
                    q0s = dd; q1s = ddy; q2s = ddy * dY[0];
                } else {
                    ar = fmaf(w, dCr[p], ar); ag = fmaf(w, dCg[p], ag); ab = fmaf(w, dCb[p], ab);
                    q0s += dd; q1s += ddy; q2s = fmaf(ddy, dY[p], q2s);
                }
                T[p] = T[p] - w;
            }
            // raw moments of dd = d L / d(log alpha) about the splat's mean; the factors that are constant per gaussian
            // (1/sig, the conic, 1/2) are applied once per gaussian by the parameter kernels, not once per (tile, splat)
            const float qx = dX * q0s;
            float v[8];
            v[0] = ar; v[1] = ag; v[2] = ab; v[3] = q0s;
            v[4] = qx; v[5] = q1s;
            v[6] = dX * qx; v[7] = dX * q1s;
            float outv;
            if (RED == 0) {
                const float d = reduce8_lds(v, lane, xaddr);
                const float t9 = wave_sum_lds(q2s, xaddr);
                outv = lane == 62 ? t9 : d;
            } else {
                const float d = reduce8_swap(v);
                const float t9 = wave_sum_to_lane63(q2s);
                // lane 62 <- t9(63): row_shl:1 into row 3 / bank 3 only; lane 63 has no source and keeps d
                outv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(d), __float_as_int(t9), 0x101, 0x8, 0x8, false));
            }
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readfirstlane((int)sid[k]);
            if (ocomp >= 0) {                                            // row base in SGPRs, per-lane byte offset in one VGPR
                if (DET) {
                    const float sc = fminf(fmaxf(outv * (ocomp >= 6 ? GS_FIXED_SCALE2 : GS_FIXED_SCALE), -9.0e18f), 9.0e18f);      // saturate, never wrap
                    char *rowp = reinterpret_cast<char *>(a.g2d_fixed) + (size_t)gid * 80;
                    atomicAdd(reinterpret_cast<unsigned long long *>(rowp + 2u * ooff), (unsigned long long)__float2ll_rn(sc));
                } else {
                    char *rowp = reinterpret_cast<char *>(a.g2d) + (size_t)gid * 40;
                    atomicAdd(reinterpret_cast<float *>(rowp + ooff), outv);
                }
            }
        }
        walked += (uint32_t)cnt; evaluated += (uint32_t)nk;
    }
    if (lane == 0 && a.walked) { atomicAdd(a.walked, (unsigned long long)walked); atomicAdd(a.walked + 1, (unsigned long long)evaluated); }
}


// Launch configurations were chosen by A/B timing on MI355X at C3 (tools/abtest.py):
// forward <unroll 2, 8 waves/SIMD> for literal lists, <unroll 2, unconstrained> with early-out;
// backward <8 waves/SIMD> literal, <unconstrained> with early-out.  `variant` selects the other
// instantiations for re-measurement.
hipError_t gs_launch_composite_fwd(const GsCompositeArgs &a, hipStream_t s) {
    const int ntiles = a.gx * a.gy;
    if (ntiles <= 0) return hipSuccess;
    const dim3 grid(a.map_mode == 2 ? ((a.gy + 7) / 8) * 8 * a.gx : ((ntiles + 7) / 8) * 8), block(64);
    const bool early = a.t_min > 0.0f;
    const int v = a.variant == 0 ? (early ? 1 : 2) : a.variant;
#define GS_F(E, U, M) do { if (a.cull) hipLaunchKernelGGL((composite_fwd_kernel<E, U, M, true>), grid, block, 0, s, a); \
                           else hipLaunchKernelGGL((composite_fwd_kernel<E, U, M, false>), grid, block, 0, s, a); } while (0)
    if (v == 1) { if (early) GS_F(true, 2, 1); else GS_F(false, 2, 1); }
    else if (v == 2) { if (early) GS_F(true, 2, 8); else GS_F(false, 2, 8); }
    else { if (early) GS_F(true, 1, 8); else GS_F(false, 1, 8); }
#undef GS_F
    return hipGetLastError();
}

hipError_t gs_launch_composite_bwd(const GsCompositeArgs &a, hipStream_t s) {
    const int ntiles = a.gx * a.gy;
    if (ntiles <= 0) return hipSuccess;
    const dim3 grid(a.map_mode == 2 ? ((a.gy + 7) / 8) * 8 * a.gx : ((ntiles + 7) / 8) * 8), block(64);
    const bool early = a.t_min > 0.0f;
    const int v = a.variant == 0 ? 1 : a.variant;      // measured best: LDS-crossbar reduction, registers unconstrained
#define GS_B2(E, M, D, Z) do { if (a.cull) hipLaunchKernelGGL((composite_bwd_kernel<E, M, D, Z, true>), grid, block, 0, s, a); \
                               else hipLaunchKernelGGL((composite_bwd_kernel<E, M, D, Z, false>), grid, block, 0, s, a); } while (0)
#define GS_B(E, M, Z) do { if (a.g2d_fixed) GS_B2(E, M, true, Z); else GS_B2(E, M, false, Z); } while (0)
    if (v == 1) { if (early) GS_B(true, 1, 0); else GS_B(false, 1, 0); }
    else if (v == 2) { if (early) GS_B(true, 8, 0); else GS_B(false, 8, 0); }
    else if (v == 3) { if (early) GS_B(true, 1, 1); else GS_B(false, 1, 1); }
    else { if (early) GS_B(true, 8, 1); else GS_B(false, 8, 1); }
#undef GS_B
#undef GS_B2
    return hipGetLastError();
}
