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
// 64-bit ballot, the per-splat gradient reduction is six DPP steps, and there is no
// workgroup barrier on the critical path.  Backward replaces splatGrads (splat.jl:271-396),
// which is not a valid adjoint of the 3-D forward: it is the derived adjoint,
// walking the list in the SAME order with the suffix colour obtained as D - prefix
// (D = C_final . dC), so T is recomputed exactly as in the forward and never divided back.
//
// Both kernels are VALU/transcendental bound (about 15 / 45 lane-ops per pixel-splat), not
// HBM bound; see DESIGN.md section 5 for the roofline accounting.
#include "gs_common.h"

#define CB 64                       // splats staged per batch
#define NEG_HALF_LOG2E (-0.72134752044448170368f)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (round robin), so
// give each XCD a contiguous band of tiles -> neighbouring tiles (which share most of their
// splat payloads) hit the same 4 MiB L2.  Pure speed heuristic, never correctness.
__device__ __forceinline__ int tile_of_block(int b, int ntiles) {
    const int per = (ntiles + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);
    return t;
}

__device__ __forceinline__ GsPayload unpack_payload(const float4 &a, const float4 &b, const float4 &c) {
    GsPayload P;
    P.mx = a.x; P.my = a.y; P.sig = a.z; P.bbx = __float_as_uint(a.w);
    P.i0 = b.x; P.i1 = b.y; P.i2 = b.z; P.i3 = b.w;
    P.r = c.x; P.g = c.y; P.b = c.z; P.bby = __float_as_uint(c.w);
    return P;
}

template <bool EARLY>
__global__ __launch_bounds__(64) void composite_fwd_kernel(GsCompositeArgs a) {
    __shared__ float4 sp[CB * 3];                                       // 3 KiB: one batch of payloads
    const int ntiles = a.gx * a.gy;
    const int tile = tile_of_block(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int px = (tile % a.gx) * GS_TILE + (lane & 15) + 1;          // 1-based, splat.jl:204
    const int py0 = (tile / a.gx) * GS_TILE + (lane >> 4) + 1;         // rows py0 + 4p
    const float fx = (float)px;
    const uint32_t s0 = a.ranges[2 * tile], s1 = a.ranges[2 * tile + 1];

    float Cr[4], Cg[4], Cb[4], T[4];
    uint32_t walked = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        Cr[p] = Cg[p] = Cb[p] = 0.0f;
        T[p] = (px <= a.W && py0 + 4 * p <= a.H) ? 1.0f : 0.0f;         // off-image pixels are inert
    }

    // prefetch batch 0 (three dwordx4 per lane; kept in registers until staged)
    const float4 *pay4 = reinterpret_cast<const float4 *>(a.payload);
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    uint32_t pos = s0 + lane;
    if (pos < s1) { const size_t g = (uint32_t)a.inst[pos]; n0 = pay4[3 * g]; n1 = pay4[3 * g + 1]; n2 = pay4[3 * g + 2]; }
    for (uint32_t base = s0; base < s1; base += CB) {
        const int cnt = (int)min((uint32_t)CB, s1 - base);
        __syncthreads();                                                // one wave: orders LDS reads/writes only
        sp[3 * lane] = n0; sp[3 * lane + 1] = n1; sp[3 * lane + 2] = n2;
        __syncthreads();
        pos = base + CB + lane;                                         // next batch in flight during the loop below
        if (pos < s1) { const size_t g = (uint32_t)a.inst[pos]; n0 = pay4[3 * g]; n1 = pay4[3 * g + 1]; n2 = pay4[3 * g + 2]; }
        for (int k = 0; k < cnt; ++k) {
            const GsPayload P = unpack_payload(sp[3 * k], sp[3 * k + 1], sp[3 * k + 2]);
            const int xmin = (int)(short)(P.bbx & 0xFFFFu), xmax = (int)(short)(P.bbx >> 16);
            const int ymin = (int)(short)(P.bby & 0xFFFFu), ymax = (int)(short)(P.bby >> 16);
            const bool hitx = (px >= xmin) && (px <= xmax);
            const float dX = fx - P.mx;
            // -1/2 log2(e) * (i0 dX^2 + (i1+i2) dX dY + i3 dY^2), dY-polynomial coefficients
            const float A0 = (NEG_HALF_LOG2E * P.i0) * dX * dX;
            const float B0 = (NEG_HALF_LOG2E * (P.i1 + P.i2)) * dX;
            const float Cq = NEG_HALF_LOG2E * P.i3;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int py = py0 + 4 * p;
                const float dY = (float)py - P.my;
                const float e = fast_exp2(fmaf(dY, fmaf(Cq, dY, B0), A0));
                bool hit = hitx && (py >= ymin) && (py <= ymax);
                if (EARLY) hit = hit && !(T[p] < a.t_min);
                const float alpha = hit ? P.sig * e : 0.0f;
                const float w = alpha * T[p];
                Cr[p] = fmaf(P.r, w, Cr[p]);
                Cg[p] = fmaf(P.g, w, Cg[p]);
                Cb[p] = fmaf(P.b, w, Cb[p]);
                T[p] = T[p] - w;                                        // == T*(1-alpha) up to rounding
            }
        }
        walked += (uint32_t)cnt;
        if (EARLY) {
            const bool live = !(T[0] < a.t_min) || !(T[1] < a.t_min) || !(T[2] < a.t_min) || !(T[3] < a.t_min);
            if (__ballot(live) == 0ull) break;                          // whole tile saturated
        }
    }
    if (lane == 0 && a.walked) atomicAdd(a.walked, (unsigned long long)walked);
    if (px <= a.W) {
        const size_t plane = (size_t)a.W * a.H;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int py = py0 + 4 * p;
            if (py <= a.H) {
                const size_t o = (size_t)(px - 1) + (size_t)a.W * (py - 1);
                if (a.image) { a.image[o] = Cr[p]; a.image[o + plane] = Cg[p]; a.image[o + 2 * plane] = Cb[p]; }
                if (a.trans) a.trans[o] = T[p];
            }
        }
    }
}

// ---------------------------------------------------------------- wave64 reductions
// Nine per-splat sums are needed.  Reducing them one by one costs 6 DPP adds each; instead
// eight of them go through a reduce-scatter butterfly built on gfx950's lane-swap
// instructions: v_permlane32_swap pairs two registers (one add sums BOTH across the wave
// halves, leaving value A in lanes 0-31 and B in lanes 32-63), v_permlane16_swap does the
// same across 16-lane rows, and four row_shr DPP adds finish inside the rows.  Result: lane
// 15 of row r holds the wave total of value ORDER[r] -- 20 VALU ops for 8 values.
typedef unsigned int gs_u2 __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v = dpp_add<0x111, 0xF, 0xF>(v);      // row_shr:1
    v = dpp_add<0x112, 0xF, 0xF>(v);      // row_shr:2
    v = dpp_add<0x114, 0xF, 0xF>(v);      // row_shr:4
    v = dpp_add<0x118, 0xF, 0xF>(v);      // row_shr:8   -> lane 15 of each row = row sum
    return v;
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = row_sum_to_lane15(v);
    v = dpp_add<0x142, 0xA, 0xF>(v);      // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC, 0xF>(v);      // row_bcast:31 into rows 2,3 -> lane 63 = total
    return v;
}
// lanes 0-31: a(l)+a(l+32) ; lanes 32-63: b(l-32)+b(l)
__device__ __forceinline__ float fold32(float a, float b) {
    const gs_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// rows 0,2: a.row(r)+a.row(r+1) ; rows 1,3: b.row(r-1)+b.row(r)
__device__ __forceinline__ float fold16(float a, float b) {
    const gs_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// -> lane 15 of rows 0..3 = totals of (v0, v2, v1, v3) in `lo`, of (v4, v6, v5, v7) in `hi`
__device__ __forceinline__ void reduce8(const float (&v)[8], float &lo, float &hi) {
    const float b0 = fold32(v[0], v[1]), b1 = fold32(v[2], v[3]), b2 = fold32(v[4], v[5]), b3 = fold32(v[6], v[7]);
    lo = row_sum_to_lane15(fold16(b0, b1));
    hi = row_sum_to_lane15(fold16(b2, b3));
}

// g2d row of a gaussian: [dr dg db dsig dmx dmy d00 d01 (d10 = d01, filled by the reader) d11]
// lane 15 of row r adds lo -> LO_COMP[r] ; lane 14 of row r adds hi -> HI_COMP[r] ; lane 61 adds the 9th
__device__ __forceinline__ int out_component(int lane) {
    const int row = lane >> 4, pos = lane & 15;
    // lo rows hold (v0,v2,v1,v3) = (dr, db, dg, dsig) ; hi rows hold (v4,v6,v5,v7) = (dmx, d00, dmy, d01)
    const int lo_comp = row == 0 ? 0 : row == 1 ? 2 : row == 2 ? 1 : 3;
    const int hi_comp = row == 0 ? 4 : row == 1 ? 6 : row == 2 ? 5 : 7;
    if (pos == 15) return lo_comp;
    if (pos == 14) return hi_comp;
    if (lane == 61) return 9;
    return -1;
}

template <bool EARLY>
__global__ __launch_bounds__(64) void composite_bwd_kernel(GsCompositeArgs a) {
    __shared__ float4 sp[CB * 3];
    __shared__ uint32_t sid[CB];
    const int ntiles = a.gx * a.gy;
    const int tile = tile_of_block(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x;
    const int px = (tile % a.gx) * GS_TILE + (lane & 15) + 1;
    const int py0 = (tile / a.gx) * GS_TILE + (lane >> 4) + 1;
    const float fx = (float)px;
    const uint32_t s0 = a.ranges[2 * tile], s1 = a.ranges[2 * tile + 1];
    const size_t plane = (size_t)a.W * a.H;
    const int ocomp = out_component(lane);
    const bool take_hi = (lane & 15) == 14, take_9 = lane == 61;

    float dCr[4], dCg[4], dCb[4], T[4], S[4], fy[4];
    uint32_t walked = 0;
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
        // S = colour still to come (dotted with dC): starts at C_final . dC
        S[p] = in ? (a.image[o] * dCr[p] + a.image[o + plane] * dCg[p] + a.image[o + 2 * plane] * dCb[p]) : 0.0f;
    }

    const float4 *pay4 = reinterpret_cast<const float4 *>(a.payload);
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    uint32_t nid = 0;
    uint32_t pos = s0 + lane;
    if (pos < s1) { nid = (uint32_t)a.inst[pos]; n0 = pay4[3 * (size_t)nid]; n1 = pay4[3 * (size_t)nid + 1]; n2 = pay4[3 * (size_t)nid + 2]; }
    for (uint32_t base = s0; base < s1; base += CB) {
        const int cnt = (int)min((uint32_t)CB, s1 - base);
        __syncthreads();
        sp[3 * lane] = n0; sp[3 * lane + 1] = n1; sp[3 * lane + 2] = n2; sid[lane] = nid;
        __syncthreads();
        pos = base + CB + lane;
        if (pos < s1) { nid = (uint32_t)a.inst[pos]; n0 = pay4[3 * (size_t)nid]; n1 = pay4[3 * (size_t)nid + 1]; n2 = pay4[3 * (size_t)nid + 2]; }
        for (int k = 0; k < cnt; ++k) {
            const GsPayload P = unpack_payload(sp[3 * k], sp[3 * k + 1], sp[3 * k + 2]);
            const int xmin = (int)(short)(P.bbx & 0xFFFFu), xmax = (int)(short)(P.bbx >> 16);
            const int ymin = (int)(short)(P.bby & 0xFFFFu), ymax = (int)(short)(P.bby >> 16);
            const bool hitx = (px >= xmin) && (px <= xmax);
            const float dX = fx - P.mx;
            const float A0 = (NEG_HALF_LOG2E * P.i0) * dX * dX;
            const float B0 = (NEG_HALF_LOG2E * (P.i1 + P.i2)) * dX;
            const float Cq = NEG_HALF_LOG2E * P.i3;
            float e[4], dY[4];
            bool any = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int py = py0 + 4 * p;
                dY[p] = fy[p] - P.my;
                const float ex = fast_exp2(fmaf(dY[p], fmaf(Cq, dY[p], B0), A0));
                bool hit = hitx && (py >= ymin) && (py <= ymax);
                if (EARLY) hit = hit && !(T[p] < a.t_min);
                any = any || hit;
                e[p] = hit ? ex : 0.0f;
            }
            if (__ballot(any) == 0ull) continue;                        // nobody in the tile touched it
            float ar = 0.0f, ag = 0.0f, ab = 0.0f, asig = 0.0f, q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float alpha = P.sig * e[p];
                const float w = alpha * T[p];
                const float cdot = fmaf(P.r, dCr[p], fmaf(P.g, dCg[p], P.b * dCb[p]));
                ar = fmaf(w, dCr[p], ar);
                ag = fmaf(w, dCg[p], ag);
                ab = fmaf(w, dCb[p], ab);
                S[p] = fmaf(-cdot, w, S[p]);                            // colour behind this splat
                const float om = 1.0f - alpha;
                const float inv = om > 0.0f ? fast_rcp(om) : 0.0f;
                const float dalpha = fmaf(T[p], cdot, -(S[p] * inv));   // dL/dalpha
                asig = fmaf(e[p], dalpha, asig);                        // alpha = sig * e
                const float dd = -(alpha * dalpha);                     // dL/ddist
                q0 += dd;
                q1 = fmaf(dd, dY[p], q1);
                q2 = fmaf(dd * dY[p], dY[p], q2);
                T[p] = T[p] - w;
            }
            // per-lane outputs (dX differs per lane; everything below is linear in the partials)
            const float mc = 0.5f * (P.i1 + P.i2);
            const float qx = dX * q0;
            float v[8];
            v[0] = ar; v[1] = ag; v[2] = ab; v[3] = asig;
            v[4] = -fmaf(P.i0, qx, mc * q1);                            // d mu_x (delta = pixel - mu)
            v[5] = -fmaf(mc, qx, P.i3 * q1);                            // d mu_y
            v[6] = 0.5f * dX * qx;                                      // d inv[0]
            v[7] = 0.5f * dX * q1;                                      // d inv[1] = d inv[2]
            float lo, hi;
            reduce8(v, lo, hi);
            const float t9 = wave_sum_to_lane63(0.5f * q2);             // d inv[3]
            // hi's totals move one lane down (lane 14), the ninth two lanes down (lane 61): one
            // atomic wave-instruction then covers the gaussian's 40-byte row with 9 active lanes
            const float hi_s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(hi), 0x101, 0xF, 0xF, true));  // row_shl:1
            const float t9_s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t9), 0x102, 0xF, 0xF, true));  // row_shl:2
            const float outv = take_9 ? t9_s : (take_hi ? hi_s : lo);
            if (ocomp >= 0) atomicAdd(a.g2d + (size_t)sid[k] * 10 + ocomp, outv);
        }
        walked += (uint32_t)cnt;
        if (EARLY) {
            const bool live = !(T[0] < a.t_min) || !(T[1] < a.t_min) || !(T[2] < a.t_min) || !(T[3] < a.t_min);
            if (__ballot(live) == 0ull) break;
        }
    }
    if (lane == 0 && a.walked) atomicAdd(a.walked, (unsigned long long)walked);
}

hipError_t gs_launch_composite_fwd(const GsCompositeArgs &a, hipStream_t s) {
    const int ntiles = a.gx * a.gy;
    if (ntiles <= 0) return hipSuccess;
    const int grid = ((ntiles + 7) / 8) * 8;
    if (a.t_min > 0.0f) hipLaunchKernelGGL(composite_fwd_kernel<true>, dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(composite_fwd_kernel<false>, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t gs_launch_composite_bwd(const GsCompositeArgs &a, hipStream_t s) {
    const int ntiles = a.gx * a.gy;
    if (ntiles <= 0) return hipSuccess;
    const int grid = ((ntiles + 7) / 8) * 8;
    if (a.t_min > 0.0f) hipLaunchKernelGGL(composite_bwd_kernel<true>, dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(composite_bwd_kernel<false>, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}
