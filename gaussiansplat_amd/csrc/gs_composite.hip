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

// ---------------------------------------------------------------- wave64 sum -> lane 63 (DPP)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __int_as_float(t);
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

__device__ __forceinline__ float bcast63(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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

    float dCr[4], dCg[4], dCb[4], T[4], S[4];
    uint32_t walked = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int py = py0 + 4 * p;
        const bool in = (px <= a.W && py <= a.H);
        const size_t o = in ? (size_t)(px - 1) + (size_t)a.W * (py - 1) : 0;
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
            float ar = 0.0f, ag = 0.0f, ab = 0.0f, asig = 0.0f, q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
            bool any = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int py = py0 + 4 * p;
                const float dY = (float)py - P.my;
                float e = fast_exp2(fmaf(dY, fmaf(Cq, dY, B0), A0));
                bool hit = hitx && (py >= ymin) && (py <= ymax);
                if (EARLY) hit = hit && !(T[p] < a.t_min);
                any = any || hit;
                e = hit ? e : 0.0f;
                const float alpha = P.sig * e;
                const float w = alpha * T[p];
                const float cdot = fmaf(P.r, dCr[p], fmaf(P.g, dCg[p], P.b * dCb[p]));
                ar = fmaf(w, dCr[p], ar);
                ag = fmaf(w, dCg[p], ag);
                ab = fmaf(w, dCb[p], ab);
                S[p] = fmaf(-cdot, w, S[p]);                            // colour behind this splat
                const float om = 1.0f - alpha;
                const float inv = om > 0.0f ? fast_rcp(om) : 0.0f;
                const float dalpha = fmaf(T[p], cdot, -(S[p] * inv));   // dL/dalpha
                asig = fmaf(e, dalpha, asig);                           // alpha = sig * e
                const float dd = -(alpha * dalpha);                     // dL/ddist
                q0 += dd;
                q1 = fmaf(dd, dY, q1);
                q2 = fmaf(dd * dY, dY, q2);
                T[p] = T[p] - w;
            }
            if (__ballot(any) == 0ull) continue;                        // nobody in the tile touched it
            // per-lane moments (dX differs per lane), then one wave reduction per quantity
            const float qx = dX * q0, qxx = dX * qx, qxy = dX * q1;
            const float Rr = wave_sum_to_lane63(ar), Rg = wave_sum_to_lane63(ag), Rb = wave_sum_to_lane63(ab);
            const float Rs = wave_sum_to_lane63(asig);
            const float Rx = wave_sum_to_lane63(qx), Ry = wave_sum_to_lane63(q1);
            const float Rxx = wave_sum_to_lane63(qxx), Rxy = wave_sum_to_lane63(qxy), Ryy = wave_sum_to_lane63(q2);
            // totals sit in lane 63: broadcast, finish the 10 outputs, and issue ONE atomic
            // wave-instruction whose lanes 0..9 cover the gaussian's contiguous 40-byte row
            const float tr = bcast63(Rr), tg = bcast63(Rg), tb = bcast63(Rb), tsg = bcast63(Rs);
            const float tx = bcast63(Rx), ty = bcast63(Ry), txx = bcast63(Rxx), txy = bcast63(Rxy), tyy = bcast63(Ryy);
            const float mc = 0.5f * (P.i1 + P.i2);
            float v = tr;
            v = lane == 1 ? tg : v;
            v = lane == 2 ? tb : v;
            v = lane == 3 ? tsg : v;
            v = lane == 4 ? -(P.i0 * tx + mc * ty) : v;                 // d mu_x  (delta = pixel - mu)
            v = lane == 5 ? -(mc * tx + P.i3 * ty) : v;                 // d mu_y
            v = lane == 6 ? 0.5f * txx : v;                             // d inv[0]
            v = (lane == 7 || lane == 8) ? 0.5f * txy : v;              // d inv[1], d inv[2]
            v = lane == 9 ? 0.5f * tyy : v;                             // d inv[3]
            if (lane < 10) atomicAdd(a.g2d + (size_t)sid[k] * 10 + lane, v);
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
