// gs_preprocess_bwd.hip -- per-gaussian chain rule from the 2-D splat gradients back to the
// model parameters (means, scales, quaternions, opacities, SH coefficients).
//
// The reference has no 3-D backward: backward.jl:3-38 launches splatGrads (splat.jl:271-396),
// the adjoint of an older 2-D fitting forward, and grads.jl:1-3 (updateColorGrads) is an empty
// stub.  What is contractual is kept: gradients ACCUMULATE into arrays shaped like the
// parameters (splat.jl:137-156) until resetGrads (splat.jl:158-173).  The mathematics is the
// derived adjoint of the reference FORWARD (projection.jl:39-155, cov2d.jl:30-45,
// splat.jl:175-193), including its quirks: R22 = 1 - 2(x^2 - z^2), the gaussian's own R in
// J*R, +0.3 on all four covariance entries, clip-space view direction for SH.
//
// Two kernels, one thread per gaussian, no atomics (each thread owns its rows); HBM-bound:
//   gs_sh_bwd_kernel    d rgb -> d shs, plus d rgb / d(clip position) handed to the second kernel
//                       (16 B/gaussian).  The SH gradients (3K floats = 192 B at degree 3, thread-strided in
//                       HBM) go through an LDS tile [256][3K+1]: conflict-free per-thread rows (odd stride),
//                       stored/added to d_shs with coalesced accesses.  The SH coefficients are read only for
//                       gaussians a pixel touched, by their own thread (the colour's dependence on the view direction).
//   gs_geom_bwd_kernel  d{mu', invCov2d, sig} -> d{means, scales, quaternions, opacities}.  The chain (3x3 / 2x3 / 2x2
//                       products, the 2x2 inverse and the quaternion terms) is evaluated in fp64 from the fp32 inputs:
//                       in fp32 its cancellations put the quaternion gradient at 2e-4 .. 2e-3 relative L2 of the fp64
//                       adjoint (C3, C5 tile samples) while every other array sits at 2e-5 .. 1e-4; ~400 flops per
//                       gaussian, so the kernel stays HBM-bound.
// Splitting keeps both under 128 VGPRs (the fused version needed 168 + spills).
// `overwrite` stores instead of accumulating: used for the first backward after resetGrads, so
// the reset needs no 4(11+3K)-byte/gaussian zero fill and this pass no read of the old gradients.
#include "gs_common.h"
#include <stdlib.h>

#ifndef GS_NT_STORES
#define GS_NT_STORES 2               // overwrite-mode gradient stores bypass the caches: 1 the SH gradients, 2 the geometry chain's too (0: A/B builds)
#endif
#define SH_C0 0.28209479177387814f
#define SH_C1 0.48860251190291990f
__constant__ float bC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
__constant__ float bC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// 2-D gradient row of gaussian g: float atomics buffer, or the deterministic fixed-point buffer
__device__ __forceinline__ void load_g2(const GsPreprocessBwdArgs &a, int64_t g, float (&o)[10]) {
    if (a.g2d_fixed) {
#pragma unroll
        for (int i = 0; i < 10; ++i) o[i] = (float)((double)a.g2d_fixed[GS_G2D_STRIDE * g + i] * gs_fixed_inv(i));
    } else {                                                           // one 64-byte row: three 16-byte loads
        const float4 *r = reinterpret_cast<const float4 *>(a.g2d + GS_G2D_STRIDE * g);
        const float4 v0 = r[0], v1 = r[1], v2 = r[2];
        o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w; o[8] = v2.x; o[9] = v2.y;
    }
}

// old + v as one rounded add that the compiler may not fuse with the products v came from: the accumulating and the
// overwriting instantiation must produce the SAME v (accumulating eight views == the sum of eight single-view gradients)
__device__ __forceinline__ float add_exact(float old, float v) {
#pragma clang fp contract(off)
    return old + v;
}
// accumulate, or (sgd_scale != 0) apply the step: the same fma gs_sgd_step would do on the stored float gradient
__device__ __forceinline__ float acc_or_step(float old, float v, float sgd_scale) {
    return sgd_scale != 0.0f ? fmaf(sgd_scale, v, old) : add_exact(old, v);
}

// FUSED: the geometry chain of the gaussian follows in the same thread (gs_backward's usual case: both phases) -- its row of 2-D gradients
// and its mean are read once, d L / d tps stays in registers, one launch less.  The two-kernel form remains for callers that run the
// phases apart (a multi-GPU host starts the exchange of the SH gradients between them).
template <int DEG, bool OVERWRITE, bool FUSED>
__global__ __launch_bounds__(256) void gs_sh_bwd_kernel(GsPreprocessBwdArgs a, GsCamera cam) {
#pragma clang fp contract(off)   // the fused and the two-kernel form must produce the same bits: no context-dependent fma formation
    constexpr int K = (DEG + 1) * (DEG + 1);
    constexpr int ROW = 3 * K + 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];       // [256][ROW]
    // Untouched gaussians.  With the transmittance early-out most gaussians of a dense view are never composited (C3: 63 %, C5: 90 %,
    // tools/touched_rows.py): their colour gradient d rgb is exactly zero, hence d shs = basis * 0 and the colour -> direction term are
    // exactly zero.  When ACCUMULATING (views after the first of a batch) their d_shs rows are neither read nor written: C4 on one GPU
    // 8.85 -> 8.71 ms per 8 views, same box (profiles/r04j_ab_untouched_rows.log).  When OVERWRITING the zeros have to be written.
    constexpr bool SKIP = !OVERWRITE;
    __shared__ uint8_t srow_live[256];
    const int64_t gb = (int64_t)blockIdx.x * blockDim.x;
    const int nb = (int)min((int64_t)blockDim.x, a.n - gb);
    float g2[10];
    if (SKIP) {
        const int64_t g0 = gb + threadIdx.x;
        bool lv = false;
        if (g0 < a.n) { load_g2(a, g0, g2); lv = g2[0] != 0.0f || g2[1] != 0.0f || g2[2] != 0.0f; }
        srow_live[threadIdx.x] = lv ? 1 : 0;
        __syncthreads();
    }
    // 3K is a multiple of 4 only for K = 4, 16 ... : use 16-byte global accesses when it is
    constexpr bool VEC = (3 * K) % 4 == 0;
    const bool vec_out = VEC && (reinterpret_cast<uintptr_t>(a.d_shs) & 15) == 0;   // e.g. a flat buffer slice at 44 n bytes
    const int64_t g = gb + threadIdx.x;
    if (g < a.n) {
        if (!SKIP) load_g2(a, g, g2);
        // a gaussian no pixel touched (off screen, or skipped by the composite because its per-view payload is not finite:
        // tz == 0, exp(scale) overflow, singular covariance) has an all-zero row: its gradient is exactly zero, and the
        // recomputed direction may be NaN, so zeros are substituted for the basis instead of forming 0 * NaN
        const bool live = g2[0] != 0.0f || g2[1] != 0.0f || g2[2] != 0.0f;
        const float grgb[3] = {g2[0], g2[1], g2[2]};
        const float *T = cam.T, *P = cam.P;
        const float m1 = a.means[3 * g], m2 = a.means[3 * g + 1], m3 = a.means[3 * g + 2];
        float t[4], p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = T[i] * m1 + T[i + 4] * m2 + T[i + 8] * m3 + T[i + 12];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = P[i] * t[0] + P[i + 4] * t[1] + P[i + 8] * t[2] + P[i + 12] * t[3];
        // ---- rgb(sh, dir(p)), splat.jl:180-193
        const float v0 = p[0] - (cam.lookAt[0] - cam.eye[0]);
        const float v1 = p[1] - (cam.lookAt[1] - cam.eye[1]);
        const float v2 = p[2] - (cam.lookAt[2] - cam.eye[2]);
        const float inrm = live ? rsqrtf(v0 * v0 + v1 * v1 + v2 * v2) : 0.0f;
        const float X = live ? v0 * inrm : 0.0f, Y = live ? v1 * inrm : 0.0f, Z = live ? v2 * inrm : 0.0f;
        float bs[K];
        bs[0] = SH_C0;
        if constexpr (DEG >= 1) { bs[1] = -Y * SH_C1; bs[2] = Z * SH_C1; bs[3] = -X * SH_C1; }
        if constexpr (DEG >= 2) {
            const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
            bs[4] = bC2[0] * xy; bs[5] = bC2[1] * yz; bs[6] = bC2[2] * (2 * zz - xx - yy); bs[7] = bC2[3] * xz; bs[8] = bC2[4] * (xx - yy);
            if constexpr (DEG >= 3) {
                bs[9] = bC3[0] * Y * (3 * xx - yy); bs[10] = bC3[1] * xy * Z; bs[11] = bC3[2] * Y * (4 * zz - xx - yy);
                bs[12] = bC3[3] * Z * (2 * zz - 3 * xx - 3 * yy); bs[13] = bC3[4] * X * (4 * zz - xx - yy);
                bs[14] = bC3[5] * Z * (xx - yy); bs[15] = bC3[6] * X * (xx - 3 * yy);
            }
        }
        float *sh = tile + threadIdx.x * ROW;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) sh[c + 3 * k] = bs[k] * grgb[c];   // the tile carries d L / d sh (stored coalesced below)
        }
        float ddir[3] = {0.0f, 0.0f, 0.0f};
        if (live) {
            // d L / d dir through the colour needs the coefficients: those of THIS gaussian, read by its own thread as twelve 16-byte
            // loads, and only if a pixel touched it (37 % of the gaussians at C3, 10 % at C5) -- cs[k] = d rgb . sh[:, k], then the
            // derivative polynomials of the basis.  (Until round 4 every row came in through the LDS tile, touched or not: 192 of the
            // kernel's 476 bytes per gaussian.  A 3 x 3 Jacobian d rgb / d dir written by the preprocess instead was measured too: the
            // backward 10 us faster than this, the preprocess 30 us slower -- 131 VGPRs, three waves per SIMD; profiles/r04s_ab_lazy_sh.log.)
            const float4 *row = reinterpret_cast<const float4 *>(a.shs + (int64_t)3 * K * g);
            float shv[3 * K];
            if constexpr ((3 * K) % 4 == 0) {
#pragma unroll
                for (int q = 0; q < 3 * K / 4; ++q) { const float4 t4 = row[q]; shv[4 * q] = t4.x; shv[4 * q + 1] = t4.y; shv[4 * q + 2] = t4.z; shv[4 * q + 3] = t4.w; }
            } else {
#pragma unroll
                for (int q = 0; q < 3 * K; ++q) shv[q] = a.shs[(int64_t)3 * K * g + q];
            }
            float cs[K];
#pragma unroll
            for (int k = 0; k < K; ++k) cs[k] = grgb[0] * shv[3 * k] + grgb[1] * shv[3 * k + 1] + grgb[2] * shv[3 * k + 2];
            if constexpr (DEG >= 1) { ddir[0] += -SH_C1 * cs[3]; ddir[1] += -SH_C1 * cs[1]; ddir[2] += SH_C1 * cs[2]; }
            if constexpr (DEG >= 2) {
                ddir[0] += bC2[0] * Y * cs[4] - 2 * bC2[2] * X * cs[6] + bC2[3] * Z * cs[7] + 2 * bC2[4] * X * cs[8];
                ddir[1] += bC2[0] * X * cs[4] + bC2[1] * Z * cs[5] - 2 * bC2[2] * Y * cs[6] - 2 * bC2[4] * Y * cs[8];
                ddir[2] += bC2[1] * Y * cs[5] + 4 * bC2[2] * Z * cs[6] + bC2[3] * X * cs[7];
            }
            if constexpr (DEG >= 3) {
                const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
                ddir[0] += 6 * bC3[0] * xy * cs[9] + bC3[1] * yz * cs[10] - 2 * bC3[2] * xy * cs[11] - 6 * bC3[3] * xz * cs[12]
                           + bC3[4] * (4 * zz - 3 * xx - yy) * cs[13] + 2 * bC3[5] * xz * cs[14] + bC3[6] * (3 * xx - 3 * yy) * cs[15];
                ddir[1] += bC3[0] * (3 * xx - 3 * yy) * cs[9] + bC3[1] * xz * cs[10] + bC3[2] * (4 * zz - xx - 3 * yy) * cs[11]
                           - 6 * bC3[3] * yz * cs[12] - 2 * bC3[4] * xy * cs[13] - 2 * bC3[5] * yz * cs[14] - 6 * bC3[6] * xy * cs[15];
                ddir[2] += bC3[1] * xy * cs[10] + 8 * bC3[2] * yz * cs[11] + bC3[3] * (6 * zz - 3 * xx - 3 * yy) * cs[12]
                           + 8 * bC3[4] * xz * cs[13] + bC3[5] * (xx - yy) * cs[14];
            }
        }
        const float dd = X * ddir[0] + Y * ddir[1] + Z * ddir[2];
        // d L / d tps[1:3] through the colour (dir = normalize(tps[1:3] - (lookAt - eye)))
        const float4 dpc_sh = make_float4((ddir[0] - X * dd) * inrm, (ddir[1] - Y * dd) * inrm, (ddir[2] - Z * dd) * inrm, 0.0f);
        if (FUSED) {
            const float (&g2f)[10] = g2;
#define GS_GEOM_DPC dpc_sh
            do {
#include "gs_geom_bwd_body.inc"
            } while (0);
#undef GS_GEOM_DPC
        } else reinterpret_cast<float4 *>(a.dpc)[g] = dpc_sh;
    }
    __syncthreads();
    if (a.d_shs) {
        if (vec_out) {
            float4 *dst = reinterpret_cast<float4 *>(a.d_shs + gb * 3 * K);
            for (int i4 = threadIdx.x; i4 < nb * (3 * K / 4); i4 += blockDim.x) {
                const int row = (i4 * 4) / (3 * K);
                if (!OVERWRITE && !srow_live[row]) continue;             // + 0 (or a step of 0): the row stays as it is
                const float *t = tile + row * ROW + (i4 * 4) % (3 * K);
                float4 v = make_float4(t[0], t[1], t[2], t[3]);
                if (!OVERWRITE) {
                    const float4 o = dst[i4];
                    if (a.sgd_scale != 0.0f) { v.x = fmaf(a.sgd_scale, v.x, o.x); v.y = fmaf(a.sgd_scale, v.y, o.y); v.z = fmaf(a.sgd_scale, v.z, o.z); v.w = fmaf(a.sgd_scale, v.w, o.w); }
                    else { v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                }
                if (OVERWRITE && GS_NT_STORES) {
                    // streamed past the caches: 192 B per gaussian that nothing on the GPU reads again this frame.  Stored normally they
                    // pushed the model's SH rows out of the 256 MB infinity cache, and the next frame's preprocess fetched them from
                    // HBM again (C3: preprocess 86 -> 73 us, same box, profiles/r04t_ab_nontemporal.log).  (Accumulating views read the
                    // rows back: cached stores.)
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f vv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(vv, reinterpret_cast<v4f *>(dst) + i4);
                } else dst[i4] = v;
            }
        } else {
            for (int idx = threadIdx.x; idx < nb * 3 * K; idx += blockDim.x) {
                if (!OVERWRITE && !srow_live[idx / (3 * K)]) continue;
                const float v = tile[(idx / (3 * K)) * ROW + idx % (3 * K)];
                if (OVERWRITE) a.d_shs[gb * 3 * K + idx] = v;
                else if (a.sgd_scale != 0.0f) a.d_shs[gb * 3 * K + idx] = fmaf(a.sgd_scale, v, a.d_shs[gb * 3 * K + idx]);
                else a.d_shs[gb * 3 * K + idx] += v;
            }
        }
    }
}

template <bool OVERWRITE>
__global__ __launch_bounds__(256) void gs_geom_bwd_kernel(GsPreprocessBwdArgs a, GsCamera cam) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n) return;
    float g2f[10];
    load_g2(a, g, g2f);                                  // colour gradient + raw moments (gs_common.h: gs_g2d_to_grads)
#define GS_GEOM_DPC (reinterpret_cast<const float4 *>(a.dpc)[g])
    do {
#include "gs_geom_bwd_body.inc"
    } while (0);
#undef GS_GEOM_DPC
}

// ---------------------------------------------------------------- colour-factored gradient exchange (multi-GPU)
// d L / d sh[k][c] of one view is basis_k(dir_view) * d rgb[c]: rank one in (k, c).  A multi-view step therefore need
// not all-reduce the 3K floats per gaussian (48 at degree 3 = 81 % of the gradient buffer): ranks exchange the THREE
// floats d rgb per (view, gaussian) -- an all-gather of 12 B instead of an all-reduce of 192 B per gaussian -- and
// every rank rebuilds sum_v basis(dir_v) (x) d rgb_v locally from the cameras it already knows.
__global__ __launch_bounds__(256) void gs_pack_drgb_kernel(const float *__restrict__ g2d, const long long *__restrict__ g2d_fixed,
                                                            float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over 3n
    if (i >= 3 * n) return;
    const int64_t g = i / 3; const int c = (int)(i - 3 * g);
    out[i] = g2d_fixed ? (float)((double)g2d_fixed[GS_G2D_STRIDE * g + c] * GS_FIXED_INV) : g2d[GS_G2D_STRIDE * g + c];
}

// cams: nviews records of 38 floats {T[16], P[16], eye[3], lookAt[3]}; drgb: [nviews][3n]; the direction and the
// basis are computed exactly as in gs_sh_bwd_kernel.
template <int DEG, bool OVERWRITE>
__global__ __launch_bounds__(256) void gs_sh_from_views_kernel(int64_t n, const float *__restrict__ means, int nviews,
                                                                const float *__restrict__ cams, const float *__restrict__ drgb,
                                                                float *__restrict__ d_shs) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    constexpr int ROW = 3 * K + 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];       // [256][ROW] accumulators
    const int64_t gb = (int64_t)blockIdx.x * blockDim.x;
    const int nb = (int)min((int64_t)blockDim.x, n - gb);
    const int64_t g = gb + threadIdx.x;
    float *acc = tile + threadIdx.x * ROW;
#pragma unroll
    for (int i = 0; i < 3 * K; ++i) acc[i] = 0.0f;
    if (g < n) {
        const float m1 = means[3 * g], m2 = means[3 * g + 1], m3 = means[3 * g + 2];
        for (int v = 0; v < nviews; ++v) {
            const float *T = cams + 38 * v, *P = T + 16, *eye = T + 32, *lookAt = T + 35;
            float t[4], p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = T[i] * m1 + T[i + 4] * m2 + T[i + 8] * m3 + T[i + 12];
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = P[i] * t[0] + P[i + 4] * t[1] + P[i + 8] * t[2] + P[i + 12] * t[3];
            const float v0 = p[0] - (lookAt[0] - eye[0]);
            const float v1 = p[1] - (lookAt[1] - eye[1]);
            const float v2 = p[2] - (lookAt[2] - eye[2]);
            const float inrm = rsqrtf(v0 * v0 + v1 * v1 + v2 * v2);
            const float X = v0 * inrm, Y = v1 * inrm, Z = v2 * inrm;
            float bs[K];
            bs[0] = SH_C0;
            if constexpr (DEG >= 1) { bs[1] = -Y * SH_C1; bs[2] = Z * SH_C1; bs[3] = -X * SH_C1; }
            if constexpr (DEG >= 2) {
                const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
                bs[4] = bC2[0] * xy; bs[5] = bC2[1] * yz; bs[6] = bC2[2] * (2 * zz - xx - yy); bs[7] = bC2[3] * xz; bs[8] = bC2[4] * (xx - yy);
                if constexpr (DEG >= 3) {
                    bs[9] = bC3[0] * Y * (3 * xx - yy); bs[10] = bC3[1] * xy * Z; bs[11] = bC3[2] * Y * (4 * zz - xx - yy);
                    bs[12] = bC3[3] * Z * (2 * zz - 3 * xx - 3 * yy); bs[13] = bC3[4] * X * (4 * zz - xx - yy);
                    bs[14] = bC3[5] * Z * (xx - yy); bs[15] = bC3[6] * X * (xx - 3 * yy);
                }
            }
            const float *gr = drgb + (size_t)v * 3 * (size_t)n + 3 * g;
            const float g0 = gr[0], g1 = gr[1], g2 = gr[2];
#pragma unroll
            for (int k = 0; k < K; ++k) { acc[3 * k] += bs[k] * g0; acc[3 * k + 1] += bs[k] * g1; acc[3 * k + 2] += bs[k] * g2; }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nb * 3 * K; idx += blockDim.x) {     // coalesced rows
        const float v = tile[(idx / (3 * K)) * ROW + idx % (3 * K)];
        if (OVERWRITE) d_shs[gb * 3 * K + idx] = v; else d_shs[gb * 3 * K + idx] += v;
    }
}

hipError_t gs_launch_pack_drgb(const float *g2d, const long long *g2d_fixed, float *out, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_pack_drgb_kernel, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, s, g2d, g2d_fixed, out, n);
    return hipGetLastError();
}

hipError_t gs_launch_sh_from_views(int64_t n, int sh_degree, const float *means, int nviews, const float *cams, const float *drgb,
                                   float *d_shs, int overwrite, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 block(256), grid((unsigned)((n + 255) / 256));
    const int K = (sh_degree + 1) * (sh_degree + 1);
    const size_t lds = sizeof(float) * 256 * (3 * K + 1);
#define GS_SV(D) do { if (overwrite) hipLaunchKernelGGL((gs_sh_from_views_kernel<D, true>), grid, block, lds, s, n, means, nviews, cams, drgb, d_shs); \
                      else hipLaunchKernelGGL((gs_sh_from_views_kernel<D, false>), grid, block, lds, s, n, means, nviews, cams, drgb, d_shs); } while (0)
    switch (sh_degree) {
        case 0: GS_SV(0); break;
        case 1: GS_SV(1); break;
        case 2: GS_SV(2); break;
        case 3: GS_SV(3); break;
        default: return hipErrorInvalidValue;
    }
#undef GS_SV
    return hipGetLastError();
}

// gaussians per workgroup of the SH kernel: its LDS tile is (3K + 1) floats per gaussian (49 at SH degree 3), so 256 gaussians
// allow three workgroups per CU, 128 six (64 / 128 / 256 measured equal: profiles/HISTORY.md)
#ifndef GS_SHBWD_THREADS
#define GS_SHBWD_THREADS 256
#endif
static int sh_bwd_threads() { return GS_SHBWD_THREADS; }
hipError_t gs_launch_preprocess_bwd(const GsPreprocessBwdArgs &a, const GsCamera &cam, hipStream_t s, int phases) {
    if (a.n <= 0) return hipSuccess;
    const int T = sh_bwd_threads();
    dim3 block(T), grid((unsigned)((a.n + T - 1) / T));
    const int K = (a.sh_degree + 1) * (a.sh_degree + 1);
    const size_t lds = sizeof(float) * T * (3 * K + 1);
#define GS_SH2(D, F) do { if (a.overwrite) hipLaunchKernelGGL((gs_sh_bwd_kernel<D, true, F>), grid, block, lds, s, a, cam); \
                         else hipLaunchKernelGGL((gs_sh_bwd_kernel<D, false, F>), grid, block, lds, s, a, cam); } while (0)
#define GS_SH(D) do { if (fused) GS_SH2(D, true); else GS_SH2(D, false); } while (0)
    const bool fused = (phases & 3) == 3;
    if (phases & 1)
    switch (a.sh_degree) {
        case 0: GS_SH(0); break;
        case 1: GS_SH(1); break;
        case 2: GS_SH(2); break;
        case 3: GS_SH(3); break;
        default: return hipErrorInvalidValue;
    }
    block = dim3(256); grid = dim3((unsigned)((a.n + 255) / 256));
    if ((phases & 2) && !fused) {
        if (a.overwrite) hipLaunchKernelGGL(gs_geom_bwd_kernel<true>, grid, block, 0, s, a, cam);
        else hipLaunchKernelGGL(gs_geom_bwd_kernel<false>, grid, block, 0, s, a, cam);
    }
    return hipGetLastError();
}
