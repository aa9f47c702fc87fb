// gs_preprocess2d.hip -- the 2-D image-fitting renderer (RendererType GAUSSIAN_2D) for gfx950.
//
// Forward: preprocess(::GaussianRenderer2D) (reference src/forward.jl:9-33) = computeCov2d_kernel
// (src/cov2d.jl:3-28: Sigma = R(theta) diag(exp s)^2 R' with +0.3 on the diagonal), computeInvCov2d
// (src/cov2d.jl:30-45) and computeBB (src/boundingbox.jl:4-36), three launches with a device sync each in the
// reference, one kernel here.  It emits the SAME 64-byte payload as the 3-D preprocess, so binning and both
// composite kernels are shared.  SplatData2D (src/splat.jl:20-26): means 2xN in [0,1]^2 (pixel position
// (w*mx, h*my), splat.jl:337-339), scales 2xN (log), rotations 1xN, opacities 1xN (used raw, splat.jl:341),
// colors 3xN.  The reference hands the [0,1] means to computeBB (forward.jl:25-31), which would pin every box
// to the image origin; the pixel position is used instead.
//
// Backward: the chain from the composite's per-gaussian sums d{rgb3, sig, mu2, inv4} to SplatGrads2D
// (src/splat.jl:28-34).  The reference's splatGrads (splat.jl:271-396) mixes several forwards
// and is not reproduced; this is the derived adjoint, checked against fp64 autograd of the restated forward.
//
// Compiled with -ffp-contract=off: tile rectangles must be bit-identical to the CPU oracle.
// HBM-bound: 36 B read + 60 B written per gaussian forward, 76 B read + 36 B written backward.
#include "gs_common.h"
#include "gs_detmath.h"

#pragma clang fp contract(off)

__device__ __forceinline__ int gs2_tile_div(float v) {
    v = fminf(fmaxf(v, -1073741824.0f), 1073741824.0f);
    return (int)v / GS_TILE;
}
__device__ __forceinline__ uint32_t gs2_pack_i16(float lo, float hi) {
    const int a = (int)fminf(fmaxf(lo, -32768.0f), 32767.0f);
    const int b = (int)fminf(fmaxf(hi, -32768.0f), 32767.0f);
    return ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
}

__global__ __launch_bounds__(256) void gs_preprocess2d_kernel(GsPreprocess2DArgs a) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n) return;
    // ---- computeCov2d_kernel, cov2d.jl:5-26
    float sn, cs;
    gs_sincosf(a.rots[g], sn, cs);
    const float R[2][2] = {{cs, -sn}, {sn, cs}};
    const float S[2][2] = {{gs_expf(a.scales[2 * g]), 0.0f}, {0.0f, gs_expf(a.scales[2 * g + 1])}};
    float Wm[2][2], Jm[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s = R[i][0] * S[0][j];
            s = s + R[i][1] * S[1][j];
            Wm[i][j] = s;                                              // :18 W = R*S
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s = Wm[i][0] * Wm[j][0];
            s = s + Wm[i][1] * Wm[j][1];
            Jm[i][j] = s;                                              // :19 J = W*W'
        }
    const float a0 = (float)((double)Jm[0][0] + 0.3);                  // :25 (Float64 literal, diagonal only)
    const float a1 = Jm[1][0], a2 = Jm[0][1];
    const float a3 = (float)((double)Jm[1][1] + 0.3);                  // :26
    // ---- computeInvCov2d, cov2d.jl:30-45
    const float det = a0 * a3 - a2 * a1;
    const float idet = 1.0f / det;
    const float inv0 = a3 * idet, inv1 = -(a1 * idet), inv2 = -(a2 * idet), inv3 = a0 * idet;
    // ---- pixel position, splat.jl:337-339
    const float mux = (float)a.W * a.means[2 * g], muy = (float)a.H * a.means[2 * g + 1];
    // ---- computeBB, boundingbox.jl:19-27
    const float halfad = (a0 + a3) / 2.0f;
    const double disc = (double)(halfad * halfad - det);
    const double sq = sqrt(gs_jlmax(0.1, disc));
    const double e1 = (double)halfad - sq, e2 = (double)halfad + sq;
    const double r = ceil(3.0 * sqrt(gs_jlmax(e1, e2)));
    const float bxmin = (float)gs_jlmax(1.0, floor(-r + (double)mux));
    const float bxmax = (float)gs_jlmin((double)a.W, ceil(r + (double)mux));
    const float bymin = (float)gs_jlmax(1.0, floor(-r + (double)muy));
    const float bymax = (float)gs_jlmin((double)a.H, ceil(r + (double)muy));
    // ---- tile rectangle, binning.jl:6-27 (non-finite boxes: dropped, as in the 3-D path)
    const bool finite_bb = isfinite(bxmin) && isfinite(bxmax) && isfinite(bymin) && isfinite(bymax);
    uint16_t rc[4] = {0, 0, 0, 0};
    if (finite_bb) {
        int bminx = gs2_tile_div(floorf(bxmin)) + 1, bmaxx = gs2_tile_div(ceilf(bxmax)) + 1;
        int bminy = gs2_tile_div(floorf(bymin)) + 1, bmaxy = gs2_tile_div(ceilf(bymax)) + 1;
        if (bminx <= bmaxx && bminy <= bmaxy) {
            bminx = max(bminx, 1); bminy = max(bminy, 1);
            bmaxx = min(bmaxx, a.gx); bmaxy = min(bmaxy, a.gy);
            if (bminx <= bmaxx && bminy <= bmaxy) {
                rc[0] = (uint16_t)bminx; rc[1] = (uint16_t)bmaxx; rc[2] = (uint16_t)bminy; rc[3] = (uint16_t)bmaxy;
            }
        }
    }
    // raw opacity (splat.jl:341) clamped to [0, 1): alpha must stay below 1 for the adjoint's 1/(1-alpha); NaN -> 0
    const float sg = fminf(fmaxf(a.opac[g], 0.0f), 0.99999994f);
    const float cr = a.colors[3 * g], cg = a.colors[3 * g + 1], cb = a.colors[3 * g + 2];
    GsPayload p;
    const bool pay_ok = isfinite(cr) && isfinite(cg) && isfinite(cb) && isfinite(sg) && isfinite(mux) && isfinite(muy) &&
                        isfinite(inv0) && isfinite(inv1) && isfinite(inv2) && isfinite(inv3);
    p.mx = mux; p.my = muy; p.sig = sg;
    p.l2s = fminf(__builtin_amdgcn_logf(sg), GS_L2S_CAP);             // log2; sg == 0 gives -inf: alpha = exp2(-inf) = 0
    p.ka = GS_NEG_HALF_LOG2E * inv0; p.kb = GS_NEG_HALF_LOG2E * (inv1 + inv2); p.kc = GS_NEG_HALF_LOG2E * inv3;
    p.r = cr; p.g = cg; p.b = cb;
    if (finite_bb && pay_ok) { p.bbx = gs2_pack_i16(bxmin, bxmax); p.bby = gs2_pack_i16(bymin, bymax); }
    else { p.bbx = 1u; p.bby = 1u; }                                   // min 1, max 0: empty
    gs_payload_box_edges(p);
    a.payload[g] = p;
    reinterpret_cast<float4 *>(a.invcov)[g] = make_float4(inv0, inv1, inv2, inv3);
    a.depth_key[g] = 0u;                                               // no depth: lists are in gaussian-index order
    reinterpret_cast<uint2 *>(a.rect)[g] = make_uint2((uint32_t)rc[0] | ((uint32_t)rc[1] << 16),
                                                       (uint32_t)rc[2] | ((uint32_t)rc[3] << 16));
    if (a.dbg.mu) {
        a.dbg.mu[2 * g] = mux; a.dbg.mu[2 * g + 1] = muy;
        a.dbg.cov2d[4 * g] = a0; a.dbg.cov2d[4 * g + 1] = a1; a.dbg.cov2d[4 * g + 2] = a2; a.dbg.cov2d[4 * g + 3] = a3;
        a.dbg.invcov[4 * g] = inv0; a.dbg.invcov[4 * g + 1] = inv1; a.dbg.invcov[4 * g + 2] = inv2; a.dbg.invcov[4 * g + 3] = inv3;
        a.dbg.bbs[4 * g] = bxmin; a.dbg.bbs[4 * g + 1] = bymin; a.dbg.bbs[4 * g + 2] = bxmax; a.dbg.bbs[4 * g + 3] = bymax;
    }
}

// d{rgb3, sig, mu2, inv4} -> d{means2, scales2, rotation, opacity, colors3}; accumulate (+=) or overwrite.
template <bool OVERWRITE>
__global__ __launch_bounds__(256) void gs_preprocess2d_bwd_kernel(GsPreprocess2DBwdArgs a) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n) return;
    float g2[10];
    if (a.g2d_fixed) {
#pragma unroll
        for (int i = 0; i < 10; ++i) g2[i] = (float)((double)a.g2d_fixed[GS_G2D_STRIDE * g + i] * gs_fixed_inv(i));
    } else {
#pragma unroll
        for (int i = 0; i < 10; ++i) g2[i] = a.g2d[GS_G2D_STRIDE * g + i];
    }
    // forward pieces again (cheaper than storing them): Sigma = Wm Wm' + 0.3 I, M = Sigma^-1
    float sn, cs;
    gs_sincosf(a.rots[g], sn, cs);
    const float e0 = gs_expf(a.scales[2 * g]), e1 = gs_expf(a.scales[2 * g + 1]);
    const float Wm[2][2] = {{cs * e0, -sn * e1}, {sn * e0, cs * e1}};
    const float c00 = Wm[0][0] * Wm[0][0] + Wm[0][1] * Wm[0][1] + 0.3f;
    const float c01 = Wm[0][0] * Wm[1][0] + Wm[0][1] * Wm[1][1];
    const float c11 = Wm[1][0] * Wm[1][0] + Wm[1][1] * Wm[1][1] + 0.3f;
    const float idet = 1.0f / (c00 * c11 - c01 * c01);
    const float M[2][2] = {{c11 * idet, -c01 * idet}, {-c01 * idet, c00 * idet}};
    const float op = a.opac[g];
    gs_g2d_to_grads(g2, fminf(fmaxf(op, 0.0f), 0.99999994f), M[0][0], M[0][1], M[1][1]);      // raw moments -> d{sig, mu, conic}
    const float G[2][2] = {{g2[6], g2[8]}, {g2[7], g2[9]}};            // column-major inv4: [r + 2c]
    // dSigma = -M' G M'
    float t1[2][2], dcov[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) t1[i][j] = M[0][i] * G[0][j] + M[1][i] * G[1][j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dcov[i][j] = -(t1[i][0] * M[j][0] + t1[i][1] * M[j][1]);
    // Sigma = Wm Wm': dWm = (dSigma + dSigma') Wm ; Wm = R(theta) diag(e)
    float dWm[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            dWm[i][j] = (dcov[i][0] + dcov[0][i]) * Wm[0][j] + (dcov[i][1] + dcov[1][i]) * Wm[1][j];
    const float R[2][2] = {{cs, -sn}, {sn, cs}}, dR[2][2] = {{-sn, -cs}, {cs, -sn}};
    const float e[2] = {e0, e1};
    float ds[2], dth = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float de = R[0][j] * dWm[0][j] + R[1][j] * dWm[1][j];
        dth = dth + (dR[0][j] * dWm[0][j] + dR[1][j] * dWm[1][j]) * e[j];
        ds[j] = de * e[j];
    }
    const float dm0 = (float)a.W * g2[4], dm1 = (float)a.H * g2[5];
    if (a.d_means) {
        if (OVERWRITE) { a.d_means[2 * g] = dm0; a.d_means[2 * g + 1] = dm1; }
        else { a.d_means[2 * g] += dm0; a.d_means[2 * g + 1] += dm1; }
    }
    if (a.d_scales) {
        if (OVERWRITE) { a.d_scales[2 * g] = ds[0]; a.d_scales[2 * g + 1] = ds[1]; }
        else { a.d_scales[2 * g] += ds[0]; a.d_scales[2 * g + 1] += ds[1]; }
    }
    if (a.d_rots) { if (OVERWRITE) a.d_rots[g] = dth; else a.d_rots[g] += dth; }
    const float dop = (op > 0.0f && op < 0.99999994f) ? g2[3] : 0.0f;       // the clamp of the forward has zero slope outside
    if (a.d_opac) { if (OVERWRITE) a.d_opac[g] = dop; else a.d_opac[g] += dop; }
    if (a.d_colors) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { if (OVERWRITE) a.d_colors[3 * g + k] = g2[k]; else a.d_colors[3 * g + k] += g2[k]; }
    }
}

hipError_t gs_launch_preprocess2d(const GsPreprocess2DArgs &a, hipStream_t stream) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_preprocess2d_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t gs_launch_preprocess2d_bwd(const GsPreprocess2DBwdArgs &a, hipStream_t stream) {
    if (a.n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.n + 255) / 256)), block(256);
    if (a.overwrite) hipLaunchKernelGGL(gs_preprocess2d_bwd_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(gs_preprocess2d_bwd_kernel<false>, grid, block, 0, stream, a);
    return hipGetLastError();
}
