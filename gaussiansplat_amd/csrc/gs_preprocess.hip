// gs_preprocess.hip -- fused per-gaussian preprocess for gfx950.
//
// Replaces five reference launches (each with a device sync) by one kernel, one thread per
// gaussian: frustumCulling (src/projection.jl:39-100), tValues (:103-155), computeInvCov2d
// (src/cov2d.jl:30-45), computeBB (src/boundingbox.jl:4-36), plus sh2color / cusigmoid
// hoisted out of the per-pixel loop (src/splat.jl:175-193; they depend on gaussian and view
// only), the depth radix key for forward.jl:103 and the tile rectangle of hitBinning
// (src/binning.jl:3-35).
//
// This translation unit is compiled with -ffp-contract=off: every fp32 operation below is
// individually rounded in the reference's written order, so tile rectangles and depth keys
// are bit-identical to the CPU oracle.  HBM-bound: 4*(11+3K) B read +
// 60 B written per gaussian; no LDS, no MFMA.
#include "gs_common.h"
#include "gs_detmath.h"

#pragma clang fp contract(off)

#define SH_C0 0.28209479177387814f
#define SH_C1 0.48860251190291990f
__constant__ float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

__device__ __forceinline__ int gs_tile_div(float v) {
    // Julia div(v, 16f0) + saturation (see oracle gso_tile_rect): v is integer valued.
    v = fminf(fmaxf(v, -1073741824.0f), 1073741824.0f);
    return (int)v / GS_TILE;                       // C integer division truncates like div()
}

__device__ __forceinline__ uint32_t gs_pack_i16(float lo, float hi) {
    int a = (int)fminf(fmaxf(lo, -32768.0f), 32767.0f);
    int b = (int)fminf(fmaxf(hi, -32768.0f), 32767.0f);
    return ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
}

template <int DEG>
__global__ __launch_bounds__(256) void gs_preprocess_kernel(GsPreprocessArgs a, GsCamera cam) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n) return;
    const float *T = cam.T, *P = cam.P;

    // ---- frustumCulling, projection.jl:46-93
    const float m1 = a.means[3 * g], m2 = a.means[3 * g + 1], m3 = a.means[3 * g + 2];
    float ts[4], tps[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = T[i] * m1;
        s = s + T[i + 4] * m2;
        s = s + T[i + 8] * m3;
        s = s + T[i + 12] * 1.0f;
        ts[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = P[i] * ts[0];
        s = s + P[i + 4] * ts[1];
        s = s + P[i + 8] * ts[2];
        s = s + P[i + 12] * ts[3];
        tps[i] = s;
    }
    const double cx = cam.W / 2.0, cy = cam.H / 2.0;                  // forward.jl:58-59 (Float64)
    const float wf = (float)cam.W, hf = (float)cam.H;
    const float mux = (float)((double)((wf * tps[0] / tps[3] + 1.0f) / 2.0f) + cx);   // projection.jl:88
    const float muy = (float)((double)((hf * tps[1] / tps[3] + 1.0f) / 2.0f) + cy);   // :89

    // ---- tValues, projection.jl:107-152
    const float tx = ts[0], ty = ts[1], tz = ts[2];
    float J[2][3];
    J[0][0] = cam.fx / tz; J[1][0] = 0.0f;
    J[0][1] = 0.0f;        J[1][1] = cam.fy / tz;
    J[0][2] = -cam.fx * tx / (tz * tz);
    J[1][2] = -cam.fy * ty / (tz * tz);
    const float qw = a.quats[4 * g], qx = a.quats[4 * g + 1], qy = a.quats[4 * g + 2], qz = a.quats[4 * g + 3];
    float R[3][3];                                                    // quatToRot, projection.jl:1-14
    R[0][0] = 1.0f - 2.0f * (qy * qy + qz * qz);
    R[1][0] = 2.0f * (qx * qy + qw * qz);
    R[2][0] = 2.0f * (qx * qz - qw * qy);
    R[0][1] = 2.0f * (qx * qy - qw * qz);
    R[1][1] = 1.0f - 2.0f * (qx * qx - qz * qz);                      // :8 (minus, as written)
    R[2][1] = 2.0f * (qy * qz + qw * qx);
    R[0][2] = 2.0f * (qx * qz + qw * qy);
    R[1][2] = 2.0f * (qy * qz - qw * qx);
    R[2][2] = 1.0f - 2.0f * (qx * qx + qy * qy);
    float S[3][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    S[0][0] = gs_expf(a.scales[3 * g]);
    S[1][1] = gs_expf(a.scales[3 * g + 1]);
    S[2][2] = gs_expf(a.scales[3 * g + 2]);
    float Wm[3][3], C3[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = R[i][0] * S[0][j];
            s = s + R[i][1] * S[1][j];
            s = s + R[i][2] * S[2][j];
            Wm[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = Wm[i][0] * Wm[j][0];
            s = s + Wm[i][1] * Wm[j][1];
            s = s + Wm[i][2] * Wm[j][2];
            C3[i][j] = s;
        }
    float JR[2][3], JCR[2][3], c2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = J[i][0] * R[0][j];
            s = s + J[i][1] * R[1][j];
            s = s + J[i][2] * R[2][j];
            JR[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = JR[i][0] * C3[0][j];
            s = s + JR[i][1] * C3[1][j];
            s = s + JR[i][2] * C3[2][j];
            JCR[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s = JCR[i][0] * JR[j][0];
            s = s + JCR[i][1] * JR[j][1];
            s = s + JCR[i][2] * JR[j][2];
            c2[i][j] = (float)((double)s + 0.3);                      // :150 (+0.3 on all four, Float64)
        }
    const float a0 = c2[0][0], a1 = c2[1][0], a2 = c2[0][1], a3 = c2[1][1];

    // ---- computeInvCov2d, cov2d.jl:30-45
    const float det = a0 * a3 - a2 * a1;
    const float idet = 1.0f / det;
    const float inv0 = a3 * idet, inv1 = -(a1 * idet), inv2 = -(a2 * idet), inv3 = a0 * idet;

    // ---- computeBB, boundingbox.jl:19-27
    const float halfad = (a0 + a3) / 2.0f;
    const double disc = (double)(halfad * halfad - det);
    const double sq = sqrt(gs_jlmax(0.1, disc));
    const double e1 = (double)halfad - sq;
    const double e2 = (double)halfad + sq;
    const double r = ceil(3.0 * sqrt(gs_jlmax(e1, e2)));
    const float bxmin = (float)gs_jlmax(1.0, floor(-r + (double)mux));
    const float bxmax = (float)gs_jlmin((double)cam.W, ceil(r + (double)mux));
    const float bymin = (float)gs_jlmax(1.0, floor(-r + (double)muy));
    const float bymax = (float)gs_jlmin((double)cam.H, ceil(r + (double)muy));

    // ---- sh2color, splat.jl:180-193 (degrees 2,3: build extension, same accumulation order)
    const float d0 = tps[0] - (cam.lookAt[0] - cam.eye[0]);
    const float d1 = tps[1] - (cam.lookAt[1] - cam.eye[1]);
    const float d2 = tps[2] - (cam.lookAt[2] - cam.eye[2]);
    const float nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    const float ninv = 1.0f / nrm;
    const float x = ninv * d0, y = ninv * d1, z = ninv * d2;
    float bs[K];
    bs[0] = SH_C0;
    if constexpr (DEG >= 1) { bs[1] = -y * SH_C1; bs[2] = z * SH_C1; bs[3] = -x * SH_C1; }
    if constexpr (DEG >= 2) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        bs[4] = kC2[0] * xy;
        bs[5] = kC2[1] * yz;
        bs[6] = kC2[2] * ((2.0f * zz - xx) - yy);
        bs[7] = kC2[3] * xz;
        bs[8] = kC2[4] * (xx - yy);
        if constexpr (DEG >= 3) {
            bs[9]  = (kC3[0] * y) * (3.0f * xx - yy);
            bs[10] = (kC3[1] * xy) * z;
            bs[11] = (kC3[2] * y) * ((4.0f * zz - xx) - yy);
            bs[12] = (kC3[3] * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
            bs[13] = (kC3[4] * x) * ((4.0f * zz - xx) - yy);
            bs[14] = (kC3[5] * z) * (xx - yy);
            bs[15] = (kC3[6] * x) * (xx - 3.0f * yy);
        }
    }
    const float *sh = a.shs + (int64_t)3 * K * g;
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = sh[c] * bs[0];
#pragma unroll
        for (int k = 1; k < K; ++k) s = s + sh[c + 3 * k] * bs[k];
        rgb[c] = (float)((double)s + 0.5);                            // :192
    }
    // ---- cusigmoid, splat.jl:175-178
    const float ez = gs_expf(a.opac[g]);
    float sg = ez / (1.0f + ez);

    // ---- tile rectangle, binning.jl:6-27 (non-finite boxes: dropped)
    const bool finite_bb = isfinite(bxmin) && isfinite(bxmax) && isfinite(bymin) && isfinite(bymax);
    uint16_t rc[4] = {0, 0, 0, 0};
    if (finite_bb) {
        int bminx = gs_tile_div(floorf(bxmin)) + 1, bmaxx = gs_tile_div(ceilf(bxmax)) + 1;
        int bminy = gs_tile_div(floorf(bymin)) + 1, bmaxy = gs_tile_div(ceilf(bymax)) + 1;
        if (bminx <= bmaxx && bminy <= bmaxy) {
            bminx = max(bminx, 1); bminy = max(bminy, 1);
            bmaxx = min(bmaxx, a.gx); bmaxy = min(bmaxy, a.gy);
            if (bminx <= bmaxx && bminy <= bmaxy) {
                rc[0] = (uint16_t)bminx; rc[1] = (uint16_t)bmaxx; rc[2] = (uint16_t)bminy; rc[3] = (uint16_t)bmaxy;
            }
        }
    }

    // ---- depth key for sortperm(-tps[3,:], lt=isless), forward.jl:103
    uint32_t key = 0u;
    if (a.order != 0) {
        const float v = (a.order == 1) ? -tps[2] : tps[2];
        const uint32_t u = gs_f2u(v);
        key = (v != v) ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
    }

    // ---- payload.  The near/far skip of splat.jl:227 becomes an empty pixel box.  A splat whose
    // per-view payload is not finite (degenerate covariance, NaN/Inf SH or opacity) is skipped the
    // same way: the reference would write NaN into its pixel box, and its example scrubs those NaNs
    // afterwards (examples/main.jl:35); the composite kernels mask arithmetically and could not keep
    // a NaN inside the box.
    GsPayload p;
    const bool depth_ok = !(tps[2] < cam.near_ || tps[2] > cam.far_);
    const bool pay_ok = isfinite(rgb[0]) && isfinite(rgb[1]) && isfinite(rgb[2]) && isfinite(sg) && isfinite(mux) && isfinite(muy) &&
                        isfinite(inv0) && isfinite(inv1) && isfinite(inv2) && isfinite(inv3);
    p.mx = mux; p.my = muy; p.sig = sg;
    // what the composite kernels' inner loops consume, formed once per (gaussian, view) instead of once per (tile, splat):
    // alpha = exp2(ka dX^2 + kb dX dY + kc dY^2 + l2s)   (same single fp32 operations the staging lanes performed in rounds 1-2)
    p.l2s = fminf(__builtin_amdgcn_logf(sg), GS_L2S_CAP);
    p.ka = GS_NEG_HALF_LOG2E * inv0; p.kb = GS_NEG_HALF_LOG2E * (inv1 + inv2); p.kc = GS_NEG_HALF_LOG2E * inv3;
    p.r = rgb[0]; p.g = rgb[1]; p.b = rgb[2];
    if (finite_bb && depth_ok && pay_ok) { p.bbx = gs_pack_i16(bxmin, bxmax); p.bby = gs_pack_i16(bymin, bymax); }
    else { p.bbx = 1u; p.bby = 1u; }                                   // min 1, max 0: empty
    gs_payload_box_edges(p);
    a.payload[g] = p;
    reinterpret_cast<float4 *>(a.invcov)[g] = make_float4(inv0, inv1, inv2, inv3);
    a.depth_key[g] = key;
    if (a.key_range) {                                                  // the frame's key range, for the two-step depth sort (gs_sort.hip)
        uint32_t *lo = a.key_range + (size_t)(blockIdx.x % GS_KEY_RANGE_SLOTS) * GS_KEY_RANGE_STRIDE;
        uint32_t *hi = lo + (size_t)GS_KEY_RANGE_SLOTS * GS_KEY_RANGE_STRIDE;
        // keys of non-finite depths (NaN, +-Inf) stay out of the range: ds_bucket clamps them into the end buckets, and a single
        // one would otherwise stretch the range until every finite key shares a bucket
        const bool finite = key > 0x007FFFFFu && key < 0xFF800000u;
        if (__ballot(1) == ~0ull) {                                     // a full wave (every wave but the grid's last) folds first
            uint32_t mn = finite ? key : 0xFFFFFFFFu, mx = finite ? key : 0u;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, d)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, d)); }
            if ((threadIdx.x & 63) == 0) { atomicMin(lo, mn); atomicMax(hi, mx); }
        } else if (finite) { atomicMin(lo, key); atomicMax(hi, key); }
    }
    reinterpret_cast<uint2 *>(a.rect)[g] = make_uint2((uint32_t)rc[0] | ((uint32_t)rc[1] << 16),
                                                       (uint32_t)rc[2] | ((uint32_t)rc[3] << 16));
    if (a.dbg.ts) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a.dbg.ts[4 * g + i] = ts[i]; a.dbg.tps[4 * g + i] = tps[i]; }
        a.dbg.mu[2 * g] = mux; a.dbg.mu[2 * g + 1] = muy;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) a.dbg.cov3d[9 * g + i + 3 * j] = C3[i][j];
        a.dbg.cov2d[4 * g] = a0; a.dbg.cov2d[4 * g + 1] = a1; a.dbg.cov2d[4 * g + 2] = a2; a.dbg.cov2d[4 * g + 3] = a3;
        a.dbg.invcov[4 * g] = inv0; a.dbg.invcov[4 * g + 1] = inv1; a.dbg.invcov[4 * g + 2] = inv2; a.dbg.invcov[4 * g + 3] = inv3;
        a.dbg.bbs[4 * g] = bxmin; a.dbg.bbs[4 * g + 1] = bymin; a.dbg.bbs[4 * g + 2] = bxmax; a.dbg.bbs[4 * g + 3] = bymax;
    }
}

hipError_t gs_launch_preprocess(const GsPreprocessArgs &a, const GsCamera &cam, hipStream_t stream) {
    if (a.n <= 0) return hipSuccess;
    const dim3 block(256), grid((unsigned)((a.n + 255) / 256));
    switch (a.sh_degree) {
        case 0: hipLaunchKernelGGL(gs_preprocess_kernel<0>, grid, block, 0, stream, a, cam); break;
        case 1: hipLaunchKernelGGL(gs_preprocess_kernel<1>, grid, block, 0, stream, a, cam); break;
        case 2: hipLaunchKernelGGL(gs_preprocess_kernel<2>, grid, block, 0, stream, a, cam); break;
        case 3: hipLaunchKernelGGL(gs_preprocess_kernel<3>, grid, block, 0, stream, a, cam); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
