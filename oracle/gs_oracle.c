/*
 * gs_oracle.c -- CPU restatement of the arhik/GaussianSplat hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see gs_oracle.h).  PARITY UNPINNED by the reference: it has
 * no tests/golden vectors and cannot be run here; this restatement is pinned by the
 * independent NumPy restatement, closed-form KATs and fp64 autograd instead.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math [-fopenmp] -shared -fPIC (Makefile).
 * Citations: /root/reference/src/<file>:<line>.
 */
#include "gs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ helpers */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Julia max/min propagate NaN (Base.max for floats). */
static inline double jl_max(double a, double b) { if (a != a || b != b) return NAN; return a > b ? a : b; }
static inline double jl_min(double a, double b) { if (a != a || b != b) return NAN; return a < b ? a : b; }

int gso_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* The spec's exp: Cody-Waite reduction + Cephes expf polynomial; every operation an
 * individually rounded fp32 op, so the HIP path and NumPy reproduce it bit-for-bit.
 * Stands in for Julia/libdevice exp (splat.jl:176, projection.jl:133-135). */
float gso_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -87.33654f) return 0.0f;          /* results below FLT_MIN flush to zero */
    float n = rintf(x * 1.44269504f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float z = r * r;
    float y = 1.9875691500e-4f;
    y = y * r + 1.3981999507e-3f;
    y = y * r + 8.3334519073e-3f;
    y = y * r + 4.1665795894e-2f;
    y = y * r + 1.6666665459e-1f;
    y = y * r + 5.0000001201e-1f;
    y = y * z;
    y = y + r;
    y = y + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2;
    int n2 = ni - n1;
    float s1 = u2f((uint32_t)(n1 + 127) << 23);
    float s2 = u2f((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

/* The spec's sin/cos (2-D renderer, cov2d.jl:6-8 CUDA.cos/CUDA.sin): k = rint(x*2/pi), three-term Cody-Waite
 * reduction by pi/2, Cephes sinf/cosf polynomials on [-pi/4, pi/4], quadrant fix-up; fp32 mul/add only. */
void gso_sincosf(float x, float *sn, float *cs) {
    if (!isfinite(x)) { *sn = NAN; *cs = NAN; return; }
    float kf = rintf(x * 0.636619772f);
    float r = x - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z + -1.6666654611e-1f;
    ps = ps * z;
    ps = ps * r;
    ps = ps + r;
    float pc = 2.443315711809948e-5f;
    pc = pc * z + -1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    long long k = (long long)kf;
    switch ((int)(k & 3)) {
        case 0: *sn = ps;  *cs = pc;  break;
        case 1: *sn = pc;  *cs = -ps; break;
        case 2: *sn = -ps; *cs = -pc; break;
        default: *sn = -pc; *cs = ps; break;
    }
}

/* ------------------------------------------------------------------ camera */

/* camera.jl:88-100 computeTransform, camera.jl:102-111 computeProjection.
 * Vectors are plain Julia Vector{Float32}: LinearAlgebra.norm accumulates the squares in
 * Float64 (generic_norm2) and normalize multiplies by inv(norm). */
static void normalize3(const float a[3], float out[3]) {
    double s = (double)(a[0] * a[0]);
    s += (double)(a[1] * a[1]);
    s += (double)(a[2] * a[2]);
    float nrm = (float)sqrt(s);
    float inv = 1.0f / nrm;
    for (int i = 0; i < 3; ++i) out[i] = a[i] * inv;
}
static void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

void gso_camera_matrices(const float eye[3], const float lookAt[3], const float up[3],
                         float fx, float fy, float near_, float far_, int W, int H,
                         gso_camera *out) {
    float d[3] = { lookAt[0] - eye[0], lookAt[1] - eye[1], lookAt[2] - eye[2] };
    float w[3], u[3], v[3], c[3];
    normalize3(d, w);                       /* camera.jl:91 */
    cross3(up, w, c);
    normalize3(c, u);                       /* camera.jl:92 */
    cross3(w, u, v);                        /* camera.jl:93 */
    const float *rows[3] = { u, v, w };
    float m[16];                            /* m[i + 4*j], rows u,v,w; row 4 all zero (:96) */
    memset(m, 0, sizeof m);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i + 4 * j] = rows[i][j];
    /* translateCamera (camera.jl:65-77) = inverse of translate(eye) = [I | -eye] */
    float ti[16];
    memset(ti, 0, sizeof ti);
    ti[0] = ti[5] = ti[10] = ti[15] = 1.0f;
    ti[12] = -eye[0]; ti[13] = -eye[1]; ti[14] = -eye[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = m[i] * ti[4 * j];
            s = s + m[i + 4] * ti[1 + 4 * j];
            s = s + m[i + 8] * ti[2 + 4 * j];
            s = s + m[i + 12] * ti[3 + 4 * j];
            out->T[i + 4 * j] = s;
        }
    memset(out->P, 0, sizeof out->P);
    out->P[0]  = 2.0f * fx / (float)W;                         /* camera.jl:105 */
    out->P[5]  = 2.0f * fy / (float)H;                         /* :106 */
    out->P[10] = (far_ + near_) / (far_ - near_);              /* :107 */
    out->P[14] = -2.0f * (far_ * near_) / (far_ - near_);      /* :108  p[3,4] */
    out->P[11] = 1.0f;                                         /* :109  p[4,3] */
    out->W = W; out->H = H; out->fx = fx; out->fy = fy;
    out->near_ = near_; out->far_ = far_;
    for (int i = 0; i < 3; ++i) { out->eye[i] = eye[i]; out->lookAt[i] = lookAt[i]; }
}

/* ------------------------------------------------------------------ SH basis */

#define SH_C0 0.28209479177387814f
#define SH_C1 0.48860251190291990f
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

/* splat.jl:190 for degree<=1; degrees 2,3 are the build's extension (standard 3DGS real
 * SH polynomials) with the operation order fixed here. */
static int sh_basis_f32(int deg, float x, float y, float z, float b[16]) {
    b[0] = SH_C0;
    if (deg < 1) return 1;
    b[1] = -y * SH_C1;
    b[2] = z * SH_C1;
    b[3] = -x * SH_C1;
    if (deg < 2) return 4;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2[0] * xy;
    b[5] = SH_C2[1] * yz;
    b[6] = SH_C2[2] * ((2.0f * zz - xx) - yy);
    b[7] = SH_C2[3] * xz;
    b[8] = SH_C2[4] * (xx - yy);
    if (deg < 3) return 9;
    b[9]  = (SH_C3[0] * y) * (3.0f * xx - yy);
    b[10] = (SH_C3[1] * xy) * z;
    b[11] = (SH_C3[2] * y) * ((4.0f * zz - xx) - yy);
    b[12] = (SH_C3[3] * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
    b[13] = (SH_C3[4] * x) * ((4.0f * zz - xx) - yy);
    b[14] = (SH_C3[5] * z) * (xx - yy);
    b[15] = (SH_C3[6] * x) * (xx - 3.0f * yy);
    return 16;
}

/* ------------------------------------------------------------------ preprocess */

void gso_preprocess(int64_t n, int sh_degree,
                    const float *means, const float *scales, const float *quats,
                    const float *opacities, const float *shs, const gso_camera *cam,
                    float *ts_o, float *tps_o, float *mu_o, float *cov3d_o, float *cov2d_o,
                    float *invcov_o, float *bbs_o, float *rgb_o, float *sig_o) {
    const float *T = cam->T, *P = cam->P;
    const int K = (sh_degree + 1) * (sh_degree + 1);
    const double cx = cam->W / 2.0, cy = cam->H / 2.0;   /* forward.jl:58-59 (Float64) */
    const float fx = cam->fx, fy = cam->fy;
    const float wf = (float)cam->W, hf = (float)cam->H;
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n; ++g) {
        /* ---- frustumCulling, projection.jl:39-100 */
        float m1 = means[3 * g], m2 = means[3 * g + 1], m3 = means[3 * g + 2];
        float ts[4], tps[4];
        for (int i = 0; i < 4; ++i) {
            float s = T[i] * m1;
            s = s + T[i + 4] * m2;
            s = s + T[i + 8] * m3;
            s = s + T[i + 12] * 1.0f;
            ts[i] = s;                                       /* :59-63 */
        }
        for (int i = 0; i < 4; ++i) {
            float s = P[i] * ts[0];
            s = s + P[i + 4] * ts[1];
            s = s + P[i + 8] * ts[2];
            s = s + P[i + 12] * ts[3];
            tps[i] = s;                                      /* :77-81 */
        }
        float mux = (float)((double)((wf * tps[0] / tps[3] + 1.0f) / 2.0f) + cx);   /* :88 */
        float muy = (float)((double)((hf * tps[1] / tps[3] + 1.0f) / 2.0f) + cy);   /* :89 */

        /* ---- tValues, projection.jl:103-155 */
        float tx = ts[0], ty = ts[1], tz = ts[2];
        float J[2][3];
        J[0][0] = fx / tz;  J[1][0] = 0.0f;                  /* :113-114 (column-major fill) */
        J[0][1] = 0.0f;     J[1][1] = fy / tz;               /* :115-116 */
        J[0][2] = -fx * tx / (tz * tz);                      /* :117 */
        J[1][2] = -fy * ty / (tz * tz);                      /* :118 */
        float qw = quats[4 * g], qx = quats[4 * g + 1], qy = quats[4 * g + 2], qz = quats[4 * g + 3];
        float R[3][3];                                       /* projection.jl:1-14, R[row][col] */
        R[0][0] = 1.0f - 2.0f * (qy * qy + qz * qz);
        R[1][0] = 2.0f * (qx * qy + qw * qz);
        R[2][0] = 2.0f * (qx * qz - qw * qy);
        R[0][1] = 2.0f * (qx * qy - qw * qz);
        R[1][1] = 1.0f - 2.0f * (qx * qx - qz * qz);         /* :8, minus as written */
        R[2][1] = 2.0f * (qy * qz + qw * qx);
        R[0][2] = 2.0f * (qx * qz + qw * qy);
        R[1][2] = 2.0f * (qy * qz - qw * qx);
        R[2][2] = 1.0f - 2.0f * (qx * qx + qy * qy);
        float S[3][3] = { { 0 } };
        S[0][0] = gso_expf(scales[3 * g]);                   /* :133-135 */
        S[1][1] = gso_expf(scales[3 * g + 1]);
        S[2][2] = gso_expf(scales[3 * g + 2]);
        float Wm[3][3], C3[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float s = R[i][0] * S[0][j];
                s = s + R[i][1] * S[1][j];
                s = s + R[i][2] * S[2][j];
                Wm[i][j] = s;                                /* :136 W = R*S */
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float s = Wm[i][0] * Wm[j][0];
                s = s + Wm[i][1] * Wm[j][1];
                s = s + Wm[i][2] * Wm[j][2];
                C3[i][j] = s;                                /* :137 cov3d = W*W' */
            }
        float JR[2][3], JCR[2][3], c2[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) {
                float s = J[i][0] * R[0][j];
                s = s + J[i][1] * R[1][j];
                s = s + J[i][2] * R[2][j];
                JR[i][j] = s;                                /* :144 */
            }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) {
                float s = JR[i][0] * C3[0][j];
                s = s + JR[i][1] * C3[1][j];
                s = s + JR[i][2] * C3[2][j];
                JCR[i][j] = s;                               /* :145 */
            }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                float s = JCR[i][0] * JR[j][0];
                s = s + JCR[i][1] * JR[j][1];
                s = s + JCR[i][2] * JR[j][2];
                c2[i][j] = (float)((double)s + 0.3);         /* :146-152, +0.3 is Float64 */
            }
        float a0 = c2[0][0], a1 = c2[1][0], a2 = c2[0][1], a3 = c2[1][1];   /* column-major */

        /* ---- computeInvCov2d, cov2d.jl:30-45 (StaticArrays 2x2 inv: adjugate * 1/det) */
        float det = a0 * a3 - a2 * a1;
        float idet = 1.0f / det;
        float inv0 = a3 * idet, inv1 = -(a1 * idet), inv2 = -(a2 * idet), inv3 = a0 * idet;

        /* ---- computeBB, boundingbox.jl:4-36 (0.1, 3.0 are Float64 literals) */
        float halfad = (a0 + a3) / 2.0f;                     /* :20 */
        double disc = (double)(halfad * halfad - det);
        double sq = sqrt(jl_max(0.1, disc));
        double e1 = (double)halfad - sq;                     /* :21 */
        double e2 = (double)halfad + sq;                     /* :22 */
        double r = ceil(3.0 * sqrt(jl_max(e1, e2)));         /* :23 */
        float bxmin = (float)jl_max(1.0, floor(-r + (double)mux));            /* :24 */
        float bxmax = (float)jl_min((double)cam->W, ceil(r + (double)mux));   /* :25 */
        float bymin = (float)jl_max(1.0, floor(-r + (double)muy));            /* :26 */
        float bymax = (float)jl_min((double)cam->H, ceil(r + (double)muy));   /* :27 */

        /* ---- sh2color, splat.jl:180-193 (hoisted: depends on gaussian + view only) */
        float d0 = tps[0] - (cam->lookAt[0] - cam->eye[0]);
        float d1 = tps[1] - (cam->lookAt[1] - cam->eye[1]);
        float d2 = tps[2] - (cam->lookAt[2] - cam->eye[2]);
        float nrm = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
        float ninv = 1.0f / nrm;
        float x = ninv * d0, y = ninv * d1, z = ninv * d2;   /* :188-189 */
        float b[16];
        sh_basis_f32(sh_degree, x, y, z, b);
        float rgb[3];
        for (int c = 0; c < 3; ++c) {
            const float *sh = shs + (int64_t)3 * K * g;       /* shs[c + 3k], splat.jl:117 */
            float s = sh[c] * b[0];
            for (int k = 1; k < K; ++k) s = s + sh[c + 3 * k] * b[k];
            rgb[c] = (float)((double)s + 0.5);               /* :192, 0.5 is Float64 */
        }
        /* ---- cusigmoid, splat.jl:175-178 */
        float ez = gso_expf(opacities[g]);
        float sg = ez / (1.0f + ez);

        if (ts_o)  for (int i = 0; i < 4; ++i) ts_o[4 * g + i] = ts[i];
        if (tps_o) for (int i = 0; i < 4; ++i) tps_o[4 * g + i] = tps[i];
        if (mu_o)  { mu_o[2 * g] = mux; mu_o[2 * g + 1] = muy; }
        if (cov3d_o) for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) cov3d_o[9 * g + i + 3 * j] = C3[i][j];
        if (cov2d_o) { cov2d_o[4 * g] = a0; cov2d_o[4 * g + 1] = a1; cov2d_o[4 * g + 2] = a2; cov2d_o[4 * g + 3] = a3; }
        if (invcov_o) { invcov_o[4 * g] = inv0; invcov_o[4 * g + 1] = inv1; invcov_o[4 * g + 2] = inv2; invcov_o[4 * g + 3] = inv3; }
        if (bbs_o) { bbs_o[4 * g] = bxmin; bbs_o[4 * g + 1] = bymin; bbs_o[4 * g + 2] = bxmax; bbs_o[4 * g + 3] = bymax; }
        if (rgb_o) { rgb_o[3 * g] = rgb[0]; rgb_o[3 * g + 1] = rgb[1]; rgb_o[3 * g + 2] = rgb[2]; }
        if (sig_o) sig_o[g] = sg;
    }
}

/* ------------------------------------------------------------------ 2-D renderer preprocess */

/* preprocess(::GaussianRenderer2D), forward.jl:9-33: computeCov2d_kernel (cov2d.jl:3-28), computeInvCov2d
 * (cov2d.jl:30-45), computeBB (boundingbox.jl:4-36).  SplatData2D (splat.jl:20-26): means 2xN in [0,1]^2,
 * scales 2xN (log), rotations 1xN (theta), opacities 1xN, colors 3xN.  The pixel position of a gaussian is
 * (w*mx, h*my) (splat.jl:337-339); the reference passes the [0,1] means to computeBB (forward.jl:25-31), which
 * would put every box at the image origin -- the spec uses the pixel position (documented deviation).
 * alpha = opacity*exp(-dist/2) with the raw opacity (splat.jl:341,345), colour = colors (no SH). */
void gso_preprocess2d(int64_t n, const float *means, const float *scales, const float *rots,
                      const float *opacities, const float *colors, int W, int H,
                      float *mu_o, float *cov2d_o, float *invcov_o, float *bbs_o, float *rgb_o, float *sig_o) {
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n; ++g) {
        float sn, cs;
        gso_sincosf(rots[g], &sn, &cs);                               /* cov2d.jl:6-8 */
        float R[2][2] = { { cs, -sn }, { sn, cs } };                    /* :9-12 */
        float S[2][2] = { { gso_expf(scales[2 * g]), 0.0f }, { 0.0f, gso_expf(scales[2 * g + 1]) } };   /* :14-17 */
        float Wm[2][2], Jm[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                float s = R[i][0] * S[0][j];
                s = s + R[i][1] * S[1][j];
                Wm[i][j] = s;                                          /* :18 W = R*S */
            }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                float s = Wm[i][0] * Wm[j][0];
                s = s + Wm[i][1] * Wm[j][1];
                Jm[i][j] = s;                                          /* :19 J = W*W' */
            }
        const float a0 = (float)((double)Jm[0][0] + 0.3);              /* :25 (+0.3 is Float64, diagonal only) */
        const float a1 = Jm[1][0], a2 = Jm[0][1];
        const float a3 = (float)((double)Jm[1][1] + 0.3);              /* :26 */
        /* computeInvCov2d, cov2d.jl:30-45 */
        const float det = a0 * a3 - a2 * a1;
        const float idet = 1.0f / det;
        const float inv0 = a3 * idet, inv1 = -(a1 * idet), inv2 = -(a2 * idet), inv3 = a0 * idet;
        /* pixel position, splat.jl:337-339 */
        const float mux = (float)W * means[2 * g], muy = (float)H * means[2 * g + 1];
        /* computeBB, boundingbox.jl:19-27 */
        const float halfad = (a0 + a3) / 2.0f;
        const double disc = (double)(halfad * halfad - det);
        const double sq = sqrt(jl_max(0.1, disc));
        const double e1 = (double)halfad - sq, e2 = (double)halfad + sq;
        const double r = ceil(3.0 * sqrt(jl_max(e1, e2)));
        const float bxmin = (float)jl_max(1.0, floor(-r + (double)mux));
        const float bxmax = (float)jl_min((double)W, ceil(r + (double)mux));
        const float bymin = (float)jl_max(1.0, floor(-r + (double)muy));
        const float bymax = (float)jl_min((double)H, ceil(r + (double)muy));
        if (mu_o) { mu_o[2 * g] = mux; mu_o[2 * g + 1] = muy; }
        if (cov2d_o) { cov2d_o[4 * g] = a0; cov2d_o[4 * g + 1] = a1; cov2d_o[4 * g + 2] = a2; cov2d_o[4 * g + 3] = a3; }
        if (invcov_o) { invcov_o[4 * g] = inv0; invcov_o[4 * g + 1] = inv1; invcov_o[4 * g + 2] = inv2; invcov_o[4 * g + 3] = inv3; }
        if (bbs_o) { bbs_o[4 * g] = bxmin; bbs_o[4 * g + 1] = bymin; bbs_o[4 * g + 2] = bxmax; bbs_o[4 * g + 3] = bymax; }
        if (rgb_o) { rgb_o[3 * g] = colors[3 * g]; rgb_o[3 * g + 1] = colors[3 * g + 1]; rgb_o[3 * g + 2] = colors[3 * g + 2]; }
        if (sig_o) sig_o[g] = fminf(fmaxf(opacities[g], 0.0f), 0.99999994f);   /* raw opacity clamped to [0, 1): the spec */
    }
}

/* ------------------------------------------------------------------ depth order */

uint32_t gso_depth_key(float clipz, int order) {
    if (order == GSO_ORDER_INDEX) return 0u;
    float v = (order == GSO_ORDER_DEPTH_DESC) ? -clipz : clipz;   /* forward.jl:103 sorts -tps[3,:] */
    if (v != v) return 0xFFFFFFFFu;                               /* isless: NaN is largest */
    uint32_t u = f2u(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);            /* -0.0 < +0.0, like isless */
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

void gso_depth_order(int64_t n, const float *tps, int order, uint32_t *perm) {
    if (order == GSO_ORDER_INDEX) { for (int64_t g = 0; g < n; ++g) perm[g] = (uint32_t)g; return; }
    uint64_t *kv = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t g = 0; g < n; ++g)
        kv[g] = ((uint64_t)gso_depth_key(tps[4 * g + 2], order) << 32) | (uint32_t)g;  /* ties: index (stable) */
    qsort(kv, (size_t)n, sizeof(uint64_t), cmp_u64);
    for (int64_t g = 0; g < n; ++g) perm[g] = (uint32_t)(kv[g] & 0xFFFFFFFFu);
    free(kv);
}

/* ------------------------------------------------------------------ binning */

/* Julia div(x::Float32, y::Float32) = round((x - rem(x, y)) / y) (binning.jl:14-17). */
static inline int32_t jl_div_tile(float v, float bs) {
    float q = rintf((v - fmodf(v, bs)) / bs);
    /* Int32(q) would throw InexactError out of range; the spec saturates (clipped to the
     * grid right after, so the result is identical whenever the reference succeeds). */
    if (q > 1.0e9f) q = 1.0e9f;
    if (q < -1.0e9f) q = -1.0e9f;
    return (int32_t)q;
}

int gso_tile_rect(const float bb[4], int tile, int gx, int gy, int32_t rect[4]) {
    /* NaN/Inf boxes make Int32() throw in the reference; the spec drops such gaussians. */
    for (int i = 0; i < 4; ++i) if (!isfinite(bb[i])) return 0;
    float bs = (float)tile;
    int32_t bminx = jl_div_tile(floorf(bb[0]), bs) + 1;      /* binning.jl:6,14 */
    int32_t bminy = jl_div_tile(floorf(bb[1]), bs) + 1;      /* :10,15 */
    int32_t bmaxx = jl_div_tile(ceilf(bb[2]), bs) + 1;       /* :8,16 */
    int32_t bmaxy = jl_div_tile(ceilf(bb[3]), bs) + 1;       /* :12,17 */
    if (bminx > bmaxx) return 0;                             /* :20-23 */
    if (bminy > bmaxy) return 0;
    if (bminx < 1) bminx = 1;                                /* :27 clip to the grid */
    if (bminy < 1) bminy = 1;
    if (bmaxx > gx) bmaxx = gx;
    if (bmaxy > gy) bmaxy = gy;
    if (bminx > bmaxx || bminy > bmaxy) return 0;
    rect[0] = bminx; rect[1] = bmaxx; rect[2] = bminy; rect[3] = bmaxy;
    return 1;
}

/* hitBinning + scan! + compactHits as sparse lists (binning.jl:3-35, forward.jl:137-141, compact.jl:3-21).
 * Threads own bands of tile ROWS: every thread walks the gaussians in list order and only touches the tiles
 * of its rows, so every tile's list is written by one thread in list order -- the result does not depend on the
 * thread count (the single-threaded build runs the same code with one band). */
int64_t gso_bin(int64_t n, const float *bbs, const float *tps, const uint32_t *perm,
                int order, int tile, int gx, int gy,
                uint32_t *ranges, uint32_t *ids, uint64_t *keys, int64_t cap) {
    const int64_t nt = (int64_t)gx * gy;
    int64_t *cnt = (int64_t *)calloc((size_t)nt + 1, sizeof(int64_t));
    int32_t *rcs = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(n > 0 ? n : 1));   /* tile rect per gaussian, rcs[4g] = 0: none */
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n; ++g) {
        int32_t rc[4];
        if (!gso_tile_rect(bbs + 4 * g, tile, gx, gy, rc)) rc[0] = rc[1] = rc[2] = rc[3] = 0;
        memcpy(rcs + 4 * g, rc, sizeof(rc));
    }
#pragma omp parallel
    {
#ifdef _OPENMP
        const int nth = omp_get_num_threads(), tid = omp_get_thread_num();
#else
        const int nth = 1, tid = 0;
#endif
        const int y0 = (int)((int64_t)gy * tid / nth) + 1, y1 = (int)((int64_t)gy * (tid + 1) / nth);   /* rows y0..y1, 1-based */
        for (int64_t g = 0; g < n && y0 <= y1; ++g) {
            const int32_t *rc = rcs + 4 * g;
            if (rc[0] == 0) continue;
            const int a = rc[2] > y0 ? rc[2] : y0, b = rc[3] < y1 ? rc[3] : y1;
            for (int ty = a; ty <= b; ++ty)
                for (int tx = rc[0]; tx <= rc[1]; ++tx) cnt[(int64_t)(ty - 1) * gx + (tx - 1)]++;
        }
    }
    int64_t total = 0;
    for (int64_t t = 0; t < nt; ++t) { int64_t c = cnt[t]; cnt[t] = total; total += c; }
    cnt[nt] = total;
    if (ranges) for (int64_t t = 0; t < nt; ++t) { ranges[2 * t] = (uint32_t)cnt[t]; ranges[2 * t + 1] = (uint32_t)cnt[t + 1]; }
    if (ids || keys) {
        int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nt > 0 ? nt : 1));
        memcpy(cur, cnt, sizeof(int64_t) * (size_t)nt);
#pragma omp parallel
        {
#ifdef _OPENMP
            const int nth = omp_get_num_threads(), tid = omp_get_thread_num();
#else
            const int nth = 1, tid = 0;
#endif
            const int y0 = (int)((int64_t)gy * tid / nth) + 1, y1 = (int)((int64_t)gy * (tid + 1) / nth);
            for (int64_t s = 0; s < n && y0 <= y1; ++s) {
                const int64_t g = perm ? (int64_t)perm[s] : s;      /* list order: index or depth rank */
                const int32_t *rc = rcs + 4 * g;
                if (rc[0] == 0) continue;
                const int a = rc[2] > y0 ? rc[2] : y0, b = rc[3] < y1 ? rc[3] : y1;
                if (a > b) continue;
                const uint32_t lo = (order == GSO_ORDER_INDEX) ? (uint32_t)g : gso_depth_key(tps[4 * g + 2], order);
                for (int ty = a; ty <= b; ++ty)
                    for (int tx = rc[0]; tx <= rc[1]; ++tx) {
                        const int64_t t = (int64_t)(ty - 1) * gx + (tx - 1);
                        const int64_t p = cur[t]++;
                        if (p < cap) {
                            if (ids) ids[p] = (uint32_t)g;
                            if (keys) keys[p] = ((uint64_t)t << 32) | lo;
                        }
                    }
            }
        }
        free(cur);
    }
    free(rcs);
    free(cnt);
    return total;
}

int64_t gso_bin_dense_literal(int64_t n, const float *bbs, int tile, int gx, int gy,
                              uint32_t *hitIdxs, int64_t maxBin) {
    const int64_t nt = (int64_t)gx * gy;
    uint8_t *hits = (uint8_t *)calloc((size_t)(nt * n + 1), 1);          /* forward.jl:120 */
    uint16_t *scan = (uint16_t *)calloc((size_t)(nt * n + 1), 2);        /* forward.jl:137 */
    for (int64_t g = 0; g < n; ++g) {                                    /* binning.jl:3-35 */
        int32_t rc[4];
        if (!gso_tile_rect(bbs + 4 * g, tile, gx, gy, rc)) continue;
        for (int i = rc[0]; i <= rc[1]; ++i)
            for (int j = rc[2]; j <= rc[3]; ++j) hits[(i - 1) + (int64_t)gx * (j - 1) + nt * g] = 1;
    }
    int64_t maxHits = 0;
    for (int64_t t = 0; t < nt; ++t) {                                   /* scan!(+, dims=3), UInt16 wraps */
        uint16_t acc = 0;
        for (int64_t g = 0; g < n; ++g) {
            acc = (uint16_t)(acc + hits[t + nt * g]);
            scan[t + nt * g] = acc;
            if (acc > maxHits) maxHits = acc;                            /* forward.jl:139 */
        }
    }
    if (hitIdxs) {
        memset(hitIdxs, 0, sizeof(uint32_t) * (size_t)(nt * maxBin));
        for (int64_t g = 0; g < n; ++g)                                  /* compact.jl:3-21 */
            for (int64_t t = 0; t < nt; ++t)
                if (hits[t + nt * g] == 1) {
                    int64_t slot = scan[t + nt * g];
                    if (slot != 0 && slot <= maxBin) hitIdxs[t + nt * (slot - 1)] = (uint32_t)(g + 1);
                }
    }
    free(hits); free(scan);
    return maxHits;
}

/* ------------------------------------------------------------------ composite forward */

void gso_composite_forward(const gso_camera *cam, int tile, int gx, int gy,
                           const uint32_t *ranges, const uint32_t *ids,
                           const float *mu, const float *invcov, const float *bbs,
                           const float *sig, const float *rgb, const float *tps,
                           float t_min, float *image, float *trans) {
    const int W = cam->W, H = cam->H;
    const int64_t plane = (int64_t)W * H;
    const int64_t nt = (int64_t)gx * gy;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t t = 0; t < nt; ++t) {
        int bx = (int)(t % gx) + 1, by = (int)(t / gx) + 1;              /* blockIdx, 1-based */
        uint32_t s0 = ranges[2 * t], s1 = ranges[2 * t + 1];
        for (int tyi = 1; tyi <= tile; ++tyi)
            for (int txi = 1; txi <= tile; ++txi) {
                int i = (bx - 1) * tile + txi;                           /* splat.jl:204-205 */
                int j = (by - 1) * tile + tyi;
                if (i > W || j > H) continue;                            /* build's guard (ragged edge) */
                float C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, Tr = 1.0f;        /* :210-213 */
                float fi = (float)i, fj = (float)j;
                for (uint32_t k = s0; k < s1; ++k) {                     /* :224 */
                    if (t_min > 0.0f && ((k - s0) % GSO_EARLY_BATCH) == 0 && Tr < t_min) break;   /* extension: early-out */
                    uint32_t b = ids[k];
                    if (tps) {                                            /* 2-D renderer: no clip z, no skip */
                        float cz = tps[4 * (int64_t)b + 2];
                        if (cz < cam->near_ || cz > cam->far_) continue; /* :227 */
                    }
                    const float *bb = bbs + 4 * (int64_t)b;
                    int hit = (bb[0] <= fi) && (fi <= bb[2]) && (bb[1] <= fj) && (fj <= bb[3]);  /* :240 */
                    if (!hit) continue;
                    const float *iv = invcov + 4 * (int64_t)b;
                    float dX = fi - mu[2 * (int64_t)b];                  /* :243 */
                    float dY = fj - mu[2 * (int64_t)b + 1];              /* :244 */
                    float v1 = iv[0] * dX + iv[2] * dY;                  /* invCov2d*delta */
                    float v2 = iv[1] * dX + iv[3] * dY;
                    float dist = 0.50f * (v1 * dX + v2 * dY);            /* :246 */
                    float alpha = sig[b] * gso_expf(-dist);              /* :247 */
                    const float *c = rgb + 3 * (int64_t)b;
                    C0 = C0 + (c[0] * alpha) * Tr;                       /* :255-257 */
                    C1 = C1 + (c[1] * alpha) * Tr;
                    C2 = C2 + (c[2] * alpha) * Tr;
                    Tr = Tr * (1.0f - alpha);                            /* :259 */
                }
                int64_t px = (int64_t)(i - 1) + (int64_t)W * (j - 1);
                image[px] = C0; image[px + plane] = C1; image[px + 2 * plane] = C2;   /* :263-265 */
                trans[px] = Tr;                                                      /* :266 */
            }
    }
}

/* ------------------------------------------------------------------ backward (fp64 adjoint) */

static const double D_C0 = 0.28209479177387814, D_C1 = 0.48860251190291990;
static const double D_C2[5] = { 1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396 };
static const double D_C3[7] = { -0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435 };

/* basis b[k] and gradient db[k][3] = d b_k / d(x,y,z) */
static int sh_basis_f64(int deg, double x, double y, double z, double b[16], double db[16][3]) {
    memset(db, 0, sizeof(double) * 16 * 3);
    b[0] = D_C0;
    if (deg < 1) return 1;
    b[1] = -y * D_C1; db[1][1] = -D_C1;
    b[2] = z * D_C1;  db[2][2] = D_C1;
    b[3] = -x * D_C1; db[3][0] = -D_C1;
    if (deg < 2) return 4;
    double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = D_C2[0] * xy;                 db[4][0] = D_C2[0] * y;  db[4][1] = D_C2[0] * x;
    b[5] = D_C2[1] * yz;                 db[5][1] = D_C2[1] * z;  db[5][2] = D_C2[1] * y;
    b[6] = D_C2[2] * (2 * zz - xx - yy); db[6][0] = -2 * D_C2[2] * x; db[6][1] = -2 * D_C2[2] * y; db[6][2] = 4 * D_C2[2] * z;
    b[7] = D_C2[3] * xz;                 db[7][0] = D_C2[3] * z;  db[7][2] = D_C2[3] * x;
    b[8] = D_C2[4] * (xx - yy);          db[8][0] = 2 * D_C2[4] * x; db[8][1] = -2 * D_C2[4] * y;
    if (deg < 3) return 9;
    b[9]  = D_C3[0] * y * (3 * xx - yy);          db[9][0] = 6 * D_C3[0] * xy;  db[9][1] = D_C3[0] * (3 * xx - 3 * yy);
    b[10] = D_C3[1] * xy * z;                     db[10][0] = D_C3[1] * yz; db[10][1] = D_C3[1] * xz; db[10][2] = D_C3[1] * xy;
    b[11] = D_C3[2] * y * (4 * zz - xx - yy);     db[11][0] = -2 * D_C3[2] * xy; db[11][1] = D_C3[2] * (4 * zz - xx - 3 * yy); db[11][2] = 8 * D_C3[2] * yz;
    b[12] = D_C3[3] * z * (2 * zz - 3 * xx - 3 * yy); db[12][0] = -6 * D_C3[3] * xz; db[12][1] = -6 * D_C3[3] * yz; db[12][2] = D_C3[3] * (6 * zz - 3 * xx - 3 * yy);
    b[13] = D_C3[4] * x * (4 * zz - xx - yy);     db[13][0] = D_C3[4] * (4 * zz - 3 * xx - yy); db[13][1] = -2 * D_C3[4] * xy; db[13][2] = 8 * D_C3[4] * xz;
    b[14] = D_C3[5] * z * (xx - yy);              db[14][0] = 2 * D_C3[5] * xz; db[14][1] = -2 * D_C3[5] * yz; db[14][2] = D_C3[5] * (xx - yy);
    b[15] = D_C3[6] * x * (xx - 3 * yy);          db[15][0] = D_C3[6] * (3 * xx - 3 * yy); db[15][1] = -6 * D_C3[6] * xy;
    return 16;
}

typedef struct {            /* fp64 per-gaussian forward state */
    double t[4], p[4], mu[2], M[4] /* inv cov, col-major */, sig, rgb[3];
} g64;

static void fwd64(int64_t g, int deg, const float *means, const float *scales, const float *quats,
                  const float *opac, const float *shs, const gso_camera *cam, g64 *o) {
    const int K = (deg + 1) * (deg + 1);
    double m[3] = { means[3 * g], means[3 * g + 1], means[3 * g + 2] };
    for (int i = 0; i < 4; ++i)
        o->t[i] = cam->T[i] * m[0] + cam->T[i + 4] * m[1] + cam->T[i + 8] * m[2] + cam->T[i + 12];
    for (int i = 0; i < 4; ++i)
        o->p[i] = cam->P[i] * o->t[0] + cam->P[i + 4] * o->t[1] + cam->P[i + 8] * o->t[2] + cam->P[i + 12] * o->t[3];
    o->mu[0] = (cam->W * o->p[0] / o->p[3] + 1.0) / 2.0 + cam->W / 2.0;
    o->mu[1] = (cam->H * o->p[1] / o->p[3] + 1.0) / 2.0 + cam->H / 2.0;
    double tx = o->t[0], ty = o->t[1], tz = o->t[2], fx = cam->fx, fy = cam->fy;
    double J[2][3] = { { fx / tz, 0, -fx * tx / (tz * tz) }, { 0, fy / tz, -fy * ty / (tz * tz) } };
    double w = quats[4 * g], x = quats[4 * g + 1], y = quats[4 * g + 2], z = quats[4 * g + 3];
    double R[3][3] = {
        { 1 - 2 * (y * y + z * z), 2 * (x * y - w * z),       2 * (x * z + w * y) },
        { 2 * (x * y + w * z),     1 - 2 * (x * x - z * z),   2 * (y * z - w * x) },
        { 2 * (x * z - w * y),     2 * (y * z + w * x),       1 - 2 * (x * x + y * y) } };
    double e[3] = { exp((double)scales[3 * g]), exp((double)scales[3 * g + 1]), exp((double)scales[3 * g + 2]) };
    double Wm[3][3], Sg[3][3], A[2][3], AS[2][3], cov[2][2];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Wm[i][j] = R[i][j] * e[j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Sg[i][j] = Wm[i][0] * Wm[j][0] + Wm[i][1] * Wm[j][1] + Wm[i][2] * Wm[j][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) A[i][j] = J[i][0] * R[0][j] + J[i][1] * R[1][j] + J[i][2] * R[2][j];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) AS[i][j] = A[i][0] * Sg[0][j] + A[i][1] * Sg[1][j] + A[i][2] * Sg[2][j];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) cov[i][j] = AS[i][0] * A[j][0] + AS[i][1] * A[j][1] + AS[i][2] * A[j][2] + 0.3;
    double det = cov[0][0] * cov[1][1] - cov[0][1] * cov[1][0];
    o->M[0] = cov[1][1] / det; o->M[1] = -cov[1][0] / det; o->M[2] = -cov[0][1] / det; o->M[3] = cov[0][0] / det;
    double ez = exp((double)opac[g]);
    o->sig = ez / (1.0 + ez);
    double v[3];
    for (int i = 0; i < 3; ++i) v[i] = o->p[i] - ((double)cam->lookAt[i] - (double)cam->eye[i]);
    double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double b[16], db[16][3];
    sh_basis_f64(deg, v[0] / nrm, v[1] / nrm, v[2] / nrm, b, db);
    for (int c = 0; c < 3; ++c) {
        double s = 0.5;
        for (int k = 0; k < K; ++k) s += (double)shs[(int64_t)3 * K * g + c + 3 * k] * b[k];
        o->rgb[c] = s;
    }
}

/* chain g2d = (drgb[3], dsig, dmu[2], dM[4]) back to the parameters of gaussian g */
static void bwd64(int64_t g, int deg, const float *means, const float *scales, const float *quats,
                  const float *opac, const float *shs, const gso_camera *cam, const double *g2,
                  double *dmeans, double *dscales, double *dquats, double *dopac, double *dshs) {
    const int K = (deg + 1) * (deg + 1);
    (void)means;
    g64 f;
    fwd64(g, deg, means, scales, quats, opac, shs, cam, &f);
    const double *grgb = g2, gsig = g2[3], *gmu = g2 + 4, *gM = g2 + 6;
    double tx = f.t[0], ty = f.t[1], tz = f.t[2], fx = cam->fx, fy = cam->fy;
    double J[2][3] = { { fx / tz, 0, -fx * tx / (tz * tz) }, { 0, fy / tz, -fy * ty / (tz * tz) } };
    double w = quats[4 * g], x = quats[4 * g + 1], y = quats[4 * g + 2], z = quats[4 * g + 3];
    double R[3][3] = {
        { 1 - 2 * (y * y + z * z), 2 * (x * y - w * z),       2 * (x * z + w * y) },
        { 2 * (x * y + w * z),     1 - 2 * (x * x - z * z),   2 * (y * z - w * x) },
        { 2 * (x * z - w * y),     2 * (y * z + w * x),       1 - 2 * (x * x + y * y) } };
    double e[3] = { exp((double)scales[3 * g]), exp((double)scales[3 * g + 1]), exp((double)scales[3 * g + 2]) };
    double Wm[3][3], Sg[3][3], A[2][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Wm[i][j] = R[i][j] * e[j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Sg[i][j] = Wm[i][0] * Wm[j][0] + Wm[i][1] * Wm[j][1] + Wm[i][2] * Wm[j][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) A[i][j] = J[i][0] * R[0][j] + J[i][1] * R[1][j] + J[i][2] * R[2][j];
    /* M = cov^-1  =>  dcov = -M^T gM M^T   (row,col): M[r][c] = f.M[r + 2c] */
    double M[2][2] = { { f.M[0], f.M[2] }, { f.M[1], f.M[3] } };
    double G[2][2] = { { gM[0], gM[2] }, { gM[1], gM[3] } };
    double t1[2][2], dcov[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) t1[i][j] = M[0][i] * G[0][j] + M[1][i] * G[1][j];       /* M^T G */
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) dcov[i][j] = -(t1[i][0] * M[j][0] + t1[i][1] * M[j][1]); /* .. M^T */
    /* cov = A Sg A^T + 0.3 */
    double ds2[2][2] = { { 2 * dcov[0][0], dcov[0][1] + dcov[1][0] }, { dcov[0][1] + dcov[1][0], 2 * dcov[1][1] } };
    double ASg[2][3], dA[2][3], dSg[3][3], dJ[2][3], dR[3][3], dWm[3][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) ASg[i][j] = A[i][0] * Sg[0][j] + A[i][1] * Sg[1][j] + A[i][2] * Sg[2][j];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) dA[i][j] = ds2[i][0] * ASg[0][j] + ds2[i][1] * ASg[1][j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) s += A[a][i] * dcov[a][b] * A[b][j];
        dSg[i][j] = s;
    }
    /* A = J R */
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) dJ[i][j] = dA[i][0] * R[j][0] + dA[i][1] * R[j][1] + dA[i][2] * R[j][2];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dR[i][j] = J[0][i] * dA[0][j] + J[1][i] * dA[1][j];
    /* Sg = Wm Wm^T */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += (dSg[i][k] + dSg[k][i]) * Wm[k][j];
        dWm[i][j] = s;
    }
    double de[3] = { 0, 0, 0 };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { dR[i][j] += dWm[i][j] * e[j]; de[j] += R[i][j] * dWm[i][j]; }
    for (int j = 0; j < 3; ++j) dscales[3 * g + j] += de[j] * e[j];
    /* R(q) with the reference's R22 = 1 - 2(x^2 - z^2) */
    double dw = 0, dx = 0, dy = 0, dz = 0;
    dy += -4 * y * dR[0][0]; dz += -4 * z * dR[0][0];
    dx += 2 * y * dR[1][0]; dy += 2 * x * dR[1][0]; dw += 2 * z * dR[1][0]; dz += 2 * w * dR[1][0];
    dx += 2 * z * dR[2][0]; dz += 2 * x * dR[2][0]; dw += -2 * y * dR[2][0]; dy += -2 * w * dR[2][0];
    dx += 2 * y * dR[0][1]; dy += 2 * x * dR[0][1]; dw += -2 * z * dR[0][1]; dz += -2 * w * dR[0][1];
    dx += -4 * x * dR[1][1]; dz += 4 * z * dR[1][1];
    dy += 2 * z * dR[2][1]; dz += 2 * y * dR[2][1]; dw += 2 * x * dR[2][1]; dx += 2 * w * dR[2][1];
    dx += 2 * z * dR[0][2]; dz += 2 * x * dR[0][2]; dw += 2 * y * dR[0][2]; dy += 2 * w * dR[0][2];
    dy += 2 * z * dR[1][2]; dz += 2 * y * dR[1][2]; dw += -2 * x * dR[1][2]; dx += -2 * w * dR[1][2];
    dx += -4 * x * dR[2][2]; dy += -4 * y * dR[2][2];
    dquats[4 * g] += dw; dquats[4 * g + 1] += dx; dquats[4 * g + 2] += dy; dquats[4 * g + 3] += dz;
    /* J(t) */
    double dt[4] = { 0, 0, 0, 0 }, dp[4] = { 0, 0, 0, 0 };
    double tz2 = tz * tz, tz3 = tz2 * tz;
    dt[0] += dJ[0][2] * (-fx / tz2);
    dt[1] += dJ[1][2] * (-fy / tz2);
    dt[2] += dJ[0][0] * (-fx / tz2) + dJ[1][1] * (-fy / tz2) + dJ[0][2] * (2 * fx * tx / tz3) + dJ[1][2] * (2 * fy * ty / tz3);
    /* mu(p) */
    double Wd = cam->W, Hd = cam->H;
    dp[0] += gmu[0] * 0.5 * Wd / f.p[3];
    dp[1] += gmu[1] * 0.5 * Hd / f.p[3];
    dp[3] += -gmu[0] * 0.5 * Wd * f.p[0] / (f.p[3] * f.p[3]) - gmu[1] * 0.5 * Hd * f.p[1] / (f.p[3] * f.p[3]);
    /* rgb(sh, dir(p)) */
    double v[3];
    for (int i = 0; i < 3; ++i) v[i] = f.p[i] - ((double)cam->lookAt[i] - (double)cam->eye[i]);
    double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double dir[3] = { v[0] / nrm, v[1] / nrm, v[2] / nrm };
    double b[16], db[16][3], ddir[3] = { 0, 0, 0 };
    sh_basis_f64(deg, dir[0], dir[1], dir[2], b, db);
    for (int k = 0; k < K; ++k) {
        double cs = 0;
        for (int c = 0; c < 3; ++c) {
            dshs[(int64_t)3 * K * g + c + 3 * k] += b[k] * grgb[c];
            cs += grgb[c] * (double)shs[(int64_t)3 * K * g + c + 3 * k];
        }
        for (int a = 0; a < 3; ++a) ddir[a] += cs * db[k][a];
    }
    double dd = dir[0] * ddir[0] + dir[1] * ddir[1] + dir[2] * ddir[2];
    for (int a = 0; a < 3; ++a) dp[a] += (ddir[a] - dir[a] * dd) / nrm;
    /* p = P t ; t = T [m;1] */
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) dt[j] += cam->P[i + 4 * j] * dp[i];
    for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int i = 0; i < 4; ++i) s += cam->T[i + 4 * j] * dt[i];
        dmeans[3 * g + j] += s;
    }
    dopac[g] += gsig * f.sig * (1.0 - f.sig);
}

/* fp64 adjoint of the composite (shared by the 3-D and the 2-D renderer): walks every pixel's list with the fp32
 * forward's discrete decisions (boxes, lists, near/far when tps != NULL, early-out rule) and accumulates
 * g2d[10*g ..] = d{rgb3, sig, mu2, inv4}. */
static void composite_adjoint64(int64_t n, const g64 *F, const gso_camera *cam, int tile, int gx, int gy,
                                const uint32_t *ranges, const uint32_t *ids, const float *bbs, const float *tps,
                                float t_min, const float *dC, double *g2d) {
    (void)n;
    const int W = cam->W, H = cam->H;
    const int64_t plane = (int64_t)W * H, nt = (int64_t)gx * gy;
#pragma omp parallel
    {
        size_t capk = 1024;
        double *al = (double *)malloc(sizeof(double) * capk * 5);   /* alpha, T, Gexp, dX, dY per contributor */
        uint32_t *who = (uint32_t *)malloc(sizeof(uint32_t) * capk);
#pragma omp for schedule(dynamic, 1)
        for (int64_t t = 0; t < nt; ++t) {
            int bx = (int)(t % gx) + 1, by = (int)(t / gx) + 1;
            uint32_t s0 = ranges[2 * t], s1 = ranges[2 * t + 1];
            if ((size_t)(s1 - s0) + 1 > capk) {
                capk = (size_t)(s1 - s0) + 1;
                al = (double *)realloc(al, sizeof(double) * capk * 5);
                who = (uint32_t *)realloc(who, sizeof(uint32_t) * capk);
            }
            for (int tyi = 1; tyi <= tile; ++tyi)
                for (int txi = 1; txi <= tile; ++txi) {
                    int i = (bx - 1) * tile + txi, j = (by - 1) * tile + tyi;
                    if (i > W || j > H) continue;
                    int64_t px = (int64_t)(i - 1) + (int64_t)W * (j - 1);
                    double dc[3] = { dC[px], dC[px + plane], dC[px + 2 * plane] };
                    double Tr = 1.0;
                    size_t m = 0;
                    float fi = (float)i, fj = (float)j;
                    for (uint32_t k = s0; k < s1; ++k) {
                        if (t_min > 0.0f && ((k - s0) % GSO_EARLY_BATCH) == 0 && Tr < (double)t_min) break;
                        uint32_t b = ids[k];
                        if (tps) {
                            float cz = tps[4 * (int64_t)b + 2];
                            if (cz < cam->near_ || cz > cam->far_) continue;
                        }
                        const float *bb = bbs + 4 * (int64_t)b;
                        if (!((bb[0] <= fi) && (fi <= bb[2]) && (bb[1] <= fj) && (fj <= bb[3]))) continue;
                        const g64 *f = &F[b];
                        double dX = (double)i - f->mu[0], dY = (double)j - f->mu[1];
                        double v1 = f->M[0] * dX + f->M[2] * dY, v2 = f->M[1] * dX + f->M[3] * dY;
                        double Gx = exp(-0.5 * (v1 * dX + v2 * dY));
                        double a = f->sig * Gx;
                        al[5 * m] = a; al[5 * m + 1] = Tr; al[5 * m + 2] = Gx; al[5 * m + 3] = dX; al[5 * m + 4] = dY;
                        who[m] = b; ++m;
                        Tr *= (1.0 - a);
                    }
                    /* back to front: B = colour (dotted with dC) of everything behind k, as
                     * seen through nothing: d(C.dC)/d alpha_k = T_k (c_k.dC - B_k) */
                    double B = 0.0;
                    for (size_t q = m; q-- > 0;) {
                        uint32_t b = who[q];
                        const g64 *f = &F[b];
                        double a = al[5 * q], Tk = al[5 * q + 1], Gx = al[5 * q + 2], dX = al[5 * q + 3], dY = al[5 * q + 4];
                        double cd = f->rgb[0] * dc[0] + f->rgb[1] * dc[1] + f->rgb[2] * dc[2];
                        double da = Tk * (cd - B);
                        B = cd * a + (1.0 - a) * B;
                        double wgt = a * Tk;
                        double ddist = -a * da;                         /* alpha = sig*exp(-dist) */
                        double ddX = ddist * 0.5 * (2 * f->M[0] * dX + (f->M[1] + f->M[2]) * dY);
                        double ddY = ddist * 0.5 * (2 * f->M[3] * dY + (f->M[1] + f->M[2]) * dX);
                        double add[10] = { wgt * dc[0], wgt * dc[1], wgt * dc[2], Gx * da, -ddX, -ddY,
                                           0.5 * dX * dX * ddist, 0.5 * dX * dY * ddist, 0.5 * dX * dY * ddist, 0.5 * dY * dY * ddist };
                        for (int c = 0; c < 10; ++c) {
#pragma omp atomic
                            g2d[10 * (int64_t)b + c] += add[c];
                        }
                    }
                }
        }
        free(al); free(who);
    }
}

void gso_backward(int64_t n, int sh_degree,
                  const float *means, const float *scales, const float *quats,
                  const float *opacities, const float *shs, const gso_camera *cam,
                  int tile, int gx, int gy, const uint32_t *ranges, const uint32_t *ids,
                  const float *bbs_in, float t_min, const float *dC,
                  double *dmeans, double *dscales, double *dquats, double *dopac,
                  double *dshs, double *g2d_out) {
    /* discrete decisions come from the fp32 forward */
    float *tps = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    float *bbs = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    gso_preprocess(n, sh_degree, means, scales, quats, opacities, shs, cam, NULL, tps, NULL, NULL, NULL, NULL, bbs, NULL, NULL);
    if (bbs_in) memcpy(bbs, bbs_in, sizeof(float) * 4 * (size_t)n);
    g64 *F = (g64 *)calloc((size_t)(n > 0 ? n : 1), sizeof(g64));
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n; ++g) fwd64(g, sh_degree, means, scales, quats, opacities, shs, cam, &F[g]);
    double *g2d = (double *)calloc((size_t)(n > 0 ? n : 1) * 10, sizeof(double));
    composite_adjoint64(n, F, cam, tile, gx, gy, ranges, ids, bbs, tps, t_min, dC, g2d);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n; ++g)
        bwd64(g, sh_degree, means, scales, quats, opacities, shs, cam, g2d + 10 * g, dmeans, dscales, dquats, dopac, dshs);
    if (g2d_out) memcpy(g2d_out, g2d, sizeof(double) * 10 * (size_t)n);
    free(g2d); free(F); free(tps); free(bbs);
}

/* ------------------------------------------------------------------ 2-D renderer backward (fp64 adjoint) */

/* Derived adjoint of gso_preprocess2d + composite (the reference's splatGrads, splat.jl:271-396, mixes several
 * forwards and is not reproduced; its CONTRACT is: gradients accumulate into arrays shaped like the parameters).
 * Sigma = R S^2 R' + 0.3 I, M = Sigma^-1, mu = (W mx, H my), alpha = opacity * exp(-dist/2), colour = colors. */
void gso_backward2d(int64_t n, const float *means, const float *scales, const float *rots,
                    const float *opacities, const float *colors, int W, int H,
                    int tile, int gx, int gy, const uint32_t *ranges, const uint32_t *ids,
                    float t_min, const float *dC,
                    double *dmeans, double *dscales, double *drots, double *dopac, double *dcolors, double *g2d_out) {
    gso_camera cam;
    memset(&cam, 0, sizeof(cam));
    cam.W = W; cam.H = H;
    float *bbs = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    gso_preprocess2d(n, means, scales, rots, opacities, colors, W, H, NULL, NULL, NULL, bbs, NULL, NULL);
    g64 *F = (g64 *)calloc((size_t)(n > 0 ? n : 1), sizeof(g64));
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n; ++g) {
        g64 *o = &F[g];
        double th = rots[g], c = cos(th), s = sin(th);
        double e1 = exp((double)scales[2 * g]), e2 = exp((double)scales[2 * g + 1]);
        double Wm[2][2] = { { c * e1, -s * e2 }, { s * e1, c * e2 } };
        double cov[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) cov[i][j] = Wm[i][0] * Wm[j][0] + Wm[i][1] * Wm[j][1] + (i == j ? 0.3 : 0.0);
        double det = cov[0][0] * cov[1][1] - cov[0][1] * cov[1][0];
        o->M[0] = cov[1][1] / det; o->M[1] = -cov[1][0] / det; o->M[2] = -cov[0][1] / det; o->M[3] = cov[0][0] / det;
        o->mu[0] = (double)W * means[2 * g]; o->mu[1] = (double)H * means[2 * g + 1];
        o->sig = fminf(fmaxf(opacities[g], 0.0f), 0.99999994f);
        for (int k = 0; k < 3; ++k) o->rgb[k] = colors[3 * g + k];
    }
    double *g2d = (double *)calloc((size_t)(n > 0 ? n : 1) * 10, sizeof(double));
    composite_adjoint64(n, F, &cam, tile, gx, gy, ranges, ids, bbs, NULL, t_min, dC, g2d);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < n; ++g) {
        const double *g2 = g2d + 10 * g, *gM = g2 + 6;
        const g64 *f = &F[g];
        for (int k = 0; k < 3; ++k) dcolors[3 * g + k] += g2[k];
        if (opacities[g] > 0.0f && opacities[g] < 0.99999994f) dopac[g] += g2[3];     /* clamp: zero slope outside */
        dmeans[2 * g] += (double)W * g2[4];
        dmeans[2 * g + 1] += (double)H * g2[5];
        /* M = cov^-1 => dcov = -M' gM M' */
        double M[2][2] = { { f->M[0], f->M[2] }, { f->M[1], f->M[3] } };
        double G[2][2] = { { gM[0], gM[2] }, { gM[1], gM[3] } };
        double t1[2][2], dcov[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) t1[i][j] = M[0][i] * G[0][j] + M[1][i] * G[1][j];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) dcov[i][j] = -(t1[i][0] * M[j][0] + t1[i][1] * M[j][1]);
        /* cov = Wm Wm' + 0.3 I, Wm = R(theta) diag(e) */
        double th = rots[g], c = cos(th), s = sin(th);
        double e[2] = { exp((double)scales[2 * g]), exp((double)scales[2 * g + 1]) };
        double R[2][2] = { { c, -s }, { s, c } }, dRdth[2][2] = { { -s, -c }, { c, -s } };
        double Wm[2][2], dWm[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) Wm[i][j] = R[i][j] * e[j];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            double acc = 0;
            for (int k = 0; k < 2; ++k) acc += (dcov[i][k] + dcov[k][i]) * Wm[k][j];
            dWm[i][j] = acc;
        }
        double dth = 0;
        for (int j = 0; j < 2; ++j) {
            double de = 0;
            for (int i = 0; i < 2; ++i) { de += R[i][j] * dWm[i][j]; dth += dRdth[i][j] * e[j] * dWm[i][j]; }
            dscales[2 * g + j] += de * e[j];
        }
        drots[g] += dth;
    }
    if (g2d_out) memcpy(g2d_out, g2d, sizeof(double) * 10 * (size_t)n);
    free(g2d); free(F); free(bbs);
}

