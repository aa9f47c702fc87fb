/* render_c.c -- libgsplat_hip.so from plain C: no Python, no torch, nothing but include/gsplat.h.
 *
 *   gcc -O2 -Iinclude examples/render_c.c -o examples/render_c -Lgaussiansplat_amd/lib -lgsplat_hip -lm \
 *       -Wl,-rpath,'$ORIGIN/../gaussiansplat_amd/lib'
 *   examples/render_c [n_gaussians] [W] [H] [frames]
 *
 * The call sequence is the reference's examples/main.jl:14-34 (getRenderer -> preprocess -> compactIdxs -> forward)
 * plus backward into library-owned gradient arrays; what a Julia `ccall` host does (julia/backend.jl) in C.
 * Prints per-frame wall time (host clock around the API calls + gs_synchronize) and checksums. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gsplat.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {                     /* xorshift64*: deterministic, no libc rand */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
}
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

#define CHECK(call) do { int rc__ = (call); if (rc__ != GS_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, gs_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    const int W = argc > 2 ? atoi(argv[2]) : 800, H = argc > 3 ? atoi(argv[3]) : 800, frames = argc > 4 ? atoi(argv[4]) : 10;
    const int deg = 3, K = 16;
    gs_ctx *ctx = NULL;
    gs_config cfg;
    gs_default_config(&cfg);
    if (gs_create(&ctx, 0, &cfg) != GS_OK) { fprintf(stderr, "gs_create: %s\n", gs_last_error(NULL)); return 2; }

    /* camera of camera.jl:24-47 with fx scaled to the image (the build's synthetic scenes) */
    const float fx = 3200.0f * W / 1920.0f, near_ = 0.1f, far_ = 100.0f, eye[3] = {1, 3, 30}, lookAt[3] = {0, 0, 0};
    float w[3] = {lookAt[0] - eye[0], lookAt[1] - eye[1], lookAt[2] - eye[2]}, up[3] = {0, 1, 0}, u[3], v[3];
    float nw = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 3; ++i) w[i] /= nw;
    u[0] = up[1] * w[2] - up[2] * w[1]; u[1] = up[2] * w[0] - up[0] * w[2]; u[2] = up[0] * w[1] - up[1] * w[0];
    float nu = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int i = 0; i < 3; ++i) u[i] /= nu;
    v[0] = w[1] * u[2] - w[2] * u[1]; v[1] = w[2] * u[0] - w[0] * u[2]; v[2] = w[0] * u[1] - w[1] * u[0];
    float T[16] = {0}, P[16] = {0};             /* column-major; row 4 of T all zero (camera.jl:96) */
    const float *ax[3] = {u, v, w};
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[r + 4 * c] = ax[r][c];
        T[r + 12] = -(ax[r][0] * eye[0] + ax[r][1] * eye[1] + ax[r][2] * eye[2]);
    }
    P[0] = 2 * fx / W; P[5] = 2 * fx / H; P[10] = (far_ + near_) / (far_ - near_); P[14] = -2 * far_ * near_ / (far_ - near_); P[11] = 1.0f;

    float *means = malloc(sizeof(float) * 3 * n), *scales = malloc(sizeof(float) * 3 * n), *quats = malloc(sizeof(float) * 4 * n);
    float *opac = malloc(sizeof(float) * n), *shs = malloc(sizeof(float) * 3 * K * n);
    const double Wv = W * 30.0 / fx, Hv = H * 30.0 / fx;
    for (int64_t g = 0; g < n; ++g) {
        means[3 * g] = (float)((urand() - 0.5) * Wv); means[3 * g + 1] = (float)((urand() - 0.5) * Hv); means[3 * g + 2] = (float)(urand() * 8 - 4);
        double q[4], nq = 0;
        for (int i = 0; i < 3; ++i) scales[3 * g + i] = (float)(-4.5 + 2.0 * urand());
        for (int i = 0; i < 4; ++i) { q[i] = urand() - 0.5; nq += q[i] * q[i]; }
        for (int i = 0; i < 4; ++i) quats[4 * g + i] = (float)(q[i] / sqrt(nq));
        opac[g] = (float)(-2.0 + 6.0 * urand());
        for (int i = 0; i < 3 * K; ++i) shs[3 * K * g + i] = (float)((urand() - 0.5) * (i < 3 ? 1.0 : 0.3));
    }
    CHECK(gs_set_model(ctx, n, deg, means, scales, quats, opac, shs, GS_MEM_HOST));
    CHECK(gs_set_camera(ctx, T, P, fx, fx, near_, far_, eye, lookAt, W, H));

    const size_t px = (size_t)W * H;
    float *image = malloc(sizeof(float) * 3 * px), *trans = malloc(sizeof(float) * px), *dC = malloc(sizeof(float) * 3 * px);
    for (size_t i = 0; i < 3 * px; ++i) dC[i] = (float)(urand() - 0.5);
    gs_grads grads;
    CHECK(gs_grads_alloc(ctx, &grads));

    double best = 1e30;
    for (int f = 0; f < frames; ++f) {
        CHECK(gs_reset_grads(ctx, &grads));
        CHECK(gs_synchronize(ctx));
        const double t0 = now_ms();
        CHECK(gs_preprocess(ctx));
        CHECK(gs_bin(ctx, 0, 0));
        CHECK(gs_forward(ctx, NULL, NULL, GS_MEM_DEVICE));           /* image stays on the device */
        CHECK(gs_backward(ctx, dC, GS_MEM_HOST, &grads));            /* dC from the host: includes the PCIe copy */
        CHECK(gs_synchronize(ctx));
        const double dt = now_ms() - t0;
        if (dt < best) best = dt;
    }
    CHECK(gs_forward(ctx, image, trans, GS_MEM_HOST));
    float *dmeans = malloc(sizeof(float) * 3 * n);
    CHECK(gs_grads_read(ctx, &grads, dmeans, NULL, NULL, NULL, NULL));
    double simg = 0, str = 0, sg = 0;
    for (size_t i = 0; i < 3 * px; ++i) simg += image[i];
    for (size_t i = 0; i < px; ++i) str += trans[i];
    for (int64_t i = 0; i < 3 * n; ++i) sg += fabs(dmeans[i]);
    int64_t wf = 0, wb = 0;
    CHECK(gs_get_work_counters(ctx, &wf, &wb));
    printf("render_c ok: n=%lld %dx%d instances=%lld walked=%lld best_frame_ms=%.3f (fwd+bwd, dC uploaded per frame) "
           "mean_image=%.6f mean_T=%.6f sum|dmeans|=%.6e\n", (long long)n, W, H, (long long)gs_num_instances(ctx), (long long)wf, best,
           simg / (3.0 * px), str / px, sg);
    const int ok = isfinite(simg) && isfinite(sg) && sg > 0 && str / px < 1.0;
    gs_destroy(ctx);
    return ok ? 0 : 3;
}
