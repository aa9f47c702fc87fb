// gs_loss.hip -- fused L1 + DSSIM loss and its image gradient, and the SGD parameter update.
//
// the step after backward.  Replaces the reference's host-side loss
// (src/loss.jl:5-72: 11x11 window exp(-r)/sqrt(2 sigma^2) normalised, SSIM via grouped `conv` with
// zero padding 5, loss = 0.9*sum|img-gt|/(2 length) + 0.1*(1-mean ssim)/2) and the commented-out
// AD + SGD lines of src/train.jl:39-46 (grads = gradient(lossFunc, img, gt); param .-= lr*grad).
// The reference relies on an AD package that is not in its Manifest; here the gradient is written
// out: with mu = k*x, s_xx = k*x^2, s_xy = k*xy and S = A1 A2 / (B1 B2),
//     dS_q/dx_p = k(q-p) [ dS/dmu_x + 2 x_p dS/ds_xx + y_p dS/ds_xy ]_q
// so dC = w1 sign(x-y) + w2 (k*G_mu + 2x k*G_xx + y k*G_xy): two LDS-tiled 11x11 stencil passes
// (pass 1 reads 8 B and writes 12 B per pixel-channel, pass 2 reads 20 B and writes 4 B; no MFMA: the window is a
// 121-tap non-separable stencil on fp32 data whose symmetric taps are folded first).
#include "gs_common.h"

#define LW 11
#define LP 5
#define LTX 64                    // tile: 64 x 16 outputs of one channel per workgroup, 4 consecutive pixels per thread
#define LTY 16
#define LHX (LTX + 2 * LP)        // 74 halo columns
#define LHY (LTY + 2 * LP)        // 26 halo rows
#define LPITCH 76                 // floats per LDS row (16-byte aligned rows for ds_read_b128)
#define LNF 6                     // folded window rows / columns: w[j][i] = w[10-j][i] = w[j][10-i]

struct GsLossArgs {
    int W, H, C;
    const float *img, *gt;
    float *g_mu, *g_xx, *g_xy;     // C*H*W each
    float *dC;
    double *acc;                   // [0] sum |img-gt|, [1] sum ssim map
    float w_l1, w_ssim;            // (1-lam)/(2n), -lam/(2n)
    float win[LW * LW];
};

// The window exp(-|d|) of loss.jl:4-11 is not separable, but it is symmetric under both reflections, so the 121 taps fold
// onto 6 x 6 weights: rows j and 10-j are added sample by sample first (shared by a thread's four outputs), then columns i
// and 10-i per output.  That is 1/3 of the multiply-adds of the plain stencil, all of them on VGPR operands (an SGPR
// weight operand halves the fma rate on gfx950), and the samples come in as 16-byte LDS reads shared by four outputs.
// Measured at 1920x1080x3: 0.85 ms for the two plain 121-tap kernels of round 1, see DESIGN.md for this version.
// a uniform value the compiler must hold in a VGPR: a v_fmac with an SGPR source issues at 4.4 cycles per wave on gfx950, 2.3 with
// VGPR sources only (tools/valu_ubench3.hip); left alone the kernarg weights stay in SGPRs (242 of the loop's 630 VALU instructions)
__device__ __forceinline__ float in_vgpr(float x) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
__device__ __forceinline__ float block_sum_256(float v, float *sm) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// halo tile of one plane: rows y0-5 .. y0+20, columns x0-5 .. x0+68, zero outside the image (loss.jl:29 pads with zeros).
// Two steps, so that a kernel can put the fetches of ALL its planes in flight before the first LDS store waits for one of them:
// as one loop (load, wait, store per element) the 8 elements per thread and plane were 16 .. 24 global-load latencies in a row
// (round 4: ssim_stats 172 -> 145 us, ssim_grad 92 -> 68 us at 1920x1080x3).  The address is clamped instead of predicated
// (every load is unconditional, the select happens on the value).
template <int NT> struct HaloIt { static constexpr int n = (LHY * LPITCH + NT - 1) / NT; };     // halo elements per thread and plane
template <int NT>
__device__ __forceinline__ void halo_fetch(float (&v)[HaloIt<NT>::n], const float *__restrict__ plane, int W, int H, int x0, int y0) {
    int hy = (int)threadIdx.x / LPITCH, hx = (int)threadIdx.x - hy * LPITCH;          // element i = thread + NT k: one division, then steps
#pragma unroll
    for (int k = 0; k < HaloIt<NT>::n; ++k) {
        const int gx = x0 + hx - LP, gy = y0 + hy - LP;
        const bool in = hx < LHX && gx >= 0 && gx < W && gy >= 0 && gy < H;          // (hy >= LHY: gy may still be inside; never stored)
        const float t = plane[(uint32_t)(min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1))];   // uniform base + 32-bit lane offset
        v[k] = in ? t : 0.0f;
        hx += NT % LPITCH; hy += NT / LPITCH;
        if (hx >= LPITCH) { hx -= LPITCH; hy += 1; }
    }
}
template <int NT>
__device__ __forceinline__ void halo_store(float (*dst)[LPITCH], const float (&v)[HaloIt<NT>::n]) {
    float *d = &dst[0][0];
#pragma unroll
    for (int k = 0; k < HaloIt<NT>::n; ++k) {
        const int i = (int)threadIdx.x + NT * k;
        if (i < LHY * LPITCH) d[i] = v[k];
    }
}
// NF (a multiple of 4) consecutive samples of a halo row starting at column col0 (a multiple of 4): a thread with NO outputs needs
// NO + 10 of them
template <int NF>
__device__ __forceinline__ void load_row(float (&v)[NF], const float (*src)[LPITCH], int row, int col0) {
    const float4 *p = reinterpret_cast<const float4 *>(&src[row][col0]);
#pragma unroll
    for (int q = 0; q < NF / 4; ++q) { const float4 t = p[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}
// acc[o] += sum_i w[i] F[o + i] over the 11 window columns of output o, with the columns folded: 5 adds + 6 fma per output
template <int NO, int NF>
__device__ __forceinline__ void fold_cols(float (&acc)[NO], const float (&F)[NF], const float (&w)[LNF]) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float a = acc[o];
#pragma unroll
        for (int c = 0; c < LP; ++c) a = fmaf(w[c], F[o + c] + F[o + 2 * LP - c], a);
        acc[o] = fmaf(w[LP], F[o + LP], a);
    }
}
// Tile of a workgroup.  The dispatcher deals workgroups to the 8 XCDs round-robin (block b runs on XCD b % 8), each XCD with its own
// L2; in plain grid order the eight tiles around a tile sit on eight different XCDs and every one fetches the shared halo from memory
// itself.  Here XCD x works through the x-th eighth of the tiles in row-major order, so the tiles in flight on an XCD are neighbours
// (stats 109 -> 105 us, grad 72 -> 68 us at 1920x1080x3).  (Persistent workgroups that fetch the next tile's halo before the stencil of
// the current one were measured too, into registers and by LDS-DMA into a second pair of buffers: 111 / 114 us for the stats kernel
// against 105 -- the kernel is bound by its 2700 VALU instructions per wave, not by the fetch; profiles/r04m_loss_persistent.log,
// r04n_loss_lds_dma.log.)
struct LossTile { int x0, y0, c; };
__device__ __forceinline__ bool loss_tile(const GsLossArgs &a, LossTile &t) {
    const int gx = (a.W + LTX - 1) / LTX, gy = (a.H + LTY - 1) / LTY, T = gx * gy * a.C;
    const int chunk = (T + 7) / 8;
    const int k = (int)(blockIdx.x >> 3);
    const int id = (int)(blockIdx.x & 7u) * chunk + k;
    if (k >= chunk || id >= T) return false;
    t.c = id / (gx * gy);
    const int r = id - t.c * (gx * gy);
    t.y0 = (r / gx) * LTY; t.x0 = (r % gx) * LTX;
    return true;
}
// The folded window (6 x 6 weights, rows padded to 8 floats) in LDS: a row is two broadcast reads straight into VGPRs, issued with the
// sample rows.  (From the kernarg segment a row was an s_load whose wait stood in front of every iteration's LDS reads, and its
// values sat in SGPRs: an SGPR source halves the issue rate of a v_fmac on gfx950.)
__device__ __forceinline__ void stage_window(float (*swin)[8], const GsLossArgs &a) {
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < LNF; ++j)
#pragma unroll
            for (int i = 0; i < LNF; ++i) swin[j][i] = a.win[j * LW + i];
    }
}
__device__ __forceinline__ void window_row(float (&w)[LNF], const float (*swin)[8], int j) {
    const float4 t = *reinterpret_cast<const float4 *>(&swin[j][0]);
    const float2 u = *reinterpret_cast<const float2 *>(&swin[j][4]);
    w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w; w[4] = u.x; w[5] = u.y;
}

__global__ __launch_bounds__(256, 3) void ssim_stats_kernel(GsLossArgs a) {
    __shared__ __attribute__((aligned(16))) float sx[LHY][LPITCH], sy[LHY][LPITCH];
    __shared__ float sm[4], sm2[4];
    __shared__ __attribute__((aligned(16))) float swin[LNF][8];
    stage_window(swin, a);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const size_t plane = (size_t)a.W * a.H;
    LossTile cur;
    if (!loss_tile(a, cur)) return;
    {
        float hx_[HaloIt<256>::n], hy_[HaloIt<256>::n];
        halo_fetch<256>(hx_, a.img + cur.c * plane, a.W, a.H, cur.x0, cur.y0);
        halo_fetch<256>(hy_, a.gt + cur.c * plane, a.W, a.H, cur.x0, cur.y0);
        halo_store<256>(sx, hx_); halo_store<256>(sy, hy_);
    }
    __syncthreads();
    float l1 = 0.0f, St = 0.0f;
    {
        float mux[4] = {0, 0, 0, 0}, muy[4] = {0, 0, 0, 0}, sxx[4] = {0, 0, 0, 0}, syy[4] = {0, 0, 0, 0}, sxy[4] = {0, 0, 0, 0};
        float xc[4], yc[4];                                                       // the outputs' own samples (L1 term)
#pragma unroll 1
        for (int j = 0; j < LNF; ++j) {
            float w[LNF];
            window_row(w, swin, j);
            float x1[16], y1[16], F[16];
            load_row<16>(x1, sx, ty + j, 4 * tx); load_row<16>(y1, sy, ty + j, 4 * tx);
            if (j < LP) {
                float x2[16], y2[16];
                load_row<16>(x2, sx, ty + 2 * LP - j, 4 * tx); load_row<16>(y2, sy, ty + 2 * LP - j, 4 * tx);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = x1[i] + x2[i];
                fold_cols(mux, F, w);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = y1[i] + y2[i];
                fold_cols(muy, F, w);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = fmaf(x1[i], x1[i], x2[i] * x2[i]);
                fold_cols(sxx, F, w);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = fmaf(y1[i], y1[i], y2[i] * y2[i]);
                fold_cols(syy, F, w);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = fmaf(x1[i], y1[i], x2[i] * y2[i]);
                fold_cols(sxy, F, w);
            } else {                                                              // the window's centre row
#pragma unroll
                for (int o = 0; o < 4; ++o) { xc[o] = x1[o + LP]; yc[o] = y1[o + LP]; }
                fold_cols(mux, x1, w);
                fold_cols(muy, y1, w);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = x1[i] * x1[i];
                fold_cols(sxx, F, w);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = y1[i] * y1[i];
                fold_cols(syy, F, w);
#pragma unroll
                for (int i = 0; i < 14; ++i) F[i] = x1[i] * y1[i];
                fold_cols(sxy, F, w);
            }
        }
        const int py = cur.y0 + ty;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int px = cur.x0 + 4 * tx + o;
            if (px < a.W && py < a.H) {
                const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;              // loss.jl:37-38
                const float s2x = sxx[o] - mux[o] * mux[o], s2y = syy[o] - muy[o] * muy[o], cxy = sxy[o] - mux[o] * muy[o];
                const float A1 = 2.0f * mux[o] * muy[o] + C1, A2 = 2.0f * cxy + C2;
                const float B1 = mux[o] * mux[o] + muy[o] * muy[o] + C1, B2 = s2x + s2y + C2;
                const float iB = 1.0f / (B1 * B2);
                const float S = A1 * A2 * iB;                                    // loss.jl:53-56
                // partials at fixed (s_xx, s_xy): A1_mu = 2 mu_y, A2_mu = -2 mu_y, B1_mu = 2 mu_x, B2_mu = -2 mu_x
                const float dmu = (2.0f * muy[o] * A2 - 2.0f * muy[o] * A1) * iB - S * (2.0f * mux[o] / B1 - 2.0f * mux[o] / B2);
                const size_t q = cur.c * plane + (size_t)py * a.W + px;
                a.g_mu[q] = dmu;
                a.g_xx[q] = -S / B2;
                a.g_xy[q] = 2.0f * A1 * iB;
                l1 += fabsf(xc[o] - yc[o]);
                St += S;
            }
        }
    }
    const float t1 = block_sum_256(l1, sm);
    const float t2 = block_sum_256(St, sm2);
    // 64 slots, 64 bytes apart (summed by the host): 6120 workgroups x 2 atomics on ONE pair of words drain one after the other
    if (threadIdx.x == 0) {
        double *slot = a.acc + (size_t)(blockIdx.x % GS_LOSS_SLOTS) * GS_LOSS_SLOT_STRIDE;   // (one pair of atomics per workgroup)
        atomicAdd(&slot[0], (double)t1); atomicAdd(&slot[1], (double)t2);
    }
}

// NO outputs per thread (consecutive pixels of one row): 64 / NO x 16 threads per 64 x 16 tile.  A thread's NO + 10 window columns come
// out of (NO + 12) / 4 16-byte LDS reads per row: 6 floats read per output and plane row at NO = 4, 2.5 at NO = 8 -- measured equal
// (68 vs 74 us at 1920x1080x3): what NO = 8 saves in LDS traffic it loses in occupancy (141 VGPRs, three waves per SIMD against five).
#ifndef LOSS_GRAD_MINW
#define LOSS_GRAD_MINW 3                 // waves per SIMD the kernel is built for at least (NO = 4 needs 92 VGPRs: five)
#endif
#ifndef LOSS_GRAD_NO
#define LOSS_GRAD_NO 4
#endif
template <int NO>
__global__ __launch_bounds__((LTX / NO) * LTY, LOSS_GRAD_MINW) void ssim_grad_kernel(GsLossArgs a) {
    constexpr int NT = (LTX / NO) * LTY, NF = NO + 12;
    __shared__ __attribute__((aligned(16))) float s0[LHY][LPITCH], s1[LHY][LPITCH], s2[LHY][LPITCH];
    __shared__ __attribute__((aligned(16))) float swin[LNF][8];
    stage_window(swin, a);
    const int tx = threadIdx.x % (LTX / NO), ty = threadIdx.x / (LTX / NO);
    const size_t plane = (size_t)a.W * a.H;
    LossTile cur;
    if (!loss_tile(a, cur)) return;
    const size_t cb = cur.c * plane;
    const int py = cur.y0 + ty;
    float xq[NO], yq[NO];                                                     // the outputs' own samples: fetched with the halos, used last
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const uint32_t q = (uint32_t)(min(py, a.H - 1) * a.W + min(cur.x0 + NO * tx + o, a.W - 1));
        xq[o] = (a.img + cb)[q]; yq[o] = (a.gt + cb)[q];
    }
    {
        float h0[HaloIt<NT>::n], h1[HaloIt<NT>::n], h2[HaloIt<NT>::n];
        halo_fetch<NT>(h0, a.g_mu + cb, a.W, a.H, cur.x0, cur.y0);            // no ssim term outside the image
        halo_fetch<NT>(h1, a.g_xx + cb, a.W, a.H, cur.x0, cur.y0);
        halo_fetch<NT>(h2, a.g_xy + cb, a.W, a.H, cur.x0, cur.y0);
        halo_store<NT>(s0, h0); halo_store<NT>(s1, h1); halo_store<NT>(s2, h2);
    }
    __syncthreads();
    {
        float c0[NO], c1[NO], c2[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) c0[o] = c1[o] = c2[o] = 0.0f;
#pragma unroll 1
        for (int j = 0; j < LNF; ++j) {   // symmetric window: k(q-p) = k(p-q)
            float w[LNF];
            window_row(w, swin, j);
            float u[NF], v[NF];
            load_row<NF>(u, s0, ty + j, NO * tx);
            if (j < LP) { load_row<NF>(v, s0, ty + 2 * LP - j, NO * tx);
#pragma unroll
                for (int i = 0; i < NO + 10; ++i) u[i] += v[i]; }
            fold_cols<NO, NF>(c0, u, w);
            load_row<NF>(u, s1, ty + j, NO * tx);
            if (j < LP) { load_row<NF>(v, s1, ty + 2 * LP - j, NO * tx);
#pragma unroll
                for (int i = 0; i < NO + 10; ++i) u[i] += v[i]; }
            fold_cols<NO, NF>(c1, u, w);
            load_row<NF>(u, s2, ty + j, NO * tx);
            if (j < LP) { load_row<NF>(v, s2, ty + 2 * LP - j, NO * tx);
#pragma unroll
                for (int i = 0; i < NO + 10; ++i) u[i] += v[i]; }
            fold_cols<NO, NF>(c2, u, w);
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int px = cur.x0 + NO * tx + o;
            if (px < a.W && py < a.H) {
                const size_t q = cb + (size_t)py * a.W + px;
                const float x = xq[o], y = yq[o], d = x - y;
                const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
                a.dC[q] = a.w_l1 * sgn + a.w_ssim * (c0[o] + 2.0f * x * c1[o] + y * c2[o]);
            }
        }
    }
}

hipError_t gs_launch_loss(const GsLossArgs &a, hipStream_t s) {
    hipError_t e = hipMemsetAsync(a.acc, 0, sizeof(double) * GS_LOSS_SLOTS * GS_LOSS_SLOT_STRIDE, s);
    if (e != hipSuccess) return e;
    const int ntiles = ((a.W + LTX - 1) / LTX) * ((a.H + LTY - 1) / LTY) * a.C;
    const int grid = 8 * ((ntiles + 7) / 8);                                  // loss_tile: an eighth of the tiles per XCD
    hipLaunchKernelGGL(ssim_stats_kernel, dim3(grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(ssim_grad_kernel<LOSS_GRAD_NO>, dim3(grid), dim3((LTX / LOSS_GRAD_NO) * LTY), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------- SGD: param .-= lr * grad (train.jl:42-46)
__global__ void sgd_kernel(float *__restrict__ p, const float *__restrict__ g, float lr, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = fmaf(-lr, g[i], p[i]);
}
hipError_t gs_launch_sgd(float *p, const float *g, float lr, size_t n, hipStream_t s) {
    if (!p || !g || n == 0) return hipSuccess;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, lr, n);
    return hipGetLastError();
}

// exported builder used by gs_api.hip
hipError_t gs_loss_run(int W, int H, int C, const float *img, const float *gt, float *maps, double *acc, float *dC, float lam,
                       const float *win121, hipStream_t s) {
    GsLossArgs a{};
    a.W = W; a.H = H; a.C = C; a.img = img; a.gt = gt;
    const size_t n = (size_t)W * H * C;
    a.g_mu = maps; a.g_xx = maps + n; a.g_xy = maps + 2 * n;
    a.dC = dC; a.acc = acc;
    a.w_l1 = (1.0f - lam) / (2.0f * (float)n);
    a.w_ssim = -lam / (2.0f * (float)n);
    for (int i = 0; i < LW * LW; ++i) a.win[i] = win121[i];
    return gs_launch_loss(a, s);
}
