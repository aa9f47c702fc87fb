// Micro-benchmark 4: the clamp output modifier against v_med3_f32 (pixel-box penalty forms of the composite loops) on gfx950.
// Same harness as valu_ubench3.hip: 8 waves per SIMD, 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;
#define REP8(M) M(a0, b0, c0) M(a1, b1, c1) M(a2, b2, c2) M(a3, b3, c3) M(a4, b0, c1) M(a5, b1, c2) M(a6, b2, c3) M(a7, b3, c0)
#define MED3(a, b, c) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define ADDCLAMP(a, b, c) asm volatile("v_add_f32_e64 %0, |%0|, -1.0 clamp" : "+v"(a));
#define ADDCLAMPV(a, b, c) asm volatile("v_add_f32_e64 %0, |%0|, %1 clamp" : "+v"(a) : "v"(b));
#define FMACLAMP(a, b, c) asm volatile("v_fma_f32 %0, |%0|, %1, -1.0 clamp" : "+v"(a) : "v"(b));
#define FMAPLAIN(a, b, c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define OLDPEN(a, b, c) { float t; asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "v"(c)); asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(t) : "v"(a)); asm volatile("v_fma_f32 %0, %1, |%2|, %0" : "+v"(a) : "v"(b), "v"(t)); }
#define NEWPEN(a, b, c) { float t; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "v"(c)); asm volatile("v_add_f32_e64 %0, |%0|, -1.0 clamp" : "+v"(t)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(t)); }
#define NEWPEN2(a, b, c) { float t; asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(c)); asm volatile("v_fma_f32 %0, |%0|, %1, -1.0 clamp" : "+v"(t) : "v"(b)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(t)); }

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float s, long long *clk) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 0.999f + 1e-6f * threadIdx.x, b1 = b0 * 1.0001f, b2 = b0 * 0.9999f, b3 = b0 * 1.0002f;
    float c0 = 1e-3f * threadIdx.x, c1 = c0 + 1e-3f, c2 = c0 + 2e-3f, c3 = c0 + 3e-3f;
    const long long t0 = wall_clock64();
    const long long k0 = clock64();
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) { REP8(MED3) }
        else if (KIND == 1) { REP8(ADDCLAMP) }
        else if (KIND == 2) { REP8(ADDCLAMPV) }
        else if (KIND == 3) { REP8(FMACLAMP) }
        else if (KIND == 4) { REP8(FMAPLAIN) }
        else if (KIND == 5) { REP8(OLDPEN) }
        else if (KIND == 6) { REP8(NEWPEN) }
        else if (KIND == 7) { REP8(NEWPEN2) }
    }
    const long long k1 = clock64();
    const long long t1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = k1 - k0; clk[1] = t1 - t0; }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + c0;
}
template <int KIND> int run(const char *name, int per_iter, float *d, long long *clk) {
    const int blocks = 256 * 4 * 8;   // 8 waves per SIMD
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, clk);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);          // wall_clock64 ticks at 100 MHz
    const double inst_per_simd = 8.0 * ITER * per_iter;
    printf("%-44s %.3f ms  %.2f GHz -> %.2f cycles per wave-instr per SIMD\n", name, ms, ghz, ms * 1e6 / inst_per_simd * ghz);
    return 0;
}
int main() {
    float *d; long long *clk; CHK(hipMalloc(&d, 256 * 4 * 8 * 64 * 4)); CHK(hipMalloc(&clk, 16));
    run<0>("v_med3_f32 d,d,v,v", 8, d, clk);
    run<1>("v_add_f32_e64 d,|d|,-1.0 clamp", 8, d, clk);
    run<2>("v_add_f32_e64 d,|d|,v clamp", 8, d, clk);
    run<3>("v_fma_f32 d,|d|,v,-1.0 clamp", 8, d, clk);
    run<4>("v_fma_f32 d,d,v,v", 8, d, clk);
    run<5>("old penalty: med3 + sub + fma|.| (3 instr)", 24, d, clk);
    run<6>("new penalty: fma + add|.|clamp + fma (3)", 24, d, clk);
    run<7>("new penalty 2: sub + fma|.|clamp + fma (3)", 24, d, clk);
    return 0;
}
