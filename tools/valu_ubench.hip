// Micro-benchmark: VALU issue rate on gfx950 for the instruction kinds the composite kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;
template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, sv = {s, s};
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) {      // 8 independent v_fma_f32
            a0 = fmaf(a0, s, s); a1 = fmaf(a1, s, s); a2 = fmaf(a2, s, s); a3 = fmaf(a3, s, s);
            a4 = fmaf(a4, s, s); a5 = fmaf(a5, s, s); a6 = fmaf(a6, s, s); a7 = fmaf(a7, s, s);
        } else if (KIND == 1) {   // 4 packed fma = 8 flops-lanes
            p0 = __builtin_elementwise_fma(p0, sv, sv); p1 = __builtin_elementwise_fma(p1, sv, sv);
            p2 = __builtin_elementwise_fma(p2, sv, sv); p3 = __builtin_elementwise_fma(p3, sv, sv);
        } else if (KIND == 2) {   // 8 exp2
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (KIND == 3) {   // 8 cmp+cndmask pairs
            a0 = a0 < s ? a1 : a0; a1 = a1 < s ? a2 : a1; a2 = a2 < s ? a3 : a2; a3 = a3 < s ? a4 : a3;
            a4 = a4 < s ? a5 : a4; a5 = a5 < s ? a6 : a5; a6 = a6 < s ? a7 : a6; a7 = a7 < s ? a0 : a7;
        } else if (KIND == 4) {   // 8 dpp adds (row_shr:1)
#define D(x) x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xF, 0xF, true))
            D(a0); D(a1); D(a2); D(a3); D(a4); D(a5); D(a6); D(a7);
        } else if (KIND == 5) {   // 8 mul
            a0 *= s; a1 *= s; a2 *= s; a3 *= s; a4 *= s; a5 *= s; a6 *= s; a7 *= s;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND> int run(const char *name, int per_iter, float *d) {
    const int blocks = 256 * 4 * 8;   // 8 waves per SIMD
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.999f);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.999f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double inst_per_simd = 8.0 * ITER * per_iter;     // waves/SIMD * iters * instr
    printf("%-14s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    return 0;
}
int main() {
    float *d; CHK(hipMalloc(&d, 256 * 4 * 8 * 64 * 4));
    run<0>("v_fma_f32", 8, d); run<1>("v_pk_fma_f32", 4, d); run<2>("v_exp_f32", 8, d); run<3>("cmp+cndmask", 16, d);
    run<4>("v_add_dpp", 8, d); run<5>("v_mul_f32", 8, d);
    return 0;
}
