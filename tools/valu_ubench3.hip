// Micro-benchmark 3: issue cost of the exact VALU instruction FORMS in the composite inner loops on gfx950
// (operand kinds matter: number of distinct VGPR sources, SGPR/abs modifiers, VOP2 vs VOP3, packed forms).
// 8 waves per SIMD, 8 independent chains per wave; cycles are at the clock measured in the same process.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(M) M(a0, b0, c0) M(a1, b1, c1) M(a2, b2, c2) M(a3, b3, c3) M(a4, b0, c1) M(a5, b1, c2) M(a6, b2, c3) M(a7, b3, c0)
#define FMA3(a, b, c) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define FMA3D(a, b, c) asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(a) : "v"(b), "v"(c), "v"(d0));
#define FMA2S(a, b, c) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "s"(s));
#define FMAABS(a, b, c) asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(a) : "v"(b), "s"(s));
#define FMAC(a, b, c) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define MUL2(a, b, c) asm volatile("v_mul_f32_e32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define MULSELF(a, b, c) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a) : "v"(b));
#define SUB2(a, b, c) asm volatile("v_sub_f32_e32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define SUBSELF(a, b, c) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MED3(a, b, c) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define MAX3(a, b, c) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define MINMAX(a, b, c) asm volatile("v_max_f32_e32 %0, %0, %1\n v_min_f32_e32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));
#define EXP(a, b, c) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a));
#define ADDABS(a, b, c) asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(a) : "v"(b));
#define SUBABS(a, b, c) asm volatile("v_sub_f32_e64 %0, %0, |%1|" : "+v"(a) : "v"(b));
#define FMA_LIT(a, b, c) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "s"(s));
#define MULS(a, b, c) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a) : "s"(s));
#define SUB1(a, b, c) asm volatile("v_sub_f32_e32 %0, 1.0, %0" : "+v"(a));
#define FMA0(a, b, c) asm volatile("v_fma_f32 %0, %0, %1, 0" : "+v"(a) : "v"(b));
#define FMAONE(a, b, c) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a) : "v"(b));
#define MULHALF(a, b, c) asm volatile("v_mul_f32_e32 %0, 0.5, %0" : "+v"(a));
#define XORLIT(a, b, c) asm volatile("v_xor_b32_e32 %0, 0x80000000, %0" : "+v"(a));
#define MOVDPP(a, b, c) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));
#define ADDDPP(a, b, c) asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));
#define CNDMASK(a, b, c) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(msk));
#define RFL(a, b, c) { int t; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t) : "v"(a)); acc += t; }
#define MAD64(a, b, c) asm volatile("v_mad_u64_u32 %0, vcc, %1, 40, %0" : "+v"(w64) : "v"(a) : "vcc");
#define SWZADD(a, b, c) { float t; asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a)); asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a) : "v"(t)); }
#define PKREP4(M) M(p0, q0, r0) M(p1, q1, r1) M(p2, q0, r1) M(p3, q1, r0)
#define PKFMA(p, q, r) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q), "v"(r));
#define PKFMASEL(p, q, r) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "+v"(p) : "v"(q), "v"(r));
#define PKFMANEG(p, q, r) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(p) : "v"(q), "v"(r));
#define PKMUL(p, q, r) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(p) : "v"(q), "v"(r));
#define PKADD(p, q, r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float s, long long *clk) {
    const unsigned long long msk = 0x5555555555555555ull;
    int acc = 0; unsigned long long w64 = threadIdx.x;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 0.999f + 1e-6f * threadIdx.x, b1 = b0 * 1.0001f, b2 = b0 * 0.9999f, b3 = b0 * 1.0002f;
    float c0 = 1e-3f * threadIdx.x, c1 = c0 + 1e-3f, c2 = c0 + 2e-3f, c3 = c0 + 3e-3f, d0 = 1e-4f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q0 = {b0, b1}, q1 = {b2, b3}, r0 = {c0, c1}, r1 = {c2, c3};
    const long long t0 = wall_clock64();
    const long long k0 = clock64();
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) { REP8(FMA3) }
        else if (KIND == 1) { REP8(FMA2S) }
        else if (KIND == 2) { REP8(FMAC) }
        else if (KIND == 3) { REP8(MUL2) }
        else if (KIND == 4) { REP8(SUB2) }
        else if (KIND == 5) { REP8(FMAABS) }
        else if (KIND == 6) { REP8(MED3) }
        else if (KIND == 7) { PKREP4(PKFMA) PKREP4(PKFMA) }
        else if (KIND == 8) { PKREP4(PKMUL) PKREP4(PKMUL) }
        else if (KIND == 9) { REP8(EXP) }
        else if (KIND == 10) { REP8(ADDABS) }
        else if (KIND == 11) { PKREP4(PKFMASEL) PKREP4(PKFMASEL) }
        else if (KIND == 12) { PKREP4(PKFMANEG) PKREP4(PKFMANEG) }
        else if (KIND == 13) { REP8(FMA3D) }
        else if (KIND == 14) { REP8(MULSELF) }
        else if (KIND == 15) { REP8(SUBSELF) }
        else if (KIND == 16) { REP8(MINMAX) }
        else if (KIND == 17) { REP8(MAX3) }
        else if (KIND == 18) { REP8(SUBABS) }
        else if (KIND == 19) { PKREP4(PKADD) PKREP4(PKADD) }
        else if (KIND == 20) { REP8(MULS) }
        else if (KIND == 21) { REP8(SUB1) }
        else if (KIND == 22) { REP8(FMA0) }
        else if (KIND == 23) { REP8(FMAONE) }
        else if (KIND == 24) { REP8(MULHALF) }
        else if (KIND == 25) { REP8(XORLIT) }
        else if (KIND == 26) { REP8(MOVDPP) }
        else if (KIND == 27) { REP8(ADDDPP) }
        else if (KIND == 28) { REP8(CNDMASK) }
        else if (KIND == 29) { REP8(RFL) }
        else if (KIND == 30) { REP8(MAD64) }
    }
    const long long k1 = clock64();
    const long long t1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = k1 - k0; clk[1] = t1 - t0; }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + b0 + c0 + q0.x + r0.x + (float)acc + (float)w64;
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
    printf("%-34s %.3f ms  %.2f GHz -> %.2f cycles per wave-instr per SIMD\n", name, ms, ghz, ms * 1e6 / inst_per_simd * ghz);
    return 0;
}
int main() {
    float *d; long long *clk; CHK(hipMalloc(&d, 256 * 4 * 8 * 64 * 4)); CHK(hipMalloc(&clk, 16));
    run<0>("v_fma_f32 v,v,v,(acc)", 8, d, clk);
    run<13>("v_fma_f32 v,v,v,v (4 distinct)", 8, d, clk);
    run<1>("v_fma_f32 v,v,s,(acc)", 8, d, clk);
    run<5>("v_fma_f32 v,|v|,s,(acc)", 8, d, clk);
    run<2>("v_fmac_f32_e32 acc,v,v", 8, d, clk);
    run<3>("v_mul_f32_e32 d,v,v", 8, d, clk);
    run<14>("v_mul_f32_e32 d,d,v", 8, d, clk);
    run<4>("v_sub_f32_e32 d,v,v", 8, d, clk);
    run<15>("v_sub_f32_e32 d,d,v", 8, d, clk);
    run<10>("v_add_f32_e64 d,d,|v|", 8, d, clk);
    run<18>("v_sub_f32_e64 d,d,|v|", 8, d, clk);
    run<6>("v_med3_f32 d,d,v,v", 8, d, clk);
    run<17>("v_max3_f32 d,d,v,v", 8, d, clk);
    run<16>("v_max_f32+v_min_f32 (clamp)", 16, d, clk);
    run<9>("v_exp_f32", 8, d, clk);
    run<7>("v_pk_fma_f32 p,q,r,(acc)", 8, d, clk);
    run<11>("v_pk_fma_f32 op_sel bcast", 8, d, clk);
    run<12>("v_pk_fma_f32 neg (T - aT)", 8, d, clk);
    run<8>("v_pk_mul_f32", 8, d, clk);
    run<19>("v_pk_add_f32", 8, d, clk);
    run<20>("v_mul_f32_e32 d,s,d (SGPR, VOP2)", 8, d, clk);
    run<21>("v_sub_f32_e32 d,1.0,d", 8, d, clk);
    run<22>("v_fma_f32 d,d,v,0", 8, d, clk);
    run<23>("v_fma_f32 d,d,v,1.0", 8, d, clk);
    run<24>("v_mul_f32_e32 d,0.5,d", 8, d, clk);
    run<25>("v_xor_b32 d,0x80000000,d", 8, d, clk);
    run<26>("s_nop1 + v_mov_b32_dpp", 8, d, clk);
    run<27>("s_nop1 + v_add_f32_dpp", 8, d, clk);
    run<28>("v_cndmask_b32_e64 (SGPR mask)", 8, d, clk);
    run<29>("v_readfirstlane_b32", 8, d, clk);
    run<30>("v_mad_u64_u32", 8, d, clk);
    return 0;
}
