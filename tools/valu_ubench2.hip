// Micro-benchmark 2: cross-lane and mask instruction costs on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, float s, unsigned long long msk) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) { u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[(i + 1) & 7]), false, false); a[i] = __uint_as_float(r.x) + __uint_as_float(r.y); }
            else if (KIND == 1) { u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[(i + 1) & 7]), false, false); a[i] = __uint_as_float(r.x) + __uint_as_float(r.y); }
            else if (KIND == 2) { a[i] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[i]), 0x041F)); }       // xor 1 (bitmask mode)
            else if (KIND == 3) { a[i] += __shfl_xor(a[i], 32); }
            else if (KIND == 4) { a[i] = __builtin_amdgcn_fmed3f(a[i], s, 1.0f); }
            else if (KIND == 5) { a[i] = fmaxf(a[i], s); }
            else if (KIND == 6) { a[i] = (msk >> (threadIdx.x & 63)) & 1 ? a[(i + 1) & 7] : a[i]; }                         // cndmask with loop-invariant mask
            else if (KIND == 7) { a[i] = __builtin_amdgcn_rcpf(a[i]); }
            else if (KIND == 8) { a[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0xB1, 0xF, 0xF, true)); }  // quad_perm [1,0,3,2]
            else if (KIND == 9) { a[i] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[(i + 1) & 7]), 0x111, 0xF, 0xF, true)); } // dpp src independent
            else if (KIND == 10) { a[i] = a[i] + fabsf(a[(i + 1) & 7] - s); }   // sub + add|abs|
            else if (KIND == 11) { a[i] = __builtin_amdgcn_readlane(__float_as_int(a[i]), 63) * 0.5f + a[i]; }
        }
    }
    float t = 0; for (int i = 0; i < 8; ++i) t += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = t;
}
template <int KIND> int run(const char *name, float *d) {
    const int blocks = 256 * 4 * 8;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0x5555555555555555ull);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0x5555555555555555ull);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    double groups = 8.0 * ITER * 8;
    printf("%-28s %.3f ms -> %.2f cyc@2.4GHz per group\n", name, ms, ms * 1e6 / groups * 2.4);
    return 0;
}
int main() {
    float *d; CHK(hipMalloc(&d, 256 * 4 * 8 * 64 * 4));
    run<0>("permlane32_swap + add", d); run<1>("permlane16_swap + add", d); run<2>("ds_swizzle + add", d); run<3>("shfl_xor32 (bpermute) + add", d);
    run<4>("v_med3_f32", d); run<5>("v_max_f32", d); run<6>("cndmask (invariant mask)", d); run<7>("v_rcp_f32", d);
    run<8>("add dpp quad_perm", d); run<9>("add + mov_dpp(other reg)", d); run<10>("sub + add|abs|", d); run<11>("readlane + fma", d);
    return 0;
}
