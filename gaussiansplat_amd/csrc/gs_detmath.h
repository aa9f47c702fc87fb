// gs_detmath.h -- deterministic fp32 helpers of the numeric spec.
//
// The preprocess path decides tile ids and depth keys, which must be bit-identical to the
// CPU oracle.  Everything here is built only from individually rounded IEEE fp32/fp64
// operations (the translation unit is compiled with -ffp-contract=off), so gfx950 and any
// IEEE host produce the same bits.  exp stands in for Julia/libdevice exp
// (reference src/splat.jl:176, src/projection.jl:133-135).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

__device__ __forceinline__ float gs_u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t gs_f2u(float f) { return __float_as_uint(f); }

// Cody-Waite range reduction + Cephes expf polynomial; results below FLT_MIN flush to 0.
__device__ __forceinline__ float gs_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return __builtin_huge_valf();
    if (x < -87.33654f) return 0.0f;
    float n = __builtin_rintf(x * 1.44269504f);
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
    float s1 = gs_u2f((uint32_t)(n1 + 127) << 23);
    float s2 = gs_u2f((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

// Julia max/min propagate NaN.
__device__ __forceinline__ double gs_jlmax(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }
__device__ __forceinline__ double gs_jlmin(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }

// The spec's sin/cos (2-D renderer, reference src/cov2d.jl:6-8 CUDA.cos / CUDA.sin): k = rint(x*2/pi), three-term
// Cody-Waite reduction by pi/2, Cephes sinf/cosf polynomials on [-pi/4, pi/4], quadrant fix-up; fp32 mul/add only
// (the CPU oracle and its NumPy twin use the same sequence of operations).
__device__ __forceinline__ void gs_sincosf(float x, float &sn, float &cs) {
    if (!(x - x == 0.0f)) { sn = __builtin_nanf(""); cs = sn; return; }      // NaN or Inf
    const float kf = __builtin_rintf(x * 0.636619772f);
    float r = x - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    const float z = r * r;
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
    const long long k = (long long)kf;
    switch ((int)(k & 3)) {
        case 0: sn = ps;  cs = pc;  break;
        case 1: sn = pc;  cs = -ps; break;
        case 2: sn = -ps; cs = -pc; break;
        default: sn = -pc; cs = ps; break;
    }
}
