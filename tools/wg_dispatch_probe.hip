// How fast does the chip START workgroups?  (GPU box)
//   hipcc --offload-arch=gfx950 -O2 -o tools/wg_dispatch_probe tools/wg_dispatch_probe.hip && ./tools/wg_dispatch_probe
// A small frame's composite launch is a few thousand one-wave workgroups that each live ~10 us (C1 backward: 2048 workgroups, waves alive 14 us
// on average, kernel 31 us).  Every wave of this probe spins `us` microseconds; the same number of waves is launched as workgroups of
// 1, 2, 4, 8 and 16 waves.  kernel time - us = what it costs to get them all started (and retired).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (ticks == 12345678ull) *sink = 1;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REPS = 50;
    for (int waves : {1024, 2048, 4096}) {
        for (unsigned long long ticks : {0ull, 1000ull}) {              // 0: empty waves; 1000 ticks of 10 ns = 10 us each
            for (int per : {1, 2, 4, 8, 16}) {
                const int wgs = waves / per;
                for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64 * per), 0, s, ticks, sink);
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64 * per), 0, s, ticks, sink);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("{\"waves\": %d, \"waves_per_workgroup\": %d, \"workgroups\": %d, \"spin_us\": %.1f, \"us_per_launch\": %.2f}\n", waves, per, wgs, ticks * 0.01, ms * 1e3 / REPS);
            }
        }
    }
    return 0;
}
