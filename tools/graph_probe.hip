// Does a hipGraph shorten a chain of short DEPENDENT kernels on this runtime?  (measure, do not model.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/graph_probe tools/graph_probe.hip && ./tools/graph_probe
// A small frame (C1: 10 k gaussians) is a chain of ~19 kernels of 2-10 us each on one stream.  The host enqueues far ahead of the GPU
// (no synchronisation inside a frame), so what separates two kernels is the GPU-side kernel boundary, not the host's launch call.
// This probe times chains of 19 kernels -- empty ones, and ones that spin ~4 us -- launched (a) eagerly, 200 chains queued back to
// back, (b) as one captured graph per chain, 200 graph launches back to back; events around the 200 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
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
    const int CHAIN = 19, REPS = 200;
    for (unsigned long long ticks : {0ull, 400ull}) {                    // 0: empty kernels; 400 ticks of 10 ns = 4 us each
        for (int blocks : {1, 256}) {
            auto chain = [&]() { for (int k = 0; k < CHAIN; ++k) hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s, ticks, sink); };
            for (int w = 0; w < 20; ++w) chain();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < REPS; ++r) chain();
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float eager = 0; CK(hipEventElapsedTime(&eager, e0, e1));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float graph = 0; CK(hipEventElapsedTime(&graph, e0, e1));
            printf("{\"kernel_us\": %.1f, \"blocks\": %d, \"chain\": %d, \"eager_us_per_chain\": %.2f, \"graph_us_per_chain\": %.2f, \"eager_us_per_kernel\": %.3f, \"graph_us_per_kernel\": %.3f}\n",
                   ticks * 0.01, blocks, CHAIN, eager * 1e3 / REPS, graph * 1e3 / REPS, eager * 1e3 / REPS / CHAIN, graph * 1e3 / REPS / CHAIN);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    }
    return 0;
}
