// Does ds_add_rtn_u32 hand out pre-values in ascending lane order when several lanes of ONE wave
// instruction hit the same LDS address?  (Needed for a one-instruction stable radix rank.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(const unsigned *dig, unsigned *bad, int rounds) {
    __shared__ unsigned cnt[256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) cnt[i] = 0;
        __syncthreads();
        if (w == 0) {
            const unsigned d = dig[(blockIdx.x * rounds + r) * 64 + lane];
            const bool active = (d >> 8) == 0;                 // some lanes inactive
            unsigned got = 0xFFFFFFFFu;
            if (active) got = atomicAdd(&cnt[d & 255], 1u);
            // expected: number of ACTIVE lower lanes with the same digit
            unsigned long long peers = __ballot(active);
            for (int b = 0; b < 8; ++b) { unsigned long long bal = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? bal : ~bal; }
            const unsigned want = __popcll(peers & ((1ull << lane) - 1ull));
            if (active && got != want) ++nbad;
        }
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    const int blocks = 2048, rounds = 64;
    std::vector<unsigned> h((size_t)blocks * rounds * 64);
    unsigned s = 12345;
    for (size_t i = 0; i < h.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned mode = (i / 64) % 4;
        unsigned d = mode == 0 ? (s >> 24) : mode == 1 ? ((s >> 24) & 7) : mode == 2 ? 5u : ((s >> 24) & 31);
        if (((s >> 8) & 15) == 0) d |= 0x100;                 // ~6% inactive lanes
        h[i] = d;
    }
    unsigned *dd, *db; CHK(hipMalloc(&dd, h.size() * 4)); CHK(hipMalloc(&db, 4));
    CHK(hipMemcpy(dd, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemset(db, 0, 4));
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dd, db, rounds);
    CHK(hipDeviceSynchronize());
    unsigned bad; CHK(hipMemcpy(&bad, db, 4, hipMemcpyDeviceToHost));
    printf("mismatches: %u of %zu lane-ops (x20 launches)\n", bad, h.size());
    return 0;
}
