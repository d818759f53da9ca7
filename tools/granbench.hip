// granbench.hip -- does the flush granularity matter?  Every wave owns 16 "blocks" of 1 536 B (12 lines) and writes
// them round by round, like the witness kernel's per-round flush, with a spin between rounds; per round every block
// advances by G bytes (G = 128: one line per block and round, 8 blocks per store instruction ... G = 1 536: a whole
// block at once).  Same bytes, same region per wave, only the size of the contiguous piece written at one time differs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int BS = 1536;  // bytes per block

template <int G>
__global__ void __launch_bounds__(256) k_gran(uint8_t *out, uint64_t nblk, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t *base = out + blk0 * BS;
    constexpr int LPP = G / 16;          // lanes per piece
    constexpr int BPI = LPP >= 64 ? 1 : (64 / LPP > 16 ? 16 : 64 / LPP);  // blocks per store instruction
    constexpr int PASSES = 16 / BPI;     // instructions to cover the 16 blocks (G <= 1024)
    for (int r = 0; r < BS / G; ++r) {
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        if (G <= 1024) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int b = p * BPI + lane / LPP, sub = lane % LPP;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(base + (size_t)b * BS + r * G + sub * 16), "v"(v) : "memory");
            }
        } else {
            for (int b = 0; b < 16; ++b)
                for (int q = lane * 16; q < G; q += 1024)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(base + (size_t)b * BS + r * G + q), "v"(v) : "memory");
        }
    }
}

int main(int argc, char **argv) {
    const int spin = argc > 1 ? atoi(argv[1]) : 150;
    const uint64_t maxblk = 1ull << 21;
    uint8_t *buf[2];
    for (auto &b : buf) CK(hipMalloc(&b, maxblk * BS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("spin %d\n", spin);
    for (int lg : {17, 21}) {
        const uint64_t nblk = 1ull << lg;
        const size_t bytes = nblk * BS, slots = (maxblk * BS) / bytes;
        const unsigned grid = (unsigned)(nblk / 64);
        for (int g : {128, 256, 512, 1024, 1536}) {
            const int reps = lg > 18 ? 10 : 100;
            auto launch = [&](int i) {
                uint8_t *b = buf[(i / slots) & 1] + (i % slots) * bytes;
                switch (g) {
                    case 128: hipLaunchKernelGGL(k_gran<128>, dim3(grid), dim3(256), 0, 0, b, nblk, spin); break;
                    case 256: hipLaunchKernelGGL(k_gran<256>, dim3(grid), dim3(256), 0, 0, b, nblk, spin * 2); break;
                    case 512: hipLaunchKernelGGL(k_gran<512>, dim3(grid), dim3(256), 0, 0, b, nblk, spin * 4); break;
                    case 1024: hipLaunchKernelGGL(k_gran<1024>, dim3(grid), dim3(256), 0, 0, b, nblk, spin * 8); break;
                    default: hipLaunchKernelGGL(k_gran<1536>, dim3(grid), dim3(256), 0, 0, b, nblk, spin * 12); break;
                }
            };
            for (int i = 0; i < 3; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(i + 3);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipGetLastError());
            printf("2^%d blocks (%6.0f MB)  piece %5d B  %8.2f us  %7.1f GB/s\n", lg, bytes / 1e6, g, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e9);
        }
    }
    return 0;
}
