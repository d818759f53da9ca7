// burstbench.hip -- store-only emulation of a column-burst flush: a workgroup of W waves owns 16*W blocks; for each of
// the three packed columns in turn it "computes" (spin) and then writes that column's whole region (16*W*stride bytes)
// contiguously in one burst.  LDS is requested so that residency matches what staging the largest column would allow.
// Compare with sizebench (per-round slices of per-wave regions) and fillbench (linear fill).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int XS = 1360, YS = 1056, ZS = 608;

__global__ void __launch_bounds__(256) k_burst(uint8_t *x, uint8_t *y, uint8_t *z, uint64_t nblk, int spin) {
    extern __shared__ uint8_t lds[];
    const int bpg = (blockDim.x >> 6) * 16;
    const uint64_t blk0 = (uint64_t)blockIdx.x * bpg;
    if (blk0 >= nblk) return;
    const uint64_t nb = nblk - blk0 < (uint64_t)bpg ? nblk - blk0 : bpg;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    uint8_t *cols[3] = {x + blk0 * XS, y + blk0 * YS, z + blk0 * ZS};
    const size_t len[3] = {nb * XS, nb * YS, nb * ZS};
    for (int c = 0; c < 3; ++c) {
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        if (spin < 0) lds[threadIdx.x] = (uint8_t)v.x;
        for (size_t p = (size_t)threadIdx.x * 16; p < len[c]; p += (size_t)blockDim.x * 16)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(cols[c] + p), "v"(v) : "memory");
    }
}

int main(int argc, char **argv) {
    const uint64_t maxblk = 1ull << 20;
    const size_t per = maxblk * (XS + YS + ZS);
    uint8_t *buf[2];
    for (auto &b : buf) CK(hipMalloc(&b, per + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_burst), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%-6s %-6s %-8s %-6s %10s %10s\n", "log2n", "waves", "lds(KB)", "spin", "us/launch", "GB/s");
    for (int spin : {0, 400, 1200})
        for (int waves : {3, 4})
            for (int ldskb : {64, 80, 128})
                for (int lg : {16, 20}) {
                    if (waves == 4 && ldskb == 64) continue;
                    const uint64_t nblk = 1ull << lg;
                    const size_t bytes = nblk * (XS + YS + ZS), slots = per / bytes;
                    const unsigned grid = (unsigned)((nblk + 16 * waves - 1) / (16 * waves));
                    const int reps = lg >= 19 ? 10 : 100;
                    auto launch = [&](int i) {
                        const size_t s = (size_t)i % (slots * 2);
                        uint8_t *b = buf[s / slots] + (s % slots) * bytes;
                        hipLaunchKernelGGL(k_burst, dim3(grid), dim3(64 * waves), (size_t)ldskb * 1024, 0, b, b + nblk * XS, b + nblk * (XS + YS), nblk, spin);
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
                    printf("%-6d %-6d %-8d %-6d %10.2f %10.1f\n", lg, waves, ldskb, spin, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e9);
                }
    return 0;
}
