// sizebench.hip -- fixed cost of a launch that only streams stores: the witness kernel's store volume and
// unit geometry (a wave owns 16 consecutive blocks of three packed columns) without any compute, at
// 2^14..2^20 blocks per launch, launches back to back over a ring of buffers larger than the Infinity
// Cache.  Separates what the memory system charges a short launch (ramp + drain) from what the
// kernel's own per-wave latency adds.  Diagnostic only.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int XS = 1360, YS = 1056, ZS = 608;

template <int SC1>  // 0 plain, 1 sc1 (write-through), 2 nontemporal
__device__ __forceinline__ void st(u32x4 *p, u32x4 v) {
    if (SC1 == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else if (SC1 == 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// every wave streams its 16-block range of each column, ROUNDS slices per column in turn (like the per-round flush)
template <int SC1>
__global__ void __launch_bounds__(256) k_units(uint8_t *x, uint8_t *y, uint8_t *z, uint64_t nblk, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t *cols[3] = {x + blk0 * XS, y + blk0 * YS, z + blk0 * ZS};
    const int len[3] = {16 * XS, 16 * YS, 16 * ZS};
    for (int r = 0; r < 10; ++r) {
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        for (int c = 0; c < 3; ++c) {
            const int lo = len[c] / 10 * r / 1024 * 1024, hi = r == 9 ? len[c] : len[c] / 10 * (r + 1) / 1024 * 1024;
            for (int p = lo + lane * 16; p < hi; p += 64 * 16) st<SC1>(reinterpret_cast<u32x4 *>(cols[c] + p), v);
        }
    }
}

int main(int argc, char **argv) {
    const int spin = argc > 1 ? atoi(argv[1]) : 0;
    const uint64_t maxblk = 1ull << 20;
    const size_t per = maxblk * (XS + YS + ZS);
    const int ring = 2;  // 2 x 3.2 GB
    uint8_t *buf[ring];
    for (int i = 0; i < ring; ++i) CK(hipMalloc(&buf[i], per + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("spin=%d\n%-8s %-6s %-6s %10s %10s\n", spin, "log2n", "waves", "sc1", "us/launch", "GB/s");
    for (int sc1 = 0; sc1 < 3; ++sc1)
        for (int waves : {4, 1})
            for (int lg = 14; lg <= 20; ++lg) {
                const uint64_t nblk = 1ull << lg;
                const size_t bytes = nblk * (XS + YS + ZS);
                const size_t slots = per / bytes;  // launches rotate through distinct regions of the ring
                const unsigned grid = (unsigned)((nblk + 16 * waves - 1) / (16 * waves));
                const int reps = lg >= 19 ? 20 : 200;
                auto launch = [&](int i) {
                    const size_t s = (size_t)i % (slots * ring);
                    uint8_t *b = buf[s / slots] + (s % slots) * bytes;
                    if (sc1 == 1) hipLaunchKernelGGL(k_units<1>, dim3(grid), dim3(64 * waves), 0, 0, b, b + nblk * XS, b + nblk * (XS + YS), nblk, spin);
                    else if (sc1 == 2) hipLaunchKernelGGL(k_units<2>, dim3(grid), dim3(64 * waves), 0, 0, b, b + nblk * XS, b + nblk * (XS + YS), nblk, spin);
                    else hipLaunchKernelGGL(k_units<0>, dim3(grid), dim3(64 * waves), 0, 0, b, b + nblk * XS, b + nblk * (XS + YS), nblk, spin);
                };
                for (int i = 0; i < 5; ++i) launch(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) launch(i + 5);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipGetLastError());
                printf("%-8d %-6d %-6d %10.2f %10.1f\n", lg, waves, sc1, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e9);
            }
    return 0;
}
