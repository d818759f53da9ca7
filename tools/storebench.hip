// storebench.hip -- which global-store patterns reach the HBM write roofline on
// MI355X?  Diagnostic only (not part of the product): decides how the witness
// kernels should stream their slabs out.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int XS = 1360, YS = 1056, ZS = 608;

// (A) every wave streams whole contiguous 16-block column ranges (what a whole-slab flush does)
__global__ void k_whole(uint8_t* x, uint8_t* y, uint8_t* z, uint64_t nblk) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t* cols[3] = {x + blk0 * XS, y + blk0 * YS, z + blk0 * ZS};
    const int len[3] = {16 * XS, 16 * YS, 16 * ZS};
    for (int c = 0; c < 3; ++c)
        for (int p = lane * 16; p < len[c]; p += 64 * 16) *reinterpret_cast<u32x4*>(cols[c] + p) = v;
}

// (B) per-round runs: 9 segments, per block a run of LEN bytes at stride S (current kernel)
template <int LEN, int GS>
__device__ __forceinline__ void runs(uint8_t* g, int lane, u32x4 v) {
    constexpr int PPB = LEN / 16, TOTAL = 16 * PPB;
#pragma unroll
    for (int p0 = 0; p0 < TOTAL; p0 += 64) {
        const int p = p0 + lane, b = p / PPB, q = p - b * PPB;
        if (p < TOTAL) *reinterpret_cast<u32x4*>(g + (size_t)b * GS + q * 16) = v;
    }
}
__global__ void k_runs(uint8_t* x, uint8_t* y, uint8_t* z, uint64_t nblk, int spin) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t *gx = x + blk0 * XS, *gy = y + blk0 * YS, *gz = z + blk0 * ZS;
    runs<176, XS>(gx, lane, v); runs<128, YS>(gy, lane, v); runs<80, ZS>(gz, lane, v);
    for (int g = 1; g <= 7; ++g) {
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;  // emulate compute between flushes
        runs<144, XS>(gx + 32 + 144 * g, lane, v); runs<112, YS>(gy + 16 + 112 * g, lane, v); runs<64, ZS>(gz + 16 + 64 * g, lane, v);
    }
    for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
    runs<176, XS>(gx + 32 + 144 * 8, lane, v); runs<144, YS>(gy + 16 + 112 * 8, lane, v); runs<80, ZS>(gz + 16 + 64 * 8, lane, v);
}

// (C) whole 128-byte lines only, but scattered: after "round" R every block's completed
// prefix is flushed up to the last whole line (sliding-window design).
__global__ void k_lines(uint8_t* x, uint8_t* y, uint8_t* z, uint64_t nblk, int spin) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t* cols[3] = {x + blk0 * XS, y + blk0 * YS, z + blk0 * ZS};
    const int strides[3] = {XS, YS, ZS};
    const int head[3] = {32, 16, 16}, rnd[3] = {144, 112, 64}, tail[3] = {32, 32, 16};
    const int sub = lane & 7, grp = lane >> 3;  // 8 lanes = one 128-byte line, 8 lines per instruction
    for (int R = 1; R <= 9; ++R) {
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        for (int c = 0; c < 3; ++c) {
            const int e0 = R == 1 ? 0 : head[c] + rnd[c] * (R - 1);
            const int e1 = head[c] + rnd[c] * R + (R == 9 ? tail[c] : 0);
            // lines of the wave's region [0, 16*stride) whose bytes all lie in some block's [.., e1) and not before e0-line
            for (int b = 0; b < 16; ++b) {
                const int base = b * strides[c];
                int lo = (base + e0) & ~127;              // line holding the first new byte (completed now or earlier?)
                if (lo < base + e0 && R > 1) lo += 0;      // partial line from last round completes now: include it
                int hi = (R == 9) ? (base + e1 + 127) & ~127 : (base + e1) & ~127;  // last round: flush the partial tail too
                if (R == 9 && b < 15) hi = (base + e1) & ~127;  // tail line shared with next block: it flushes it
                if (b > 0 && lo < base) lo = lo;          // line shared with previous block is written by this block
                for (int l = lo + grp * 128; l < hi; l += 8 * 128) *reinterpret_cast<u32x4*>(cols[c] + l + sub * 16) = v;
            }
        }
    }
}


// (E) runs / lines with S rounds per flush.  Segment boundaries: [0, head+S*rnd), ..., last includes tail.
template <bool LINES>
__global__ void k_seg(uint8_t* x, uint8_t* y, uint8_t* z, uint64_t nblk, int S) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t* cols[3] = {x + blk0 * XS, y + blk0 * YS, z + blk0 * ZS};
    const int strides[3] = {XS, YS, ZS};
    const int head[3] = {32, 16, 16}, rnd[3] = {144, 112, 64};
    for (int R0 = 0; R0 < 9; R0 += S) {
        const int R1 = R0 + S >= 9 ? 9 : R0 + S;
        for (int c = 0; c < 3; ++c) {
            const int e0 = R0 == 0 ? 0 : head[c] + rnd[c] * R0;
            const int e1 = R1 == 9 ? strides[c] : head[c] + rnd[c] * R1;
            if (LINES) {
                const int sub = lane & 7, grp = lane >> 3;
                for (int b = 0; b < 16; ++b) {
                    const int base = b * strides[c];
                    const int lo = (base + e0) & ~127;
                    const int hi = (R1 == 9 && b == 15) ? (base + e1 + 127) & ~127 : (base + e1) & ~127;
                    for (int l = lo + grp * 128; l < hi; l += 8 * 128) *reinterpret_cast<u32x4*>(cols[c] + l + sub * 16) = v;
                }
            } else {
                const int ppb = (e1 - e0) / 16, total = 16 * ppb;
                for (int p = lane; p < total; p += 64) {
                    const int b = p / ppb, q = p - b * ppb;
                    *reinterpret_cast<u32x4*>(cols[c] + (size_t)b * strides[c] + e0 + q * 16) = v;
                }
            }
        }
    }
}


// (F) the candidate real pattern: flush after every round, whole 128-byte lines only,
// lane = (block = lane>>3 (+8 second half), 16-byte piece = lane&7): 8 lines per store instruction.
__global__ void k_lines8(uint8_t* x, uint8_t* y, uint8_t* z, uint64_t nblk, int S, int spin) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t* cols[3] = {x + blk0 * XS, y + blk0 * YS, z + blk0 * ZS};
    const int strides[3] = {XS, YS, ZS};
    const int head[3] = {32, 16, 16}, rnd[3] = {144, 112, 64};
    const int sub = lane & 7;
    for (int R0 = 0; R0 < 9; R0 += S) {
        const int R1 = R0 + S >= 9 ? 9 : R0 + S;
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        for (int c = 0; c < 3; ++c) {
            const int e0 = R0 == 0 ? 0 : head[c] + rnd[c] * R0;
            const int e1 = R1 == 9 ? strides[c] : head[c] + rnd[c] * R1;
            for (int h = 0; h < 2; ++h) {
                const int b = (lane >> 3) + 8 * h;
                const int base = b * strides[c];
                const int lo = R0 == 0 ? (base + 127) >> 7 : (base + e0) >> 7;
                const int hi = R1 == 9 ? (base + strides[c] + 127) >> 7 : (base + e1) >> 7;
                for (int k = lo; k < hi; ++k) *reinterpret_cast<u32x4*>(cols[c] + 128 * k + sub * 16) = v;
            }
        }
    }
}

// (D) plain contiguous stream, grid-stride (upper bound)
__global__ void k_stream(u32x4* out, size_t n16) {
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}

int main(int argc, char** argv) {
    const uint64_t nblk = 1ull << (argc > 1 ? atoi(argv[1]) : 20);
    const bool quick = argc > 2;
    uint8_t *x, *y, *z;
    CK(hipMalloc(&x, nblk * XS + 4096)); CK(hipMalloc(&y, nblk * YS + 4096)); CK(hipMalloc(&z, nblk * ZS + 4096));
    const double bytes = (double)nblk * (XS + YS + ZS);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double nbytes) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 40;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGetLastError());
        printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms / reps, nbytes / (ms / reps * 1e-3) / 1e9);
    };
    timeit("stream contiguous (2048 WG x 256)", [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (u32x4*)x, (size_t)(nblk * XS / 16)); }, (double)nblk * XS);
    for (int waves : {1, 4}) {
        for (int ldskb : {0, 16, 32, 48, 64}) {
            char name[128];
            const unsigned grid = (unsigned)((nblk + 16 * waves - 1) / (16 * waves));
            snprintf(name, sizeof name, "whole  waves/WG=%d lds=%dKB", waves, ldskb);
            timeit(name, [&] { hipLaunchKernelGGL(k_whole, dim3(grid), dim3(64 * waves), ldskb * 1024, 0, x, y, z, nblk); }, bytes);
        }
    }
    for (int spin : {0, 200}) {
        if (quick) break;
        for (int waves : {1, 4}) {
            for (int ldskb : {0, 8, 16, 32}) {
                char name[128];
                const unsigned grid = (unsigned)((nblk + 16 * waves - 1) / (16 * waves));
                snprintf(name, sizeof name, "runs   waves/WG=%d lds=%dKB spin=%d", waves, ldskb, spin);
                timeit(name, [&] { hipLaunchKernelGGL(k_runs, dim3(grid), dim3(64 * waves), ldskb * 1024, 0, x, y, z, nblk, spin); }, bytes);
                snprintf(name, sizeof name, "lines  waves/WG=%d lds=%dKB spin=%d", waves, ldskb, spin);
                timeit(name, [&] { hipLaunchKernelGGL(k_lines, dim3(grid), dim3(64 * waves), ldskb * 1024, 0, x, y, z, nblk, spin); }, bytes);
            }
        }
    }
    for (int S : {1, 2, 3, 5, 9}) {
        if (quick) break;
        for (int waves : {1, 4}) {
            char name[128];
            const unsigned grid = (unsigned)((nblk + 16 * waves - 1) / (16 * waves));
            snprintf(name, sizeof name, "seg-runs  S=%d waves/WG=%d", S, waves);
            timeit(name, [&] { hipLaunchKernelGGL(k_seg<false>, dim3(grid), dim3(64 * waves), 0, 0, x, y, z, nblk, S); }, bytes);
            snprintf(name, sizeof name, "seg-lines S=%d waves/WG=%d", S, waves);
            timeit(name, [&] { hipLaunchKernelGGL(k_seg<true>, dim3(grid), dim3(64 * waves), 0, 0, x, y, z, nblk, S); }, bytes);
        }
    }
    for (int spin : {0, 100}) for (int S : {1, 2, 3}) {
        for (int waves : {1, 4}) for (int ldskb : {0, 24, 48}) {
            char name[128];
            const unsigned grid = (unsigned)((nblk + 16 * waves - 1) / (16 * waves));
            snprintf(name, sizeof name, "lines8 S=%d waves/WG=%d lds=%dKB spin=%d", S, waves, ldskb, spin);
            timeit(name, [&] { hipLaunchKernelGGL(k_lines8, dim3(grid), dim3(64 * waves), ldskb * 1024, 0, x, y, z, nblk, S, spin); }, bytes);
        }
    }
    return 0;
}
