// unitbench.hip -- store-only emulation of candidate divisions of labour for the witness kernel (round 2 study).
// Question: the round-1 kernel fills per-wave regions (16 blocks x 3 columns) over ten rounds and runs at the
// "scattered order" rate.  What does a ONE-SHOT unit reach -- a workgroup that owns nb consecutive blocks, computes
// first (spin, no stores) and then bursts its nb*stride bytes of every column contiguously and exits, workgroups
// dispatched in block order -- as a function of nb, with LDS requested as a real staging of nb slabs would need?
// Compared in one process with (a) the round-sliced per-wave pattern of round 1 and (b) a linear 4 KB one-shot fill
// of the same bytes.  Streams: 3 (shared key: x, y, z packed) or 7 (per-block keys: + words, kx, ky, kz).
// Diagnostic only.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/unitbench tools/unitbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Streams {
    int flavour;  // 0 plain, 1 nontemporal, 2 sc1
    int xcd;  // 1: workgroups that share an XCD (blockIdx % 8) take one contiguous eighth of the groups
    int ns;
    uint8_t *base[7];
    int stride[7];
};

__device__ __forceinline__ uint32_t group_of(const Streams &s, uint32_t id, uint32_t ngroups) {
    if (!s.xcd) return id;
    const uint32_t q = ngroups / 8, r = ngroups % 8, x = id % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
}
__device__ int g_flavour_dev;  // unused; the flavour travels in Streams.flavour
__device__ __forceinline__ void st_f(uint8_t *p, u32x4 v, int flavour) {
    if (flavour == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else if (flavour == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = v;
}
#define st_sc1(p, v) st_f((p), (v), s.flavour)

// one-shot unit: spin, then burst every stream's nb*stride bytes
__global__ void __launch_bounds__(256) k_unit(Streams s, uint64_t nblk, int nb, int spin) {
    extern __shared__ uint8_t lds[];
    const uint64_t blk0 = (uint64_t)blockIdx.x * nb;
    if (blk0 >= nblk) return;
    const int n = nblk - blk0 < (uint64_t)nb ? (int)(nblk - blk0) : nb;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
    if (spin < 0) lds[threadIdx.x] = (uint8_t)v.x;
    for (int c = 0; c < s.ns; ++c) {
        uint8_t *g = s.base[c] + blk0 * s.stride[c];
        const int len = n * s.stride[c];
        for (int p = threadIdx.x * 16; p < len; p += blockDim.x * 16) st_sc1(g + p, v);
    }
}

// round-1 pattern: a wave owns 16 blocks; ten rounds, after each it stores the KB-aligned tenth of each column range
__global__ void __launch_bounds__(256) k_rounds(Streams s, uint64_t nblk, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)group_of(s, blockIdx.x, gridDim.x) * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    for (int c = 3; c < s.ns; ++c) {  // key slab: one contiguous flush per column before the rounds
        uint8_t *g = s.base[c] + blk0 * s.stride[c];
        for (int p = lane * 16; p < 16 * s.stride[c]; p += 64 * 16) st_sc1(g + p, v);
    }
    for (int r = 0; r < 10; ++r) {
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        for (int c = 0; c < 3; ++c) {
            uint8_t *g = s.base[c] + blk0 * s.stride[c];
            const int len = 16 * s.stride[c];
            const int lo = len / 10 * r / 1024 * 1024, hi = r == 9 ? len : len / 10 * (r + 1) / 1024 * 1024;
            for (int p = lo + lane * 16; p < hi; p += 64 * 16) st_sc1(g + p, v);
        }
    }
}

// column bursts: a wave owns 16 blocks; per column in turn (key columns first): spin (the column's records are
// computed into an LDS image), then the column's whole 16-block range leaves in one contiguous burst
__global__ void __launch_bounds__(256) k_colburst(Streams s, uint64_t nblk, int spin) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)group_of(s, blockIdx.x, gridDim.x) * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    for (int i = 0; i < 2 * spin; ++i) v.x = v.x * 1664525u + 1013904223u;  // phase A: the AES rounds themselves
    if (spin < 0) lds[threadIdx.x] = (uint8_t)v.x;
    const int order[7] = {3, 4, 5, 6, 0, 1, 2};
    for (int oi = 0; oi < 7; ++oi) {
        const int c = order[oi];
        if (c >= s.ns) continue;
        if (c < 3) for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        uint8_t *g = s.base[c] + blk0 * s.stride[c];
        for (int p = lane * 16; p < 16 * s.stride[c]; p += 64 * 16) st_sc1(g + p, v);
    }
}

// paced linear sweep: a wave owns 16 blocks and emits them block by block in address order (x, y, z of block 0, then
// block 1, ...) with a spin in front of every block: every column range is written once, linearly, over the wave's life
__global__ void __launch_bounds__(256) k_sweep(Streams s, uint64_t nblk, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint64_t blk0 = ((uint64_t)blockIdx.x * waves + wave) * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    for (int i = 0; i < 2 * spin; ++i) v.x = v.x * 1664525u + 1013904223u;
    for (int b = 0; b < 16; b += 2) {  // two blocks per step: 2 720 B of x = 2.7 store instructions
        for (int i = 0; i < spin / 4; ++i) v.x = v.x * 1664525u + 1013904223u;
        for (int c = 0; c < s.ns; ++c) {
            uint8_t *g = s.base[c] + (blk0 + b) * s.stride[c];
            for (int p = lane * 16; p < 2 * s.stride[c]; p += 64 * 16) st_sc1(g + p, v);
        }
    }
}

// one stream per workgroup: unit u = (chunk u / ns, stream u % ns) writes nb*stride bytes of ONE stream and exits, so the
// launch is ns interleaved linear fronts (mode 0), or stream after stream (mode 1: all chunks of x, then all of y, ...)
__global__ void __launch_bounds__(256) k_fronts(Streams s, uint64_t nblk, int nb, int mode) {
    const uint64_t nchunks = (nblk + nb - 1) / nb;
    const uint64_t u = blockIdx.x;
    const int c = mode ? (int)(u / nchunks) : (int)(u % s.ns);
    const uint64_t j = mode ? u % nchunks : u / s.ns;
    const uint64_t blk0 = j * nb;
    if (blk0 >= nblk || c >= s.ns) return;
    const int n = nblk - blk0 < (uint64_t)nb ? (int)(nblk - blk0) : nb;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    uint8_t *g = s.base[c] + blk0 * s.stride[c];
    const int len = n * s.stride[c];
    for (int p = threadIdx.x * 16; p < len; p += blockDim.x * 16) st_sc1(g + p, v);
}

// linear fill: workgroup i writes the 4 KB chunk i of one buffer
__global__ void __launch_bounds__(256) k_fill(uint8_t *out, size_t total, int flavour = 2) {
    const size_t p = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    if (p < total) st_f(out + p, v, flavour);
}

static int g_xcd = 0, g_flavour = 2;
int main(int argc, char **argv) {
    const int only_lg = argc > 1 ? atoi(argv[1]) : 0;
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");  // few variants, few launches: for rocprofv3 --pmc passes
    const uint64_t maxblk = 1ull << 20;
    const int S3[3] = {1360, 1056, 608}, S7[7] = {1360, 1056, 608, 96, 400, 240, 200};
    const size_t per = maxblk * 3960 + (1 << 20);
    uint8_t *buf[2];
    for (auto &b : buf) CK(hipMalloc(&b, per));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_unit), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_colburst), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto timeit = [&](int reps, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(i + 3);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / reps;
    };
    for (int ns : {3, 7})
        for (int lg : {16, 20}) {
            if (only_lg && lg != only_lg) continue;
            if (quick && ns != 7) continue;
            const uint64_t nblk = 1ull << lg;
            const int *S = ns == 3 ? S3 : S7;
            size_t bpb = 0;
            for (int c = 0; c < ns; ++c) bpb += S[c];
            const size_t bytes = nblk * bpb, slots = per / bytes;
            const int reps = quick ? 3 : lg >= 19 ? 10 : 100;
            auto streams = [&](int i) {
                const size_t sl = (size_t)i % (slots * 2);
                uint8_t *b = buf[sl / slots] + (sl % slots) * bytes;
                Streams s;
                s.flavour = g_flavour;
                s.xcd = g_xcd;
                s.ns = ns;
                for (int c = 0; c < 7; ++c) { s.base[c] = nullptr; s.stride[c] = 0; }
                for (int c = 0; c < ns; ++c) { s.base[c] = b; s.stride[c] = S[c]; b += nblk * S[c]; }
                return s;
            };
            const bool fronts_only = argc > 2 && (!strcmp(argv[2], "fronts") || !strcmp(argv[2], "rank") || !strcmp(argv[2], "flavour"));
            if (!fronts_only) {
                const double us = timeit(reps, [&](int i) {
                    const size_t sl = (size_t)i % (slots * 2);
                    hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, buf[sl / slots] + (sl % slots) * bytes, bytes);
                });
                printf("ns %d lg %2d  linear 4 KB fill                              %9.2f us %8.1f GB/s\n", ns, lg, us, bytes / us / 1e3);
            }
            for (int waves : {4, 3})
                for (int spin : {0, 120}) {
                    if (fronts_only) continue;
                    if (quick && (waves != 3 || spin)) continue;
                    const double us = timeit(reps, [&](int i) {
                        hipLaunchKernelGGL(k_rounds, dim3((unsigned)((nblk + 16 * waves - 1) / (16 * waves))), dim3(64 * waves), 0, 0, streams(i), nblk, spin);
                    });
                    printf("ns %d lg %2d  round-sliced, %d waves x 16 blocks, spin %4d      %9.2f us %8.1f GB/s\n", ns, lg, waves, spin, us, bytes / us / 1e3);
                }
            for (int rep = 0; rep < (quick ? 1 : 3); ++rep)
                for (int waves : {3, 1})
                    for (int spin : {0, 60, 120}) {
                        if (fronts_only) continue;
                        if (quick && (waves != 3 || spin != 60)) continue;
                        const size_t lds = (size_t)waves * 22 * 1024;
                        double us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_colburst, dim3((unsigned)((nblk + 16 * waves - 1) / (16 * waves))), dim3(64 * waves), lds, 0, streams(i), nblk, spin);
                        });
                        printf("ns %d lg %2d  column bursts, %d waves x 16 blocks, lds %6zu spin %4d  %9.2f us %8.1f GB/s\n", ns, lg, waves, lds, spin, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_sweep, dim3((unsigned)((nblk + 16 * waves - 1) / (16 * waves))), dim3(64 * waves), 0, 0, streams(i), nblk, spin);
                        });
                        printf("ns %d lg %2d  paced sweep,   %d waves x 16 blocks,            spin %4d  %9.2f us %8.1f GB/s\n", ns, lg, waves, spin, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_rounds, dim3((unsigned)((nblk + 16 * waves - 1) / (16 * waves))), dim3(64 * waves), 0, 0, streams(i), nblk, spin / 2);
                        });
                        printf("ns %d lg %2d  round-sliced,  %d waves x 16 blocks,            spin %4d  %9.2f us %8.1f GB/s\n", ns, lg, waves, spin / 2, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            const size_t sl = (size_t)i % (slots * 2);
                            hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, buf[sl / slots] + (sl % slots) * bytes, bytes);
                        });
                        printf("ns %d lg %2d  linear 4 KB fill                                          %9.2f us %8.1f GB/s\n", ns, lg, us, bytes / us / 1e3);
                        fflush(stdout);
                    }
            if (argc > 2 && !strcmp(argv[2], "flavour")) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rounds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                for (int rep = 0; rep < 2; ++rep)
                    for (int fl : {0, 1, 2}) {
                        g_flavour = fl;
                        g_xcd = 1;
                        const char *fn = fl == 0 ? "plain" : fl == 1 ? "nt   " : "sc1  ";
                        double us = timeit(reps, [&](int i) {
                            const size_t sl = (size_t)i % (slots * 2);
                            hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, buf[sl / slots] + (sl % slots) * bytes, bytes, fl);
                        });
                        printf("ns %d lg %2d  %s linear 4 KB fill                                   %9.2f us %8.1f GB/s\n", ns, lg, fn, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_rounds, dim3((unsigned)((nblk + 47) / 48)), dim3(192), (size_t)3 * 22 * 1024, 0, streams(i), nblk, 30);
                        });
                        printf("ns %d lg %2d  %s round-sliced, 3 waves x 16 blocks, lds 66 KB, spin 30  %9.2f us %8.1f GB/s\n", ns, lg, fn, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_colburst, dim3((unsigned)((nblk + 15) / 16)), dim3(64), (size_t)22 * 1024, 0, streams(i), nblk, 60);
                        });
                        printf("ns %d lg %2d  %s column bursts, 1 wave x 16 blocks, lds 22 KB, spin 60  %9.2f us %8.1f GB/s\n", ns, lg, fn, us, bytes / us / 1e3);
                        for (int nb : {4, 8, 16}) {
                            us = timeit(reps, [&](int i) {
                                hipLaunchKernelGGL(k_unit, dim3((unsigned)((nblk + nb - 1) / nb)), dim3(64), (size_t)nb * bpb, 0, streams(i), nblk, nb, 0);
                            });
                            printf("ns %d lg %2d  %s one-shot nb %2d, 64 threads, lds %6zu                 %9.2f us %8.1f GB/s\n", ns, lg, fn, nb, (size_t)nb * bpb, us, bytes / us / 1e3);
                        }
                        fflush(stdout);
                    }
                g_flavour = 2;
                g_xcd = 0;
                continue;
            }
            if (argc > 2 && !strcmp(argv[2], "rank")) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rounds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                for (int rep = 0; rep < 3; ++rep) {
                    for (int xcd : {0, 1}) {
                        g_xcd = xcd;
                        const size_t lds = 3 * 22 * 1024;  // the real kernel's residency: two 3-wave groups per CU
                        double us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_rounds, dim3((unsigned)((nblk + 47) / 48)), dim3(192), lds, 0, streams(i), nblk, 30);
                        });
                        printf("ns %d lg %2d  xcd %d  round-sliced, 3 waves x 16 blocks, lds 66 KB, spin 30     %9.2f us %8.1f GB/s\n", ns, lg, xcd, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_colburst, dim3((unsigned)((nblk + 47) / 48)), dim3(192), lds, 0, streams(i), nblk, 60);
                        });
                        printf("ns %d lg %2d  xcd %d  column bursts, 3 waves x 16 blocks, lds 66 KB, spin 60    %9.2f us %8.1f GB/s\n", ns, lg, xcd, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_colburst, dim3((unsigned)((nblk + 15) / 16)), dim3(64), lds / 3, 0, streams(i), nblk, 60);
                        });
                        printf("ns %d lg %2d  xcd %d  column bursts, 1 wave x 16 blocks, lds 22 KB, spin 60     %9.2f us %8.1f GB/s\n", ns, lg, xcd, us, bytes / us / 1e3);
                        us = timeit(reps, [&](int i) {
                            hipLaunchKernelGGL(k_colburst, dim3((unsigned)((nblk + 15) / 16)), dim3(64), 44 * 1024, 0, streams(i), nblk, 60);
                        });
                        printf("ns %d lg %2d  xcd %d  column bursts, 1 wave x 16 blocks, lds 44 KB, spin 60     %9.2f us %8.1f GB/s\n", ns, lg, xcd, us, bytes / us / 1e3);
                    }
                    g_xcd = 0;
                    const double us = timeit(reps, [&](int i) {
                        const size_t sl = (size_t)i % (slots * 2);
                        hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, buf[sl / slots] + (sl % slots) * bytes, bytes);
                    });
                    printf("ns %d lg %2d  linear 4 KB fill                                                    %9.2f us %8.1f GB/s\n", ns, lg, us, bytes / us / 1e3);
                    fflush(stdout);
                }
                continue;
            }
            if (argc > 2 && !strcmp(argv[2], "fronts")) {
                for (int rep = 0; rep < 2; ++rep) {
                    for (int mode : {0, 1})
                        for (int nb : {4, 8, 16, 64})
                            for (int threads : {64, 256}) {
                                const uint64_t nchunks = (nblk + nb - 1) / nb;
                                const double us = timeit(reps, [&](int i) {
                                    hipLaunchKernelGGL(k_fronts, dim3((unsigned)(nchunks * ns)), dim3(threads), 0, 0, streams(i), nblk, nb, mode);
                                });
                                printf("ns %d lg %2d  %s nb %2d threads %3d                  %9.2f us %8.1f GB/s\n", ns, lg,
                                       mode ? "stream after stream," : "interleaved fronts,  ", nb, threads, us, bytes / us / 1e3);
                            }
                    for (int nb : {4, 8, 16, 32})
                        for (int threads : {64, 256}) {
                            const size_t lds = (size_t)nb * bpb > 160 * 1024 ? 0 : (size_t)nb * bpb;
                            const double us = timeit(reps, [&](int i) {
                                hipLaunchKernelGGL(k_unit, dim3((unsigned)((nblk + nb - 1) / nb)), dim3(threads), lds, 0, streams(i), nblk, nb, 0);
                            });
                            printf("ns %d lg %2d  one-shot all streams, nb %2d threads %3d lds %6zu       %9.2f us %8.1f GB/s\n", ns, lg, nb, threads, lds, us, bytes / us / 1e3);
                        }
                    double us = timeit(reps, [&](int i) {
                        hipLaunchKernelGGL(k_rounds, dim3((unsigned)((nblk + 47) / 48)), dim3(192), 0, 0, streams(i), nblk, 0);
                    });
                    printf("ns %d lg %2d  round-sliced, 3 waves x 16 blocks                      %9.2f us %8.1f GB/s\n", ns, lg, us, bytes / us / 1e3);
                    us = timeit(reps, [&](int i) {
                        const size_t sl = (size_t)i % (slots * 2);
                        hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, buf[sl / slots] + (sl % slots) * bytes, bytes);
                    });
                    printf("ns %d lg %2d  linear 4 KB fill                                       %9.2f us %8.1f GB/s\n", ns, lg, us, bytes / us / 1e3);
                }
                continue;
            }
            if (argc > 2 && !strcmp(argv[2], "new")) continue;
            for (int nb : {2, 4, 8, 16, 32, 48})
                for (int threads : {64, 256}) {
                    if (threads == 256 && nb < 8) continue;
                    if (quick && (nb != 16 || threads != 64)) continue;
                    for (int ldsmode : {0, 1}) {  // 0: no LDS (residency by waves), 1: whole slabs staged (nb * bpb bytes)
                        const size_t lds = ldsmode ? (size_t)nb * bpb : 0;
                        if (lds > 160 * 1024) continue;
                        for (int spin : {0, 400, 1600}) {
                            if (ldsmode == 0 && spin) continue;
                            if (quick && (ldsmode == 0 || spin)) continue;
                            const double us = timeit(reps, [&](int i) {
                                hipLaunchKernelGGL(k_unit, dim3((unsigned)((nblk + nb - 1) / nb)), dim3(threads), lds, 0, streams(i), nblk, nb, spin);
                            });
                            printf("ns %d lg %2d  one-shot nb %2d threads %3d lds %6zu spin %4d   %9.2f us %8.1f GB/s\n", ns, lg, nb, threads, lds, spin, us,
                                   bytes / us / 1e3);
                            fflush(stdout);
                        }
                    }
                }
        }
    return 0;
}
