// probebench.hip -- round 3: is "how fast the many-front write pattern runs" a property of an ALLOCATION?
// N separate hipMalloc buffers of the same size; on each, a one-column version of the round-sliced emulation (a wave owns
// 16 x stride bytes and writes them in ten slices; 2^20/col_div "blocks") and a linear 4 KiB fill, several times, in an
// interleaved order.  If the per-buffer times are stable over repeats and differ between buffers, placement is a property
// of the allocation and can be PROBED: allocate candidates, keep the fast ones.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probebench tools/probebench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void st(uint8_t *p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__global__ void __launch_bounds__(64) k_fronts(uint8_t *base, uint32_t nblk, int stride) {
    const int lane = threadIdx.x;
    const uint32_t ng = gridDim.x, id = blockIdx.x, q = ng / 8, rr = ng % 8, x = id % 8;
    const uint32_t grp = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + id / 8;
    const size_t blk0 = (size_t)grp * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t *g = base + blk0 * stride;
    const int len = 16 * stride;
    for (int r = 0; r < 10; ++r) {
        const int lo = len / 10 * r / 1024 * 1024, hi = r == 9 ? len : len / 10 * (r + 1) / 1024 * 1024;
        for (int p = lo + lane * 16; p < hi; p += 64 * 16) st(g + p, v);
    }
}
__global__ void __launch_bounds__(256) k_fill(uint8_t *out, size_t total) {
    const size_t p = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    if (p < total) st(out + p, v);
}

// "spacing" mode: is it the PHYSICAL DISTANCE between the regions written at once?  One column (2^20 blocks x stride) whose
// eight XCD windows (an eighth of the column each, what xcd_remap 1 makes every XCD write) are eight separate physical
// chunks with filler allocations of S bytes created between them (held while measuring), so that consecutive windows lie
// about S apart in whatever order the driver hands physical memory out.
static int spacing_mode(int stride) {
    const uint32_t nblk = 1u << 20;
    const size_t bytes = (size_t)nblk * stride, win = bytes / 8;  // 2^17 blocks x stride: a multiple of 2 MiB for every stride used
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / 4;
    };
    printf("one column of %zu bytes as 8 chunks (one per XCD window), filler of S bytes between consecutive chunks\n", bytes);
    for (int rep = 0; rep < 3; ++rep)
        for (size_t S : {(size_t)0, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30, (size_t)0}) {
            void *va = nullptr;
            CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> fillers;
            for (int w = 0; w < 8; ++w) {
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, win, &prop, 0));
                CK(hipMemMap((uint8_t *)va + (size_t)w * win, win, 0, h, 0));
                CK(hipMemRelease(h));
                if (S && w < 7) {
                    hipMemGenericAllocationHandle_t f;
                    CK(hipMemCreate(&f, S, &prop, 0));
                    fillers.push_back(f);
                }
            }
            CK(hipMemSetAccess(va, bytes, &acc, 1));
            uint8_t *b = (uint8_t *)va;
            const double f = timeit([&] { hipLaunchKernelGGL(k_fronts, dim3(nblk / 16), dim3(64), 0, 0, b, nblk, stride); });
            const double l = timeit([&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, b, bytes); });
            printf("rep %d  filler %5zu MiB   fronts %7.1f us  fill %7.1f us  ratio %.3f\n", rep, S >> 20, f, l, f / l);
            fflush(stdout);
            for (auto h : fillers) CK(hipMemRelease(h));
            CK(hipMemUnmap(va, bytes));
            CK(hipMemAddressFree(va, bytes));
        }
    return 0;
}

// "map" mode: the whole device memory as 256 MiB tiles in the order the driver hands them out; the one-column many-front
// pattern against a linear fill on each, twice.  '.' = pattern within 8 % of the fill, '#' = slower, digits = the ratio's
// first decimal beyond 1.0 when the two repeats disagree.
static int map_mode(int stride, size_t tile_mib, int max_tiles) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t tile = tile_mib << 20;
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    int nt = (int)((free_b - ((size_t)6 << 30)) / tile);
    if (max_tiles > 0 && max_tiles < nt) nt = max_tiles;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, tile * nt, 0, nullptr, 0));
    int made = 0;
    for (int i = 0; i < nt; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, tile, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemMap((uint8_t *)va + (size_t)i * tile, tile, 0, h, 0));
        CK(hipMemRelease(h));
        ++made;
    }
    CK(hipMemSetAccess(va, tile * made, &acc, 1));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const uint32_t tb = (uint32_t)(tile / stride / 16 * 16);
    printf("%d tiles of %zu MiB (%.1f GiB of %.1f GiB free), stride %d\n", made, tile_mib, made * (double)tile / (1 << 30), free_b / (double)(1 << 30), stride);
    std::vector<float> r[2];
    for (int rep = 0; rep < 2; ++rep) {
        r[rep].resize(made);
        for (int i = 0; i < made; ++i) {
            uint8_t *b = (uint8_t *)va + (size_t)i * tile;
            hipLaunchKernelGGL(k_fronts, dim3(tb / 16), dim3(64), 0, 0, b, tb, stride);
            CK(hipEventRecord(e0));
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_fronts, dim3(tb / 16), dim3(64), 0, 0, b, tb, stride);
            CK(hipEventRecord(e1));
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)tb * stride + 4095) / 4096)), dim3(256), 0, 0, b, (size_t)tb * stride);
            CK(hipEventRecord(e2));
            CK(hipEventSynchronize(e2));
            float f, l;
            CK(hipEventElapsedTime(&f, e0, e1));
            CK(hipEventElapsedTime(&l, e1, e2));
            r[rep][i] = f / l;
        }
    }
    int nfast = 0;
    for (int i = 0; i < made; ++i) {
        const bool a = r[0][i] <= 1.08f, b2 = r[1][i] <= 1.08f;
        nfast += a && b2;
        if (i % 64 == 0) printf("\n%5d  ", i);
        if (a == b2) putchar(a ? '.' : '#');
        else putchar('0' + (int)std::min(9.0f, std::max(0.0f, (std::max(r[0][i], r[1][i]) - 1.0f) * 10)));
    }
    printf("\n%d of %d tiles run the pattern within 8 %% of their fill in both repeats\n", nfast, made);
    int hist[2][12] = {};
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < made; ++i) hist[rep][std::min(11, std::max(0, (int)((r[rep][i] - 0.96f) / 0.02f)))]++;
    for (int rep = 0; rep < 2; ++rep) {
        printf("repeat %d, ratio histogram from 0.96 in steps of 0.02:", rep);
        for (int b = 0; b < 12; ++b) printf(" %d", hist[rep][b]);
        printf("\n");
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "spacing")) return spacing_mode(argc > 2 ? atoi(argv[2]) : 1360);
    if (argc > 1 && !strcmp(argv[1], "map")) return map_mode(argc > 2 ? atoi(argv[2]) : 1360, argc > 3 ? (size_t)atoi(argv[3]) : 256, argc > 4 ? atoi(argv[4]) : 0);
    const int N = argc > 1 ? atoi(argv[1]) : 16;
    const int stride = argc > 2 ? atoi(argv[2]) : 1360;
    const uint32_t nblk = 1u << 20;
    const size_t bytes = (size_t)nblk * stride;
    std::vector<uint8_t *> buf(N);
    for (auto &b : buf) CK(hipMalloc(&b, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / 4;
    };
    printf("%d buffers of %zu bytes (stride %d): many-front pattern / linear fill, us per pass, three repeats\n", N, bytes, stride);
    std::vector<std::vector<double>> tf(N), tl(N);
    for (int rep = 0; rep < 3; ++rep)
        for (int i = 0; i < N; ++i) {
            tf[i].push_back(timeit([&] { hipLaunchKernelGGL(k_fronts, dim3(nblk / 16), dim3(64), 0, 0, buf[i], nblk, stride); }));
            tl[i].push_back(timeit([&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, buf[i], bytes); }));
        }
    for (int i = 0; i < N; ++i)
        printf("buffer %2d  %p   fronts %7.1f %7.1f %7.1f  (%5.0f GB/s)   fill %7.1f %7.1f %7.1f  (%5.0f GB/s)\n", i, (void *)buf[i], tf[i][0], tf[i][1], tf[i][2],
               bytes / tf[i][1] / 1e3, tl[i][0], tl[i][1], tl[i][2], bytes / tl[i][1] / 1e3);
    for (auto b : buf) CK(hipFree(b));
    // ---- tiles: the same question at finer granularity, with the virtual-memory API (what a probing allocator would use)
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t tile : {(size_t)64 << 20, (size_t)256 << 20, (size_t)1024 << 20}) {
        const int nt = (int)(((size_t)24 << 30) / tile);  // 24 GiB worth of tiles
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, tile * nt, 0, nullptr, 0));
        for (int i = 0; i < nt; ++i) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, tile, &prop, 0));
            CK(hipMemMap((uint8_t *)va + (size_t)i * tile, tile, 0, h, 0));
            CK(hipMemRelease(h));
        }
        CK(hipMemSetAccess(va, tile * nt, &acc, 1));
        const uint32_t tb = (uint32_t)(tile / stride / 16 * 16);  // blocks per tile
        printf("tiles of %zu MiB (%d of them): fronts/fill time ratio per tile (two repeats)\n", tile >> 20, nt);
        std::vector<double> r1(nt), r2(nt);
        for (int rep = 0; rep < 2; ++rep)
            for (int i = 0; i < nt; ++i) {
                uint8_t *b = (uint8_t *)va + (size_t)i * tile;
                const double f = timeit([&] { hipLaunchKernelGGL(k_fronts, dim3(tb / 16), dim3(64), 0, 0, b, tb, stride); });
                const double l = timeit([&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)tb * stride + 4095) / 4096)), dim3(256), 0, 0, b, (size_t)tb * stride); });
                (rep ? r2 : r1)[i] = f / l;
            }
        for (int i = 0; i < nt; ++i) printf("%s%.2f/%.2f", i % 16 ? "  " : "\n  ", r1[i], r2[i]);
        printf("\n");
        CK(hipMemUnmap(va, tile * nt));
        CK(hipMemAddressFree(va, tile * nt));
    }
    return 0;
}
