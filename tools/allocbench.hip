// allocbench.hip -- round 3: HOW the seven output columns are backed decides 10-15 % of the witness kernel's launch.
//
// Found while running the seed + fill emulation (tools/seedfill.hip): in ONE process on ONE box the product kernel took
// 767 us when all columns of a set were carved from one hipMalloc and 667 us when every column had its own hipMalloc;
// batch size, column alignment and skews inside one allocation changed nothing.  This tool measures the product (sc1 and
// nontemporal stores), its stores-only mode and the round-sliced emulation over different BACKINGS of the same virtual
// layout, built with the virtual-memory API where needed (hipMemAddressReserve / hipMemCreate / hipMemMap):
//   one       one hipMalloc per set, columns back to back                      (what a C host would naturally do)
//   percol    one hipMalloc per column                                         (what torch tensors are)
//   vmm S     one contiguous VA range per set, backed by physical chunks of S bytes each (hipMemCreate per chunk)
//   vmmcol S  one VA range per column, chunks of S bytes
//   vmmxcd    one VA range per column, one physical chunk per XCD window (an eighth of the column, rounded to the granule)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/allocbench tools/allocbench.hip -ldl
// Run:   tools/allocbench [tools/libaesw_diag.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include "../include/aesw.h"
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d: %s\n", hipGetErrorString(e_), __LINE__, #x); exit(1); } } while (0)

constexpr int NS = 7;
__constant__ int c_stride[NS] = {1360, 1056, 608, 96, 400, 240, 200};
static const int h_stride[NS] = {1360, 1056, 608, 96, 400, 240, 200};
struct Cols { uint8_t *base[NS]; };

__device__ __forceinline__ void st(uint8_t *p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__global__ void __launch_bounds__(64) k_rounds(Cols o, uint32_t nblk) {
    const int lane = threadIdx.x;
    const uint32_t ng = gridDim.x, id = blockIdx.x, q = ng / 8, rr = ng % 8, x = id % 8;
    const uint32_t grp = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + id / 8;
    const size_t blk0 = (size_t)grp * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    for (int c = 3; c < NS; ++c) {
        uint8_t *g = o.base[c] + blk0 * c_stride[c];
        for (int p = lane * 16; p < 16 * c_stride[c]; p += 64 * 16) st(g + p, v);
    }
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 3; ++c) {
            uint8_t *g = o.base[c] + blk0 * c_stride[c];
            const int len = 16 * c_stride[c];
            const int lo = len / 10 * r / 1024 * 1024, hi = r == 9 ? len : len / 10 * (r + 1) / 1024 * 1024;
            for (int p = lo + lane * 16; p < hi; p += 64 * 16) st(g + p, v);
        }
}
// variants of the round-sliced emulation that break the lockstep of neighbouring waves:
//   rot:  wave g starts with slice (g % 10) of its column ranges instead of slice 0 (same bytes, rotated order)
//   odd:  the slices are cut at 128-byte granularity with a per-wave odd offset instead of 1 KiB boundaries
__global__ void __launch_bounds__(64) k_rounds_var(Cols o, uint32_t nblk, int rot, int gran) {
    const int lane = threadIdx.x;
    const uint32_t ng = gridDim.x, id = blockIdx.x, q = ng / 8, rr = ng % 8, x = id % 8;
    const uint32_t grp = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + id / 8;
    const size_t blk0 = (size_t)grp * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    for (int c = 3; c < NS; ++c) {
        uint8_t *g = o.base[c] + blk0 * c_stride[c];
        for (int p = lane * 16; p < 16 * c_stride[c]; p += 64 * 16) st(g + p, v);
    }
    const int r0 = rot ? (int)(grp % 10) : 0;
    for (int rr_ = 0; rr_ < 10; ++rr_) {
        const int r = (rr_ + r0) % 10;
        for (int c = 0; c < 3; ++c) {
            uint8_t *g = o.base[c] + blk0 * c_stride[c];
            const int len = 16 * c_stride[c];
            const int lo = len / 10 * r / gran * gran, hi = r == 9 ? len : len / 10 * (r + 1) / gran * gran;
            for (int p = lo + lane * 16; p < hi; p += 64 * 16) st(g + p, v);
        }
    }
}
template <int ID>
__global__ void __launch_bounds__(64) k_rounds_id(Cols o, uint32_t nblk) {
    const int lane = threadIdx.x;
    const uint32_t ng = gridDim.x, id = blockIdx.x, q = ng / 8, rr = ng % 8, x = id % 8;
    const uint32_t grp = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + id / 8;
    const size_t blk0 = (size_t)grp * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane + ID};
    for (int c = 3; c < NS; ++c) {
        uint8_t *g = o.base[c] + blk0 * c_stride[c];
        for (int p = lane * 16; p < 16 * c_stride[c]; p += 64 * 16) st(g + p, v);
    }
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 3; ++c) {
            uint8_t *g = o.base[c] + blk0 * c_stride[c];
            const int len = 16 * c_stride[c];
            const int lo = len / 10 * r / 1024 * 1024, hi = r == 9 ? len : len / 10 * (r + 1) / 1024 * 1024;
            for (int p = lo + lane * 16; p < hi; p += 64 * 16) st(g + p, v);
        }
}
__global__ void __launch_bounds__(256) k_fill(uint8_t *out, size_t total) {
    const size_t p = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    if (p < total) st(out + p, v);
}

static size_t g_gran = 2 << 20;
static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_acc;
static size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// a VA range of `bytes` backed by physical chunks whose sizes are given (each rounded up to the granule)
static uint8_t *vmm_range(const std::vector<size_t> &chunks) {
    size_t total = 0;
    for (size_t c : chunks) total += up(c, g_gran);
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    size_t off = 0;
    for (size_t c : chunks) {
        const size_t sz = up(c, g_gran);
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, sz, &g_prop, 0));
        CK(hipMemMap(reinterpret_cast<uint8_t *>(va) + off, sz, 0, h, 0));
        CK(hipMemRelease(h));  // the mapping keeps the memory alive
        off += sz;
    }
    CK(hipMemSetAccess(va, total, &g_acc, 1));
    return reinterpret_cast<uint8_t *>(va);
}
// the same, but the physical chunks are created first (in order) and then mapped into the VA range in a pseudo-random
// order: whatever pattern the physical allocator follows, neighbouring virtual chunks land on unrelated frames
static uint8_t *vmm_range_shuffled(size_t bytes, size_t chunk, uint64_t seed) {
    const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &g_prop, 0));
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
    for (size_t i = n - 1; i > 0; --i) {  // Fisher-Yates with xorshift64*
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const size_t j = (size_t)((x * 0x2545f4914f6cdd1dull) % (i + 1));
        std::swap(order[i], order[j]);
    }
    for (size_t i = 0; i < n; ++i) {
        CK(hipMemMap(reinterpret_cast<uint8_t *>(va) + i * chunk, chunk, 0, h[order[i]], 0));
        CK(hipMemRelease(h[order[i]]));
    }
    CK(hipMemSetAccess(va, total, &g_acc, 1));
    return reinterpret_cast<uint8_t *>(va);
}
// ---- probed tiles: allocate more physical tiles than needed, time a many-front write and a linear fill on each (mapped at a
// scratch address), keep the tiles with the lowest fronts / fill ratio, map them as one VA range, release the rest
__global__ void __launch_bounds__(64) k_probe_fronts(uint8_t *base, uint32_t nblk) {
    const int lane = threadIdx.x;
    const uint32_t ng = gridDim.x, id = blockIdx.x, q = ng / 8, rr = ng % 8, x = id % 8;
    const uint32_t grp = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + id / 8;
    const size_t blk0 = (size_t)grp * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    uint8_t *g = base + blk0 * 1360;
    for (int r = 0; r < 10; ++r) {
        const int lo = 2176 * r / 1024 * 1024, hi = r == 9 ? 21760 : 2176 * (r + 1) / 1024 * 1024;
        for (int p = lo + lane * 16; p < hi; p += 64 * 16) st(g + p, v);
    }
}
__global__ void __launch_bounds__(256) k_probe_fill(uint8_t *out, size_t total) {
    const size_t p = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    if (p < total) st(out + p, v);
}
static std::vector<double> g_last_ratios;
static uint8_t *vmm_probed(size_t bytes, size_t tile, double over) {
    const size_t need = (bytes + tile - 1) / tile, cand = (size_t)(need * over) + 1;
    std::vector<hipMemGenericAllocationHandle_t> h(cand);
    std::vector<std::pair<double, size_t>> score(cand);
    void *scratch = nullptr;
    CK(hipMemAddressReserve(&scratch, tile, 0, nullptr, 0));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const uint32_t tb = (uint32_t)(tile / 1360 / 16 * 16);
    for (size_t i = 0; i < cand; ++i) {
        CK(hipMemCreate(&h[i], tile, &g_prop, 0));
        CK(hipMemMap(scratch, tile, 0, h[i], 0));
        CK(hipMemSetAccess(scratch, tile, &g_acc, 1));
        uint8_t *b = reinterpret_cast<uint8_t *>(scratch);
        hipLaunchKernelGGL(k_probe_fronts, dim3(tb / 16), dim3(64), 0, 0, b, tb);  // warm
        CK(hipEventRecord(e0));
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_probe_fronts, dim3(tb / 16), dim3(64), 0, 0, b, tb);
        CK(hipEventRecord(e1));
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)(((size_t)tb * 1360 + 4095) / 4096)), dim3(256), 0, 0, b, (size_t)tb * 1360);
        CK(hipEventRecord(e2));
        CK(hipEventSynchronize(e2));
        float f, l;
        CK(hipEventElapsedTime(&f, e0, e1));
        CK(hipEventElapsedTime(&l, e1, e2));
        score[i] = {f / l, i};
        CK(hipMemUnmap(scratch, tile));
    }
    CK(hipMemAddressFree(scratch, tile));
    std::sort(score.begin(), score.end());
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, need * tile, 0, nullptr, 0));
    g_last_ratios.clear();
    for (size_t k = 0; k < cand; ++k) {
        const size_t i = score[k].second;
        if (k < need) {
            CK(hipMemMap(reinterpret_cast<uint8_t *>(va) + k * tile, tile, 0, h[i], 0));
            g_last_ratios.push_back(score[k].first);
        }
        CK(hipMemRelease(h[i]));  // the rejected tiles go back to the driver here
    }
    CK(hipMemSetAccess(va, need * tile, &g_acc, 1));
    g_last_ratios.push_back(score[cand - 1].first);  // the worst candidate, for the log
    return reinterpret_cast<uint8_t *>(va);
}
static std::vector<size_t> even_chunks(size_t bytes, size_t chunk) {
    std::vector<size_t> v;
    for (size_t o = 0; o < bytes; o += chunk) v.push_back(std::min(chunk, bytes - o));
    return v;
}

int main(int argc, char **argv) {
    const char *libpath = argc > 1 ? argv[1] : "tools/libaesw_diag.so";
    const uint32_t nblk = 1u << 20;
    size_t bpb = 0;
    for (int s : h_stride) bpb += s;
    const size_t bytes = (size_t)nblk * bpb, alg = (size_t)nblk * 3992;
    const int NBUF = 2;
    int dev = 0;
    CK(hipGetDevice(&dev));
    memset(&g_prop, 0, sizeof g_prop);
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = dev;
    g_acc.location = g_prop.location;
    g_acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemGetAllocationGranularity(&g_gran, &g_prop, hipMemAllocationGranularityRecommended));
    printf("VMM granule (recommended): %zu bytes\n", g_gran);

    uint8_t *pt, *keys;
    CK(hipMalloc(&pt, (size_t)nblk * 16));
    CK(hipMalloc(&keys, (size_t)nblk * 16));
    CK(hipMemset(pt, 0x5a, (size_t)nblk * 16));
    CK(hipMemset(keys, 0xc3, (size_t)nblk * 16));

    struct Backing { std::string name; Cols set[NBUF]; };
    std::vector<Backing> backs;
    auto carve = [&](uint8_t *b) { Cols c; for (int s = 0; s < NS; ++s) { c.base[s] = b; b += (size_t)nblk * h_stride[s]; } return c; };
    {   // one hipMalloc per set
        Backing k{"one hipMalloc per set, columns back to back", {}};
        for (auto &c : k.set) { uint8_t *b; CK(hipMalloc(&b, bytes)); c = carve(b); }
        backs.push_back(k);
    }
    {   // one hipMalloc per column
        Backing k{"one hipMalloc per column", {}};
        for (auto &c : k.set) for (int s = 0; s < NS; ++s) CK(hipMalloc(&c.base[s], (size_t)nblk * h_stride[s]));
        backs.push_back(k);
    }
    for (double over : {2.0, 3.0}) {
        char nm[128];
        snprintf(nm, sizeof nm, "PROBED 256 MiB tiles: best of %.0fx candidates per set", over);
        Backing k{nm, {}};
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (auto &c : k.set) {
            c = carve(vmm_probed(bytes, (size_t)256 << 20, over));
            printf("  probed set: kept ratios %.3f .. %.3f, worst candidate %.3f\n", g_last_ratios.front(), g_last_ratios[g_last_ratios.size() - 2], g_last_ratios.back());
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("built '%s' x %d sets in %.2f s\n", nm, NBUF, (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9);
        backs.push_back(k);
    }
    for (int rep_ = 0; rep_ < 1; ++rep_)
        for (size_t chunk : {(size_t)2 << 20, (size_t)16 << 20}) {
            char nm[128];
            snprintf(nm, sizeof nm, "one VA range per set, %zu MiB chunks mapped in SHUFFLED order (seed %d)", chunk >> 20, rep_);
            Backing k{nm, {}};
            for (int b = 0; b < NBUF; ++b) k.set[b] = carve(vmm_range_shuffled(bytes, chunk, 17 * rep_ + b + chunk));
            backs.push_back(k);
        }
    for (size_t chunk : {(size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
        char nm[128];
        snprintf(nm, sizeof nm, "one VA range per set, physical chunks of %zu KiB", chunk >> 10);
        Backing k{nm, {}};
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (auto &c : k.set) c = carve(vmm_range(even_chunks(bytes, chunk)));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("built '%s' x %d sets in %.2f s\n", nm, NBUF, (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9);
        backs.push_back(k);
    }
    for (size_t chunk : {(size_t)2 << 20, (size_t)32 << 20}) {
        char nm[128];
        snprintf(nm, sizeof nm, "one VA range per column, physical chunks of %zu MiB", chunk >> 20);
        Backing k{nm, {}};
        for (auto &c : k.set) for (int s = 0; s < NS; ++s) c.base[s] = vmm_range(even_chunks((size_t)nblk * h_stride[s], chunk));
        backs.push_back(k);
    }
    {   // per column, one physical chunk per XCD window (an eighth of the column)
        Backing k{"one VA range per column, one physical chunk per XCD window (1/8 column)", {}};
        for (auto &c : k.set) for (int s = 0; s < NS; ++s) c.base[s] = vmm_range(even_chunks((size_t)nblk * h_stride[s], up((size_t)nblk * h_stride[s] / 8, g_gran)));
        backs.push_back(k);
    }
    {   // per column, ONE physical chunk (should equal "one hipMalloc per column")
        Backing k{"one VA range per column, one physical chunk", {}};
        for (auto &c : k.set) for (int s = 0; s < NS; ++s) c.base[s] = vmm_range({(size_t)nblk * h_stride[s]});
        backs.push_back(k);
    }

    if (argc > 2 && !strcmp(argv[2], "pmc")) {
        // for rocprofv3 --pmc / --kernel-trace: the emulation over backing i runs as kernel k_rounds_id<i>
        for (size_t i = 0; i < backs.size(); ++i) printf("k_rounds_id<%zu> = %s\n", i, backs[i].name.c_str());
        for (int rep_ = 0; rep_ < 4; ++rep_)
            for (size_t i = 0; i < backs.size(); ++i) {
                const Cols &o = backs[i].set[rep_ % NBUF];
                const dim3 g(nblk / 16), t(64);
                switch (i) {
                case 0: hipLaunchKernelGGL(k_rounds_id<0>, g, t, 0, 0, o, nblk); break;
                case 1: hipLaunchKernelGGL(k_rounds_id<1>, g, t, 0, 0, o, nblk); break;
                case 2: hipLaunchKernelGGL(k_rounds_id<2>, g, t, 0, 0, o, nblk); break;
                case 3: hipLaunchKernelGGL(k_rounds_id<3>, g, t, 0, 0, o, nblk); break;
                case 4: hipLaunchKernelGGL(k_rounds_id<4>, g, t, 0, 0, o, nblk); break;
                case 5: hipLaunchKernelGGL(k_rounds_id<5>, g, t, 0, 0, o, nblk); break;
                case 6: hipLaunchKernelGGL(k_rounds_id<6>, g, t, 0, 0, o, nblk); break;
                case 7: hipLaunchKernelGGL(k_rounds_id<7>, g, t, 0, 0, o, nblk); break;
                case 8: hipLaunchKernelGGL(k_rounds_id<8>, g, t, 0, 0, o, nblk); break;
                case 9: hipLaunchKernelGGL(k_rounds_id<9>, g, t, 0, 0, o, nblk); break;
                case 10: hipLaunchKernelGGL(k_rounds_id<10>, g, t, 0, 0, o, nblk); break;
                case 11: hipLaunchKernelGGL(k_rounds_id<11>, g, t, 0, 0, o, nblk); break;
                case 12: hipLaunchKernelGGL(k_rounds_id<12>, g, t, 0, 0, o, nblk); break;
                case 13: hipLaunchKernelGGL(k_rounds_id<13>, g, t, 0, 0, o, nblk); break;
                case 14: hipLaunchKernelGGL(k_rounds_id<14>, g, t, 0, 0, o, nblk); break;
                default: hipLaunchKernelGGL(k_rounds_id<15>, g, t, 0, 0, o, nblk); break;
                }
                CK(hipDeviceSynchronize());
            }
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, 0, backs[0].set[0].base[0], bytes);
        CK(hipDeviceSynchronize());
        return 0;
    }
    void *lib = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { printf("cannot load %s: %s\n", libpath, dlerror()); return 1; }
    auto create = reinterpret_cast<decltype(&aesw_create)>(dlsym(lib, "aesw_create"));
    auto setopt = reinterpret_cast<decltype(&aesw_set_option)>(dlsym(lib, "aesw_set_option"));
    auto enc = reinterpret_cast<decltype(&aesw_encrypt_witness_device)>(dlsym(lib, "aesw_encrypt_witness_device"));
    uint8_t sbox[256], m2[256], m3[256];
    for (int i = 0; i < 256; ++i) { sbox[i] = (uint8_t)(i * 7 + 3); m2[i] = (uint8_t)((i << 1) ^ ((i & 0x80) ? 0x1b : 0)); m3[i] = (uint8_t)(m2[i] ^ i); }
    aesw_ctx *ctx[3];
    for (auto &c : ctx) if (create(&c, 0, sbox, m2, m3) != AESW_OK) { printf("aesw_create failed\n"); return 1; }
    setopt(ctx[0], "store_mode", 2);
    setopt(ctx[1], "store_mode", 1);
    const bool diag = setopt(ctx[2], "store_mode", 5) == AESW_OK;
    // "xcd" mode: the product (nontemporal stores) under different workgroup -> block group orders
    const int xmodes[] = {1, 0, 4, 32, 256, 2048};
    aesw_ctx *xctx[6];
    for (int i = 0; i < 6; ++i) {
        if (create(&xctx[i], 0, sbox, m2, m3) != AESW_OK) { printf("aesw_create failed\n"); return 1; }
        setopt(xctx[i], "store_mode", 1);
        if (setopt(xctx[i], "xcd_remap", xmodes[i]) != AESW_OK) { printf("xcd_remap %d refused\n", xmodes[i]); return 1; }
    }

    hipStream_t st0;
    CK(hipStreamCreate(&st0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        const int reps = 6;
        for (int i = 0; i < 2; ++i) launch(i);
        CK(hipStreamSynchronize(st0));
        std::vector<double> t;
        for (int k = 0; k < 5; ++k) {
            CK(hipEventRecord(e0, st0));
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(e1, st0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3 / reps);
        }
        std::sort(t.begin(), t.end());
        return t[2];
    };
    auto product = [&](aesw_ctx *c, const Cols &o) {
        aesw_key_slab ks{o.base[3], o.base[4], o.base[5], o.base[6]};
        const int rc = enc(c, pt, keys, 1, nblk, AESW_LAYOUT_PACKED, o.base[0], o.base[1], o.base[2], nullptr, &ks, st0);
        if (rc != AESW_OK) { printf("encrypt rc %d\n", rc); exit(1); }
    };
    if (argc > 2 && !strcmp(argv[2], "lockstep")) {
        printf("%-78s %10s %10s %10s\n", "backing \\ emulation variant (us per launch)", "lockstep", "rotated", "rot+128B");
        for (int round = 0; round < 2; ++round)
            for (const Backing &b : backs) {
                const double a = timeit([&](int i) { hipLaunchKernelGGL(k_rounds_var, dim3(nblk / 16), dim3(64), 0, st0, b.set[i % NBUF], nblk, 0, 1024); });
                const double r = timeit([&](int i) { hipLaunchKernelGGL(k_rounds_var, dim3(nblk / 16), dim3(64), 0, st0, b.set[i % NBUF], nblk, 1, 1024); });
                const double o = timeit([&](int i) { hipLaunchKernelGGL(k_rounds_var, dim3(nblk / 16), dim3(64), 0, st0, b.set[i % NBUF], nblk, 1, 128); });
                printf("%-78s %10.1f %10.1f %10.1f\n", b.name.c_str(), a, r, o);
                fflush(stdout);
            }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "xcd")) {
        printf("%-78s", "backing \\ xcd_remap (product, nontemporal stores, us per launch)");
        for (int m : xmodes) printf(" %8d", m);
        printf("\n");
        for (int round = 0; round < 2; ++round)
            for (const Backing &b : backs) {
                printf("%-78s", b.name.c_str());
                for (int i = 0; i < 6; ++i) printf(" %8.1f", timeit([&](int k) { product(xctx[i], b.set[k % NBUF]); }));
                printf("\n");
                fflush(stdout);
            }
        return 0;
    }
    printf("%-78s %10s %10s %10s %10s   (us per launch of 2^20 blocks; GB/s algorithmic in brackets)\n", "backing", "sc1", "nt", "stores", "emulation");
    for (int round = 0; round < 3; ++round) {
        for (const Backing &b : backs) {
            const double a = timeit([&](int i) { product(ctx[0], b.set[i % NBUF]); });
            const double n = timeit([&](int i) { product(ctx[1], b.set[i % NBUF]); });
            const double s = diag ? timeit([&](int i) { product(ctx[2], b.set[i % NBUF]); }) : 0.0;
            const double e = timeit([&](int i) { hipLaunchKernelGGL(k_rounds, dim3(nblk / 16), dim3(64), 0, st0, b.set[i % NBUF], nblk); });
            printf("%-78s %6.1f (%4.0f) %6.1f (%4.0f) %6.1f (%4.0f) %6.1f (%4.0f)\n", b.name.c_str(), a, alg / a / 1e3, n, alg / n / 1e3, s, s ? alg / s / 1e3 : 0.0, e, alg / e / 1e3);
            fflush(stdout);
        }
        const double f = timeit([&](int i) { hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, st0, backs[0].set[i % NBUF].base[0], bytes); });
        printf("%-78s %6.1f (%4.0f)\n", "linear 4 KiB fill of the same bytes (first backing)", f, alg / f / 1e3);
    }
    return 0;
}
