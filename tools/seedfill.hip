// seedfill.hip -- round 3, the "seed + fill" experiment of VERDICT r02 item 5, with its kill criterion.
//
// Question: the witness kernel is a many-stream writer (a wave owns 16 blocks of every column for ten rounds) and sits
// 10-20 % under a linear fill.  Would a two-kernel division be faster?
//   seed : the AES rounds only; per block it writes a SEED (the eleven states s_0..s_10, the eleven round keys, the
//          ciphertext: 368 B, stored as 384 B) -- ~0.35 KB per block instead of 3 960 B;
//   fill : expand_fr's proven geometry -- one-shot workgroups that each write ONE 4 KiB chunk of ONE output column with
//          one store per wave, chunks in address order -- rebuilding every 16-byte piece from the seeds of the 3-44
//          blocks the chunk covers (seeds staged in LDS, a table-driven row -> (round, record, byte) map in the real
//          thing; here: the loads, a barrier, two LDS reads, an optional spin, the store).
// The seeds must not make a round trip through HBM (+19 % traffic), so the batch is cut into chunks whose seeds stay
// cache resident, pipelined over two or more streams: seed(i+1) runs beside fill(i).  Everything here EMULATES the
// memory side (no AES): it is an upper bound for what such a design could reach.  In the same process: the product
// kernel and its stores-only mode (store_mode 5 of a -DAESW_DIAGNOSTIC build, through the C ABI), the round-sliced
// emulation of tools/unitbench.hip, and a linear 4 KiB fill of the same bytes.
// Kill criterion (VERDICT): build the real thing only if the best seed + fill emulation reaches >= 1.10 x the
// product's algorithmic GB/s on two boxes.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/seedfill tools/seedfill.hip -ldl
// Run:   tools/seedfill [path/to/libaesw_diag.so]      (python tools/parts.py --build-only builds that library)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/aesw.h"
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NS = 7;
constexpr int SEED = 384;   // bytes per block: 176 states + 176 round keys + 16 ct, padded
constexpr int PANEL = 512;  // blocks per panel: 512 * stride is a multiple of 4096 for every stream
__constant__ int c_stride[NS] = {1360, 1056, 608, 96, 400, 240, 200};
static const int h_stride[NS] = {1360, 1056, 608, 96, 400, 240, 200};
// chunks per panel and stream: 170 132 76 12 50 30 25 = 495
__constant__ int c_cum[NS + 1] = {0, 170, 302, 378, 390, 440, 470, 495};
constexpr int CPP = 495;

struct Cols { uint8_t *base[NS]; };

__device__ __forceinline__ void st(uint8_t *p, u32x4 v, int flavour) {
    if (flavour == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else if (flavour == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = v;
}

// seed kernel: a wave owns 16 blocks, spins (the AES rounds), writes 16 * 384 B contiguously
__global__ void __launch_bounds__(256) k_seed(uint8_t *seed, uint32_t nblk, int spin, int flavour) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t blk0 = wave * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
    uint8_t *g = seed + (size_t)blk0 * SEED;
#pragma unroll
    for (int p = 0; p < 16 * SEED; p += 64 * 16) st(g + p + lane * 16, v, flavour);
}

// fill kernel: workgroup = one 4 KiB chunk of one stream.  xcd = 1: workgroups that share an XCD (id % 8) take whole
// panels, so a panel's seeds are fetched into ONE XCD's L2 and re-read there.  readmode 0: no seed reads (pure fill in
// this order), 1: stage the covered blocks' seeds in LDS, barrier, two LDS reads per lane
__global__ void __launch_bounds__(256) k_fill_seed(Cols o, const uint8_t *seed, uint32_t nblk, uint32_t npanels, int xcd, int readmode,
                                                   int spin, int flavour) {
    __shared__ u32x4 lds[640];
    uint32_t u = blockIdx.x, panel, r;
    if (xcd) {
        const uint32_t x = u % 8, q = u / 8;  // panels are dealt to XCDs round-robin: panel = 8 * (q / CPP) + x
        panel = 8 * (q / CPP) + x;
        r = q % CPP;
    } else {
        panel = u / CPP;
        r = u % CPP;
    }
    if (panel >= npanels) return;
    int c = 0;
#pragma unroll
    for (int i = 1; i < NS; ++i) c += r >= (uint32_t)c_cum[i];
    const uint32_t j = r - c_cum[c];  // chunk inside the panel
    const int stride = c_stride[c];
    const uint32_t off = j * 4096u;  // byte offset inside the panel's range of this stream
    uint8_t *g = o.base[c] + (size_t)panel * PANEL * stride + off + threadIdx.x * 16;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    if (readmode) {
        const uint32_t b_lo = off / stride, b_hi = (off + 4095u) / stride;  // blocks (panel relative) this chunk covers
        const bool keycol = c >= 3;
        const uint32_t per = keycol ? 12 : 24;  // 16-byte pieces of a seed this column needs (round keys only / everything)
        const uint32_t nload = (b_hi - b_lo + 1) * per;
        const uint8_t *s = seed + ((size_t)panel * PANEL + b_lo) * SEED + (keycol ? 176 : 0);
        for (uint32_t i = threadIdx.x; i < nload; i += 256) lds[i] = *reinterpret_cast<const u32x4 *>(s + (i / per) * SEED + (i % per) * 16);
        __syncthreads();
        const u32x4 a = lds[(threadIdx.x * 37u) % nload], b = lds[(threadIdx.x * 11u + 5u) % nload];
        v.x ^= a.x ^ b.y;
        v.y ^= a.z ^ b.w;
    }
    for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
    st(g, v, flavour);
}

// the round-sliced pattern of tools/unitbench.hip (one wave x 16 blocks per workgroup, XCD-contiguous order, sc1)
__global__ void __launch_bounds__(64) k_rounds(Cols o, uint32_t nblk) {
    const int lane = threadIdx.x;
    const uint32_t ng = gridDim.x, id = blockIdx.x, q = ng / 8, rr = ng % 8, x = id % 8;
    const uint32_t grp = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + id / 8;
    const size_t blk0 = (size_t)grp * 16;
    if (blk0 >= nblk) return;
    u32x4 v = {1u, 2u, 3u, (uint32_t)lane};
    for (int c = 3; c < NS; ++c) {
        uint8_t *g = o.base[c] + blk0 * c_stride[c];
        for (int p = lane * 16; p < 16 * c_stride[c]; p += 64 * 16) st(g + p, v, 2);
    }
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 3; ++c) {
            uint8_t *g = o.base[c] + blk0 * c_stride[c];
            const int len = 16 * c_stride[c];
            const int lo = len / 10 * r / 1024 * 1024, hi = r == 9 ? len : len / 10 * (r + 1) / 1024 * 1024;
            for (int p = lo + lane * 16; p < hi; p += 64 * 16) st(g + p, v, 2);
        }
}

__global__ void __launch_bounds__(256) k_fill(uint8_t *out, size_t total, int flavour) {
    const size_t p = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    if (p < total) st(out + p, v, flavour);
}

int main(int argc, char **argv) {
    const char *libpath = argc > 1 ? argv[1] : "tools/libaesw_diag.so";
    const bool sizes_mode = argc > 2 && !strcmp(argv[2], "sizes");
    const uint32_t nmax = (1u << 20) + 4096;  // buffers are sized for the largest batch of the "sizes" mode
    uint32_t nblk = 1u << 20;
    size_t bpb = 0;
    for (int s : h_stride) bpb += s;
    size_t bytes = (size_t)nblk * bpb;       // written cells: 3 960 B per block
    size_t alg = (size_t)nblk * 3992;        // algorithmic bytes of the product (written + 32 B read)
    const int NBUF = 3;                            // rotate output sets: > 256 MiB between rewrites
    uint8_t *buf[NBUF], *seed, *pt, *keys;
    for (auto &b : buf) CK(hipMalloc(&b, (size_t)nmax * bpb + (64 << 20)));
    CK(hipMalloc(&seed, (size_t)nmax * SEED));
    CK(hipMalloc(&pt, (size_t)nmax * 16));
    CK(hipMalloc(&keys, (size_t)nmax * 16));
    CK(hipMemset(pt, 0x5a, (size_t)nmax * 16));
    CK(hipMemset(keys, 0xc3, (size_t)nmax * 16));
    size_t col_skew = 0;  // "sizes" mode: extra bytes between consecutive columns of a set
    // "sizes" mode, second question: ONE allocation per buffer set (columns carved back to back) or one hipMalloc per column?
    uint8_t *sep[NBUF][NS];
    bool separate = false;
    if (sizes_mode)
        for (auto &set : sep)
            for (int c = 0; c < NS; ++c) CK(hipMalloc(&set[c], (size_t)nmax * h_stride[c] + (2 << 20)));
    auto cols_of = [&](int i, uint32_t first_block) {
        Cols c;
        uint8_t *b = buf[i % NBUF];
        if (separate) {
            for (int s = 0; s < NS; ++s) c.base[s] = sep[i % NBUF][s] + (size_t)first_block * h_stride[s];
            return c;
        }
        for (int s = 0; s < NS; ++s) { c.base[s] = b + (size_t)first_block * h_stride[s]; b += (size_t)nblk * h_stride[s] + col_skew; }
        return c;
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipStream_t st0;
    CK(hipStreamCreate(&st0));
    // median of 7 timed batches of `reps` launches each
    auto timeit = [&](int reps, auto launch) {
        for (int i = 0; i < 2; ++i) launch(i, st0);
        CK(hipStreamSynchronize(st0));
        std::vector<double> t;
        for (int k = 0; k < 7; ++k) {
            CK(hipEventRecord(e0, st0));
            for (int i = 0; i < reps; ++i) launch(k * reps + i, st0);
            CK(hipEventRecord(e1, st0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3 / reps);
        }
        std::sort(t.begin(), t.end());
        return t[3];
    };
    auto report = [&](const char *name, double us) {
        printf("%-92s %9.2f us  %7.1f GB/s written  %7.1f GB/s algorithmic (%.3f of 8 TB/s)\n", name, us, bytes / us / 1e3, alg / us / 1e3, alg / us / 8e6);
        fflush(stdout);
    };

    // ---- the product through the C ABI (same process) ----------------------------------------------------------
    void *lib = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    aesw_ctx *ctx[3] = {nullptr, nullptr, nullptr};  // product (sc1), stores only, product with nontemporal stores
    decltype(&aesw_encrypt_witness_device) enc = nullptr;
    if (lib) {
        auto create = reinterpret_cast<decltype(&aesw_create)>(dlsym(lib, "aesw_create"));
        auto setopt = reinterpret_cast<decltype(&aesw_set_option)>(dlsym(lib, "aesw_set_option"));
        enc = reinterpret_cast<decltype(&aesw_encrypt_witness_device)>(dlsym(lib, "aesw_encrypt_witness_device"));
        uint8_t sbox[256], m2[256], m3[256];
        for (int i = 0; i < 256; ++i) {  // any table set with xtime mul tables takes the product's fast path; values are irrelevant here
            sbox[i] = (uint8_t)(i * 7 + 3);
            m2[i] = (uint8_t)((i << 1) ^ ((i & 0x80) ? 0x1b : 0));
            m3[i] = (uint8_t)(m2[i] ^ i);
        }
        for (int k = 0; k < 3; ++k) {
            if (create(&ctx[k], 0, sbox, m2, m3) != AESW_OK) { printf("aesw_create failed\n"); return 1; }
        }
        setopt(ctx[0], "store_mode", 2);
        setopt(ctx[2], "store_mode", 1);
        if (setopt(ctx[1], "store_mode", 5) != AESW_OK) { printf("%s is not a -DAESW_DIAGNOSTIC build: no stores-only mode\n", libpath); ctx[1] = nullptr; }
    } else {
        printf("no %s (%s): product rows skipped\n", libpath, dlerror());
    }
    auto product = [&](int which, int i, hipStream_t s) {
        Cols c = cols_of(i, 0);
        aesw_key_slab ks{c.base[3], c.base[4], c.base[5], c.base[6]};
        const int rc = enc(ctx[which], pt, keys, 1, nblk, AESW_LAYOUT_PACKED, c.base[0], c.base[1], c.base[2], nullptr, &ks, s);
        if (rc != AESW_OK) { printf("encrypt rc %d\n", rc); exit(1); }
    };

    // ---- seed + fill pipelines --------------------------------------------------------------------------------------
    // chunk = 2^lgc blocks; nstreams streams take chunks round-robin (seed(i); fill(i) in stream order), fork/join
    // around the whole batch with events so that the timing stream sees one unit of work
    const int MAXS = 4;
    hipStream_t ps[MAXS];
    hipEvent_t fork_ev, join_ev[MAXS];
    for (auto &s : ps) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    for (auto &jev : join_ev) CK(hipEventCreateWithFlags(&jev, hipEventDisableTiming));
    struct Variant { int lgc, nstreams, xcd, readmode, spin_fill, spin_seed, seed_flavour, fill_flavour, with_seed; };
    auto pipeline = [&](const Variant &v, int i, hipStream_t s) {
        const uint32_t cb = 1u << v.lgc, nchunks = nblk / cb;
        CK(hipEventRecord(fork_ev, s));
        for (int k = 0; k < v.nstreams; ++k) CK(hipStreamWaitEvent(ps[k], fork_ev, 0));
        for (uint32_t ch = 0; ch < nchunks; ++ch) {
            hipStream_t q = ps[ch % v.nstreams];
            uint8_t *sd = seed + (size_t)ch * cb * SEED;  // every chunk has its own seed range: no reuse hazard between streams
            if (v.with_seed) hipLaunchKernelGGL(k_seed, dim3((cb / 16 + 3) / 4), dim3(256), 0, q, sd, cb, v.spin_seed, v.seed_flavour);
            const uint32_t npanels = cb / PANEL;
            const uint32_t grid = v.xcd ? ((npanels + 7) / 8) * 8 * CPP : npanels * CPP;
            hipLaunchKernelGGL(k_fill_seed, dim3(grid), dim3(256), 0, q, cols_of(i, ch * cb), sd, cb, npanels, v.xcd, v.readmode, v.spin_fill,
                               v.fill_flavour);
        }
        for (int k = 0; k < v.nstreams; ++k) {
            CK(hipEventRecord(join_ev[k], ps[k]));
            CK(hipStreamWaitEvent(s, join_ev[k], 0));
        }
    };
    // a graph of one pipeline pass removes the host's launch cost (2 * nchunks launches) from the measurement
    auto graph_of = [&](const Variant &v, int i) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st0, hipStreamCaptureModeGlobal));
        pipeline(v, i, st0);
        CK(hipStreamEndCapture(st0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        return ge;
    };

    const int reps = 5;
    if (sizes_mode) {
        // Is the product's rate tied to the batch size being a power of two?  With 2^20 blocks every XCD's write window of a
        // column starts a multiple of 1 MiB behind the previous one (2^17 blocks x stride) and every column carved back to
        // back starts on a multiple of 16 MiB: 56 fronts that advance in lockstep at a fixed power-of-two distance.
        const uint32_t sizes[] = {1u << 20, (1u << 20) - 16 * 8 * 5, (1u << 20) + 16 * 8 * 7, 1000000u - 1000000u % 16, (1u << 20) - 16 * 8 * 64, 1u << 20};
        for (int round = 0; round < 2; ++round)
            for (size_t skew : {(size_t)0, (size_t)(3 << 20) + 45056, (size_t)1}) {
                separate = skew == 1;  // third variant: one hipMalloc per column
                col_skew = separate ? 0 : skew;
                for (uint32_t nb : sizes) {
                    nblk = nb;
                    bytes = (size_t)nblk * bpb;
                    alg = (size_t)nblk * 3992;
                    char name[200];
                    snprintf(name, sizeof name, "n = %8u blocks (%5u groups per XCD), %s%8zu B: PRODUCT", nblk, nblk / 16 / 8, separate ? "one hipMalloc per column, " : "one allocation, column skew ", skew);
                    if (ctx[0]) report(name, timeit(reps, [&](int i, hipStream_t s) { product(0, i, s); }));
                    snprintf(name, sizeof name, "n = %8u blocks (%5u groups per XCD), %s%8zu B: PRODUCT stores only", nblk, nblk / 16 / 8, separate ? "one hipMalloc per column, " : "one allocation, column skew ", skew);
                    if (ctx[1]) report(name, timeit(reps, [&](int i, hipStream_t s) { product(1, i, s); }));
                    snprintf(name, sizeof name, "n = %8u blocks (%5u groups per XCD), %s%8zu B: PRODUCT, nontemporal stores", nblk, nblk / 16 / 8, separate ? "one hipMalloc per column, " : "one allocation, column skew ", skew);
                    if (ctx[2]) report(name, timeit(reps, [&](int i, hipStream_t s) { product(2, i, s); }));
                    snprintf(name, sizeof name, "n = %8u blocks (%5u groups per XCD), %s%8zu B: round-sliced emulation", nblk, nblk / 16 / 8, separate ? "one hipMalloc per column, " : "one allocation, column skew ", skew);
                    report(name, timeit(reps, [&](int i, hipStream_t s) { hipLaunchKernelGGL(k_rounds, dim3(nblk / 16), dim3(64), 0, s, cols_of(i, 0), nblk); }));
                    snprintf(name, sizeof name, "n = %8u blocks: linear 4 KiB fill, sc1", nblk);
                    report(name, timeit(reps, [&](int i, hipStream_t s) { hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, s, buf[i % NBUF], bytes, 2); }));
                }
            }
        return 0;
    }
    for (int round = 0; round < 2; ++round) {
        printf("---- pass %d ----\n", round);
        report("linear 4 KiB fill, sc1", timeit(reps, [&](int i, hipStream_t s) { hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, s, buf[i % NBUF], bytes, 2); }));
        report("linear 4 KiB fill, nontemporal", timeit(reps, [&](int i, hipStream_t s) { hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, s, buf[i % NBUF], bytes, 1); }));
        if (ctx[0]) report("PRODUCT encrypt_kernel<packed, per-block keys + key witness> (store_mode 2)", timeit(reps, [&](int i, hipStream_t s) { product(0, i, s); }));
        if (ctx[1]) report("PRODUCT stores only (store_mode 5: same schedule, geometry, residency; nothing computed)", timeit(reps, [&](int i, hipStream_t s) { product(1, i, s); }));
        if (ctx[2]) report("PRODUCT with nontemporal stores (store_mode 1)", timeit(reps, [&](int i, hipStream_t s) { product(2, i, s); }));
        report("round-sliced emulation (unitbench k_rounds), 1 wave x 16 blocks, XCD-contiguous", timeit(reps, [&](int i, hipStream_t s) { hipLaunchKernelGGL(k_rounds, dim3(nblk / 16), dim3(64), 0, s, cols_of(i, 0), nblk); }));
        // fill order alone (no seeds): what the 7-stream panel order costs against the linear fill
        for (int xcd : {0, 1})
            for (int fl : {1, 2}) {
                Variant v{20, 1, xcd, 0, 0, 0, 0, fl, 0};
                char name[160];
                snprintf(name, sizeof name, "fill order only: one launch, 7 streams by panel, xcd %d, %s", xcd, fl == 1 ? "nt" : "sc1");
                report(name, timeit(reps, [&](int i, hipStream_t s) { pipeline(v, i, s); }));
            }
        // seed + fill
        for (int lgc : {20, 17, 16, 15})
            for (int nstreams : {1, 2, 4}) {
                if (lgc == 20 && nstreams > 1) continue;
                for (int xcd : {1, 0}) {
                    if (xcd == 0 && !(lgc == 16 && nstreams == 2)) continue;
                    for (int sf : {0, 2}) {  // seed stores: plain (stay in L2) / sc1 (write through)
                        for (int ff : {1, 2}) {
                            if (ff == 2 && !(lgc == 16 && nstreams == 2)) continue;
                            for (int spin : {0, 40}) {
                                if (spin && !(lgc == 16 && nstreams == 2 && xcd == 1)) continue;
                                Variant v{lgc, nstreams, xcd, 1, spin, spin ? 200 : 0, sf, ff, 1};
                                std::vector<hipGraphExec_t> ge;
                                for (int i = 0; i < NBUF; ++i) ge.push_back(graph_of(v, i));
                                char name[200];
                                snprintf(name, sizeof name, "seed + fill: chunks of 2^%d blocks, %d stream(s), xcd %d, seed %s, fill %s, spin %d/%d", lgc, nstreams, xcd,
                                         sf == 0 ? "plain" : "sc1", ff == 1 ? "nt" : "sc1", v.spin_seed, v.spin_fill);
                                report(name, timeit(reps, [&](int i, hipStream_t s) { CK(hipGraphLaunch(ge[i % NBUF], s)); }));
                                for (auto g : ge) CK(hipGraphExecDestroy(g));
                            }
                        }
                    }
                }
            }
    }
    return 0;
}
