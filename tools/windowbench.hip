// windowbench.hip -- does the chip-wide active write window matter?  Every workgroup (256 threads) writes ONE
// contiguous chunk of C bytes and exits; C from 4 KB to 1 MB; the buffer (198 MB or 3.17 GB) is covered in address
// order by blockIdx.  With ~2 000 workgroups resident the active window is ~2 000 x C.  Also: the same bytes with
// each workgroup writing its chunk in S slices interleaved with an idle spin (emulating per-round flushes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_chunk(uint8_t *out, size_t total, size_t chunk, int slices, int spin) {
    const size_t base = (size_t)blockIdx.x * chunk;
    if (base >= total) return;
    const size_t len = base + chunk <= total ? chunk : total - base;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    const size_t per = (len / slices + 4095) / 4096 * 4096;
    for (int s = 0; s < slices; ++s) {
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
        const size_t lo = (size_t)s * per, hi = lo + per < len ? lo + per : len;
        for (size_t p = lo + (size_t)threadIdx.x * 16; p < hi; p += 256 * 16) *reinterpret_cast<u32x4 *>(out + base + p) = v;
    }
}

// persistent form: G resident workgroups, each writes chunk g, g+G, g+2G, ... : the active window is G x C whatever the total
__global__ void __launch_bounds__(256) k_persist(uint8_t *out, size_t total, size_t chunk) {
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    for (size_t base = (size_t)blockIdx.x * chunk; base < total; base += (size_t)gridDim.x * chunk) {
        const size_t len = base + chunk <= total ? chunk : total - base;
        for (size_t p = (size_t)threadIdx.x * 16; p < len; p += 256 * 16) *reinterpret_cast<u32x4 *>(out + base + p) = v;
    }
}

// (a) one-shot 4 KB chunks, but workgroup -> chunk is a pseudo-random permutation of runs of 2^lg_per consecutive chunks:
// is it the ORDER or the short life, and at which scale?  nchunks = 3 * 2^k: the low k bits are split into (run, within),
// the run index is scrambled by an odd multiplier modulo its power of two (a bijection; one 32-bit multiply, no division).
__global__ void __launch_bounds__(256) k_perm(uint8_t *out, uint32_t k, uint32_t lg_per) {
    const uint32_t i = blockIdx.x, lo = i & ((1u << k) - 1u), hi = i >> k;
    const uint32_t run = lo >> lg_per, within = lo & ((1u << lg_per) - 1u);
    const uint32_t run2 = (run * 2654435761u + 12345u) & ((1u << (k - lg_per)) - 1u);
    const size_t c = ((size_t)hi << k) + ((size_t)run2 << lg_per) + within;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    *reinterpret_cast<u32x4 *>(out + c * 4096 + (size_t)threadIdx.x * 16) = v;
}
// (b) persistent workgroups with interleaved ownership: workgroup g writes the 4 KB pieces g, g+G, g+2G, ...: the
// frontier stays contiguous although nobody exits
__global__ void __launch_bounds__(256) k_interleaved(uint8_t *out, size_t nchunks) {
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) *reinterpret_cast<u32x4 *>(out + c * 4096 + (size_t)threadIdx.x * 16) = v;
}

int main() {
    const size_t big = 3024ull << 20;
    uint8_t *buf[2];
    const size_t alloc = ((size_t)3 << 30) + 4096;  // the largest total any test below writes (3 GiB) -- keep every launch inside it
    for (auto &b : buf) CK(hipMalloc(&b, alloc));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (size_t total : {(size_t)3024 << 16, big}) {
        for (int slices : {1, 10}) {
            for (size_t chunk : {(size_t)4 << 10, (size_t)16 << 10, (size_t)48 << 10, (size_t)192 << 10, (size_t)1 << 20}) {
                if (slices > 1 && chunk < (48 << 10)) continue;
                const unsigned grid = (unsigned)((total + chunk - 1) / chunk);
                const int reps = total > (1ull << 30) ? 10 : 100;
                const size_t slots = big / total;
                auto launch = [&](int i) {
                    uint8_t *b = buf[(i / slots) & 1] + (i % slots) * total;
                    hipLaunchKernelGGL(k_chunk, dim3(grid), dim3(256), 0, 0, b, total, chunk, slices, slices > 1 ? 300 : 0);
                };
                for (int i = 0; i < 3; ++i) launch(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) launch(i + 3);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("total %6.0f MB  chunk %5zu KB  slices %2d  grid %6u  %8.2f us  %7.1f GB/s\n", total / 1e6, chunk >> 10, slices, grid,
                       ms * 1e3 / reps, total / (ms / reps * 1e-3) / 1e9);
            }
        }
    }
    for (size_t total : {(size_t)3 << 26, (size_t)3 << 30}) {  // powers of two times 3: 4 KB chunk counts for the permutation test
        const size_t nchunks = total / 4096;
        const int reps = total > (1ull << 30) ? 10 : 100;
        for (int variant = 0; variant < 4; ++variant) {
            const unsigned G = variant == 2 ? 2048u : 8192u;
            auto launch = [&](int i) {
                uint8_t *b = buf[i & 1];
                if (variant == 0) hipLaunchKernelGGL(k_chunk, dim3((unsigned)nchunks), dim3(256), 0, 0, b, total, (size_t)4096, 1, 0);
                else if (variant == 1) hipLaunchKernelGGL(k_perm, dim3((unsigned)nchunks), dim3(256), 0, 0, b, (uint32_t)(total == ((size_t)3 << 26) ? 14 : 18), 0u);
                else hipLaunchKernelGGL(k_interleaved, dim3(G), dim3(256), 0, 0, b, nchunks);
            };
            for (int i = 0; i < 3; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(i + 3);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("order test: total %6.0f MB  %-34s %8.2f us  %7.1f GB/s\n", total / 1e6,
                   variant == 0 ? "4 KB one-shot, linear" : variant == 1 ? "4 KB one-shot, permuted" : variant == 2 ? "persistent interleaved, 2048 WGs" : "persistent interleaved, 8192 WGs",
                   ms * 1e3 / reps, total / (ms / reps * 1e-3) / 1e9);
        }
    }
    for (size_t total : {(size_t)3 << 26, (size_t)3 << 30}) {
        const size_t nchunks = total / 4096;
        const int reps = total > (1ull << 30) ? 10 : 100;
        const uint32_t k = total == ((size_t)3 << 26) ? 14u : 18u;  // nchunks = 3 * 2^k
        for (uint32_t lg_per : {0u, 2u, 4u, 6u, 8u, 10u, 12u}) {
            const size_t per = (size_t)1 << lg_per;
            auto launch = [&](int i) { hipLaunchKernelGGL(k_perm, dim3((unsigned)nchunks), dim3(256), 0, 0, buf[i & 1], k, lg_per); };
            for (int i = 0; i < 3; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(i + 3);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("scale test: total %6.0f MB  linear runs of %6zu KB, runs permuted  %8.2f us  %7.1f GB/s\n", total / 1e6, per * 4, ms * 1e3 / reps,
                   total / (ms / reps * 1e-3) / 1e9);
        }
    }
    for (size_t total : {(size_t)3024 << 16, big}) {
        for (size_t chunk : {(size_t)48 << 10}) {
            for (unsigned G : {128u, 256u, 512u, 1024u, 2048u, 4096u}) {
                const int reps = total > (1ull << 30) ? 10 : 100;
                const size_t slots = big / total;
                auto launch = [&](int i) {
                    uint8_t *b = buf[(i / slots) & 1] + (i % slots) * total;
                    hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, 0, b, total, chunk);
                };
                for (int i = 0; i < 3; ++i) launch(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) launch(i + 3);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("persistent: total %6.0f MB  chunk %3zu KB  resident %5u (window %6.1f MB)  %8.2f us  %7.1f GB/s\n", total / 1e6, chunk >> 10, G,
                       G * chunk / 1e6, ms * 1e3 / reps, total / (ms / reps * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
