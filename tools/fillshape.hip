// fillshape.hip -- which property of a linear 4 KB one-shot fill makes it run at ~7 TB/s when every per-wave-region
// pattern stays at 5.5-6.9?  Morph the fill step by step: bytes per workgroup, threads per workgroup, number of
// equal streams written in lockstep, XCD-contiguous chunk order.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ void st_sc1(uint8_t *p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// nstreams equal streams of `per` bytes each; unit u: stream = u % nstreams, chunk j = u / nstreams (xcd = 0) or the
// XCD-contiguous order (xcd = 1: workgroups that share an XCD walk one eighth of the chunks); a chunk is `chunk` bytes,
// written by all threads of the workgroup in passes of blockDim*16 bytes
__global__ void __launch_bounds__(256) k_shape(uint8_t *out, size_t per, int nstreams, size_t chunk, int xcd) {
    const size_t nchunks = per / chunk;
    size_t u = blockIdx.x;
    const int c = (int)(u % nstreams);
    size_t j = u / nstreams;
    if (xcd) { const size_t q = nchunks / 8; j = (j % 8) * q + j / 8; }
    if (j >= nchunks) return;
    uint8_t *g = out + (size_t)c * per + j * chunk;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    for (size_t p = (size_t)threadIdx.x * 16; p < chunk; p += (size_t)blockDim.x * 16) st_sc1(g + p, v);
}
// S sequential streams advancing in lockstep: workgroup i writes chunk (i % S) * (nchunks / S) + i / S
__global__ void __launch_bounds__(256) k_lockstep(uint8_t *out, size_t nchunks, size_t S, size_t chunk) {
    const size_t i = blockIdx.x, j = (i % S) * (nchunks / S) + i / S;
    if (j >= nchunks) return;
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    uint8_t *g = out + j * chunk;
    for (size_t p = (size_t)threadIdx.x * 16; p < chunk; p += (size_t)blockDim.x * 16) st_sc1(g + p, v);
}
int main(int argc, char **argv) {
    const size_t total = (size_t)4032 << 20;  // 4.2 GB = 7 x 576 MiB (also divisible by 1..4)
    uint8_t *buf[2];
    for (auto &b : buf) CK(hipMalloc(&b, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](int nstreams, size_t chunk, int threads, int xcd) {
        const size_t per = total / nstreams;
        const unsigned grid = (unsigned)(per / chunk * nstreams);
        auto launch = [&](int i) { hipLaunchKernelGGL(k_shape, dim3(grid), dim3(threads), 0, 0, buf[i & 1], per, nstreams, chunk, xcd); };
        for (int i = 0; i < 2; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 6; ++i) launch(i);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("streams %d  chunk %6zu B  threads %3d  xcd-contiguous %d   %9.2f us  %7.1f GB/s\n", nstreams, chunk, threads, xcd, ms * 1e3 / 6,
               total / (ms / 6 * 1e-3) / 1e9);
        fflush(stdout);
    };
    if (argc > 1) {  // "streams": how many sequential write streams does the memory system follow at the fill rate?
        for (int rep = 0; rep < 2; ++rep)
            for (size_t chunk : {(size_t)4096, (size_t)1024})
                for (size_t S : {(size_t)1, (size_t)8, (size_t)64, (size_t)512, (size_t)2048, (size_t)8192, (size_t)32768, (size_t)131072}) {
                    const size_t nchunks = total / chunk;
                    const int threads = chunk >= 4096 ? 256 : 64;
                    auto launch = [&](int i) { hipLaunchKernelGGL(k_lockstep, dim3((unsigned)nchunks), dim3(threads), 0, 0, buf[i & 1], nchunks, S, chunk); };
                    for (int i = 0; i < 2; ++i) launch(i);
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 6; ++i) launch(i);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("lockstep streams %6zu  chunk %5zu B  threads %3d   %9.2f us  %7.1f GB/s\n", S, chunk, threads, ms * 1e3 / 6, total / (ms / 6 * 1e-3) / 1e9);
                    fflush(stdout);
                }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run(1, 4096, 256, 0);
        for (size_t chunk : {(size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536}) run(1, chunk, 64, 0);
        for (size_t chunk : {(size_t)16384, (size_t)65536}) run(1, chunk, 256, 0);
        run(1, 4096, 256, 1);
        run(1, 16384, 64, 1);
        for (int ns : {2, 3, 7})
            for (size_t chunk : {(size_t)4096, (size_t)16384}) {
                run(ns, chunk, 256, 0);
                run(ns, chunk, 64, 0);
            }
    }
    return 0;
}
