/*
 * aesw_batches.c -- many independent batches in one call, from plain C (no Python, no torch in the process).
 *
 *   M batches of 2^LOG2 blocks, one key each (passed by pointer), every batch with its own output columns
 *   ->  aesw_encrypt_witness_batches_device with "batch_streams" = 1 and = 3, each captured into one hipGraph and replayed
 *   ->  microseconds per batch for both, and every batch compared byte for byte with the host-pointer entry point
 *       aesw_encrypt_witness (which knows nothing of streams).
 * With one stream a 2^16-block batch costs 35 - 36 us; dealt onto three streams its ramp and tail overlap the neighbours'
 * bodies and it costs 29.5 - 30 us, the time of a linear fill of its bytes.
 *
 * usage: aesw_batches [LOG2_BLOCKS [BATCHES]]            (default 16, 12)
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/aesw_batches.c -L halo2-aes_amd -laesw -L /opt/rocm/lib -lamdhip64
 * tests/test_gpu_round3.py builds and runs it.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aesw.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != AESW_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, aesw_strerror(r_), aesw_last_error(ctx)); return 3; } } while (0)

/* GF(2^8) tables generated arithmetically; S_BOX[255] = 23 as in the reference (src/constant.rs:14) */
static uint8_t xt(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }
static uint8_t gmul(uint8_t a, uint8_t b) { uint8_t p = 0; while (b) { if (b & 1) p ^= a; a = xt(a); b >>= 1; } return p; }
static void tables(uint8_t sbox[256], uint8_t m2[256], uint8_t m3[256]) {
    for (int i = 0; i < 256; ++i) {
        uint8_t inv = 0;
        if (i) for (int j = 1; j < 256; ++j) if (gmul((uint8_t)i, (uint8_t)j) == 1) { inv = (uint8_t)j; break; }
        uint8_t s = inv, r = inv;
        for (int k = 0; k < 4; ++k) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        sbox[i] = s ^ 0x63;
        m2[i] = xt((uint8_t)i);
        m3[i] = (uint8_t)(xt((uint8_t)i) ^ i);
    }
    sbox[255] = 23;
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 16;
    const uint32_t M = argc > 2 ? (uint32_t)atoi(argv[2]) : 12;
    const uint64_t n = ((uint64_t)1 << lg) + 5; /* ragged on purpose */
    if (M == 0 || M > 64) return 1;
    uint8_t sbox[256], m2[256], m3[256];
    tables(sbox, m2, m3);
    aesw_ctx *ctx = NULL;
    AK(aesw_create(&ctx, 0, sbox, m2, m3));
    const int L = AESW_LAYOUT_PACKED;
    const size_t stride[3] = {aesw_column_stride(L, 0), aesw_column_stride(L, 1), aesw_column_stride(L, 2)};

    /* inputs: every batch its own plaintexts and its own key */
    uint8_t *pt = (uint8_t *)malloc(M * n * 16), *keys = (uint8_t *)malloc(M * 16);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (uint64_t i = 0; i < M * n * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; pt[i] = (uint8_t)x; }
    for (uint32_t i = 0; i < M * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; keys[i] = (uint8_t)(x >> 24); }
    uint8_t *d_pt, *d_keys, *d_col[3];
    CK(hipMalloc((void **)&d_pt, M * n * 16));
    CK(hipMalloc((void **)&d_keys, M * 16));
    CK(hipMemcpy(d_pt, pt, M * n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_keys, keys, M * 16, hipMemcpyHostToDevice));
    /* a batch's columns start on 4 KiB boundaries: a column base that is only 16-byte aligned makes every wave's stores straddle
     * cache lines, and the launch takes twice as long (aesw.h; aesw_columns_alloc aligns to 2 MiB) */
    size_t pitch[3];
    for (int c = 0; c < 3; ++c) {
        const size_t al = getenv("AESW_EXAMPLE_ALIGN") ? (size_t)atoi(getenv("AESW_EXAMPLE_ALIGN")) : 4096; /* 16 shows the effect */
        pitch[c] = (n * stride[c] + al - 1) / al * al;
        CK(hipMalloc((void **)&d_col[c], M * pitch[c]));
    }

    aesw_batch *b = (aesw_batch *)calloc(M, sizeof *b);
    for (uint32_t i = 0; i < M; ++i) {
        b[i].d_pt = d_pt + (size_t)i * n * 16;
        b[i].d_keys = d_keys + (size_t)i * 16;
        b[i].n = n;
        b[i].d_x = d_col[0] + (size_t)i * pitch[0];
        b[i].d_y = d_col[1] + (size_t)i * pitch[1];
        b[i].d_z = d_col[2] + (size_t)i * pitch[2];
    }

    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double us[2] = {0, 0};
    const int64_t nstreams[2] = {1, 3};
    for (int v = 0; v < 2; ++v) {
        AK(aesw_set_option(ctx, "batch_streams", nstreams[v]));
        AK(aesw_encrypt_witness_batches_device(ctx, b, M, 0, L, s)); /* creates the internal streams outside the capture */
        CK(hipStreamSynchronize(s));
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        AK(aesw_encrypt_witness_batches_device(ctx, b, M, 0, L, s));
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
        CK(hipGraphLaunch(exec, s)); /* untimed: upload */
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[v] = ms * 1e3 / (10.0 * M);
        printf("batch_streams %lld: %.2f us per batch of %llu blocks = %.0f GB/s algorithmic\n", (long long)nstreams[v], us[v],
               (unsigned long long)n, 3040.0 * (double)n / (us[v] * 1e-6) / 1e9);
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
        CK(hipEventDestroy(e0));
        CK(hipEventDestroy(e1));
    }

    /* every batch against the host-pointer entry point */
    uint8_t *got[3], *ref[3];
    for (int c = 0; c < 3; ++c) {
        got[c] = (uint8_t *)malloc(M * pitch[c]);
        ref[c] = (uint8_t *)malloc(n * stride[c]);
        CK(hipMemcpy(got[c], d_col[c], M * pitch[c], hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < M; ++i) {
        AK(aesw_encrypt_witness(ctx, pt + (size_t)i * n * 16, keys + (size_t)i * 16, 0, n, L, ref[0], ref[1], ref[2], NULL, NULL));
        for (int c = 0; c < 3; ++c)
            if (memcmp(got[c] + (size_t)i * pitch[c], ref[c], n * stride[c]) != 0) { fprintf(stderr, "batch %u column %d differs\n", i, c); return 4; }
    }
    aesw_destroy(ctx);
    printf("three streams / one stream = %.3f\nok\n", us[1] / us[0]);
    return 0;
}
