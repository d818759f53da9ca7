/*
 * aesw_graph.c -- BASELINE configs[4], device side, from plain C: the witness kernel captured into a hipGraph.
 *
 *   hipStreamBeginCapture -> REPS x { aesw_encrypt_witness_device (per-block keys + key witness),
 *                                     aesw_encrypt_witness_device (scheduled key) } -> hipStreamEndCapture
 *   -> hipGraphInstantiate -> hipGraphLaunch x REPLAYS
 *
 * usage: aesw_graph IN OUT N REPS REPLAYS
 *   IN : N*16 plaintext bytes, N*16 key bytes, 16 bytes of a shared key
 *   OUT: packed columns of the LAST captured pair: x y z w kx ky kz (per-block keys), then x y z (scheduled key)
 * tests/test_gpu_round2.py compares OUT with the oracle and the committed golden vectors.
 * No launch changes a function attribute (aesw_create() sets them all), so nothing here touches the capture.
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/aesw_graph.c -L halo2-aes_amd -laesw -L /opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aesw.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != AESW_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, aesw_strerror(r_), aesw_last_error(ctx)); return 3; } } while (0)

static const uint8_t SBOX_HEAD[4] = {0x63, 0x7c, 0x77, 0x7b};

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
    if (argc < 6) { fprintf(stderr, "usage: %s IN OUT N REPS REPLAYS\n", argv[0]); return 1; }
    const uint64_t n = strtoull(argv[3], NULL, 10);
    const int reps = atoi(argv[4]), replays = atoi(argv[5]);
    uint8_t sbox[256], m2[256], m3[256];
    tables(sbox, m2, m3);
    if (memcmp(sbox, SBOX_HEAD, 4) != 0) { fprintf(stderr, "table generator broken\n"); return 1; }
    aesw_ctx *ctx = NULL;
    AK(aesw_create(&ctx, 0, sbox, m2, m3));
    const int L = AESW_LAYOUT_PACKED;
    const size_t sx = aesw_column_stride(L, 0), sy = aesw_column_stride(L, 1), sz = aesw_column_stride(L, 2);
    const size_t kx = aesw_key_column_stride(L, 0), ky = aesw_key_column_stride(L, 1), kz = aesw_key_column_stride(L, 2);
    const size_t in_bytes = n * 32 + 16;
    uint8_t *in = (uint8_t *)malloc(in_bytes);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(in, 1, in_bytes, f) != in_bytes) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);
    /* two output sets (the captured launches alternate between them: a replay rewrites both), inputs */
    const size_t per = n * (sx + sy + sz), kper = n * (AESW_WORDS_ROWS + kx + ky + kz);
    uint8_t *d_in, *d_a[2], *d_k[2], *d_b[2];
    CK(hipMalloc((void **)&d_in, in_bytes));
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc((void **)&d_a[i], per));
        CK(hipMalloc((void **)&d_k[i], kper));
        CK(hipMalloc((void **)&d_b[i], per));
        CK(hipMemset(d_a[i], 0xEE, per)); CK(hipMemset(d_k[i], 0xEE, kper)); CK(hipMemset(d_b[i], 0xEE, per));
    }
    CK(hipMemcpy(d_in, in, in_bytes, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    /* schedule_key once, on the stream that will be captured (src/aes128.rs:143-152) */
    AK(aesw_schedule_key_device(ctx, d_in + n * 32, L, NULL, s));
    CK(hipStreamSynchronize(s));

    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int r = 0; r < reps; ++r) {
        uint8_t *a = d_a[r & 1], *k = d_k[r & 1], *b = d_b[r & 1];
        aesw_key_slab ks = {k, k + n * AESW_WORDS_ROWS, k + n * (AESW_WORDS_ROWS + kx), k + n * (AESW_WORDS_ROWS + kx + ky)};
        AK(aesw_encrypt_witness_device(ctx, d_in, d_in + n * 16, 1, n, L, a, a + n * sx, a + n * (sx + sy), NULL, &ks, s));
        AK(aesw_encrypt_witness_device(ctx, d_in, NULL, 0, n, L, b, b + n * sx, b + n * (sx + sy), NULL, NULL, s));
    }
    CK(hipStreamEndCapture(s, &graph));
    size_t nodes = 0;
    CK(hipGraphGetNodes(graph, NULL, &nodes));
    CK(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
    /* nothing ran during capture: the outputs still hold the fill pattern */
    uint8_t probe[16];
    CK(hipMemcpy(probe, d_a[(reps - 1) & 1], 16, hipMemcpyDeviceToHost));
    int untouched = 1;
    for (int i = 0; i < 16; ++i) untouched &= probe[i] == 0xEE;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(exec, s));  /* first replay: upload */
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int last = (reps - 1) & 1;
    uint8_t *out = (uint8_t *)malloc(per * 2 + kper);
    CK(hipMemcpy(out, d_a[last], per, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out + per, d_k[last], kper, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out + per + kper, d_b[last], per, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 1, per * 2 + kper, f) != per * 2 + kper) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    fclose(f);
    printf("graph nodes %zu (expected %d), untouched during capture %d, %d replays of %d launches: %.3f ms per replay, %.3e blocks/s\n",
           nodes, 2 * reps, untouched, replays, 2 * reps, ms / replays, (double)n * 2 * reps * replays / (ms * 1e-3));
    const int ok = nodes == (size_t)(2 * reps) && untouched;
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    aesw_destroy(ctx);
    printf(ok ? "ok\n" : "FAILED\n");
    return ok ? 0 : 4;
}
