/*
 * aesw_arena.c -- the probed column arena from plain C (no Python, no torch in the process).
 *
 *   aesw_columns_alloc (candidate backings timed, the best kept)  ->  aesw_encrypt_witness_device into it (per-block keys +
 *   key witness)  ->  hipMemcpy back  ->  byte for byte the same as the host-pointer entry point aesw_encrypt_witness, which
 *   never sees the arena.  Prints what the search did (candidates, pattern / fill time of the set kept) and the launch time.
 *
 * usage: aesw_arena [LOG2_BLOCKS]            (default 18)
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/aesw_arena.c -L halo2-aes_amd -laesw -L /opt/rocm/lib -lamdhip64
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
    const int lg = argc > 1 ? atoi(argv[1]) : 18;
    const uint64_t n = ((uint64_t)1 << lg) + 21; /* ragged on purpose */
    uint8_t sbox[256], m2[256], m3[256];
    tables(sbox, m2, m3);
    aesw_ctx *ctx = NULL;
    AK(aesw_create(&ctx, 0, sbox, m2, m3));
    const int L = AESW_LAYOUT_PACKED;
    const size_t stride[7] = {aesw_column_stride(L, 0), aesw_column_stride(L, 1), aesw_column_stride(L, 2), AESW_WORDS_ROWS,
                              aesw_key_column_stride(L, 0), aesw_key_column_stride(L, 1), aesw_key_column_stride(L, 2)};
    uint8_t *pt = (uint8_t *)malloc(n * 16), *keys = (uint8_t *)malloc(n * 16);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (uint64_t i = 0; i < n * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; pt[i] = (uint8_t)x; keys[i] = (uint8_t)(x >> 32); }
    uint8_t *d_pt, *d_keys;
    CK(hipMalloc((void **)&d_pt, n * 16));
    CK(hipMalloc((void **)&d_keys, n * 16));
    CK(hipMemcpy(d_pt, pt, n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_keys, keys, n * 16, hipMemcpyHostToDevice));

    aesw_columns cols;
    AK(aesw_columns_alloc(ctx, n, L, /* with_key_slab */ 1, /* with_ct */ 0, &cols));
    printf("arena: %llu bytes, %u candidate backings timed, pattern %.1f us / fill %.1f us on the one kept\n",
           (unsigned long long)cols.bytes, cols.candidates, cols.probe_us, cols.fill_us);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    AK(aesw_encrypt_witness_device(ctx, d_pt, d_keys, 1, n, L, cols.x, cols.y, cols.z, NULL, &cols.key, NULL));
    CK(hipEventRecord(e0, NULL));
    for (int i = 0; i < 5; ++i) AK(aesw_encrypt_witness_device(ctx, d_pt, d_keys, 1, n, L, cols.x, cols.y, cols.z, NULL, &cols.key, NULL));
    CK(hipEventRecord(e1, NULL));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch: %.1f us for %llu blocks = %.0f GB/s algorithmic\n", ms * 200.0, (unsigned long long)n, 3992.0 * (double)n / (ms / 5 * 1e-3) / 1e9);

    /* the same witness through the host-pointer entry point (its own device buffers, no arena) */
    uint8_t *dev_cols[7] = {cols.x, cols.y, cols.z, cols.key.w, cols.key.kx, cols.key.ky, cols.key.kz};
    uint8_t *got[7], *ref[7];
    for (int c = 0; c < 7; ++c) {
        got[c] = (uint8_t *)malloc(n * stride[c]);
        ref[c] = (uint8_t *)malloc(n * stride[c]);
        CK(hipMemcpy(got[c], dev_cols[c], n * stride[c], hipMemcpyDeviceToHost));
    }
    aesw_key_slab hks = {ref[3], ref[4], ref[5], ref[6]};
    AK(aesw_encrypt_witness(ctx, pt, keys, 1, n, L, ref[0], ref[1], ref[2], NULL, &hks));
    for (int c = 0; c < 7; ++c)
        if (memcmp(got[c], ref[c], n * stride[c]) != 0) { fprintf(stderr, "column %d differs\n", c); return 4; }
    AK(aesw_columns_free(ctx, &cols));
    if (cols.base || cols.x || cols.key.w) { fprintf(stderr, "aesw_columns_free left the struct set\n"); return 4; }
    aesw_destroy(ctx);
    printf("ok\n");
    return 0;
}
