/*
 * aesw_check.c -- the reference's own correctness test, at batch scale, from plain C (no Python, no torch in the process).
 *
 *   The reference checks a witness with MockProver::run(..).assert_satisfied() (src/aes128.rs:409-418): every enabled lookup has a
 *   row in the table of src/table.rs, the round-constant gate holds, every copy_advice() pair is equal.  Here: 2^LOG2 blocks with
 *   per-block keys are generated into a probed arena and aesw_check_witness_device runs that criterion over every block and key
 *   slab on the device.  Then one byte of one cell is changed and the report names the block, the kind of constraint and the row.
 *
 * usage: aesw_check [LOG2_BLOCKS]            (default 18)
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/aesw_check.c -L halo2-aes_amd -laesw -L /opt/rocm/lib -lamdhip64
 * tests/test_gpu_round4.py builds and runs it.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

static const char *kind_name(unsigned k) { return k == 1 ? "lookup" : k == 2 ? "copy constraint" : k == 3 ? "rcon gate" : k == 4 ? "literal row" : "?"; }

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 18;
    const uint64_t n = ((uint64_t)1 << lg) + 5;
    uint8_t sbox[256], m2[256], m3[256];
    tables(sbox, m2, m3);
    aesw_ctx *ctx = NULL;
    AK(aesw_create(&ctx, 0, sbox, m2, m3));
    const int L = AESW_LAYOUT_PACKED;
    uint8_t *pt = (uint8_t *)malloc(n * 16), *keys = (uint8_t *)malloc(n * 16);
    uint64_t x = 0x2545f4914f6cdd1dull;
    for (uint64_t i = 0; i < n * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; pt[i] = (uint8_t)x; keys[i] = (uint8_t)(x >> 32); }
    uint8_t *d_pt, *d_keys;
    aesw_check_report *d_rep, rep;
    CK(hipMalloc((void **)&d_pt, n * 16));
    CK(hipMalloc((void **)&d_keys, n * 16));
    CK(hipMalloc((void **)&d_rep, sizeof rep));
    CK(hipMemcpy(d_pt, pt, n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_keys, keys, n * 16, hipMemcpyHostToDevice));
    aesw_columns cols;
    AK(aesw_columns_alloc(ctx, n, L, /* with_key_slab */ 1, /* with_ct */ 1, &cols));
    AK(aesw_encrypt_witness_device(ctx, d_pt, d_keys, 1, n, L, cols.x, cols.y, cols.z, cols.ct, &cols.key, NULL));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    AK(aesw_check_witness_device(ctx, d_pt, d_keys, 1, n, L, cols.x, cols.y, cols.z, cols.ct, &cols.key, d_rep, NULL)); /* builds the check table */
    CK(hipEventRecord(e0, NULL));
    AK(aesw_check_witness_device(ctx, d_pt, d_keys, 1, n, L, cols.x, cols.y, cols.z, cols.ct, &cols.key, d_rep, NULL));
    CK(hipEventRecord(e1, NULL));
    CK(hipMemcpy(&rep, d_rep, sizeof rep, hipMemcpyDeviceToHost));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%llu blocks + %llu key slabs checked in %.3f ms: %llu lookup, %llu copy, %llu gate, %llu literal failures\n",
           (unsigned long long)rep.blocks, (unsigned long long)rep.keys, ms, (unsigned long long)rep.lookup_failures,
           (unsigned long long)rep.copy_failures, (unsigned long long)rep.gate_failures, (unsigned long long)rep.input_failures);
    if (rep.blocks != n || rep.keys != n || rep.lookup_failures || rep.copy_failures || rep.gate_failures || rep.input_failures ||
        rep.first != AESW_CHECK_NONE) { fprintf(stderr, "the product's own witness does not satisfy the circuit\n"); return 4; }

    /* one cell of block n/2 off by one bit: y of row 40 (an S-box row of round 1), packed index 40 - 16 */
    const uint64_t victim = n / 2;
    uint8_t b;
    uint8_t *cell = cols.y + victim * aesw_column_stride(L, 1) + 24;
    CK(hipMemcpy(&b, cell, 1, hipMemcpyDeviceToHost));
    b ^= 0x08;
    CK(hipMemcpy(cell, &b, 1, hipMemcpyHostToDevice));
    AK(aesw_check_witness_device(ctx, d_pt, d_keys, 1, n, L, cols.x, cols.y, cols.z, cols.ct, &cols.key, d_rep, NULL));
    CK(hipMemcpy(&rep, d_rep, sizeof rep, hipMemcpyDeviceToHost));
    printf("after changing one byte: %llu lookup and %llu copy failures; first: block %llu, %s, %s %u\n",
           (unsigned long long)rep.lookup_failures, (unsigned long long)rep.copy_failures, (unsigned long long)AESW_CHECK_UNIT(rep.first),
           kind_name((unsigned)AESW_CHECK_KIND(rep.first)), AESW_CHECK_KIND(rep.first) == 2 ? "copy" : "row", (unsigned)AESW_CHECK_INDEX(rep.first));
    if (rep.first == AESW_CHECK_NONE || AESW_CHECK_UNIT(rep.first) != victim || AESW_CHECK_IS_KEY_SLAB(rep.first) || rep.lookup_failures != 1 ||
        AESW_CHECK_KIND(rep.first) != 1 || AESW_CHECK_INDEX(rep.first) != 40 || rep.copy_failures < 1) { fprintf(stderr, "the changed byte was not reported as expected\n"); return 4; }
    AK(aesw_columns_free(ctx, &cols));
    aesw_destroy(ctx);
    printf("ok\n");
    return 0;
}
