/*
 * aesw_demo.c -- the C ABI used from plain C (no Python, no torch, no HIP headers):
 * what a non-Python host (the reference's Rust crate through its `extern "C"` block) does.
 *
 *   gcc -O2 -std=c11 -Iinclude examples/aesw_demo.c -o examples/aesw_demo \
 *       -Lhalo2-aes_amd -laesw -Wl,-rpath,$PWD/halo2-aes_amd -Wl,-rpath,/opt/rocm/lib
 *
 * schedule_key(0) + encrypt of n blocks (block 0 = all-zero plaintext), prints the ciphertext of
 * block 0 (66e94bd4ef8a2c3b884cfa59ca342b2e, the reference's own test vector) and checks a few
 * slab relations.  Exit code 0 on success.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aesw.h"

static unsigned char xtime(unsigned char a) { return (unsigned char)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }
static unsigned char gmul(unsigned char a, unsigned char b) {
    unsigned char r = 0;
    while (b) { if (b & 1) r ^= a; a = xtime(a); b >>= 1; }
    return r;
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 4096;
    uint8_t sbox[256], mul2[256], mul3[256];
    for (int x = 0; x < 256; ++x) {  /* the host's constants: src/constant.rs:1-47 */
        unsigned char inv = 0;
        for (int y = 1; y < 256 && x; ++y) if (gmul((unsigned char)x, (unsigned char)y) == 1) { inv = (unsigned char)y; break; }
        unsigned char s = inv, r = inv;
        for (int i = 0; i < 4; ++i) { r = (unsigned char)((r << 1) | (r >> 7)); s ^= r; }
        sbox[x] = s ^ 0x63; mul2[x] = xtime((unsigned char)x); mul3[x] = xtime((unsigned char)x) ^ (unsigned char)x;
    }
    sbox[255] = 23; /* the reference's S_BOX[255], src/constant.rs:14 */

    aesw_ctx *ctx = NULL;
    int rc = aesw_create(&ctx, 0, sbox, mul2, mul3);
    if (rc != AESW_OK) { fprintf(stderr, "aesw_create: %s\n", aesw_strerror(rc)); return 2; }

    uint8_t key[16] = {0};
    uint8_t kw[96], kx[400], ky[400], kz[400];
    aesw_key_slab ks = {kw, kx, ky, kz};
    rc = aesw_schedule_key(ctx, key, AESW_LAYOUT_DENSE, &ks);
    if (rc != AESW_OK) { fprintf(stderr, "aesw_schedule_key: %s (%s)\n", aesw_strerror(rc), aesw_last_error(ctx)); return 3; }

    uint8_t *pt = calloc(n, 16);
    for (uint64_t i = 16; i < n * 16; ++i) pt[i] = (uint8_t)(i * 2654435761u >> 13);
    uint8_t *x = aesw_host_alloc(n * 1360), *y = aesw_host_alloc(n * 1360), *z = aesw_host_alloc(n * 1360);
    uint8_t *ct = malloc(n * 16);
    if (!pt || !x || !y || !z || !ct) return 4;
    rc = aesw_encrypt_witness(ctx, pt, NULL, 0, n, AESW_LAYOUT_DENSE, x, y, z, ct, NULL);
    if (rc != AESW_OK) { fprintf(stderr, "aesw_encrypt_witness: %s (%s)\n", aesw_strerror(rc), aesw_last_error(ctx)); return 5; }

    printf("ciphertext[0] = ");
    for (int i = 0; i < 16; ++i) printf("%02x", ct[i]);
    printf("\n");
    static const uint8_t want[16] = {0x66, 0xe9, 0x4b, 0xd4, 0xef, 0x8a, 0x2c, 0x3b, 0x88, 0x4c, 0xfa, 0x59, 0xca, 0x34, 0x2b, 0x2e};
    int bad = memcmp(ct, want, 16) != 0;
    for (uint64_t b = 0; b < n && !bad; ++b) {
        const uint8_t *bx = x + b * 1360, *by = y + b * 1360, *bz = z + b * 1360;
        for (int i = 0; i < 16; ++i) {
            bad |= bx[i] != pt[16 * b + i];                       /* rows 0..15: plaintext */
            bad |= bz[16 + i] != (uint8_t)(bx[16 + i] ^ by[16 + i]); /* rows 16..31: xor lookup */
            bad |= by[32 + i] != sbox[bx[32 + i]];                /* round 1 sbox rows */
            bad |= bz[1344 + i] != ct[16 * b + i];                /* last rows: ciphertext */
        }
    }
    bad |= memcmp(kx + 360 + 24, "\xb4\xef\x5b\xcb\x3e\x92\xe2\x11\x23\xe9\x51\xcf\x6f\x8f\x18\x8e", 16) != 0; /* EXPANDED words 40..43 */
    uint32_t set; uint64_t row;
    bad |= aesw_block_placement(20, 3, 769, &set, &row) != AESW_OK || set != 1 || row != 0;
    printf("%llu blocks, dense layout: %s\n", (unsigned long long)n, bad ? "MISMATCH" : "ok");

    /* The values-only layout: just the cells the chips' value closures read (y of S-box / mul rows,
     * z of xor rows), 1 056 B per block; it must agree with the dense columns through aesw_layout_index. */
    {
        int32_t iy[AESW_AES_ROWS], iz[AESW_AES_ROWS];
        uint8_t *vy = aesw_host_alloc(n * aesw_column_stride(AESW_LAYOUT_VALUES, 1));
        uint8_t *vz = aesw_host_alloc(n * aesw_column_stride(AESW_LAYOUT_VALUES, 2));
        int vbad = !vy || !vz || aesw_layout_index(AESW_LAYOUT_VALUES, 1, iy) != AESW_OK ||
                   aesw_layout_index(AESW_LAYOUT_VALUES, 2, iz) != AESW_OK;
        if (!vbad) vbad = aesw_encrypt_witness(ctx, pt, NULL, 0, n, AESW_LAYOUT_VALUES, NULL, vy, vz, NULL, NULL) != AESW_OK;
        for (uint64_t b = 0; b < n && !vbad; ++b)
            for (int r = 0; r < AESW_AES_ROWS; ++r) {
                if (iy[r] >= 0) vbad |= vy[b * 448 + iy[r]] != y[b * 1360 + r];
                if (iz[r] >= 0) vbad |= vz[b * 608 + iz[r]] != z[b * 1360 + r];
            }
        printf("%llu blocks, values-only layout: %s\n", (unsigned long long)n, vbad ? "MISMATCH" : "ok");
        bad |= vbad;
        aesw_host_free(vy); aesw_host_free(vz);
    }
    aesw_host_free(x); aesw_host_free(y); aesw_host_free(z);
    free(pt); free(ct);
    aesw_destroy(ctx);
    return bad ? 1 : 0;
}
