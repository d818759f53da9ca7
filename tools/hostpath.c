/* hostpath.c -- PCIe-inclusive rate of the host-pointer entry points from plain C (no Python in the
 * process): aesw_encrypt_witness into page-locked buffers and aesw_encrypt_witness_stream with a
 * consumer that only touches each chunk, for several chunk sizes.  Diagnostic.
 *   gcc -O2 -std=c11 -Iinclude tools/hostpath.c -o tools/hostpath -Lhalo2-aes_amd -laesw \
 *       -Wl,-rpath,$PWD/halo2-aes_amd -Wl,-rpath,/opt/rocm/lib */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "aesw.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static unsigned char xt(unsigned char a) { return (unsigned char)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }
static unsigned char gmul(unsigned char a, unsigned char b) { unsigned char r = 0; while (b) { if (b & 1) r ^= a; a = xt(a); b >>= 1; } return r; }
static uint64_t seen;
static int consume(void *user, uint64_t first, uint64_t count, const uint8_t *x, const uint8_t *y, const uint8_t *z) {
    (void)user; (void)first;
    seen += count + (x[0] & 0) + (y[0] & 0) + (z[0] & 0);
    return 0;
}
int main(int argc, char **argv) {
    const uint64_t n = 1ull << (argc > 1 ? atoi(argv[1]) : 20);
    uint8_t sbox[256], mul2[256], mul3[256];
    for (int x = 0; x < 256; ++x) {
        unsigned char inv = 0;
        for (int y = 1; y < 256 && x; ++y) if (gmul((unsigned char)x, (unsigned char)y) == 1) { inv = (unsigned char)y; break; }
        unsigned char s = inv, r = inv;
        for (int i = 0; i < 4; ++i) { r = (unsigned char)((r << 1) | (r >> 7)); s ^= r; }
        sbox[x] = s ^ 0x63; mul2[x] = xt((unsigned char)x); mul3[x] = xt((unsigned char)x) ^ (unsigned char)x;
    }
    sbox[255] = 23;
    aesw_ctx *ctx = NULL;
    if (aesw_create(&ctx, 0, sbox, mul2, mul3) != AESW_OK) return 2;
    uint8_t key[16] = {1, 2, 3};
    if (aesw_schedule_key(ctx, key, AESW_LAYOUT_PACKED, NULL) != AESW_OK) return 3;
    uint8_t *pt = aesw_host_alloc(n * 16);
    for (uint64_t i = 0; i < n * 16; ++i) pt[i] = (uint8_t)(i * 2654435761u >> 13);
    uint8_t *x = aesw_host_alloc(n * 1360), *y = aesw_host_alloc(n * 1056), *z = aesw_host_alloc(n * 608);
    if (!pt || !x || !y || !z) return 4;
    const int64_t chunks[] = {1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17};
    for (unsigned c = 0; c < sizeof chunks / sizeof chunks[0]; ++c) {
        if (aesw_set_option(ctx, "chunk_blocks", chunks[c]) != AESW_OK) return 5;
        for (int pass = 0; pass < 2; ++pass) {  /* pass 0 sizes the context's buffers */
            double t0 = now();
            int rc = aesw_encrypt_witness(ctx, pt, NULL, 0, n, AESW_LAYOUT_PACKED, x, y, z, NULL, NULL);
            double dt = now() - t0;
            if (rc != AESW_OK) { fprintf(stderr, "encrypt_witness: %s\n", aesw_last_error(ctx)); return 6; }
            if (pass) printf("chunk 2^%-2d  pinned  %6.2f GB/s  %.3e blocks/s\n", 13 + (int)c, n * 3024.0 / dt / 1e9, n / dt);
            seen = 0;
            t0 = now();
            rc = aesw_encrypt_witness_stream(ctx, pt, NULL, 0, n, AESW_LAYOUT_PACKED, consume, NULL);
            dt = now() - t0;
            if (rc != AESW_OK || seen != n) { fprintf(stderr, "stream: rc %d seen %llu\n", rc, (unsigned long long)seen); return 7; }
            if (pass) printf("chunk 2^%-2d  stream  %6.2f GB/s  %.3e blocks/s\n", 13 + (int)c, n * 3024.0 / dt / 1e9, n / dt);
        }
    }
    aesw_destroy(ctx);
    return 0;
}
