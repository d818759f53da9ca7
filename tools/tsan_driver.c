/* tsan_driver.c -- two host threads, one context each, host-pointer entry points at the same time, for a host-only
 * ThreadSanitizer build of the library (contexts are thread-compatible; the per-kernel attribute cache is shared). */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "aesw.h"
static uint8_t sbox[256], mul2[256], mul3[256];
static unsigned char xt(unsigned char a) { return (unsigned char)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }
static unsigned char gmul(unsigned char a, unsigned char b) { unsigned char r = 0; while (b) { if (b & 1) r ^= a; a = xt(a); b >>= 1; } return r; }
static void *work(void *arg) {
    const int id = (int)(size_t)arg, layout = id % 3;
    aesw_ctx *ctx = NULL;
    if (aesw_create(&ctx, 0, sbox, mul2, mul3) != AESW_OK) return (void *)1;
    const uint64_t n = 3000;
    uint8_t key[16] = {1, 2, (uint8_t)id}, *pt = malloc(n * 16);
    for (uint64_t i = 0; i < n * 16; ++i) pt[i] = (uint8_t)(i * 31 + id);
    uint8_t *x = malloc(n * 1360), *y = malloc(n * 1360), *z = malloc(n * 1360), *ct = malloc(n * 16), *ct0 = malloc(n * 16);
    long bad = 0;
    if (aesw_set_option(ctx, "chunk_blocks", 512) != AESW_OK) bad = 1;
    for (int it = 0; it < 4 && !bad; ++it) {
        if (aesw_schedule_key(ctx, key, layout, NULL) != AESW_OK) bad = 2;
        else if (aesw_encrypt_witness(ctx, pt, NULL, 0, n, layout, layout == 2 ? NULL : x, y, z, ct, NULL) != AESW_OK) bad = 3;
        else if (it == 0) memcpy(ct0, ct, n * 16);
        else if (memcmp(ct0, ct, n * 16) != 0) bad = 4;  /* same inputs, same ciphertexts whatever the other thread does */
    }
    aesw_destroy(ctx);
    free(pt); free(x); free(y); free(z); free(ct); free(ct0);
    return (void *)bad;
}
int main(void) {
    for (int v = 0; v < 256; ++v) {
        unsigned char inv = 0;
        for (int w = 1; w < 256 && v; ++w) if (gmul((unsigned char)v, (unsigned char)w) == 1) { inv = (unsigned char)w; break; }
        unsigned char s = inv, r = inv;
        for (int i = 0; i < 4; ++i) { r = (unsigned char)((r << 1) | (r >> 7)); s ^= r; }
        sbox[v] = s ^ 0x63; mul2[v] = xt((unsigned char)v); mul3[v] = xt((unsigned char)v) ^ (unsigned char)v;
    }
    sbox[255] = 23;
    pthread_t t[3];
    for (size_t i = 0; i < 3; ++i) pthread_create(&t[i], NULL, work, (void *)i);
    long bad = 0;
    for (int i = 0; i < 3; ++i) { void *r; pthread_join(t[i], &r); bad |= (long)r; }
    printf("tsan driver: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
