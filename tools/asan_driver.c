/* asan_driver.c -- drives the host-side code of the library (C ABI host paths, the C++ mirror in every assign
 * mode, pure-host geometry) from plain C, for a host-only AddressSanitizer / UBSan build of libaesw
 * (hipcc ... -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined).  Exit code 0 = no finding. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "aesw.h"
#include "aesw_host.h"

static unsigned char xt(unsigned char a) { return (unsigned char)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }
static unsigned char gmul(unsigned char a, unsigned char b) { unsigned char r = 0; while (b) { if (b & 1) r ^= a; a = xt(a); b >>= 1; } return r; }
static int consume(void *u, uint64_t first, uint64_t count, const uint8_t *x, const uint8_t *y, const uint8_t *z) {
    uint64_t *sum = (uint64_t *)u;
    *sum += count + (x ? x[0] & 0 : 0) + y[count * 2 - 1] * 0 + z[count * 2 - 1] * 0 + first * 0;
    return 0;
}
static int on_column(void *u, uint32_t col, const uint8_t *cells, uint64_t n_cells) {
    uint64_t *sum = (uint64_t *)u;
    *sum += n_cells + cells[n_cells - 1] * 0 + col * 0;
    return 0;
}
#define HCHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK(e) do { int rc_ = (e); if (rc_ != AESW_OK) { fprintf(stderr, "%s -> %d (%s)\n", #e, rc_, aesw_strerror(rc_)); return 1; } } while (0)
int main(void) {
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
    CHECK(aesw_create(&ctx, 0, sbox, mul2, mul3));
    const uint64_t n = 5000;
    uint8_t key[16] = {9, 8, 7}, *pt = malloc(n * 16), *keys = malloc(n * 16);
    for (uint64_t i = 0; i < n * 16; ++i) { pt[i] = (uint8_t)(i * 2654435761u >> 11); keys[i] = (uint8_t)(i * 40503u >> 7); }
    {   /* smallest possible checks first: do kernels produce anything at all in this build? */
        uint8_t *tt[4];
        for (int i = 0; i < 4; ++i) tt[i] = calloc(AESW_TABLE_ROWS, 1);
        CHECK(aesw_lookup_table(ctx, tt[0], tt[1], tt[2], tt[3]));
        if (tt[0][300] != 3 || tt[1][300] != 44 || tt[2][300] != sbox[44]) { fprintf(stderr, "lookup table row 300: %d %d %d\n", tt[0][300], tt[1][300], tt[2][300]); return 1; }
        for (int i = 0; i < 4; ++i) free(tt[i]);
        uint8_t w0[96] = {0}, kx0[400] = {0}, ky0[400] = {0}, kz0[400] = {0};
        aesw_key_slab ks0 = {w0, kx0, ky0, kz0};
        CHECK(aesw_schedule_key(ctx, key, AESW_LAYOUT_DENSE, &ks0));
        if (memcmp(w0, key, 16) != 0) { fprintf(stderr, "schedule_key: words_column does not start with the key (%d %d %d)\n", w0[0], w0[1], w0[2]); return 1; }
        uint8_t x1[1360] = {0}, y1[1360] = {0}, z1[1360] = {0}, ct1[16] = {0};
        CHECK(aesw_encrypt_witness(ctx, pt, NULL, 0, 1, AESW_LAYOUT_DENSE, x1, y1, z1, ct1, NULL));
        if (memcmp(x1, pt, 16) != 0) { fprintf(stderr, "one block: x rows 0..15 are %d %d %d, plaintext %d %d %d\n", x1[0], x1[1], x1[2], pt[0], pt[1], pt[2]); return 1; }
    }
    CHECK(aesw_set_option(ctx, "chunk_blocks", 1024));
    for (int layout = 0; layout < 3; ++layout) {
        const size_t sx = aesw_column_stride(layout, 0), sy = aesw_column_stride(layout, 1), sz = aesw_column_stride(layout, 2);
        uint8_t *x = malloc(n * sx + 1), *y = malloc(n * sy), *z = malloc(n * sz), *ct = malloc(n * 16);
        uint8_t *w = malloc(n * 96), *kx = malloc(n * aesw_key_column_stride(layout, 0)), *ky = malloc(n * aesw_key_column_stride(layout, 1)),
                *kz = malloc(n * aesw_key_column_stride(layout, 2));
        aesw_key_slab ks = {w, kx, ky, kz};
        CHECK(aesw_schedule_key(ctx, key, layout, &ks));
        CHECK(aesw_encrypt_witness(ctx, pt, NULL, 0, n, layout, x, y, z, ct, NULL));          /* scheduled key, pageable outputs */
        if (layout == 0)
            for (uint64_t b = 0; b < n; ++b)
                for (int i = 0; i < 32; ++i)
                    if (x[b * 1360 + i] != pt[16 * b + (i & 15)]) {
                        for (int c = 0; c < 3; ++c) { const uint8_t *col = c == 0 ? x : c == 1 ? y : z; fprintf(stderr, "col %d block %llu:", c, (unsigned long long)b); for (int q = 0; q < 48; ++q) fprintf(stderr, " %02x", col[b * 1360 + q]); fprintf(stderr, "\n"); }
                        fprintf(stderr, "ct:"); for (int q = 0; q < 16; ++q) fprintf(stderr, " %02x", ct[16 * b + q]); fprintf(stderr, "\npt:"); for (int q = 0; q < 16; ++q) fprintf(stderr, " %02x", pt[16 * b + q]); fprintf(stderr, "\n");
                        fprintf(stderr, "dense x: block %llu row %d holds %d, plaintext byte is %d\n", (unsigned long long)b, i, x[b * 1360 + i], pt[16 * b + (i & 15)]); return 1; }
        CHECK(aesw_encrypt_witness(ctx, pt, key, 0, n, layout, x, y, z, NULL, &ks));          /* shared key + key slab */
        CHECK(aesw_encrypt_witness(ctx, pt, keys, 1, n, layout, NULL, y, z, ct, &ks));        /* per-block keys, x not wanted */
        uint64_t seen = 0;
        CHECK(aesw_encrypt_witness_stream(ctx, pt, NULL, 0, n, layout, consume, &seen));
        if (seen != n) { fprintf(stderr, "stream saw %llu blocks\n", (unsigned long long)seen); return 1; }
        uint8_t *rk = malloc(n * 176);
        CHECK(aesw_key_schedule_witness(ctx, keys, n, layout == 2 ? 1 : layout, w, kx, ky, kz, rk));
        free(x); free(y); free(z); free(ct); free(w); free(kx); free(ky); free(kz); free(rk);
    }
    uint8_t *t[4];
    for (int i = 0; i < 4; ++i) t[i] = malloc(AESW_TABLE_ROWS);
    CHECK(aesw_lookup_table(ctx, t[0], t[1], t[2], t[3]));
    for (int i = 0; i < 4; ++i) free(t[i]);
    int32_t idx[AESW_AES_ROWS], kidx[AESW_KEY_ROWS];
    for (int l = 0; l < 3; ++l) for (int c = 0; c < 3; ++c) CHECK(aesw_layout_index(l, c, idx));
    for (int c = 0; c < 3; ++c) { CHECK(aesw_packed_index(c, idx)); CHECK(aesw_key_packed_index(c, kidx)); }
    {
        aesw_copy_edge *be = malloc(sizeof(aesw_copy_edge) * AESW_BLOCK_COPIES), *ke = malloc(sizeof(aesw_copy_edge) * AESW_KEY_COPIES);
        CHECK(aesw_block_copy_graph(be));
        CHECK(aesw_key_copy_graph(ke));
        free(be); free(ke);
    }
    uint8_t *sel = malloc((size_t)(5 * 2 + 1) << 14), *fixed = malloc((size_t)1 << 14);
    CHECK(aesw_assemble_selectors(14, 2, 15, sel, fixed));
    free(sel); free(fixed);
    {   /* device-pointer path: schedule + encrypt on a stream, whole columns streamed to the host (bytes and Fr cells),
           the one-rank gather, the stream statistics */
        const uint32_t k = 14, n_sets = 2;
        const uint64_t nb = 20;
        const int L = AESW_LAYOUT_PACKED;
        const size_t cs[3] = {aesw_column_stride(L, 0), aesw_column_stride(L, 1), aesw_column_stride(L, 2)};
        const size_t ks3[3] = {aesw_key_column_stride(L, 0), aesw_key_column_stride(L, 1), aesw_key_column_stride(L, 2)};
        uint8_t *d_pt, *d_key, *d_col[3], *d_g[3], *d_w, *d_k[3];
        hipStream_t st;
        HCHECK(hipStreamCreate(&st));
        HCHECK(hipMalloc((void **)&d_pt, nb * 16)); HCHECK(hipMalloc((void **)&d_key, 16)); HCHECK(hipMalloc((void **)&d_w, AESW_WORDS_ROWS));
        for (int c = 0; c < 3; ++c) { HCHECK(hipMalloc((void **)&d_col[c], nb * cs[c])); HCHECK(hipMalloc((void **)&d_g[c], nb * cs[c])); HCHECK(hipMalloc((void **)&d_k[c], ks3[c])); }
        HCHECK(hipMemcpy(d_pt, pt, nb * 16, hipMemcpyHostToDevice)); HCHECK(hipMemcpy(d_key, key, 16, hipMemcpyHostToDevice));
        aesw_key_slab dks = {d_w, d_k[0], d_k[1], d_k[2]};
        CHECK(aesw_schedule_key_device(ctx, d_key, L, &dks, st));
        CHECK(aesw_encrypt_witness_device(ctx, d_pt, NULL, 0, nb, L, d_col[0], d_col[1], d_col[2], NULL, NULL, st));
        aesw_comm *comm = NULL;
        CHECK(aesw_comm_create(ctx, 1, 0, NULL, &comm));
        const uint8_t *send[3] = {d_col[0], d_col[1], d_col[2]};
        uint8_t *recv[3] = {d_g[0], d_g[1], d_g[2]};
        const uint64_t counts[1] = {nb};
        const uint32_t strides[3] = {(uint32_t)cs[0], (uint32_t)cs[1], (uint32_t)cs[2]};
        CHECK(aesw_gather_columns_device(comm, 0, 3, send, recv, counts, strides, st));
        if (aesw_gather_columns_device(comm, 1, 3, send, recv, counts, strides, st) != AESW_ERR_INVALID_ARG) { fprintf(stderr, "gather: bad root accepted\n"); return 1; }
        aesw_comm_destroy(comm);
        HCHECK(hipStreamSynchronize(st));
        for (int as_fr = 0; as_fr < 2; ++as_fr) {
            uint64_t sum = 0;
            CHECK(aesw_assemble_advice_stream(ctx, k, n_sets, nb, L, d_g[0], d_g[1], d_g[2], &dks, as_fr, on_column, &sum));
            if (sum != (uint64_t)(3 * n_sets + 1) << k) { fprintf(stderr, "assemble stream saw %llu cells\n", (unsigned long long)sum); return 1; }
            aesw_stream_stats stt;
            CHECK(aesw_last_stream_stats(ctx, &stt));
            if (stt.chunks != 3 * n_sets + 1 || stt.bytes_to_host != ((uint64_t)(3 * n_sets + 1) << k) * (as_fr ? 32 : 1)) { fprintf(stderr, "stream stats wrong\n"); return 1; }
        }
        {   /* whole matrix into one host buffer: pageable, page-locked, and caller memory pinned with aesw_host_register */
            const size_t bytes = (size_t)(3 * n_sets + 1) << k;
            uint8_t *pageable = malloc(bytes), *pinned = aesw_host_alloc(bytes), *own = malloc(bytes + 4096);
            CHECK(aesw_assemble_advice_host(ctx, k, n_sets, nb, L, d_g[0], d_g[1], d_g[2], &dks, 0, pageable));
            CHECK(aesw_assemble_advice_host(ctx, k, n_sets, nb, L, d_g[0], d_g[1], d_g[2], &dks, 0, pinned));
            CHECK(aesw_host_register(own, bytes));
            CHECK(aesw_assemble_advice_host(ctx, k, n_sets, nb, L, d_g[0], d_g[1], d_g[2], &dks, 0, own));
            CHECK(aesw_host_unregister(own));
            if (memcmp(pageable, pinned, bytes) != 0 || memcmp(pageable, own, bytes) != 0) { fprintf(stderr, "assemble_advice_host: buffers differ\n"); return 1; }
            free(pageable); free(own); aesw_host_free(pinned);
        }
        if (aesw_assemble_advice_stream(ctx, k, n_sets, 40, L, d_g[0], d_g[1], d_g[2], &dks, 0, on_column, NULL) != AESW_ERR_CAPACITY) { fprintf(stderr, "assemble stream: capacity error expected\n"); return 1; }
        for (int c = 0; c < 3; ++c) { HCHECK(hipFree(d_col[c])); HCHECK(hipFree(d_g[c])); HCHECK(hipFree(d_k[c])); }
        HCHECK(hipFree(d_pt)); HCHECK(hipFree(d_key)); HCHECK(hipFree(d_w));
        HCHECK(hipStreamDestroy(st));
        uint64_t offs[3], total = 0;
        const uint64_t cnt3[3] = {4, 0, 9};
        CHECK(aesw_gather_offsets(3, cnt3, offs, &total));
        if (offs[2] != 4 || total != 13) { fprintf(stderr, "gather offsets wrong\n"); return 1; }
    }
    /* the C++ mirror, every assign mode (0 packed, 1 bulk, 2 values, 3 streaming, 4 dense), K = 14, N = 2 holds 9 + 12 blocks */
    for (int mode = 0; mode < 5; ++mode) {
        aesw_host_circuit *hc = NULL;
        const int rc = aesw_host_aes_circuit_run(ctx, 14, 2, key, pt, 20, 1, 0, mode, &hc);
        if (rc != AESW_OK) { fprintf(stderr, "circuit_run mode %d -> %d: %s\n", mode, rc, aesw_host_last_error()); return 1; }
        char msg[256];
        CHECK(aesw_host_circuit_verify(hc, msg, sizeof msg));
        aesw_host_circuit_free(hc);
    }
    {   /* the whole circuit from columns and keygen data */
        aesw_host_circuit *hc = NULL;
        char msg[256];
        CHECK(aesw_host_aes_circuit_columns(ctx, 14, 2, key, pt, 20, &hc));
        CHECK(aesw_host_circuit_verify(hc, msg, sizeof msg));
        aesw_host_circuit_free(hc);
        if (aesw_host_aes_circuit_columns(ctx, 14, 2, key, pt, 40, &hc) != AESW_ERR_CAPACITY) { fprintf(stderr, "columns: capacity error expected\n"); return 1; }
    }
    {   /* keygen pass, capacity panic, missing key */
        aesw_host_circuit *hc = NULL;
        CHECK(aesw_host_aes_circuit_run(ctx, 14, 2, key, pt, 20, 0, 0, 0, &hc));
        aesw_host_circuit_free(hc);
        if (aesw_host_aes_circuit_run(ctx, 14, 2, key, pt, 40, 1, 0, 3, &hc) != AESW_ERR_CAPACITY) { fprintf(stderr, "capacity panic expected\n"); return 1; }
        if (aesw_host_aes_circuit_run(ctx, 14, 2, key, pt, 5, 1, 1, 0, &hc) != AESW_ERR_NO_KEY) { fprintf(stderr, "no-key panic expected\n"); return 1; }
        CHECK(aesw_host_key_circuit_run(ctx, 12, key, &hc));
        aesw_host_circuit_free(hc);
    }
    {   /* round 3: the probed arena (virtual-memory candidates, whole sets and single columns), a launch into it, and the batch
         * entry point on 1 and 3 internal streams; compared with the host-pointer path */
        const uint64_t na = (1u << 16) + 48, nbat = 5, nsm = 700;
        uint8_t *hp = malloc(na * 16), *hk = malloc(na * 16), *d_p, *d_kk;
        for (uint64_t i = 0; i < na * 16; ++i) { hp[i] = (uint8_t)(i * 7 + 1); hk[i] = (uint8_t)(i * 13 + 5); }
        HCHECK(hipMalloc((void **)&d_p, na * 16));
        HCHECK(hipMalloc((void **)&d_kk, na * 16));
        HCHECK(hipMemcpy(d_p, hp, na * 16, hipMemcpyHostToDevice));
        HCHECK(hipMemcpy(d_kk, hk, na * 16, hipMemcpyHostToDevice));
        const size_t sx = aesw_column_stride(AESW_LAYOUT_PACKED, 0);
        uint8_t *ref_x = malloc(na * sx), *ref_y = malloc(na * 1056), *ref_z = malloc(na * 608), *got_x = malloc(na * sx);
        CHECK(aesw_encrypt_witness(ctx, hp, hk, 1, na, AESW_LAYOUT_PACKED, ref_x, ref_y, ref_z, NULL, NULL));
        CHECK(aesw_set_option(ctx, "arena_cache", 0));  /* every allocation of this loop must SEARCH (the same shape three times) */
        for (int unit = 0; unit < 3; ++unit) {
            CHECK(aesw_set_option(ctx, "arena_unit", unit));
            CHECK(aesw_set_option(ctx, "arena_probe", 2));
            aesw_columns cols;
            CHECK(aesw_columns_alloc(ctx, na, AESW_LAYOUT_PACKED, 1, 1, &cols));
            if (!cols.x || !cols.key.kz || !cols.ct || cols.candidates == 0) { fprintf(stderr, "arena: members missing\n"); return 1; }
            CHECK(aesw_encrypt_witness_device(ctx, d_p, d_kk, 1, na, AESW_LAYOUT_PACKED, cols.x, cols.y, cols.z, cols.ct, &cols.key, NULL));
            HCHECK(hipMemcpy(got_x, cols.x, na * sx, hipMemcpyDeviceToHost));
            if (memcmp(got_x, ref_x, na * sx) != 0) { fprintf(stderr, "arena unit %d: x differs\n", unit); return 1; }
            CHECK(aesw_columns_free(ctx, &cols));
        }
        CHECK(aesw_set_option(ctx, "arena_unit", 2));
        CHECK(aesw_set_option(ctx, "arena_probe", -1));
        {
            aesw_columns kc;  /* key slabs alone */
            CHECK(aesw_columns_alloc(ctx, 1000, AESW_LAYOUT_PACKED, 2, 0, &kc));
            if (kc.x || !kc.key.w) { fprintf(stderr, "key-only arena: wrong members\n"); return 1; }
            CHECK(aesw_columns_free(ctx, &kc));
        }
        aesw_batch bt[5];
        uint8_t *d_o[3];
        const size_t pitch[3] = {(nsm * 1360 + 127) / 128 * 128, (nsm * 1056 + 127) / 128 * 128, (nsm * 608 + 127) / 128 * 128};
        for (int c = 0; c < 3; ++c) HCHECK(hipMalloc((void **)&d_o[c], nbat * pitch[c]));
        for (uint64_t i = 0; i < nbat; ++i) {
            bt[i].d_pt = d_p + i * nsm * 16; bt[i].d_keys = d_kk + i * nsm * 16; bt[i].n = nsm;
            bt[i].d_x = d_o[0] + i * pitch[0]; bt[i].d_y = d_o[1] + i * pitch[1]; bt[i].d_z = d_o[2] + i * pitch[2];
            bt[i].d_ct = NULL; bt[i].d_key_slab = NULL;
        }
        for (int ns = 1; ns <= 3; ns += 2) {
            CHECK(aesw_set_option(ctx, "batch_streams", ns));
            HCHECK(hipMemset(d_o[0], 0, nbat * pitch[0]));
            CHECK(aesw_encrypt_witness_batches_device(ctx, bt, (uint32_t)nbat, 1, AESW_LAYOUT_PACKED, NULL));
            HCHECK(hipDeviceSynchronize());
            for (uint64_t i = 0; i < nbat; ++i) {
                HCHECK(hipMemcpy(got_x, d_o[0] + i * pitch[0], nsm * sx, hipMemcpyDeviceToHost));
                if (memcmp(got_x, ref_x + i * nsm * sx, nsm * sx) != 0) { fprintf(stderr, "batches (%d streams): batch %llu differs\n", ns, (unsigned long long)i); return 1; }
            }
        }
        if (aesw_encrypt_witness_batches_device(ctx, NULL, 2, 1, AESW_LAYOUT_PACKED, NULL) != AESW_ERR_INVALID_ARG) { fprintf(stderr, "batches: NULL accepted\n"); return 1; }
        {   /* round 4: the placement cache (free keeps the backing, the same shape takes it over, another shape searches, the
             * size bound evicts, switching it off releases) and a search under a 1 ms budget */
            CHECK(aesw_set_option(ctx, "arena_cache", 1));
            CHECK(aesw_set_option(ctx, "arena_probe", 2));
            aesw_columns a1, a2, a3;
            int64_t v = 0;
            CHECK(aesw_columns_alloc(ctx, na, AESW_LAYOUT_PACKED, 1, 1, &a1));
            uint8_t *first_x = a1.x;
            if (a1.candidates == 0) { fprintf(stderr, "cache: first allocation did not search\n"); return 1; }
            CHECK(aesw_columns_free(ctx, &a1));
            CHECK(aesw_columns_alloc(ctx, na, AESW_LAYOUT_PACKED, 1, 1, &a2));
            if (a2.candidates != 0 || a2.x != first_x) { fprintf(stderr, "cache: same shape was not handed back\n"); return 1; }
            CHECK(aesw_encrypt_witness_device(ctx, d_p, d_kk, 1, na, AESW_LAYOUT_PACKED, a2.x, a2.y, a2.z, a2.ct, &a2.key, NULL));
            HCHECK(hipMemcpy(got_x, a2.x, na * sx, hipMemcpyDeviceToHost));
            if (memcmp(got_x, ref_x, na * sx) != 0) { fprintf(stderr, "cached arena: x differs\n"); return 1; }
            CHECK(aesw_columns_alloc(ctx, na, AESW_LAYOUT_PACKED, 0, 0, &a3));  /* another shape while a2 is live */
            if (a3.candidates == 0) { fprintf(stderr, "cache: another shape did not search\n"); return 1; }
            CHECK(aesw_columns_free(ctx, &a2));
            CHECK(aesw_columns_free(ctx, &a3));
            CHECK(aesw_get_option(ctx, "arena_cached_bytes", &v));
            if (v <= 0) { fprintf(stderr, "cache: nothing cached after two frees\n"); return 1; }
            CHECK(aesw_set_option(ctx, "arena_cache_max_mb", 300));  /* evicts the older (larger) entry */
            CHECK(aesw_get_option(ctx, "arena_cached_bytes", &v));
            if (v > (int64_t)300 << 20) { fprintf(stderr, "cache: size bound not applied\n"); return 1; }
            CHECK(aesw_set_option(ctx, "arena_cache", 0));
            CHECK(aesw_get_option(ctx, "arena_cached_bytes", &v));
            if (v != 0) { fprintf(stderr, "cache: not released\n"); return 1; }
            CHECK(aesw_set_option(ctx, "arena_probe_budget_ms", 1));
            CHECK(aesw_columns_alloc(ctx, na, AESW_LAYOUT_PACKED, 1, 0, &a1));
            if (a1.candidates != 1) { fprintf(stderr, "budget: %u candidates\n", a1.candidates); return 1; }
            CHECK(aesw_columns_free(ctx, &a1));
            CHECK(aesw_set_option(ctx, "arena_probe_budget_ms", 3000));
            CHECK(aesw_set_option(ctx, "arena_cache", 1));
            CHECK(aesw_set_option(ctx, "arena_cache_max_mb", 65536));
            CHECK(aesw_set_option(ctx, "arena_probe", -1));
            CHECK(aesw_columns_alloc(ctx, na, AESW_LAYOUT_PACKED, 0, 0, &a1));  /* left in the cache for aesw_destroy to release */
            CHECK(aesw_columns_free(ctx, &a1));
        }
        {   /* round 4: aesw_check_witness_device over the product's own output (per-block keys, packed): satisfied; one byte off: not */
            aesw_columns cc;
            aesw_check_report *d_rep, rep;
            CHECK(aesw_columns_alloc(ctx, na, AESW_LAYOUT_PACKED, 1, 1, &cc));
            HCHECK(hipMalloc((void **)&d_rep, sizeof rep));
            CHECK(aesw_encrypt_witness_device(ctx, d_p, d_kk, 1, na, AESW_LAYOUT_PACKED, cc.x, cc.y, cc.z, cc.ct, &cc.key, NULL));
            CHECK(aesw_check_witness_device(ctx, d_p, d_kk, 1, na, AESW_LAYOUT_PACKED, cc.x, cc.y, cc.z, cc.ct, &cc.key, d_rep, NULL));
            HCHECK(hipMemcpy(&rep, d_rep, sizeof rep, hipMemcpyDeviceToHost));
            if (rep.blocks != na || rep.keys != na || rep.lookup_failures || rep.copy_failures || rep.gate_failures || rep.input_failures || rep.first != AESW_CHECK_NONE) {
                fprintf(stderr, "check: the product's witness does not satisfy its constraints\n"); return 1;
            }
            HCHECK(hipMemset(cc.z + 5 * 608 + 20, 0xEE, 1));
            CHECK(aesw_check_witness_device(ctx, d_p, d_kk, 1, na, AESW_LAYOUT_PACKED, cc.x, cc.y, cc.z, cc.ct, &cc.key, d_rep, NULL));
            HCHECK(hipMemcpy(&rep, d_rep, sizeof rep, hipMemcpyDeviceToHost));
            if (rep.first == AESW_CHECK_NONE || AESW_CHECK_UNIT(rep.first) != 5 || AESW_CHECK_IS_KEY_SLAB(rep.first)) { fprintf(stderr, "check: a changed byte went unnoticed\n"); return 1; }
            if (aesw_check_witness_device(ctx, d_p, d_kk, 1, na, AESW_LAYOUT_VALUES, cc.x, cc.y, cc.z, NULL, &cc.key, d_rep, NULL) != AESW_ERR_INVALID_ARG) { fprintf(stderr, "check: VALUES layout accepted\n"); return 1; }
            HCHECK(hipFree(d_rep));
            CHECK(aesw_columns_free(ctx, &cc));
            {   /* ... and the host-pointer form over three stages, the last one ragged: ref_* hold the witness of (hp, hk) */
                uint8_t *hw = malloc(na * 96), *hkx = malloc(na * 400), *hky = malloc(na * 240), *hkz = malloc(na * 200);
                aesw_key_slab hks2 = {hw, hkx, hky, hkz};
                CHECK(aesw_encrypt_witness(ctx, hp, hk, 1, na, AESW_LAYOUT_PACKED, ref_x, ref_y, ref_z, NULL, &hks2));
                CHECK(aesw_set_option(ctx, "chunk_blocks", 30000));
                CHECK(aesw_check_witness(ctx, hp, hk, 1, na, AESW_LAYOUT_PACKED, ref_x, ref_y, ref_z, NULL, &hks2, &rep));
                if (rep.blocks != na || rep.keys != na || rep.first != AESW_CHECK_NONE) { fprintf(stderr, "host check: not satisfied\n"); return 1; }
                ref_y[(na - 1) * 1056 + 3] ^= 1;
                CHECK(aesw_check_witness(ctx, hp, hk, 1, na, AESW_LAYOUT_PACKED, ref_x, ref_y, ref_z, NULL, &hks2, &rep));
                if (rep.first == AESW_CHECK_NONE || AESW_CHECK_UNIT(rep.first) != na - 1) { fprintf(stderr, "host check: the last block's change went unnoticed\n"); return 1; }
                ref_y[(na - 1) * 1056 + 3] ^= 1;
                CHECK(aesw_set_option(ctx, "chunk_blocks", 1 << 15));
                free(hw); free(hkx); free(hky); free(hkz);
            }
        }
        {   /* round 4: "stream_check": the stream entry point with a device check per chunk, scheduled key and per-block keys */
            aesw_check_report sr;
            uint64_t nchunks = 0;
            CHECK(aesw_set_option(ctx, "stream_check", 1));
            CHECK(aesw_set_option(ctx, "chunk_blocks", 20000));
            CHECK(aesw_schedule_key(ctx, hk, AESW_LAYOUT_PACKED, NULL));
            CHECK(aesw_encrypt_witness_stream(ctx, hp, NULL, 0, na, AESW_LAYOUT_PACKED, consume, &nchunks));
            CHECK(aesw_last_stream_check(ctx, &sr));
            if (sr.blocks != na || sr.keys != 1 || sr.first != AESW_CHECK_NONE) { fprintf(stderr, "stream_check (scheduled key): %llu blocks\n", (unsigned long long)sr.blocks); return 1; }
            CHECK(aesw_encrypt_witness_stream(ctx, hp, hk, 1, na, AESW_LAYOUT_DENSE, consume, &nchunks));
            CHECK(aesw_last_stream_check(ctx, &sr));
            if (sr.blocks != na || sr.keys != na || sr.first != AESW_CHECK_NONE) { fprintf(stderr, "stream_check (per-block keys, dense)\n"); return 1; }
            CHECK(aesw_set_option(ctx, "stream_check", 0));
            CHECK(aesw_set_option(ctx, "chunk_blocks", 1 << 15));
        }
        {   /* round 4: the scheduled key's slot ring.  Twenty reader streams (more than a slot tracks: folding), re-schedules
             * on rings of 1, 2 and 4 slots, a lone launch dealt out by "split_small"; the last key's output is checked */
            hipStream_t st[20];
            for (int i = 0; i < 20; ++i) HCHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            uint8_t *d_o2[3];
            for (int c = 0; c < 3; ++c) HCHECK(hipMalloc((void **)&d_o2[c], na * aesw_column_stride(AESW_LAYOUT_PACKED, c)));
            for (int ring = 1; ring <= 4; ring <<= 1) {
                CHECK(aesw_set_option(ctx, "key_slots", ring));
                for (int k = 0; k < 6; ++k) {
                    CHECK(aesw_schedule_key_device(ctx, d_kk + 16 * k, AESW_LAYOUT_PACKED, NULL, st[k % 3]));
                    for (int i = 0; i < 20; ++i)  /* every stream writes its own 64-block range */
                        CHECK(aesw_encrypt_witness_device(ctx, d_p + (size_t)i * 64 * 16, NULL, 0, 64, AESW_LAYOUT_PACKED, d_o2[0] + (size_t)i * 64 * 1360,
                                                          d_o2[1] + (size_t)i * 64 * 1056, d_o2[2] + (size_t)i * 64 * 608, NULL, NULL, st[i]));
                }
            }
            CHECK(aesw_set_option(ctx, "split_small", 3));
            CHECK(aesw_encrypt_witness_device(ctx, d_p, NULL, 0, na, AESW_LAYOUT_PACKED, d_o2[0], d_o2[1], d_o2[2], NULL, NULL, st[0]));
            CHECK(aesw_set_option(ctx, "split_small", 0));
            HCHECK(hipDeviceSynchronize());
            CHECK(aesw_encrypt_witness(ctx, hp, hk + 16 * 5, 0, na, AESW_LAYOUT_PACKED, ref_x, ref_y, ref_z, NULL, NULL));  /* key 5 as a shared key */
            HCHECK(hipMemcpy(got_x, d_o2[0], na * sx, hipMemcpyDeviceToHost));
            if (memcmp(got_x, ref_x, na * sx) != 0) { fprintf(stderr, "slot ring / split_small: x differs from the last scheduled key's witness\n"); return 1; }
            int64_t waits = 0, slots = 0;
            CHECK(aesw_get_option(ctx, "key_reader_waits", &waits));
            CHECK(aesw_get_option(ctx, "key_slots_allocated", &slots));
            if (waits < 20 || slots < 4 || slots > 16) { fprintf(stderr, "slot ring: waits %lld slots %lld\n", (long long)waits, (long long)slots); return 1; }
            for (int c = 0; c < 3; ++c) HCHECK(hipFree(d_o2[c]));
            for (int i = 0; i < 20; ++i) HCHECK(hipStreamDestroy(st[i]));  /* readers of the current slot are still tracked: aesw_destroy frees their events */
        }
        for (int c = 0; c < 3; ++c) HCHECK(hipFree(d_o[c]));
        HCHECK(hipFree(d_p)); HCHECK(hipFree(d_kk));
        free(hp); free(hk); free(ref_x); free(ref_y); free(ref_z); free(got_x);
    }
    aesw_destroy(ctx);
    free(pt); free(keys);
    printf("asan driver: ok\n");
    fflush(stdout);  /* a failure in the runtimes' own teardown must not swallow the verdict */
    return 0;
}
