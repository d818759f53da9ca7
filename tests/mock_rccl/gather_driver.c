/* gather_driver.c -- TEST INFRASTRUCTURE: calls aesw_gather_columns_device as rank R of N against the recording mock
 * of librccl (tests/mock_rccl/mock_rccl.c).  usage: gather_driver N R MAXMSG c0 c1 ... c(N-1)   (root = $GATHER_ROOT, default 0)
 * Prints the send / recv base pointers so that the test can turn logged addresses into offsets.
 * With $GATHER_DATA set (and the functional stand-in shm_rccl.c as librccl): every rank fills its send ranges with a pattern of
 * (rank, column, byte index), the root checks every byte of the gathered columns and prints "data ok". */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "aesw.h"
static uint8_t pat(int rank, int col, uint64_t i) { return (uint8_t)(i * 131u + (uint64_t)rank * 17u + (uint64_t)col * 59u + (i >> 8)); }
#define AK(x) do { int r_ = (x); if (r_ != AESW_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, aesw_comm_last_error()); return 3; } } while (0)
int main(int argc, char **argv) {
    const int n = atoi(argv[1]), rank = atoi(argv[2]);
    const uint64_t maxmsg = strtoull(argv[3], NULL, 10);
    uint64_t counts[16], offs[16], total = 0;
    const int root = getenv("GATHER_ROOT") ? atoi(getenv("GATHER_ROOT")) : 0;
    if (n < 1 || n > 16 || rank < 0 || rank >= n || root < 0 || root >= n || argc < 4 + n) return 4;
    for (int i = 0; i < n; ++i) counts[i] = strtoull(argv[4 + i], NULL, 10);
    uint8_t sbox[256], m2[256], m3[256];
    for (int i = 0; i < 256; ++i) { sbox[i] = (uint8_t)i; m2[i] = (uint8_t)(i * 2); m3[i] = (uint8_t)(i * 3); }
    aesw_ctx *ctx = NULL;
    AK(aesw_create(&ctx, 0, sbox, m2, m3));
    AK(aesw_gather_offsets(n, counts, offs, &total));
    const uint32_t strides[3] = {1360, 1056, 608};
    uint8_t *send[3], *recv[3];
    for (int c = 0; c < 3; ++c) {
        if (hipMalloc((void **)&send[c], counts[rank] * strides[c] + 16) != hipSuccess || hipMalloc((void **)&recv[c], total * strides[c] + 16) != hipSuccess) return 2;
        printf("col %d send %llu recv %llu\n", c, (unsigned long long)(size_t)send[c], (unsigned long long)(size_t)recv[c]);
    }
    const int data = getenv("GATHER_DATA") != NULL;
    if (data)
        for (int c = 0; c < 3; ++c) {
            const size_t nb = counts[rank] * strides[c];
            uint8_t *h = (uint8_t *)malloc(nb + 1);
            for (size_t i = 0; i < nb; ++i) h[i] = pat(rank, c, i);
            if (hipMemcpy(send[c], h, nb, hipMemcpyHostToDevice) != hipSuccess || hipMemset(recv[c], 0xEE, total * strides[c]) != hipSuccess) return 2;
            free(h);
        }
    uint8_t id[AESW_COMM_ID_BYTES];
    AK(aesw_comm_unique_id(id));
    aesw_comm *comm = NULL;
    AK(aesw_comm_create(ctx, n, rank, id, &comm));
    AK(aesw_comm_set_max_message(comm, maxmsg));
    AK(aesw_gather_columns_device(comm, root, 3, (const uint8_t *const *)send, recv, counts, strides, NULL));
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    if (data && rank == root) {
        for (int c = 0; c < 3; ++c) {
            const size_t nb = total * strides[c];
            uint8_t *h = (uint8_t *)malloc(nb + 1);
            if (hipMemcpy(h, recv[c], nb, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            for (int r = 0; r < n; ++r)
                for (size_t i = 0; i < counts[r] * strides[c]; ++i)
                    if (h[offs[r] * strides[c] + i] != pat(r, c, i)) { fprintf(stderr, "column %d: byte %zu of rank %d's range is wrong\n", c, i, r); return 5; }
            free(h);
        }
        printf("data ok\n");
    }
    aesw_comm_destroy(comm);
    aesw_destroy(ctx);
    printf("ok\n");
    return 0;
}
