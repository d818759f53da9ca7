/* gather_driver.c -- TEST INFRASTRUCTURE: calls aesw_gather_columns_device as rank R of N against the recording mock
 * of librccl (tests/mock_rccl/mock_rccl.c).  usage: gather_driver N R MAXMSG c0 c1 ... c(N-1)   (root = $GATHER_ROOT, default 0)
 * Prints the send / recv base pointers so that the test can turn logged addresses into offsets. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "aesw.h"
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
    uint8_t id[AESW_COMM_ID_BYTES];
    AK(aesw_comm_unique_id(id));
    aesw_comm *comm = NULL;
    AK(aesw_comm_create(ctx, n, rank, id, &comm));
    AK(aesw_comm_set_max_message(comm, maxmsg));
    AK(aesw_gather_columns_device(comm, root, 3, (const uint8_t *const *)send, recv, counts, strides, NULL));
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    aesw_comm_destroy(comm);
    aesw_destroy(ctx);
    printf("ok\n");
    return 0;
}
