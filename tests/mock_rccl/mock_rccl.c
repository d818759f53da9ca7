/* mock_rccl.c -- TEST INFRASTRUCTURE: a recording stand-in for librccl, loaded by libaesw's dlopen("librccl.so") in
 * tests/test_gpu_round2.py::test_gather_call_sequence_for_three_ranks.  No multi-GPU box is available to the builder,
 * so the multi-rank leg of aesw_gather_columns_device (peer loop, offsets, <= max_message pieces, one group) is checked
 * by recording what it would ask RCCL to do.  Every call appends one line to $MOCK_RCCL_LOG. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;

static void logf_(const char *fmt, unsigned long long a, unsigned long long b, unsigned long long c) {
    const char *path = getenv("MOCK_RCCL_LOG");
    if (!path) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    fprintf(f, fmt, a, b, c);
    fclose(f);
}
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5a, sizeof *id); logf_("id %llu %llu %llu\n", 0, 0, 0); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    *comm = malloc(8);
    logf_("init %llu %llu %llu\n", (unsigned long long)nranks, (unsigned long long)rank, (unsigned long long)(unsigned char)id.internal[0]);
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); logf_("destroy %llu %llu %llu\n", 0, 0, 0); return 0; }
ncclResult_t ncclGroupStart(void) { logf_("group_start %llu %llu %llu\n", 0, 0, 0); return 0; }
ncclResult_t ncclGroupEnd(void) { logf_("group_end %llu %llu %llu\n", 0, 0, 0); return 0; }
ncclResult_t ncclSend(const void *buf, size_t count, int dtype, int peer, ncclComm_t comm, void *stream) {
    (void)comm; (void)stream;
    logf_(dtype == 1 ? "send %llu %llu %llu\n" : "send_badtype %llu %llu %llu\n", (unsigned long long)peer, (unsigned long long)(size_t)buf, (unsigned long long)count);
    return 0;
}
ncclResult_t ncclRecv(void *buf, size_t count, int dtype, int peer, ncclComm_t comm, void *stream) {
    (void)comm; (void)stream;
    logf_(dtype == 1 ? "recv %llu %llu %llu\n" : "recv_badtype %llu %llu %llu\n", (unsigned long long)peer, (unsigned long long)(size_t)buf, (unsigned long long)count);
    return 0;
}
const char *ncclGetErrorString(ncclResult_t r) { (void)r; return "mock"; }
