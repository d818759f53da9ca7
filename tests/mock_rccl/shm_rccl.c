/* shm_rccl.c -- TEST INFRASTRUCTURE: a FUNCTIONAL stand-in for librccl.  Where mock_rccl.c only records what
 * aesw_gather_columns_device would ask RCCL to do, this one does it: a message is a file in $SHM_RCCL_DIR (a tmpfs directory),
 * written by ncclSend (device -> host -> file) and consumed by ncclRecv (file -> host -> device), so that N processes sharing
 * ONE GPU can run the C ABI's gather for real and the root's columns can be compared byte for byte (no multi-GPU box is
 * available to the builder).  Semantics kept from NCCL: operations between ncclGroupStart and ncclGroupEnd are issued together
 * at ncclGroupEnd; the k-th send to a peer pairs with the k-th receive from it, and a receive whose size differs from the paired
 * send's is an error (what a sender / receiver that cut a range into different pieces would produce).  Work is ordered behind the
 * stream it is given (hipStreamSynchronize) and done synchronously.  No xGMI, no performance: correctness of the call sequence only. */
#define _GNU_SOURCE
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
enum { OK = 0, ERR_SYSTEM = 2, ERR_ARG = 4 };

typedef struct { int send; void *buf; size_t count; int peer; void *stream; } op_t;
static op_t ops[8192];
static int nops, in_group, my_rank = -1, n_ranks;
static unsigned long seq_send[64], seq_recv[64];
static char last_error[256] = "ok";

static const char *dir(void) { const char *d = getenv("SHM_RCCL_DIR"); return d ? d : "/dev/shm"; }

static ncclResult_t do_send(const op_t *o) {
    if (hipStreamSynchronize((hipStream_t)o->stream) != hipSuccess) return ERR_SYSTEM;
    void *host = malloc(o->count ? o->count : 1);
    if (!host || hipMemcpy(host, o->buf, o->count, hipMemcpyDeviceToHost) != hipSuccess) { free(host); return ERR_SYSTEM; }
    char tmp[512], fin[512];
    snprintf(tmp, sizeof tmp, "%s/m_%d_%d_%lu.tmp", dir(), my_rank, o->peer, seq_send[o->peer]);
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%lu.msg", dir(), my_rank, o->peer, seq_send[o->peer]);
    ++seq_send[o->peer];
    FILE *f = fopen(tmp, "wb");
    if (!f || fwrite(host, 1, o->count, f) != o->count) { if (f) fclose(f); free(host); snprintf(last_error, sizeof last_error, "cannot write %s", tmp); return ERR_SYSTEM; }
    fclose(f);
    free(host);
    return rename(tmp, fin) == 0 ? OK : ERR_SYSTEM;  /* atomic: a receiver never sees half a message */
}

static ncclResult_t do_recv(const op_t *o) {
    char fin[512];
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%lu.msg", dir(), o->peer, my_rank, seq_recv[o->peer]);
    ++seq_recv[o->peer];
    struct stat st;
    const time_t t0 = time(NULL);
    while (stat(fin, &st) != 0) {
        if (time(NULL) - t0 > 60) { snprintf(last_error, sizeof last_error, "no message %s within 60 s (the peer never sent it)", fin); return ERR_SYSTEM; }
        usleep(200);
    }
    if ((size_t)st.st_size != o->count) {
        snprintf(last_error, sizeof last_error, "%s holds %lld bytes, the receive expects %zu: sender and receiver cut the range differently", fin, (long long)st.st_size, o->count);
        return ERR_ARG;
    }
    void *host = malloc(o->count ? o->count : 1);
    FILE *f = fopen(fin, "rb");
    if (!host || !f || fread(host, 1, o->count, f) != o->count) { if (f) fclose(f); free(host); return ERR_SYSTEM; }
    fclose(f);
    unlink(fin);
    if (hipStreamSynchronize((hipStream_t)o->stream) != hipSuccess) { free(host); return ERR_SYSTEM; }
    const hipError_t e = hipMemcpy(o->buf, host, o->count, hipMemcpyHostToDevice);
    free(host);
    return e == hipSuccess ? OK : ERR_SYSTEM;
}

static ncclResult_t run(void) {
    ncclResult_t rc = OK;
    for (int i = 0; i < nops && rc == OK; ++i) if (ops[i].send) rc = do_send(&ops[i]);   /* all sends first: they never block */
    for (int i = 0; i < nops && rc == OK; ++i) if (!ops[i].send) rc = do_recv(&ops[i]);
    nops = 0;
    return rc;
}

static ncclResult_t post(int send, void *buf, size_t count, int dtype, int peer, void *stream) {
    if (dtype != 1 /* ncclUint8 */ || peer < 0 || peer >= n_ranks || peer == my_rank || nops >= 8192) { snprintf(last_error, sizeof last_error, "bad send/recv argument"); return ERR_ARG; }
    ops[nops++] = (op_t){send, buf, count, peer, stream};
    return in_group ? OK : run();
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5a, sizeof *id); return OK; }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    (void)id;
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ERR_ARG;
    my_rank = rank; n_ranks = nranks;
    *comm = malloc(8);
    return OK;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); return OK; }
ncclResult_t ncclGroupStart(void) { in_group = 1; return OK; }
ncclResult_t ncclGroupEnd(void) { in_group = 0; return run(); }
ncclResult_t ncclSend(const void *buf, size_t count, int dtype, int peer, ncclComm_t comm, void *stream) { (void)comm; return post(1, (void *)buf, count, dtype, peer, stream); }
ncclResult_t ncclRecv(void *buf, size_t count, int dtype, int peer, ncclComm_t comm, void *stream) { (void)comm; return post(0, buf, count, dtype, peer, stream); }
const char *ncclGetErrorString(ncclResult_t r) { (void)r; return last_error; }
