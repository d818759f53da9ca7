// aesw_comm.cpp -- the multi-GPU exchange step of include/aesw.h: gather every rank's column ranges on one GPU
// with RCCL send/recv over xGMI (BASELINE configs[3]).  One process per GPU; the host hands the
// 128-byte unique id from rank 0 to the other ranks by its own means (MPI, a socket, torch.distributed ...).
//
// RCCL is bound at run time (dlopen): libaesw.so stays loadable on a single-GPU host without librccl, and a
// process that already holds a copy (PyTorch ships one) shares it instead of mapping a second.  A communicator
// of ONE rank never touches RCCL: its gather is the root's own device-to-device copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/aesw.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already mapped into the process first (RTLD_NOLOAD), then the system's
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.handle)
            for (const char *n : names)
                if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.handle) {
            r.error = std::string("librccl not found: ") + dlerror();
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

thread_local std::string g_comm_error;

int fail(const std::string &what) {
    g_comm_error = what;
    return AESW_ERR_COMM;
}

}  // namespace

struct aesw_comm {
    int device = -1, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    uint64_t max_message = 1ull << 30;  // a rank's range of one column travels in pieces of at most this many bytes
};

extern "C" {

const char *aesw_comm_last_error(void) { return g_comm_error.c_str(); }

int aesw_comm_unique_id(uint8_t id[AESW_COMM_ID_BYTES]) {
    if (!id) return AESW_ERR_INVALID_ARG;
    Rccl *r = rccl();
    if (!r->error.empty()) return fail(r->error);
    static_assert(AESW_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    const ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) return fail(std::string("ncclGetUniqueId: ") + r->GetErrorString(e));
    std::memcpy(id, u.internal, AESW_COMM_ID_BYTES);
    return AESW_OK;
}

int aesw_comm_create(aesw_ctx *ctx, int nranks, int rank, const uint8_t id[AESW_COMM_ID_BYTES], aesw_comm **out) {
    if (!ctx || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return AESW_ERR_INVALID_ARG;
    *out = nullptr;
    aesw_comm *c = new (std::nothrow) aesw_comm;
    if (!c) return AESW_ERR_NOMEM;
    c->device = aesw_device(ctx);
    c->nranks = nranks;
    c->rank = rank;
    if (nranks > 1) {
        Rccl *r = rccl();
        if (!r->error.empty()) { delete c; return fail(r->error); }
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (hipSetDevice(c->device) != hipSuccess) { delete c; return AESW_ERR_NO_DEVICE; }
        ncclUniqueId u;
        std::memcpy(u.internal, id, AESW_COMM_ID_BYTES);
        const ncclResult_t e = r->CommInitRank(&c->comm, nranks, u, rank);  // collective: every rank calls it
        if (prev >= 0) (void)hipSetDevice(prev);
        if (e != ncclSuccess) { delete c; return fail(std::string("ncclCommInitRank: ") + r->GetErrorString(e)); }
    }
    *out = c;
    return AESW_OK;
}

void aesw_comm_destroy(aesw_comm *c) {
    if (!c) return;
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    delete c;
}

int aesw_comm_set_max_message(aesw_comm *c, uint64_t bytes) {
    if (!c || bytes == 0) return AESW_ERR_INVALID_ARG;
    c->max_message = bytes;
    return AESW_OK;
}

// Pure host: where rank r's blocks start in the gathered columns (exclusive prefix sum of counts).
int aesw_gather_offsets(int nranks, const uint64_t *counts, uint64_t *offsets, uint64_t *total) {
    if (nranks < 1 || !counts || !offsets) return AESW_ERR_INVALID_ARG;
    uint64_t acc = 0;
    for (int r = 0; r < nranks; ++r) {
        offsets[r] = acc;
        if (counts[r] > UINT64_MAX - acc) return AESW_ERR_INVALID_ARG;
        acc += counts[r];
    }
    if (total) *total = acc;
    return AESW_OK;
}

int aesw_gather_columns_device(aesw_comm *c, int root, int n_cols, const uint8_t *const *d_send, uint8_t *const *d_recv,
                               const uint64_t *counts, const uint32_t *strides, void *stream) {
    if (!c || root < 0 || root >= c->nranks || n_cols < 1 || n_cols > 64 || !d_send || !counts || !strides) return AESW_ERR_INVALID_ARG;
    const bool is_root = c->rank == root;
    if (is_root && !d_recv) return AESW_ERR_INVALID_ARG;
    std::vector<uint64_t> offs(c->nranks);
    if (aesw_gather_offsets(c->nranks, counts, offs.data(), nullptr) != AESW_OK) return AESW_ERR_INVALID_ARG;
    for (int i = 0; i < n_cols; ++i)
        if ((strides[i] && counts[c->rank] && !d_send[i]) || (is_root && strides[i] && !d_recv[i])) return AESW_ERR_INVALID_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(c->device) != hipSuccess) return AESW_ERR_NO_DEVICE;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    // the root's own range: a device-to-device copy on the same stream (skipped when it already sits in place)
    if (is_root)
        for (int i = 0; i < n_cols; ++i) {
            const uint64_t bytes = counts[root] * strides[i];
            uint8_t *dst = d_recv[i] + offs[root] * strides[i];
            if (bytes && dst != d_send[i]) {
                const hipError_t e = hipMemcpyAsync(dst, d_send[i], bytes, hipMemcpyDeviceToDevice, s);
                if (e != hipSuccess) return fail(std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
            }
        }
    if (c->nranks == 1) return AESW_OK;
    Rccl *r = rccl();
    // One group: the root's receives from all peers progress concurrently, each peer over its own xGMI link.
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return fail(std::string("ncclGroupStart: ") + r->GetErrorString(e));
    ncclResult_t first_bad = ncclSuccess;
    auto note = [&](ncclResult_t x) { if (x != ncclSuccess && first_bad == ncclSuccess) first_bad = x; };
    for (int i = 0; i < n_cols; ++i) {
        if (is_root) {
            for (int p = 0; p < c->nranks; ++p) {
                if (p == root) continue;
                const uint64_t bytes = counts[p] * strides[i];
                uint8_t *dst = d_recv[i] + offs[p] * strides[i];
                for (uint64_t o = 0; o < bytes; o += c->max_message)
                    note(r->Recv(dst + o, (size_t)(bytes - o < c->max_message ? bytes - o : c->max_message), ncclUint8, p, c->comm, s));
            }
        } else {
            const uint64_t bytes = counts[c->rank] * strides[i];
            for (uint64_t o = 0; o < bytes; o += c->max_message)
                note(r->Send(d_send[i] + o, (size_t)(bytes - o < c->max_message ? bytes - o : c->max_message), ncclUint8, root, c->comm, s));
        }
    }
    e = r->GroupEnd();  // always close the group, even after a failed enqueue
    if (first_bad != ncclSuccess) return fail(std::string("ncclSend/ncclRecv: ") + r->GetErrorString(first_bad));
    if (e != ncclSuccess) return fail(std::string("ncclGroupEnd: ") + r->GetErrorString(e));
    return AESW_OK;
}

}  // extern "C"
