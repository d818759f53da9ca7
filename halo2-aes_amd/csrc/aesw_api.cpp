// aesw_api.cpp -- the C ABI of include/aesw.h on top of the gfx950 kernels.
// Host code only (compiled by hipcc for the HIP runtime API).  There is no CPU
// compute path: without a usable device every computing entry point fails.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <ctime>
#include <mutex>
#include <new>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/aesw.h"
#include "aesw_internal.h"
#include "aesw_layout.h"
#include "aesw_check.h"

using namespace aesw;

#include "aesw_ctx.h"

namespace {

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
bool valid_layout(int l) { return aesw_valid_layout(l); }

uint8_t xtime(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }

// ---- bn256::Fr Montgomery table: v -> v * 2^256 mod r, 32 B little-endian ----
// r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
// (halo2curves 0.6.1 bn256::Fr, Cargo.lock:779-781).
struct U256 { uint64_t l[4]; };
const U256 FR_MOD = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};

bool geq(const U256 &a, const U256 &b) {
    for (int i = 3; i >= 0; --i) {
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    }
    return true;
}
// (a + b) mod r for a, b < r  (r < 2^254 so the sum fits in 256 bits)
U256 add_mod(const U256 &a, const U256 &b) {
    U256 s;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (unsigned __int128)a.l[i] + b.l[i];
        s.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq(s, FR_MOD)) {
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; ++i) {
            unsigned __int128 d = (unsigned __int128)s.l[i] - FR_MOD.l[i] - (uint64_t)br;
            s.l[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    return s;
}
void build_fr_lut(uint8_t out[256 * 32]) {
    U256 R = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; ++i) R = add_mod(R, R);  // 2^256 mod r
    U256 acc = {{0, 0, 0, 0}};
    for (int v = 0; v < 256; ++v) {
        std::memcpy(out + 32 * v, acc.l, 32);  // little-endian limbs
        acc = add_mod(acc, R);
    }
}

}  // namespace

static void vmm_release_arena(void *va, size_t total) {
    (void)hipMemUnmap(va, total);
    (void)hipMemAddressFree(va, total);
}

extern "C" {

int aesw_version(void) { return AESW_VERSION; }

const char *aesw_strerror(int status) {
    switch (status) {
    case AESW_OK: return "ok";
    case AESW_ERR_INVALID_ARG: return "invalid argument";
    case AESW_ERR_NO_DEVICE: return "no usable gfx950 device (this library has no CPU path)";
    case AESW_ERR_HIP: return "HIP runtime error";
    case AESW_ERR_NOMEM: return "out of memory";
    case AESW_ERR_CAPACITY: return "AES calls too many. doesn't fit in the rows";
    case AESW_ERR_NO_KEY: return "Keys should be scheduled";
    case AESW_ERR_MISMATCH: return "host value disagrees with the device witness";
    case AESW_ERR_UNSATISFIED: return "constraint system not satisfied";
    case AESW_ERR_COMM: return "RCCL unavailable or a collective call failed";
    default: return "unknown status";
    }
}

// ---- the scheduled key's round-key slots (aesw_ctx.h) ----------------------------------------------------------
namespace {
constexpr size_t KEY_CHUNK_SLOTS = 16, KEY_SLOT_BYTES = 256, KEY_MAX_READERS = 16;

bool stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    return cs != hipStreamCaptureStatusNone;
}

// Both streams are being captured into the SAME graph (one forked from the other with an event, like the internal streams of
// the batch entry point from the caller's).
bool same_capture(hipStream_t a, hipStream_t b) {
    hipStreamCaptureStatus sa = hipStreamCaptureStatusNone, sb = hipStreamCaptureStatusNone;
    unsigned long long ia = 0, ib = 0;
    if (hipStreamGetCaptureInfo(a, &sa, &ia) != hipSuccess || hipStreamGetCaptureInfo(b, &sb, &ib) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return sa == hipStreamCaptureStatusActive && sb == hipStreamCaptureStatusActive && ia == ib;
}

// While some stream of this thread is being captured (global mode), allocation calls are refused: run them relaxed.
struct RelaxedCapture {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    bool on;
    RelaxedCapture() { on = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!on) (void)hipGetLastError(); }
    ~RelaxedCapture() { if (on && hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
};

// A slot nobody has used yet (fresh memory; a new chunk every KEY_CHUNK_SLOTS slots).
int key_new_slot(aesw_ctx *ctx, int *out) {
    RelaxedCapture relaxed;
    const size_t used = ctx->key_slots.size();
    if (used == ctx->key_chunks.size() * KEY_CHUNK_SLOTS) {
        uint8_t *c = nullptr;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&c), KEY_CHUNK_SLOTS * KEY_SLOT_BYTES));
        ctx->key_chunks.push_back(c);
    }
    aesw_ctx::KeySlot sl;
    sl.d = ctx->key_chunks.back() + (used % KEY_CHUNK_SLOTS) * KEY_SLOT_BYTES;
    HIP_TRY(ctx, hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    ctx->key_slots.push_back(sl);
    *out = (int)used;
    return AESW_OK;
}

// The slot an un-captured schedule writes next: the ring's next slot (spare slots and fresh ones fill the ring up to
// "key_slots"; a slot a capture has pinned meanwhile is replaced).  The caller waits for the slot's readers.
int key_next_ring_slot(aesw_ctx *ctx, int *out) {
    auto &ring = ctx->key_ring_slots;
    while ((int)ring.size() > ctx->key_ring) { ctx->key_spare.push_back(ring.back()); ring.pop_back(); }
    auto take = [&](int *idx) -> int {
        while (!ctx->key_spare.empty()) {
            const int i = ctx->key_spare.back();
            ctx->key_spare.pop_back();
            if (!ctx->key_slots[i].pinned) { *idx = i; return AESW_OK; }
        }
        return key_new_slot(ctx, idx);
    };
    if ((int)ring.size() < ctx->key_ring) {
        int idx = -1;
        const int rc = take(&idx);
        if (rc != AESW_OK) return rc;
        ring.push_back(idx);
        ctx->key_pos = (int)ring.size() - 1;
    } else {
        ctx->key_pos = (ctx->key_pos + 1) % (int)ring.size();
        if (ctx->key_slots[ring[ctx->key_pos]].pinned) {
            int idx = -1;
            const int rc = take(&idx);
            if (rc != AESW_OK) return rc;
            ring[ctx->key_pos] = idx;
        }
    }
    *out = ring[ctx->key_pos];
    return AESW_OK;
}

// An un-captured launch on `s` reads slot `sl`: the schedule that reuses the slot will wait for it.  One event per distinct
// stream: a stream's later record is ordered behind its earlier launches, so re-recording loses nobody.
int key_track_reader(aesw_ctx *ctx, aesw_ctx::KeySlot &sl, hipStream_t s) {
    for (auto &r : sl.readers)
        if (r.s == s) { HIP_TRY(ctx, hipEventRecord(r.e, s)); return AESW_OK; }
    if (sl.readers.size() >= KEY_MAX_READERS) {
        // fold the oldest reader into this stream: `s` waits for it BEHIND the launch just issued, so the event recorded
        // next on `s` stands for both
        HIP_TRY(ctx, hipStreamWaitEvent(s, sl.readers.front().e, 0));
        ctx->event_pool.push_back(sl.readers.front().e);
        sl.readers.erase(sl.readers.begin());
    }
    hipEvent_t e = nullptr;
    if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
    else HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const hipError_t rc = hipEventRecord(e, s);
    if (rc != hipSuccess) { ctx->event_pool.push_back(e); return fail_hip(ctx, rc, "hipEventRecord(key reader)"); }
    sl.readers.push_back(aesw_ctx::KeyReader{s, e});
    return AESW_OK;
}
}  // namespace

const char *aesw_last_error(const aesw_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int aesw_device_count(int *count) {
    if (!count) return AESW_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return n > 0 ? AESW_OK : AESW_ERR_NO_DEVICE;
}

int aesw_create(aesw_ctx **out, int device, const uint8_t sbox[256], const uint8_t mul2[256],
                const uint8_t mul3[256]) {
    if (!out || !sbox || !mul2 || !mul3) return AESW_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return AESW_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return AESW_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return AESW_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return AESW_ERR_NO_DEVICE;  // code objects are gfx950 only
    aesw_ctx *ctx = new (std::nothrow) aesw_ctx;
    if (!ctx) return AESW_ERR_NOMEM;
    ctx->device = device;
    DeviceGuard g(device);
    if (!g.ok) { delete ctx; return AESW_ERR_NO_DEVICE; }
    uint8_t host[768];
    std::memcpy(host, sbox, 256);
    std::memcpy(host + 256, mul2, 256);
    std::memcpy(host + 512, mul3, 256);
    ctx->xt = true;
    for (int i = 0; i < 256; ++i)
        if (mul2[i] != xtime((uint8_t)i) || mul3[i] != (uint8_t)(xtime((uint8_t)i) ^ i)) ctx->xt = false;
    uint8_t lut[256 * 32];
    build_fr_lut(lut);
    int rc = AESW_OK;
    auto T = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == AESW_OK) rc = fail_hip(ctx, e, what);
    };
    T(hipMalloc(reinterpret_cast<void **>(&ctx->d_tables), 768), "hipMalloc(tables)");
    T(hipMalloc(reinterpret_cast<void **>(&ctx->d_fr_lut), sizeof lut), "hipMalloc(fr_lut)");
    // the flush schedules depend on the layout only: searched once per process (~20 ms each), uploaded per context
    static std::vector<uint32_t> host_ftab[3];
    static std::once_flag ftab_once;
    std::call_once(ftab_once, [] {
        for (int l = 0; l < 3; ++l) {
            host_ftab[l].resize((size_t)flush_table_words(l));
            build_flush_tables(l, host_ftab[l].data());
        }
    });
    for (int l = 0; l < 3 && rc == AESW_OK; ++l) {
        const std::vector<uint32_t> &ft = host_ftab[l];
        T(hipMalloc(reinterpret_cast<void **>(&ctx->d_ftab[l]), ft.size() * sizeof(uint32_t)), "hipMalloc(flush table)");
        if (rc == AESW_OK) T(hipMemcpy(ctx->d_ftab[l], ft.data(), ft.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy(flush table)");
    }
    for (int li = 0; li < 2 && rc == AESW_OK; ++li) {  // check tables of the DENSE / PACKED layout (aesw_check.h): 24 KiB each
        std::vector<uint32_t> ct((size_t)CHK_WORDS);
        build_check_table(li == 0 ? AESW_LAYOUT_DENSE : AESW_LAYOUT_PACKED, ct.data());
        T(hipMalloc(reinterpret_cast<void **>(&ctx->d_chktab[li]), ct.size() * sizeof(uint32_t)), "hipMalloc(check table)");
        if (rc == AESW_OK) T(hipMemcpy(ctx->d_chktab[li], ct.data(), ct.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy(check table)");
    }
    if (rc == AESW_OK) { int first = -1; rc = key_new_slot(ctx, &first); if (rc == AESW_OK) ctx->key_spare.push_back(first); }  // the first chunk of round-key slots
    if (rc == AESW_OK) T(warm_launch_attributes(), "hipFuncSetAttribute(max dynamic LDS)");
    if (rc == AESW_OK) T(hipMemcpy(ctx->d_tables, host, 768, hipMemcpyHostToDevice), "hipMemcpy(tables)");
    if (rc == AESW_OK) T(hipMemcpy(ctx->d_fr_lut, lut, sizeof lut, hipMemcpyHostToDevice), "hipMemcpy(fr_lut)");
    if (rc != AESW_OK) {
        aesw_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return AESW_OK;
}

void aesw_destroy(aesw_ctx *ctx) {
    if (!ctx) return;
    {
        DeviceGuard g(ctx->device);
        if (ctx->s_compute) (void)hipStreamDestroy(ctx->s_compute);
        if (ctx->s_copy) (void)hipStreamDestroy(ctx->s_copy);
        for (int j = 0; j < 8; ++j) {
            if (ctx->s_batch[j]) (void)hipStreamDestroy(ctx->s_batch[j]);
            if (ctx->ev_join[j]) (void)hipEventDestroy(ctx->ev_join[j]);
        }
        if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
        for (int i = 0; i < 2; ++i)
            if (ctx->bounce[i]) (void)hipHostFree(ctx->bounce[i]);
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        for (auto &a : ctx->vmm_arenas)
            for (auto &r : a.ranges) { if (r.vmm) vmm_release_arena(r.p, r.bytes); else (void)hipFree(r.p); }
        aesw_arena_cache_trim(ctx, 0);
        if (ctx->d_tables) (void)hipFree(ctx->d_tables);
        if (ctx->d_fr_lut) (void)hipFree(ctx->d_fr_lut);
        for (auto &sl : ctx->key_slots) {
            if (sl.ready) (void)hipEventDestroy(sl.ready);
            for (auto &r : sl.readers) (void)hipEventDestroy(r.e);
        }
        for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
        for (uint8_t *c : ctx->key_chunks) (void)hipFree(c);
        for (uint32_t *t : ctx->d_ftab)
            if (t) (void)hipFree(t);
        for (uint32_t *t : ctx->d_chktab)
            if (t) (void)hipFree(t);
    }
    delete ctx;
}

int aesw_device(const aesw_ctx *ctx) { return ctx ? ctx->device : -1; }
int aesw_uses_xtime_path(const aesw_ctx *ctx) { return ctx && ctx->xt ? 1 : 0; }

// ---- geometry ------------------------------------------------------------------

uint32_t aesw_column_stride(int layout, int col) {
    if (!valid_layout(layout) || col < 0 || col > 2) return 0;
    if (layout == AESW_LAYOUT_DENSE) return AESW_AES_ROWS;
    if (layout == AESW_LAYOUT_VALUES) return col == 0 ? Geo<VALUES>::XS : col == 1 ? Geo<VALUES>::YS : Geo<VALUES>::ZS;
    return col == 0 ? Geo<PACKED>::XS : col == 1 ? Geo<PACKED>::YS : Geo<PACKED>::ZS;
}

uint32_t aesw_key_column_stride(int layout, int col) {
    if (!valid_layout(layout) || col < 0 || col > 2) return 0;
    if (layout == AESW_LAYOUT_DENSE) return AESW_KEY_ROWS;
    return col == 0 ? Geo<PACKED>::KXS : col == 1 ? Geo<PACKED>::KYS : Geo<PACKED>::KZS;
}

int aesw_packed_index(int col, int32_t idx[AESW_AES_ROWS]) {
    if (col < 0 || col > 2 || !idx) return AESW_ERR_INVALID_ARG;
    uint8_t mask[AES_ROWS];
    encrypt_assigned_mask(col, mask);
    int32_t n = 0;
    for (int r = 0; r < AES_ROWS; ++r) idx[r] = mask[r] ? n++ : -1;
    return AESW_OK;
}

int aesw_layout_index(int layout, int col, int32_t idx[AESW_AES_ROWS]) {
    if (!valid_layout(layout) || col < 0 || col > 2 || !idx) return AESW_ERR_INVALID_ARG;
    if (layout == AESW_LAYOUT_DENSE) {
        for (int r = 0; r < AES_ROWS; ++r) idx[r] = r;
        return AESW_OK;
    }
    if (layout == AESW_LAYOUT_PACKED) return aesw_packed_index(col, idx);
    uint8_t mask[AES_ROWS];
    encrypt_values_mask(col, mask);
    int32_t n = 0;
    for (int r = 0; r < AES_ROWS; ++r) idx[r] = mask[r] ? n++ : -1;
    return AESW_OK;
}

int aesw_key_packed_index(int col, int32_t idx[AESW_KEY_ROWS]) {
    if (col < 0 || col > 2 || !idx) return AESW_ERR_INVALID_ARG;
    uint8_t mask[KEY_ROWS];
    key_assigned_mask(col, mask);
    int32_t n = 0;
    for (int r = 0; r < KEY_ROWS; ++r) idx[r] = mask[r] ? n++ : -1;
    return AESW_OK;
}

static_assert(sizeof(aesw_copy_edge) == sizeof(CopyEdge) && sizeof(CopyEdge) == 8, "copy edge layout");
int aesw_block_copy_graph(aesw_copy_edge edges[AESW_BLOCK_COPIES]) {
    if (!edges) return AESW_ERR_INVALID_ARG;
    static_assert(AESW_BLOCK_COPIES == BLOCK_COPIES, "block copies");
    return block_copy_graph(reinterpret_cast<CopyEdge *>(edges)) == BLOCK_COPIES ? AESW_OK : AESW_ERR_INVALID_ARG;
}
int aesw_key_copy_graph(aesw_copy_edge edges[AESW_KEY_COPIES]) {
    if (!edges) return AESW_ERR_INVALID_ARG;
    static_assert(AESW_KEY_COPIES == KEY_COPIES, "key copies");
    return key_copy_graph(reinterpret_cast<CopyEdge *>(edges)) == KEY_COPIES ? AESW_OK : AESW_ERR_INVALID_ARG;
}

int aesw_selector_tags(uint8_t enc_tag[AESW_AES_ROWS], uint8_t key_tag[AESW_KEY_ROWS], uint8_t q_eq_rcon[AESW_WORDS_ROWS],
                       uint8_t rcon_fixed[AESW_WORDS_ROWS]) {
    uint8_t e[AES_ROWS], k[KEY_ROWS], q[WORDS_ROWS], c[WORDS_ROWS];
    encrypt_selector_tags(e);
    key_selector_tags(k, q, c);
    if (enc_tag) std::memcpy(enc_tag, e, sizeof e);
    if (key_tag) std::memcpy(key_tag, k, sizeof k);
    if (q_eq_rcon) std::memcpy(q_eq_rcon, q, sizeof q);
    if (rcon_fixed) std::memcpy(rcon_fixed, c, sizeof c);
    return AESW_OK;
}

// FixedAes128Config::aes_callable, src/aes128.rs:303-325: set 0 is charged
// KEY_SCHEDULE_ROWS (1760) of its 2^K rows, every set holds whole 1360-row
// blocks.  Rows: set 0 starts behind the 400 rows the key schedule really uses.
static uint64_t set_capacity(uint32_t k, uint32_t set) {
    uint64_t max_row = (uint64_t)1 << k;
    if (set == 0) {
        if (max_row < AESW_KEY_SCHEDULE_ROWS) return 0;
        max_row -= AESW_KEY_SCHEDULE_ROWS;
    }
    return max_row / AESW_AES_ROWS;
}

uint64_t aesw_block_capacity(uint32_t k, uint32_t n_sets) {
    if (k > 40 || n_sets == 0) return 0;
    uint64_t total = 0;
    for (uint32_t s = 0; s < n_sets; ++s) total += set_capacity(k, s);
    return total;
}

int aesw_block_placement(uint32_t k, uint32_t n_sets, uint64_t b, uint32_t *set, uint64_t *row) {
    if (k > 40 || n_sets == 0 || !set || !row) return AESW_ERR_INVALID_ARG;
    uint64_t left = b;
    for (uint32_t s = 0; s < n_sets; ++s) {
        const uint64_t cap = set_capacity(k, s);
        if (left < cap) {
            *set = s;
            *row = (s == 0 ? AESW_KEY_ROWS : 0) + left * AESW_AES_ROWS;
            return AESW_OK;
        }
        left -= cap;
    }
    return AESW_ERR_CAPACITY;
}

int aesw_assemble_selectors(uint32_t k, uint32_t n_sets, uint64_t n_blocks, uint8_t *selectors, uint8_t *fixed) {
    if (k < 2 || k > 32 || n_sets == 0 || n_sets > 1024 || !selectors) return AESW_ERR_INVALID_ARG;
    if (n_blocks > aesw_block_capacity(k, n_sets)) return AESW_ERR_CAPACITY;
    const uint64_t rows = (uint64_t)1 << k;
    uint8_t enc[AES_ROWS], key[KEY_ROWS], q[WORDS_ROWS], rc[WORDS_ROWS];
    encrypt_selector_tags(enc);
    key_selector_tags(key, q, rc);
    std::memset(selectors, 0, (size_t)(5 * n_sets + 1) * rows);
    if (fixed) std::memset(fixed, 0, rows);
    auto sel = [&](uint32_t set, int tag) { return selectors + (size_t)(5 * set + (tag - 1)) * rows; };  // tag 1..5 = selector order
    if (rows >= KEY_ROWS)
        for (uint32_t r = 0; r < KEY_ROWS; ++r)
            if (key[r]) sel(0, key[r])[r] = 1;
    for (uint32_t r = 0; r < WORDS_ROWS && r < rows; ++r) {
        selectors[(size_t)(5 * n_sets) * rows + r] = q[r];
        if (fixed) fixed[r] = rc[r];
    }
    for (uint64_t b = 0; b < n_blocks; ++b) {
        uint32_t set;
        uint64_t row;
        if (aesw_block_placement(k, n_sets, b, &set, &row) != AESW_OK) return AESW_ERR_CAPACITY;
        for (uint32_t r = 0; r < AES_ROWS; ++r)
            if (enc[r]) sel(set, enc[r])[row + r] = 1;
    }
    return AESW_OK;
}

// ---- options --------------------------------------------------------------------

int aesw_set_option(aesw_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return AESW_ERR_INVALID_ARG;
    if (!std::strcmp(name, "waves_shared")) { if (value < 0 || value > 4) return AESW_ERR_INVALID_ARG; ctx->waves_shared = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "waves_pbk")) { if (value < 0 || value > 4) return AESW_ERR_INVALID_ARG; ctx->waves_pbk = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "nt_stores")) { ctx->nt = value != 0 ? 1 : 0; return AESW_OK; }
    if (!std::strcmp(name, "store_mode")) {
        // 3 = "leave the flush out" (output is garbage): exists only in -DAESW_DIAGNOSTIC builds of the library (tools/)
#ifdef AESW_DIAGNOSTIC
        const int max_mode = 5;
#else
        const int max_mode = 2;
#endif
        if (value < 0 || value > max_mode) return AESW_ERR_INVALID_ARG;
        ctx->nt = (int)value;
        return AESW_OK;
    }
    if (!std::strcmp(name, "key_store_mode")) { if (value < 0 || value > 2) return AESW_ERR_INVALID_ARG; ctx->key_nt = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "fr_geometry")) { if (value < 0 || value > 2) return AESW_ERR_INVALID_ARG; ctx->fr_geo = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "fr_store_mode")) { if (value < 0 || value > 2) return AESW_ERR_INVALID_ARG; ctx->fr_nt = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "assemble_geometry")) { if (value < 0 || value > 4) return AESW_ERR_INVALID_ARG; ctx->asm_geo = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "grid_cap")) { if (value < 0 || value > 0x7fffffff) return AESW_ERR_INVALID_ARG; ctx->grid_cap = value; return AESW_OK; }
    if (!std::strcmp(name, "xcd_remap")) { if (value < 0 || value > (1 << 24)) return AESW_ERR_INVALID_ARG; ctx->xcd_remap = (uint32_t)value; return AESW_OK; }
    if (!std::strcmp(name, "lds_pad")) { if (value < 0 || value > 120 * 1024) return AESW_ERR_INVALID_ARG; ctx->lds_pad = value; return AESW_OK; }
    if (!std::strcmp(name, "arena_align_log2")) { if (value != 0 && (value < 7 || value > 32)) return AESW_ERR_INVALID_ARG; ctx->arena_align_log2 = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "arena_probe")) { if (value < -1 || value > 64) return AESW_ERR_INVALID_ARG; ctx->arena_probe = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "arena_unit")) { if (value < 0 || value > 2) return AESW_ERR_INVALID_ARG; ctx->arena_unit = (int)value; return AESW_OK; }
#ifdef AESW_TRACE
    if (!std::strcmp(name, "trace_ptr")) { ctx->trace = reinterpret_cast<uint64_t *>(value); return AESW_OK; }
#endif
    if (!std::strcmp(name, "force_table_path")) { if (value) ctx->xt = false; return AESW_OK; }
    if (!std::strcmp(name, "chunk_blocks")) { if (value < 64) return AESW_ERR_INVALID_ARG; ctx->chunk_blocks = value; return AESW_OK; }
    if (!std::strcmp(name, "batch_streams")) { if (value < 1 || value > 8) return AESW_ERR_INVALID_ARG; ctx->batch_streams = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "copy_threads")) { if (value < -1 || value > 64) return AESW_ERR_INVALID_ARG; ctx->copy_threads = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "key_slots")) { if (value < 1 || value > 64) return AESW_ERR_INVALID_ARG; ctx->key_ring = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "split_small")) { if (value < 0 || value > 8) return AESW_ERR_INVALID_ARG; ctx->split_small = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "stream_check")) { if (value != 0 && value != 1) return AESW_ERR_INVALID_ARG; ctx->stream_check = (int)value; return AESW_OK; }
    if (!std::strcmp(name, "stream_poison")) { if (value < 0) return AESW_ERR_INVALID_ARG; ctx->stream_poison = value; return AESW_OK; }
    if (!std::strcmp(name, "arena_cache")) {  // 0 also releases what is cached now
        if (value != 0 && value != 1) return AESW_ERR_INVALID_ARG;
        ctx->arena_cache_on = (int)value;
        if (!value) aesw_arena_cache_trim(ctx, 0);
        return AESW_OK;
    }
    if (!std::strcmp(name, "arena_cache_max_mb")) {
        if (value < 0 || value > ((int64_t)1 << 30)) return AESW_ERR_INVALID_ARG;
        ctx->arena_cache_max_bytes = (uint64_t)value << 20;
        aesw_arena_cache_trim(ctx, ctx->arena_cache_max_bytes);
        return AESW_OK;
    }
    if (!std::strcmp(name, "arena_probe_budget_ms")) { if (value < 0 || value > 600000) return AESW_ERR_INVALID_ARG; ctx->arena_probe_budget_ms = value; return AESW_OK; }
    return AESW_ERR_INVALID_ARG;
}

static int auto_waves(const aesw_ctx *ctx, int layout, bool pbk);
// Pageable destinations of the host-pointer entry points: a stage arrives in the page-locked bounce buffer by DMA and is
// moved on from there by the CPU.  One thread moves ~20 GB/s (less into memory it touches for the first time), the link
// delivers 55: the move is cut into 4 MiB slices handed out to "copy_threads" threads (the caller is one of them).
struct CopyJob { uint8_t *dst; const uint8_t *src; size_t bytes; };
static int auto_copy_threads(const aesw_ctx *ctx) {
    if (ctx->copy_threads >= 0) return ctx->copy_threads < 1 ? 1 : ctx->copy_threads;
    cpu_set_t set;
    int usable = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0) usable = CPU_COUNT(&set);
    const int t = usable / 4;  // leave the host its cores: a quarter of what this process may run on, 1 ... 4
    return t < 1 ? 1 : (t > 4 ? 4 : t);
}
static void parallel_copy(const std::vector<CopyJob> &jobs, int threads) noexcept {
    constexpr size_t SLICE = (size_t)4 << 20;
    std::vector<CopyJob> slices;
    std::vector<std::thread> pool;
    try {
        for (const CopyJob &j : jobs)
            for (size_t o = 0; o < j.bytes; o += SLICE) slices.push_back(CopyJob{j.dst + o, j.src + o, j.bytes - o < SLICE ? j.bytes - o : SLICE});
        pool.reserve(threads > 1 ? (size_t)threads - 1 : 0);
    } catch (...) {  // no memory for the bookkeeping: copy on this thread (nothing has been started yet)
        for (const CopyJob &j : jobs) std::memcpy(j.dst, j.src, j.bytes);
        return;
    }
    if ((int)slices.size() < threads) threads = (int)slices.size();
    std::atomic<size_t> next{0};
    auto work = [&]() noexcept {
        for (size_t i = next.fetch_add(1); i < slices.size(); i = next.fetch_add(1)) std::memcpy(slices[i].dst, slices[i].src, slices[i].bytes);
    };
    for (int t = 1; t < threads; ++t) {
        try { pool.emplace_back(work); } catch (...) { break; }  // no thread to be had: the others and the caller copy the rest
    }
    work();
    for (std::thread &t : pool) t.join();
}

static int auto_waves_key(const aesw_ctx *ctx, int layout, bool want_rk);

int aesw_get_option(const aesw_ctx *ctx, const char *name, int64_t *value) {
    if (!ctx || !name || !value) return AESW_ERR_INVALID_ARG;
    if (!std::strcmp(name, "waves_shared")) { *value = ctx->waves_shared; return AESW_OK; }
    if (!std::strcmp(name, "waves_pbk")) { *value = ctx->waves_pbk; return AESW_OK; }
    // what a launch really uses (0 = auto resolved, values above the layout's maximum clamped): packed layout
    if (!std::strcmp(name, "effective_waves_shared")) { *value = auto_waves(ctx, AESW_LAYOUT_PACKED, false); return AESW_OK; }
    if (!std::strcmp(name, "effective_waves_pbk")) { *value = auto_waves(ctx, AESW_LAYOUT_PACKED, true); return AESW_OK; }
    if (!std::strcmp(name, "effective_waves_key")) { *value = auto_waves_key(ctx, AESW_LAYOUT_PACKED, false); return AESW_OK; }
    if (!std::strcmp(name, "nt_stores")) { *value = ctx->nt == 1; return AESW_OK; }
    if (!std::strcmp(name, "store_mode")) { *value = ctx->nt; return AESW_OK; }
    if (!std::strcmp(name, "key_store_mode")) { *value = ctx->key_nt; return AESW_OK; }
    if (!std::strcmp(name, "fr_store_mode")) { *value = ctx->fr_nt; return AESW_OK; }
    if (!std::strcmp(name, "fr_geometry")) { *value = ctx->fr_geo; return AESW_OK; }
    if (!std::strcmp(name, "assemble_geometry")) { *value = ctx->asm_geo; return AESW_OK; }
    if (!std::strcmp(name, "grid_cap")) { *value = ctx->grid_cap; return AESW_OK; }
    if (!std::strcmp(name, "xcd_remap")) { *value = ctx->xcd_remap; return AESW_OK; }
    if (!std::strcmp(name, "lds_pad")) { *value = ctx->lds_pad; return AESW_OK; }
    if (!std::strcmp(name, "arena_align_log2")) { *value = ctx->arena_align_log2; return AESW_OK; }
    if (!std::strcmp(name, "arena_probe")) { *value = ctx->arena_probe; return AESW_OK; }
    if (!std::strcmp(name, "arena_unit")) { *value = ctx->arena_unit; return AESW_OK; }
    if (!std::strcmp(name, "force_table_path")) { *value = ctx->xt ? 0 : 1; return AESW_OK; }
    if (!std::strcmp(name, "chunk_blocks")) { *value = ctx->chunk_blocks; return AESW_OK; }
    if (!std::strcmp(name, "batch_streams")) { *value = ctx->batch_streams; return AESW_OK; }
    if (!std::strcmp(name, "copy_threads")) { *value = ctx->copy_threads; return AESW_OK; }
    if (!std::strcmp(name, "effective_copy_threads")) { *value = auto_copy_threads(ctx); return AESW_OK; }
    if (!std::strcmp(name, "key_slots")) { *value = ctx->key_ring; return AESW_OK; }
    if (!std::strcmp(name, "split_small")) { *value = ctx->split_small; return AESW_OK; }
    if (!std::strcmp(name, "stream_check")) { *value = ctx->stream_check; return AESW_OK; }
    if (!std::strcmp(name, "stream_poison")) { *value = ctx->stream_poison; return AESW_OK; }
    if (!std::strcmp(name, "arena_cache")) { *value = ctx->arena_cache_on; return AESW_OK; }
    if (!std::strcmp(name, "arena_cache_max_mb")) { *value = (int64_t)(ctx->arena_cache_max_bytes >> 20); return AESW_OK; }
    if (!std::strcmp(name, "arena_probe_budget_ms")) { *value = ctx->arena_probe_budget_ms; return AESW_OK; }
    if (!std::strcmp(name, "arena_cache_hits")) { *value = (int64_t)ctx->arena_cache_hits; return AESW_OK; }  // read-only statistics
    if (!std::strcmp(name, "arena_cached_bytes")) {
        uint64_t b = 0;
        for (const auto &c : ctx->arena_cache) b += c.cols.bytes;
        *value = (int64_t)b;
        return AESW_OK;
    }
    if (!std::strcmp(name, "key_reader_waits")) { *value = (int64_t)ctx->key_waits; return AESW_OK; }  // read-only statistics
    if (!std::strcmp(name, "key_slots_allocated")) { *value = (int64_t)ctx->key_slots.size(); return AESW_OK; }
    if (!std::strcmp(name, "key_slots_pinned")) {
        int64_t n = 0;
        for (const auto &sl : ctx->key_slots) n += sl.pinned ? 1 : 0;
        *value = n;
        return AESW_OK;
    }
    return AESW_ERR_INVALID_ARG;
}

// ---- device-pointer entry points ------------------------------------------------

// Waves per group when the option is 0 (auto): as many 16-block waves as keep
// 6-8 waves resident per CU given the LDS windows (DESIGN.md "occupancy").
static int auto_waves(const aesw_ctx *ctx, int layout, bool pbk) {
    // upper bound: a group's staging must stay below 64 KiB (16-bit LDS addresses in the flush descriptors)
    const int max_waves = layout == AESW_LAYOUT_DENSE ? 2 : layout == AESW_LAYOUT_VALUES ? 4 : 3;
    int w;
    // per-block keys: one-wave groups (7 resident per CU instead of two 3-wave groups) measured +1.3 ... +2.4 % at 2^20
    // blocks on two boxes and -0.6 % on a third (tools/sweep.py 20 c2 packed waves); shared key: 3-wave groups
    if (pbk) w = ctx->waves_pbk ? ctx->waves_pbk : 1;
    else w = ctx->waves_shared ? ctx->waves_shared : (layout == AESW_LAYOUT_DENSE ? 2 : 3);
    return w > max_waves ? max_waves : w;
}

// key_kernel alone (tools/keysweep.py, 2^20 keys): packed 4-wave groups, dense 2-wave groups
static int auto_waves_key(const aesw_ctx *ctx, int layout, bool want_rk) {
    // packed, witness only (no round-key output: no 2.8 KB round-key staging per wave since round 4): three 3-wave groups fit a CU
    // (45.7 KB each) and run 143.7 us against 146.3 for 4-wave groups and 148.0 / 154.9 for 2 / 1 (tools/keyarena.py,
    // profiles/r04_study/key_kernel_kz.md); with round keys a 3-wave group is 54 KB (two per CU): 4-wave groups as before
    if (ctx->waves_pbk) return ctx->waves_pbk;
    if (layout == AESW_LAYOUT_DENSE) return 2;
    return want_rk ? 4 : 3;
}

int aesw_schedule_key_device(aesw_ctx *ctx, const uint8_t *d_key, int layout, const aesw_key_slab *ks, void *stream) {
    if (!ctx || !valid_layout(layout) || !d_key || !aligned4(d_key)) return AESW_ERR_INVALID_ARG;
    KeyOut ko{nullptr, nullptr, nullptr, nullptr};
    if (ks) {
        ko = KeyOut{ks->w, ks->kx, ks->ky, ks->kz};
        if ((ko.w && !aligned16(ko.w)) || (ko.kx && !aligned16(ko.kx)) || (ko.ky && !aligned16(ko.ky)) ||
            (ko.kz && !aligned16(ko.kz)))
            return AESW_ERR_INVALID_ARG;
    }
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool cap = stream_capturing(s);
    int slot = -1;
    if (cap) {
        // a captured schedule writes its slot on every replay of the graph, whenever that is: a slot of its own, never reused
        const int rc = key_new_slot(ctx, &slot);
        if (rc != AESW_OK) return rc;
        ctx->key_slots[slot].pinned = true;
    } else {
        const int rc = key_next_ring_slot(ctx, &slot);
        if (rc != AESW_OK) return rc;
        // write-after-read: every launch that may still read this slot's previous key, on whatever stream, comes first
        aesw_ctx::KeySlot &sl = ctx->key_slots[slot];
        hipError_t e = hipSuccess;
        for (auto &r : sl.readers) {
            if (e == hipSuccess) e = hipStreamWaitEvent(s, r.e, 0);
            if (e != hipSuccess) break;
            ++ctx->key_waits;
        }
        if (e != hipSuccess) {
            // keep the readers: the slot has not been written, and the ring must come back to it with them intact
            ctx->key_pos = (ctx->key_pos + (int)ctx->key_ring_slots.size() - 1) % (int)ctx->key_ring_slots.size();
            return fail_hip(ctx, e, "hipStreamWaitEvent(key readers)");
        }
        for (auto &r : sl.readers) ctx->event_pool.push_back(r.e);
        sl.readers.clear();
    }
    aesw_ctx::KeySlot &sl = ctx->key_slots[slot];
    KeyParams kp{d_key, ctx->d_tables, ko, sl.d, 1, 0, 0};
    HIP_TRY(ctx, launch_key(kp, layout, ctx->xt, 1, ctx->key_nt, 0u, s));
    // a later encrypt on ANOTHER stream (the host-pointer entry points use the context's own) waits for these round keys
    if (!cap) HIP_TRY(ctx, hipEventRecord(sl.ready, s));
    sl.writer = s;
    ctx->key_cur = slot;
    ctx->have_key = true;
    return AESW_OK;
}

int aesw_encrypt_witness_device(aesw_ctx *ctx, const uint8_t *d_pt, const uint8_t *d_keys, int per_block_keys,
                                uint64_t n, int layout, uint8_t *d_x, uint8_t *d_y, uint8_t *d_z, uint8_t *d_ct,
                                const aesw_key_slab *ks, void *stream) {
    if (!ctx || !valid_layout(layout)) return AESW_ERR_INVALID_ARG;
    if (!d_keys) {
        if (per_block_keys) return AESW_ERR_INVALID_ARG;
        if (!ctx->have_key) return AESW_ERR_NO_KEY;  // "Keys should be scheduled", src/aes128.rs:170
    }
    if (n == 0) return AESW_OK;
    const bool has_x = aesw_column_stride(layout, 0) != 0;  // AESW_LAYOUT_VALUES has no x column: d_x is ignored
    if (!d_pt || (has_x && !d_x) || !d_y || !d_z) return AESW_ERR_INVALID_ARG;
    if ((has_x && !aligned16(d_x)) || !aligned16(d_y) || !aligned16(d_z) || !aligned4(d_pt) || (d_keys && !aligned4(d_keys)) ||
        (d_ct && !aligned4(d_ct)))
        return AESW_ERR_INVALID_ARG;
    KeyOut ko{nullptr, nullptr, nullptr, nullptr};
    if (ks) {
        ko = KeyOut{ks->w, ks->kx, ks->ky, ks->kz};
        if ((ko.w && !aligned16(ko.w)) || (ko.kx && !aligned16(ko.kx)) || (ko.ky && !aligned16(ko.ky)) ||
            (ko.kz && !aligned16(ko.kz)))
            return AESW_ERR_INVALID_ARG;
    }
    const bool kemit = ko.w || ko.kx || ko.ky || ko.kz;
    if (!d_keys && kemit) return AESW_ERR_INVALID_ARG;  // the key slab of a scheduled key comes from aesw_schedule_key*
    if (ctx->split_small > 1 && !ctx->in_split && !per_block_keys && !kemit && n >= ((uint64_t)1 << 15) && n <= ((uint64_t)1 << 17)) {
        // "split_small": the lone small batch as 2-3 sub-launches on the internal streams (fork / join as in the batch entry
        // point).  Sub-ranges are multiples of 48 blocks -- whole 3-wave groups, and 48 x 1360 / 1056 / 608 are multiples of the
        // 128-byte line, so no two sub-launches share a line of any column.
        const uint32_t parts = (uint32_t)ctx->split_small;
        const uint64_t per = ((n + parts - 1) / parts + 47) / 48 * 48;
        aesw_batch b[8];
        uint32_t cnt = 0;
        const uint64_t sx = aesw_column_stride(layout, 0), sy = aesw_column_stride(layout, 1), sz = aesw_column_stride(layout, 2);
        for (uint64_t lo = 0; lo < n && cnt < 8; lo += per, ++cnt) {
            const uint64_t m = n - lo < per ? n - lo : per;
            b[cnt] = aesw_batch{d_pt + lo * 16, d_keys, m, d_x ? d_x + lo * sx : nullptr, d_y + lo * sy, d_z + lo * sz,
                                d_ct ? d_ct + lo * 16 : nullptr, nullptr};
        }
        const int keep = ctx->batch_streams;
        ctx->batch_streams = (int)cnt;
        ctx->in_split = true;
        const int rc = aesw_encrypt_witness_batches_device(ctx, b, cnt, 0, layout, stream);
        ctx->in_split = false;
        ctx->batch_streams = keep;
        return rc;
    }
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!per_block_keys && kemit) {
        // shared key: its schedule witness is one key slab
        KeyParams kp{d_keys, ctx->d_tables, ko, nullptr, 1, 0, 0};
        HIP_TRY(ctx, launch_key(kp, layout, ctx->xt, 1, ctx->key_nt, 0u, s));
    }
    const int km = per_block_keys ? 0 : (d_keys ? 1 : 2);
    const bool cap = km == 2 && stream_capturing(s);
    if (km == 2 && s != ctx->key_slots[ctx->key_cur].writer) {
        // the round keys were written on another stream: order this launch behind them
        aesw_ctx::KeySlot &sl = ctx->key_slots[ctx->key_cur];
        if (cap && same_capture(s, sl.writer)) {
            // `s` was forked (with an event) from the capture on the key's own stream -- the internal streams of the batch entry
            // point and of "split_small" are: whatever ordered the key in front of that capture orders it in front of `s` too.
            // (Asking the key's event would be an error here: its stream is the one being captured.)
        } else if (cap) {
            // a captured launch cannot take a dependency on work outside its graph.  If the key launch has already finished,
            // there is nothing to depend on; otherwise refuse instead of dropping the wait silently
            RelaxedCapture relaxed;
            const bool done = !sl.pinned && hipEventQuery(sl.ready) == hipSuccess;
            (void)hipGetLastError();
            if (!done) {
                ctx->last_error = "scheduled-key encrypt captured on a stream other than the one aesw_schedule_key_device ran on, "
                                  "and the key launch has not finished (or was itself captured): synchronise first, or capture both on one stream";
                return AESW_ERR_INVALID_ARG;
            }
        } else if (!sl.pinned) {
            HIP_TRY(ctx, hipStreamWaitEvent(s, sl.ready, 0));
        }  // (a slot written by a captured schedule has no event: the caller orders its graph launches, include/aesw.h)
    }
    EncParams p{d_pt, d_keys, km == 2 ? reinterpret_cast<const uint32_t *>(ctx->key_slots[ctx->key_cur].d) : nullptr, ctx->d_tables, ctx->d_ftab[layout], d_x, d_y, d_z, d_ct,
                per_block_keys ? ko : KeyOut{nullptr, nullptr, nullptr, nullptr}, n, 0, 0};
#ifdef AESW_TRACE
    p.trace = ctx->trace;
#endif
    HIP_TRY(ctx, launch_encrypt(p, layout, ctx->xt, km, per_block_keys && kemit, auto_waves(ctx, layout, per_block_keys != 0),
                                ctx->nt, (uint32_t)ctx->grid_cap, ctx->xcd_remap, (uint32_t)ctx->lds_pad, s));
    if (km == 2) {  // this launch reads the current slot: nothing may overwrite the slot under it
        aesw_ctx::KeySlot &sl = ctx->key_slots[ctx->key_cur];
        if (cap) sl.pinned = true;  // read on every replay of the graph, whenever that is: the ring never reuses the slot
        else {
            const int rc = key_track_reader(ctx, sl, s);
            if (rc != AESW_OK) return rc;
        }
    }
    return AESW_OK;
}

int aesw_encrypt_witness_batches_device(aesw_ctx *ctx, const aesw_batch *batches, uint32_t count, int per_block_keys, int layout,
                                        void *stream) {
    if (!ctx || !valid_layout(layout) || (count && !batches)) return AESW_ERR_INVALID_ARG;
    if (count == 0) return AESW_OK;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    const uint32_t ns = (uint32_t)ctx->batch_streams < count ? (uint32_t)ctx->batch_streams : count;
    if (ns <= 1) {  // nothing to overlap: plain launches on the caller's stream
        for (uint32_t i = 0; i < count; ++i) {
            const aesw_batch &b = batches[i];
            const int rc = aesw_encrypt_witness_device(ctx, b.d_pt, b.d_keys, per_block_keys, b.n, layout, b.d_x, b.d_y, b.d_z, b.d_ct, b.d_key_slab, stream);
            if (rc != AESW_OK) return rc;
        }
        return AESW_OK;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!ctx->ev_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    for (uint32_t j = 0; j < ns; ++j) {
        if (!ctx->s_batch[j]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_batch[j], hipStreamNonBlocking));
        if (!ctx->ev_join[j]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join[j], hipEventDisableTiming));
    }
    // fork: the internal streams start behind what the caller's stream holds
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
    for (uint32_t j = 0; j < ns; ++j) HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_batch[j], ctx->ev_fork, 0));
    int rc = AESW_OK;
    const bool outer = !ctx->in_split;
    ctx->in_split = true;  // the launches below already sit on the internal streams: "split_small" must not deal them out again
    for (uint32_t i = 0; i < count && rc == AESW_OK; ++i) {
        const aesw_batch &b = batches[i];
        rc = aesw_encrypt_witness_device(ctx, b.d_pt, b.d_keys, per_block_keys, b.n, layout, b.d_x, b.d_y, b.d_z, b.d_ct, b.d_key_slab,
                                         ctx->s_batch[i % ns]);
    }
    if (outer) ctx->in_split = false;
    // join, also after a failed launch: what was issued must be ordered before whatever the caller does next on `stream`
    for (uint32_t j = 0; j < ns; ++j) {
        const hipError_t e1 = hipEventRecord(ctx->ev_join[j], ctx->s_batch[j]);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, ctx->ev_join[j], 0) : e1;
        if (e2 != hipSuccess && rc == AESW_OK) rc = fail_hip(ctx, e2, "join of the batch streams");
    }
    return rc;
}

int aesw_key_schedule_witness_device(aesw_ctx *ctx, const uint8_t *d_keys, uint64_t n, int layout, uint8_t *d_w,
                                     uint8_t *d_kx, uint8_t *d_ky, uint8_t *d_kz, uint8_t *d_rk, void *stream) {
    if (!ctx || !valid_layout(layout)) return AESW_ERR_INVALID_ARG;
    if (n == 0) return AESW_OK;
    if (!d_keys || !aligned4(d_keys)) return AESW_ERR_INVALID_ARG;
    if ((d_w && !aligned16(d_w)) || (d_kx && !aligned16(d_kx)) || (d_ky && !aligned16(d_ky)) ||
        (d_kz && !aligned16(d_kz)) || (d_rk && !aligned16(d_rk)))
        return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    KeyParams kp{d_keys, ctx->d_tables, KeyOut{d_w, d_kx, d_ky, d_kz}, d_rk, n, 0, 0};
    HIP_TRY(ctx, launch_key(kp, layout, ctx->xt, auto_waves_key(ctx, layout, d_rk != nullptr), ctx->key_nt, ctx->xcd_remap, reinterpret_cast<hipStream_t>(stream)));
    return AESW_OK;
}

int aesw_lookup_table_device(aesw_ctx *ctx, uint8_t *d_t0, uint8_t *d_t1, uint8_t *d_t2, uint8_t *d_t3, void *stream) {
    if (!ctx || !d_t0 || !d_t1 || !d_t2 || !d_t3) return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    HIP_TRY(ctx, launch_table(ctx->d_tables, d_t0, d_t1, d_t2, d_t3, reinterpret_cast<hipStream_t>(stream)));
    return AESW_OK;
}

namespace {
int fill_assemble_params(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, uint64_t n_blocks, int layout, const uint8_t *d_x, const uint8_t *d_y,
                         const uint8_t *d_z, const aesw_key_slab *ks, AssembleParams *p) {
    if (!ctx || !valid_layout(layout) || layout == AESW_LAYOUT_VALUES /* whole columns need every cell */ || k < 2 || k > 32 ||
        n_sets == 0 || n_sets > 1024)
        return AESW_ERR_INVALID_ARG;
    if (n_blocks && (!d_x || !d_y || !d_z)) return AESW_ERR_INVALID_ARG;
    if (n_blocks > aesw_block_capacity(k, n_sets)) return AESW_ERR_CAPACITY;  // panic in the reference, src/aes128.rs:160-162
    *p = AssembleParams{};
    p->x = d_x; p->y = d_y; p->z = d_z;
    if (ks) { p->kw = ks->w; p->kx = ks->kx; p->ky = ks->ky; p->kz = ks->kz; }
    p->fr_lut = ctx->d_fr_lut;
    p->n_blocks = n_blocks;
    p->k = k;
    p->n_sets = n_sets;
    p->col_first = 0;
    p->col_count = 3 * n_sets + 1;
    p->sx = aesw_column_stride(layout, 0); p->sy = aesw_column_stride(layout, 1); p->sz = aesw_column_stride(layout, 2);
    p->kxs = aesw_key_column_stride(layout, 0); p->kys = aesw_key_column_stride(layout, 1); p->kzs = aesw_key_column_stride(layout, 2);
    p->packed = layout == AESW_LAYOUT_PACKED;
    p->geometry = ctx->asm_geo;
    return AESW_OK;
}
uint64_t now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
}  // namespace

int aesw_assemble_advice_device(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, uint64_t n_blocks, int layout, const uint8_t *d_x,
                                const uint8_t *d_y, const uint8_t *d_z, const aesw_key_slab *ks, int as_fr, uint8_t *d_out,
                                void *stream) {
    if (!d_out || !aligned16(d_out)) return AESW_ERR_INVALID_ARG;
    AssembleParams p;
    const int rc = fill_assemble_params(ctx, k, n_sets, n_blocks, layout, d_x, d_y, d_z, ks, &p);
    if (rc != AESW_OK) return rc;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    p.out = d_out;
    HIP_TRY(ctx, launch_assemble(p, as_fr != 0, ctx->fr_nt, reinterpret_cast<hipStream_t>(stream)));
    return AESW_OK;
}

static int check_witness_impl(aesw_ctx *ctx, const uint8_t *d_pt, const uint8_t *d_keys, int per_block_keys, uint64_t n, int layout,
                              const uint8_t *d_x, const uint8_t *d_y, const uint8_t *d_z, const uint8_t *d_ct, const aesw_key_slab *ks,
                              aesw_check_report *d_report, void *stream, bool skip_shared_key);

int aesw_check_witness_device(aesw_ctx *ctx, const uint8_t *d_pt, const uint8_t *d_keys, int per_block_keys, uint64_t n, int layout,
                              const uint8_t *d_x, const uint8_t *d_y, const uint8_t *d_z, const uint8_t *d_ct, const aesw_key_slab *ks,
                              aesw_check_report *d_report, void *stream) {
    return check_witness_impl(ctx, d_pt, d_keys, per_block_keys, n, layout, d_x, d_y, d_z, d_ct, ks, d_report, stream, false);
}

static int check_witness_impl(aesw_ctx *ctx, const uint8_t *d_pt, const uint8_t *d_keys, int per_block_keys, uint64_t n, int layout,
                              const uint8_t *d_x, const uint8_t *d_y, const uint8_t *d_z, const uint8_t *d_ct, const aesw_key_slab *ks,
                              aesw_check_report *d_report, void *stream, bool skip_shared_key) {
    static_assert(sizeof(aesw_check_report) == 7 * sizeof(uint64_t), "the kernels address the report as seven u64");
    if (!ctx || !d_report || (layout != AESW_LAYOUT_DENSE && layout != AESW_LAYOUT_PACKED)) return AESW_ERR_INVALID_ARG;
    if (per_block_keys && n && !d_keys) return AESW_ERR_INVALID_ARG;
    if (n && (!d_pt || !d_x || !d_y || !d_z || !ks || !ks->w || !ks->kx || !ks->ky || !ks->kz)) return AESW_ERR_INVALID_ARG;
    if (n && (!aligned4(d_pt) || !aligned4(d_x) || !aligned4(d_y) || !aligned4(d_z) || !aligned4(ks->w) || !aligned4(ks->kx) || !aligned4(ks->ky) ||
              !aligned4(ks->kz) || (reinterpret_cast<uintptr_t>(d_report) & 7u)))
        return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    const int li = layout == AESW_LAYOUT_DENSE ? 0 : 1;  // the check tables were uploaded by aesw_create(): nothing is allocated here
    const CheckGeo cg = check_geo(layout);
    CheckParams p{};
    p.pt = d_pt; p.keys = d_keys; p.x = d_x; p.y = d_y; p.z = d_z; p.ct = d_ct;
    if (ks) { p.kw = ks->w; p.kx = ks->kx; p.ky = ks->ky; p.kz = ks->kz; }
    p.table = ctx->d_chktab[li];
    p.tab768 = ctx->d_tables;
    p.report = reinterpret_cast<uint64_t *>(d_report);
    p.n = n;
    p.per_block_keys = per_block_keys ? 1u : 0u;
    p.skip_shared_key = skip_shared_key ? 1u : 0u;
    p.sx = cg.sx; p.sy = cg.sy; p.sz = cg.sz; p.kxs = cg.kxs; p.kys = cg.kys; p.kzs = cg.kzs; p.bi = cg.bi;
    p.img = (cg.bi + cg.ki + 15u) & ~15u;
    HIP_TRY(ctx, launch_check(p, reinterpret_cast<hipStream_t>(stream)));
    return AESW_OK;
}

int aesw_expand_fr_device(aesw_ctx *ctx, const uint8_t *d_cells, uint64_t n_cells, uint8_t *d_fr, void *stream) {
    if (!ctx) return AESW_ERR_INVALID_ARG;
    if (n_cells == 0) return AESW_OK;
    if (!d_cells || !d_fr || !aligned16(d_fr)) return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    HIP_TRY(ctx, launch_expand_fr(d_cells, n_cells, ctx->d_fr_lut, d_fr, ctx->fr_nt, ctx->fr_geo, reinterpret_cast<hipStream_t>(stream)));
    return AESW_OK;
}

// ---- host-pointer entry points ----------------------------------------------------
// Pipeline: blocks are cut into chunks; chunk i's kernel runs on s_compute
// while chunk i-1's columns travel D2H on s_copy (two device buffer sets).

namespace {

struct DevBuf {
    uint8_t *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), n ? n : 16); }
};

int ensure_streams(aesw_ctx *ctx) {
    if (!ctx->s_compute) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_compute, hipStreamNonBlocking));
    if (!ctx->s_copy) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking));
    return AESW_OK;
}

bool is_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

int ensure_bounce(aesw_ctx *ctx, size_t bytes) {
    if (ctx->bounce_bytes >= bytes) return AESW_OK;
    for (int i = 0; i < 2; ++i) {
        if (ctx->bounce[i]) (void)hipHostFree(ctx->bounce[i]);
        ctx->bounce[i] = nullptr;
    }
    ctx->bounce_bytes = 0;
    for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->bounce[i]), bytes, hipHostMallocDefault));
    ctx->bounce_bytes = bytes;
    return AESW_OK;
}

int ensure_scratch(aesw_ctx *ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return AESW_OK;
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->scratch), bytes));
    ctx->scratch_bytes = bytes;
    return AESW_OK;
}

// One output column of the host path: where chunk data goes and how.
struct HostCol {
    uint8_t *dst;       // caller buffer (null = not wanted)
    const uint8_t *dev[2];
    size_t stride;      // bytes per block
    bool direct;        // caller buffer is page-locked: DMA straight into it
    size_t boff;        // offset inside the bounce buffer
};

}  // namespace

int aesw_encrypt_witness(aesw_ctx *ctx, const uint8_t *pt, const uint8_t *keys, int per_block_keys, uint64_t n,
                         int layout, uint8_t *x, uint8_t *y, uint8_t *z, uint8_t *ct, const aesw_key_slab *ks) {
    if (!ctx || !valid_layout(layout)) return AESW_ERR_INVALID_ARG;
    if (n == 0) return AESW_OK;
    if (!pt) return AESW_ERR_INVALID_ARG;  // x / y / z: a null column is computed but not copied back
    if (!keys && (per_block_keys || (ks && (ks->w || ks->kx || ks->ky || ks->kz)))) return AESW_ERR_INVALID_ARG;
    if (!keys && !ctx->have_key) return AESW_ERR_NO_KEY;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    int rc = ensure_streams(ctx);
    if (rc != AESW_OK) return rc;
    const size_t sx = aesw_column_stride(layout, 0), sy = aesw_column_stride(layout, 1), sz = aesw_column_stride(layout, 2);
    const size_t kxs = aesw_key_column_stride(layout, 0), kys = aesw_key_column_stride(layout, 1),
                 kzs = aesw_key_column_stride(layout, 2);
    const bool kemit = ks && (ks->w || ks->kx || ks->ky || ks->kz);
    const bool pbk = per_block_keys != 0;
    uint64_t chunk = (uint64_t)ctx->chunk_blocks;
    if (chunk > n) chunk = n;
    chunk = (chunk + 63) / 64 * 64;

    // carve the context's device scratch: inputs, ciphertext, two sets of output columns
    struct Carve { uint8_t *p = nullptr; };
    Carve d_pt, d_keys, d_ct, dx[2], dy[2], dz[2], dw[2], dkx[2], dky[2], dkz[2];
    {
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        const size_t o_pt = take(n * 16), o_keys = take(pbk ? n * 16 : 16), o_ct = take(ct ? n * 16 : 0);
        size_t o_x[2], o_y[2], o_z[2], o_w[2], o_kx[2], o_ky[2], o_kz[2];
        for (int i = 0; i < 2; ++i) {
            o_x[i] = take(chunk * sx); o_y[i] = take(chunk * sy); o_z[i] = take(chunk * sz);
            const size_t kn = kemit ? (pbk ? chunk : 1) : 0;  // per-block keys: a key slab per block; shared key: one
            o_w[i] = take(kn * WORDS_ROWS); o_kx[i] = take(kn * kxs);
            o_ky[i] = take(kn * kys); o_kz[i] = take(kn * kzs);
        }
        rc = ensure_scratch(ctx, off ? off : 256);
        if (rc != AESW_OK) return rc;
        uint8_t *b = ctx->scratch;
        d_pt.p = b + o_pt; d_keys.p = b + o_keys; d_ct.p = b + o_ct;
        for (int i = 0; i < 2; ++i) {
            dx[i].p = b + o_x[i]; dy[i].p = b + o_y[i]; dz[i].p = b + o_z[i];
            dw[i].p = b + o_w[i]; dkx[i].p = b + o_kx[i]; dky[i].p = b + o_ky[i]; dkz[i].p = b + o_kz[i];
        }
    }
    HostCol cols[7] = {
        {sx ? x : nullptr, {dx[0].p, dx[1].p}, sx, false, 0}, {y, {dy[0].p, dy[1].p}, sy, false, 0}, {z, {dz[0].p, dz[1].p}, sz, false, 0},
        {pbk && kemit ? ks->w : nullptr, {dw[0].p, dw[1].p}, WORDS_ROWS, false, 0},
        {pbk && kemit ? ks->kx : nullptr, {dkx[0].p, dkx[1].p}, kxs, false, 0},
        {pbk && kemit ? ks->ky : nullptr, {dky[0].p, dky[1].p}, kys, false, 0},
        {pbk && kemit ? ks->kz : nullptr, {dkz[0].p, dkz[1].p}, kzs, false, 0}};
    size_t bounce_need = 0;
    for (HostCol &c : cols) {
        if (!c.dst) continue;
        c.direct = is_pinned(c.dst);
        if (!c.direct) {
            c.boff = bounce_need;
            bounce_need += (chunk * c.stride + 255) / 256 * 256;
        }
    }
    if (bounce_need) {
        rc = ensure_bounce(ctx, bounce_need);
        if (rc != AESW_OK) return rc;
    }
    // Whatever happens below, no copy may still be reading or writing the caller's buffers when we return.
    struct SyncGuard {
        aesw_ctx *c;
        ~SyncGuard() {
            (void)hipStreamSynchronize(c->s_copy);
            (void)hipStreamSynchronize(c->s_compute);
        }
    } sync_guard{ctx};
    hipEvent_t done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    struct EvGuard {
        hipEvent_t *a, *b;
        ~EvGuard() { for (int i = 0; i < 2; ++i) { if (a[i]) (void)hipEventDestroy(a[i]); if (b[i]) (void)hipEventDestroy(b[i]); } }
    } evg{done, copied};
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
    }
    HIP_TRY(ctx, hipMemcpyAsync(d_pt.p, pt, n * 16, hipMemcpyHostToDevice, ctx->s_compute));
    if (keys) HIP_TRY(ctx, hipMemcpyAsync(d_keys.p, keys, pbk ? n * 16 : 16, hipMemcpyHostToDevice, ctx->s_compute));

    if (!pbk && kemit) {
        // shared key: one key slab, staged in stage 0's (still unused) key buffers of the scratch, copied back on s_compute
        rc = aesw_key_schedule_witness_device(ctx, d_keys.p, 1, layout, dw[0].p, dkx[0].p, dky[0].p, dkz[0].p, nullptr, ctx->s_compute);
        if (rc != AESW_OK) return rc;
        if (ks->w) HIP_TRY(ctx, hipMemcpyAsync(ks->w, dw[0].p, WORDS_ROWS, hipMemcpyDeviceToHost, ctx->s_compute));
        if (ks->kx) HIP_TRY(ctx, hipMemcpyAsync(ks->kx, dkx[0].p, kxs, hipMemcpyDeviceToHost, ctx->s_compute));
        if (ks->ky) HIP_TRY(ctx, hipMemcpyAsync(ks->ky, dky[0].p, kys, hipMemcpyDeviceToHost, ctx->s_compute));
        if (ks->kz) HIP_TRY(ctx, hipMemcpyAsync(ks->kz, dkz[0].p, kzs, hipMemcpyDeviceToHost, ctx->s_compute));
    }

    // Drain stage s: wait for its D2H, then move bounce data into pageable destinations.
    uint64_t stage_b0[2] = {0, 0}, stage_m[2] = {0, 0};
    bool stage_busy[2] = {false, false};
    auto drain = [&](int s) -> int {
        if (!stage_busy[s]) return AESW_OK;
        HIP_TRY(ctx, hipEventSynchronize(copied[s]));
        std::vector<CopyJob> jobs;
        for (const HostCol &c : cols)
            if (c.dst && !c.direct) jobs.push_back(CopyJob{c.dst + stage_b0[s] * c.stride, ctx->bounce[s] + c.boff, (size_t)(stage_m[s] * c.stride)});
        if (!jobs.empty()) parallel_copy(jobs, auto_copy_threads(ctx));
        stage_busy[s] = false;
        return AESW_OK;
    };
    uint64_t b0 = 0;
    int it = 0;
    while (b0 < n) {
        const uint64_t m = n - b0 < chunk ? n - b0 : chunk;
        const int s = it & 1;
        rc = drain(s);  // device buffers and bounce buffer of this stage are free again
        if (rc != AESW_OK) return rc;
        aesw_key_slab dks{dw[s].p, dkx[s].p, dky[s].p, dkz[s].p};
        rc = aesw_encrypt_witness_device(ctx, d_pt.p + 16 * b0, !keys ? nullptr : (pbk ? d_keys.p + 16 * b0 : d_keys.p), per_block_keys, m,
                                         layout, dx[s].p, dy[s].p, dz[s].p, ct ? d_ct.p + 16 * b0 : nullptr,
                                         pbk && kemit ? &dks : nullptr, ctx->s_compute);
        if (rc != AESW_OK) return rc;
        HIP_TRY(ctx, hipEventRecord(done[s], ctx->s_compute));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_copy, done[s], 0));
        for (const HostCol &c : cols) {
            if (!c.dst) continue;
            uint8_t *to = c.direct ? c.dst + b0 * c.stride : ctx->bounce[s] + c.boff;
            HIP_TRY(ctx, hipMemcpyAsync(to, c.dev[s], m * c.stride, hipMemcpyDeviceToHost, ctx->s_copy));
        }
        HIP_TRY(ctx, hipEventRecord(copied[s], ctx->s_copy));
        stage_b0[s] = b0;
        stage_m[s] = m;
        stage_busy[s] = true;
        b0 += m;
        ++it;
    }
    rc = drain(it & 1);
    if (rc != AESW_OK) return rc;
    rc = drain((it + 1) & 1);
    if (rc != AESW_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_copy));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_compute));
    if (ct) HIP_TRY(ctx, hipMemcpy(ct, d_ct.p, n * 16, hipMemcpyDeviceToHost));
    return AESW_OK;
}

int aesw_encrypt_witness_stream(aesw_ctx *ctx, const uint8_t *pt, const uint8_t *keys, int per_block_keys, uint64_t n, int layout,
                                aesw_chunk_fn consume, void *user) {
    if (!ctx || !valid_layout(layout) || !consume) return AESW_ERR_INVALID_ARG;
    if (n == 0) return AESW_OK;
    if (!pt || (!keys && per_block_keys)) return AESW_ERR_INVALID_ARG;
    if (!keys && !ctx->have_key) return AESW_ERR_NO_KEY;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    int rc = ensure_streams(ctx);
    if (rc != AESW_OK) return rc;
    const size_t strides[3] = {aesw_column_stride(layout, 0), aesw_column_stride(layout, 1), aesw_column_stride(layout, 2)};
    const bool pbk = per_block_keys != 0;
    uint64_t chunk = (uint64_t)ctx->chunk_blocks;
    if (chunk > n) chunk = n;
    chunk = (chunk + 63) / 64 * 64;
    // device scratch: inputs + two sets of columns; page-locked bounce: two sets of columns
    size_t off = 0, col_off[2][3], boff[3], bneed = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_pt = take(n * 16), o_keys = take(pbk ? n * 16 : 16);
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < 3; ++c) col_off[s][c] = take(chunk * strides[c]);
    for (int c = 0; c < 3; ++c) { boff[c] = bneed; bneed += (chunk * strides[c] + 255) / 256 * 256; }
    // "stream_check": every chunk is checked on the device behind its kernel (aesw_check.h).  Needs the key slab(s) the blocks' AddRoundKey
    // rows copy from -- one for a shared / scheduled key (made once, below), one per block with per-block keys (emitted by the chunk's
    // own launch into two more scratch sets) -- and one report per chunk, summed after the last one.
    const bool checking = ctx->stream_check && layout != AESW_LAYOUT_VALUES;
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    const size_t kstr[3] = {aesw_key_column_stride(layout, 0), aesw_key_column_stride(layout, 1), aesw_key_column_stride(layout, 2)};
    size_t ks_off[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, o_rep = 0;
    if (checking) {
        const uint64_t nk = pbk ? chunk : 1;
        for (int s = 0; s < (pbk ? 2 : 1); ++s) {
            ks_off[s][0] = take(nk * WORDS_ROWS);
            for (int c = 0; c < 3; ++c) ks_off[s][1 + c] = take(nk * kstr[c]);
        }
        if (!pbk) for (int c = 0; c < 4; ++c) ks_off[1][c] = ks_off[0][c];
        o_rep = take(n_chunks * sizeof(aesw_check_report));
    }
    ctx->stream_report = aesw_check_report{0, 0, 0, 0, 0, 0, AESW_CHECK_NONE};
    rc = ensure_scratch(ctx, off);
    if (rc != AESW_OK) return rc;
    rc = ensure_bounce(ctx, bneed);
    if (rc != AESW_OK) return rc;
    uint8_t *d = ctx->scratch;
    struct SyncGuard {
        aesw_ctx *c;
        ~SyncGuard() { (void)hipStreamSynchronize(c->s_copy); (void)hipStreamSynchronize(c->s_compute); }
    } sync_guard{ctx};
    HIP_TRY(ctx, hipMemcpyAsync(d + o_pt, pt, n * 16, hipMemcpyHostToDevice, ctx->s_compute));
    if (keys) HIP_TRY(ctx, hipMemcpyAsync(d + o_keys, keys, pbk ? n * 16 : 16, hipMemcpyHostToDevice, ctx->s_compute));
    auto slab_of = [&](int s) { return aesw_key_slab{d + ks_off[s][0], d + ks_off[s][1], d + ks_off[s][2], d + ks_off[s][3]}; };
    const uint8_t *d_key16 = nullptr;  // the 16 key bytes of a shared / scheduled key on the device (the literal rows of words_column)
    if (checking && !pbk) {
        // a scheduled key's bytes are the first round key of its slot (rk[0] = the key, src/key_schedule.rs:107-114)
        d_key16 = keys ? d + o_keys : ctx->key_slots[ctx->key_cur].d;
        if (!keys) {
            aesw_ctx::KeySlot &sl = ctx->key_slots[ctx->key_cur];
            if (!sl.pinned && sl.writer != ctx->s_compute) HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_compute, sl.ready, 0));
        }
        const aesw_key_slab one = slab_of(0);
        KeyParams kp{d_key16, ctx->d_tables, KeyOut{one.w, one.kx, one.ky, one.kz}, nullptr, 1, 0, 0};
        HIP_TRY(ctx, launch_key(kp, layout, ctx->xt, 1, ctx->key_nt, 0u, ctx->s_compute));
        if (!keys) { const int r = key_track_reader(ctx, ctx->key_slots[ctx->key_cur], ctx->s_compute); if (r != AESW_OK) return r; }
    }
    uint64_t chunk_index = 0;
    // per stage: kernel start / end, copy start / end (timed: aesw_last_stream_stats reports where the time went)
    hipEvent_t started[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, copy0[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    struct EvGuard {
        hipEvent_t *e[4];
        ~EvGuard() { for (auto *v : e) for (int i = 0; i < 2; ++i) if (v[i]) (void)hipEventDestroy(v[i]); }
    } evg{{started, done, copy0, copied}};
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, hipEventCreate(&started[i]));
        HIP_TRY(ctx, hipEventCreate(&done[i]));
        HIP_TRY(ctx, hipEventCreate(&copy0[i]));
        HIP_TRY(ctx, hipEventCreate(&copied[i]));
    }
    aesw_stream_stats st = {};
    const uint64_t t_begin = now_ns();
    uint64_t first[2] = {0, 0}, count[2] = {0, 0};
    bool busy[2] = {false, false};
    auto issue = [&](int s, uint64_t b0, uint64_t m) -> int {
        HIP_TRY(ctx, hipEventRecord(started[s], ctx->s_compute));
        const aesw_key_slab stage_slab = slab_of(s);
        int r = aesw_encrypt_witness_device(ctx, d + o_pt + 16 * b0, !keys ? nullptr : (pbk ? d + o_keys + 16 * b0 : d + o_keys), per_block_keys, m,
                                            layout, d + col_off[s][0], d + col_off[s][1], d + col_off[s][2], nullptr,
                                            checking && pbk ? &stage_slab : nullptr, ctx->s_compute);
        if (r != AESW_OK) return r;
        if (ctx->stream_poison > 0 && (uint64_t)ctx->stream_poison - 1 >= b0 && (uint64_t)ctx->stream_poison - 1 < b0 + m) {
            // diagnostic: two cells of one block are overwritten between the kernel and the check (tests/test_gpu_round4.py shows the
            // stream check names that block, by its batch-wide index, in whatever chunk it lies)
            const uint64_t pb = (uint64_t)ctx->stream_poison - 1 - b0;
            HIP_TRY(ctx, hipMemsetAsync(d + col_off[s][1] + pb * strides[1] + 5, 0x5A, 1, ctx->s_compute));
            HIP_TRY(ctx, hipMemsetAsync(d + col_off[s][2] + pb * strides[2] + 7, 0xA5, 1, ctx->s_compute));
        }
        if (checking) {
            r = check_witness_impl(ctx, d + o_pt + 16 * b0, pbk ? d + o_keys + 16 * b0 : d_key16, per_block_keys, m, layout, d + col_off[s][0],
                                   d + col_off[s][1], d + col_off[s][2], nullptr, &stage_slab,
                                   reinterpret_cast<aesw_check_report *>(d + o_rep) + chunk_index, ctx->s_compute, !pbk && chunk_index != 0);
            if (r != AESW_OK) return r;
            if (!keys) { r = key_track_reader(ctx, ctx->key_slots[ctx->key_cur], ctx->s_compute); if (r != AESW_OK) return r; }
            ++chunk_index;
        }
        HIP_TRY(ctx, hipEventRecord(done[s], ctx->s_compute));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_copy, done[s], 0));
        HIP_TRY(ctx, hipEventRecord(copy0[s], ctx->s_copy));
        for (int c = 0; c < 3; ++c)
            if (strides[c])
                HIP_TRY(ctx, hipMemcpyAsync(ctx->bounce[s] + boff[c], d + col_off[s][c], m * strides[c], hipMemcpyDeviceToHost, ctx->s_copy));
        HIP_TRY(ctx, hipEventRecord(copied[s], ctx->s_copy));
        first[s] = b0; count[s] = m; busy[s] = true;
        return AESW_OK;
    };
    // two stages in flight: while the host consumes stage s, stage s^1 is computed and copied
    uint64_t b0 = 0;
    int it = 0;
    for (; it < 2 && b0 < n; ++it) {
        const uint64_t m = n - b0 < chunk ? n - b0 : chunk;
        rc = issue(it, b0, m);
        if (rc != AESW_OK) return rc;
        b0 += m;
    }
    for (int s = 0;; s ^= 1) {
        if (!busy[s]) break;
        const uint64_t t0 = now_ns();
        HIP_TRY(ctx, hipEventSynchronize(copied[s]));
        const uint64_t t1 = now_ns();
        busy[s] = false;
        float ms = 0;
        if (hipEventElapsedTime(&ms, started[s], done[s]) == hipSuccess) st.kernel_ns += (uint64_t)(ms * 1e6);
        if (hipEventElapsedTime(&ms, copy0[s], copied[s]) == hipSuccess) st.d2h_ns += (uint64_t)(ms * 1e6);
        const int cr = consume(user, first[s], count[s], strides[0] ? ctx->bounce[s] + boff[0] : nullptr /* AESW_LAYOUT_VALUES: no x */,
                               ctx->bounce[s] + boff[1], ctx->bounce[s] + boff[2]);
        const uint64_t t2 = now_ns();
        st.wait_ns += t1 - t0;
        st.consumer_ns += t2 - t1;
        st.chunks += 1;
        st.bytes_to_host += count[s] * (strides[0] + strides[1] + strides[2]);
        if (cr != 0) { st.wall_ns = now_ns() - t_begin; ctx->stats = st; return AESW_ERR_MISMATCH; }
        if (b0 < n) {
            const uint64_t m = n - b0 < chunk ? n - b0 : chunk;
            rc = issue(s, b0, m);
            if (rc != AESW_OK) return rc;
            b0 += m;
        }
    }
    st.wall_ns = now_ns() - t_begin;
    ctx->stats = st;
    if (checking) {  // every chunk's kernel and check have finished (their columns have been copied): sum the reports
        std::vector<aesw_check_report> reps((size_t)n_chunks);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->s_compute));  // (a non-blocking stream: the copy below does not wait for it by itself)
        HIP_TRY(ctx, hipMemcpy(reps.data(), d + o_rep, reps.size() * sizeof(aesw_check_report), hipMemcpyDeviceToHost));
        aesw_check_report &t = ctx->stream_report;
        for (uint64_t i = 0; i < n_chunks; ++i) {
            const aesw_check_report &r = reps[i];
            t.blocks += r.blocks; t.keys += r.keys;
            t.lookup_failures += r.lookup_failures; t.copy_failures += r.copy_failures;
            t.gate_failures += r.gate_failures; t.input_failures += r.input_failures;
            if (r.first != AESW_CHECK_NONE) {
                const uint64_t unit = (r.first >> 20) + ((!pbk && ((r.first >> 19) & 1)) ? 0 : i * chunk);
                const uint64_t f = unit << 20 | (r.first & 0xfffffu);
                if (f < t.first) t.first = f;
            }
        }
    }
    return AESW_OK;
}

int aesw_last_stream_check(const aesw_ctx *ctx, aesw_check_report *out) {
    if (!ctx || !out) return AESW_ERR_INVALID_ARG;
    *out = ctx->stream_report;
    return AESW_OK;
}

int aesw_last_stream_stats(const aesw_ctx *ctx, aesw_stream_stats *out) {
    if (!ctx || !out) return AESW_ERR_INVALID_ARG;
    *out = ctx->stats;
    return AESW_OK;
}

// Whole advice columns of a K/N circuit to the host, column by column ("the host can bulk-copy into
// halo2's advice polynomials"): column j is assembled on s_compute into one of two device buffers, travels D2H on
// s_copy into one of two page-locked buffers, and is handed to `consume` while column j+1 is assembled and copied.
int aesw_assemble_advice_stream(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, uint64_t n_blocks, int layout, const uint8_t *d_x,
                                const uint8_t *d_y, const uint8_t *d_z, const aesw_key_slab *ks, int as_fr, aesw_column_fn consume,
                                void *user) {
    if (!consume) return AESW_ERR_INVALID_ARG;
    AssembleParams p;
    int rc = fill_assemble_params(ctx, k, n_sets, n_blocks, layout, d_x, d_y, d_z, ks, &p);
    if (rc != AESW_OK) return rc;
    if (k > 28) return AESW_ERR_INVALID_ARG;  // one column must fit the staging buffers
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    rc = ensure_streams(ctx);
    if (rc != AESW_OK) return rc;
    const uint64_t rows = (uint64_t)1 << k;
    const size_t col_bytes = (size_t)rows * (as_fr ? AESW_FR_BYTES : 1);
    const size_t slot = (col_bytes + 255) / 256 * 256;
    rc = ensure_scratch(ctx, 2 * slot);
    if (rc != AESW_OK) return rc;
    rc = ensure_bounce(ctx, slot);
    if (rc != AESW_OK) return rc;
    struct SyncGuard {
        aesw_ctx *c;
        ~SyncGuard() { (void)hipStreamSynchronize(c->s_copy); (void)hipStreamSynchronize(c->s_compute); }
    } sync_guard{ctx};
    hipEvent_t ev[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};  // per stage: kernel start/end, copy start/end
    struct EvGuard {
        hipEvent_t (*e)[4];
        ~EvGuard() { for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) if (e[i][j]) (void)hipEventDestroy(e[i][j]); }
    } evg{ev};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) HIP_TRY(ctx, hipEventCreate(&ev[i][j]));
    // the caller's slabs were produced on some stream of theirs: they must be complete before this call (documented)
    const uint32_t ncols = 3 * n_sets + 1;
    aesw_stream_stats st = {};
    const uint64_t t_begin = now_ns();
    auto issue = [&](int s, uint32_t col) -> int {
        AssembleParams q = p;
        q.col_first = col;
        q.col_count = 1;
        q.out = ctx->scratch + (size_t)s * slot;
        HIP_TRY(ctx, hipEventRecord(ev[s][0], ctx->s_compute));
        HIP_TRY(ctx, launch_assemble(q, as_fr != 0, ctx->fr_nt, ctx->s_compute));
        HIP_TRY(ctx, hipEventRecord(ev[s][1], ctx->s_compute));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_copy, ev[s][1], 0));
        HIP_TRY(ctx, hipEventRecord(ev[s][2], ctx->s_copy));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->bounce[s], q.out, col_bytes, hipMemcpyDeviceToHost, ctx->s_copy));
        HIP_TRY(ctx, hipEventRecord(ev[s][3], ctx->s_copy));
        return AESW_OK;
    };
    uint32_t next = 0;
    for (; next < 2 && next < ncols; ++next) {
        rc = issue((int)next, next);
        if (rc != AESW_OK) return rc;
    }
    for (uint32_t col = 0; col < ncols; ++col) {
        const int s = (int)(col & 1);
        const uint64_t t0 = now_ns();
        HIP_TRY(ctx, hipEventSynchronize(ev[s][3]));
        const uint64_t t1 = now_ns();
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[s][0], ev[s][1]) == hipSuccess) st.kernel_ns += (uint64_t)(ms * 1e6);
        if (hipEventElapsedTime(&ms, ev[s][2], ev[s][3]) == hipSuccess) st.d2h_ns += (uint64_t)(ms * 1e6);
        const int r = consume(user, col, ctx->bounce[s], rows);
        const uint64_t t2 = now_ns();
        st.wait_ns += t1 - t0;
        st.consumer_ns += t2 - t1;
        st.chunks += 1;
        st.bytes_to_host += col_bytes;
        if (r != 0) { st.wall_ns = now_ns() - t_begin; ctx->stats = st; return AESW_ERR_MISMATCH; }
        if (next < ncols) {
            rc = issue(s, next++);
            if (rc != AESW_OK) return rc;
        }
    }
    st.wall_ns = now_ns() - t_begin;
    ctx->stats = st;
    return AESW_OK;
}

// The same columns straight into ONE host buffer (column after column): DMA directly when the buffer is page-locked
// (aesw_host_alloc, or the host's own advice-polynomial memory after aesw_host_register), through the bounce buffers
// otherwise.  Column j+1 is assembled while column j travels.
int aesw_assemble_advice_host(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, uint64_t n_blocks, int layout, const uint8_t *d_x,
                              const uint8_t *d_y, const uint8_t *d_z, const aesw_key_slab *ks, int as_fr, uint8_t *out) {
    if (!out) return AESW_ERR_INVALID_ARG;
    AssembleParams p;
    int rc = fill_assemble_params(ctx, k, n_sets, n_blocks, layout, d_x, d_y, d_z, ks, &p);
    if (rc != AESW_OK) return rc;
    if (k > 28) return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    rc = ensure_streams(ctx);
    if (rc != AESW_OK) return rc;
    const uint64_t rows = (uint64_t)1 << k;
    const size_t col_bytes = (size_t)rows * (as_fr ? AESW_FR_BYTES : 1);
    const size_t slot = (col_bytes + 255) / 256 * 256;
    const uint32_t ncols = 3 * n_sets + 1;
    const bool direct = is_pinned(out) && is_pinned(out + (size_t)ncols * col_bytes - 1);
    rc = ensure_scratch(ctx, 2 * slot);
    if (rc != AESW_OK) return rc;
    if (!direct) {
        rc = ensure_bounce(ctx, slot);
        if (rc != AESW_OK) return rc;
    }
    struct SyncGuard {
        aesw_ctx *c;
        ~SyncGuard() { (void)hipStreamSynchronize(c->s_copy); (void)hipStreamSynchronize(c->s_compute); }
    } sync_guard{ctx};
    hipEvent_t done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    struct EvGuard {
        hipEvent_t *a, *b;
        ~EvGuard() { for (int i = 0; i < 2; ++i) { if (a[i]) (void)hipEventDestroy(a[i]); if (b[i]) (void)hipEventDestroy(b[i]); } }
    } evg{done, copied};
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
    }
    bool busy[2] = {false, false};
    uint32_t held[2] = {0, 0};
    auto drain = [&](int s) -> int {  // the device slot (and bounce buffer) of stage s is free again after this
        if (!busy[s]) return AESW_OK;
        HIP_TRY(ctx, hipEventSynchronize(copied[s]));
        if (!direct) parallel_copy({CopyJob{out + (size_t)held[s] * col_bytes, ctx->bounce[s], (size_t)col_bytes}}, auto_copy_threads(ctx));
        busy[s] = false;
        return AESW_OK;
    };
    for (uint32_t col = 0; col < ncols; ++col) {
        const int s = (int)(col & 1);
        rc = drain(s);
        if (rc != AESW_OK) return rc;
        AssembleParams q = p;
        q.col_first = col;
        q.col_count = 1;
        q.out = ctx->scratch + (size_t)s * slot;
        HIP_TRY(ctx, launch_assemble(q, as_fr != 0, ctx->fr_nt, ctx->s_compute));
        HIP_TRY(ctx, hipEventRecord(done[s], ctx->s_compute));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_copy, done[s], 0));
        HIP_TRY(ctx, hipMemcpyAsync(direct ? out + (size_t)col * col_bytes : ctx->bounce[s], q.out, col_bytes, hipMemcpyDeviceToHost, ctx->s_copy));
        HIP_TRY(ctx, hipEventRecord(copied[s], ctx->s_copy));
        // no stream wait on copied[s] here: the next launch goes into the OTHER slot and may run while this column travels;
        // this slot is written again only after drain(s) has host-synchronised copied[s]
        busy[s] = true;
        held[s] = col;
    }
    rc = drain(0);
    if (rc != AESW_OK) return rc;
    return drain(1);
}

int aesw_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return AESW_ERR_INVALID_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return AESW_ERR_HIP;
    }
    return AESW_OK;
}

int aesw_host_unregister(void *p) {
    if (!p) return AESW_ERR_INVALID_ARG;
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return AESW_ERR_HIP;
    }
    return AESW_OK;
}

int aesw_key_schedule_witness(aesw_ctx *ctx, const uint8_t *keys, uint64_t n, int layout, uint8_t *w, uint8_t *kx,
                              uint8_t *ky, uint8_t *kz, uint8_t *rk) {
    if (!ctx || !valid_layout(layout)) return AESW_ERR_INVALID_ARG;
    if (n == 0) return AESW_OK;
    if (!keys) return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    const size_t kxs = aesw_key_column_stride(layout, 0), kys = aesw_key_column_stride(layout, 1),
                 kzs = aesw_key_column_stride(layout, 2);
    DevBuf dk, dw, dkx, dky, dkz, drk;
    HIP_TRY(ctx, dk.alloc(n * 16));
    if (w) HIP_TRY(ctx, dw.alloc(n * WORDS_ROWS));
    if (kx) HIP_TRY(ctx, dkx.alloc(n * kxs));
    if (ky) HIP_TRY(ctx, dky.alloc(n * kys));
    if (kz) HIP_TRY(ctx, dkz.alloc(n * kzs));
    if (rk) HIP_TRY(ctx, drk.alloc(n * RK_BYTES));
    HIP_TRY(ctx, hipMemcpy(dk.p, keys, n * 16, hipMemcpyHostToDevice));
    int rc = aesw_key_schedule_witness_device(ctx, dk.p, n, layout, w ? dw.p : nullptr, kx ? dkx.p : nullptr,
                                              ky ? dky.p : nullptr, kz ? dkz.p : nullptr, rk ? drk.p : nullptr, nullptr);
    if (rc != AESW_OK) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (w) HIP_TRY(ctx, hipMemcpy(w, dw.p, n * WORDS_ROWS, hipMemcpyDeviceToHost));
    if (kx) HIP_TRY(ctx, hipMemcpy(kx, dkx.p, n * kxs, hipMemcpyDeviceToHost));
    if (ky) HIP_TRY(ctx, hipMemcpy(ky, dky.p, n * kys, hipMemcpyDeviceToHost));
    if (kz) HIP_TRY(ctx, hipMemcpy(kz, dkz.p, n * kzs, hipMemcpyDeviceToHost));
    if (rk) HIP_TRY(ctx, hipMemcpy(rk, drk.p, n * RK_BYTES, hipMemcpyDeviceToHost));
    return AESW_OK;
}

int aesw_check_witness(aesw_ctx *ctx, const uint8_t *pt, const uint8_t *keys, int per_block_keys, uint64_t n, int layout, const uint8_t *x,
                       const uint8_t *y, const uint8_t *z, const uint8_t *ct, const aesw_key_slab *ks, aesw_check_report *report) {
    if (!ctx || !report || (layout != AESW_LAYOUT_DENSE && layout != AESW_LAYOUT_PACKED)) return AESW_ERR_INVALID_ARG;
    if (per_block_keys && n && !keys) return AESW_ERR_INVALID_ARG;
    if (n && (!pt || !x || !y || !z || !ks || !ks->w || !ks->kx || !ks->ky || !ks->kz)) return AESW_ERR_INVALID_ARG;
    *report = aesw_check_report{0, 0, 0, 0, 0, 0, AESW_CHECK_NONE};
    if (n == 0) return AESW_OK;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    const CheckGeo cg = check_geo(layout);
    const uint64_t chunk = (uint64_t)ctx->chunk_blocks < n ? (uint64_t)ctx->chunk_blocks : n;
    const uint64_t nk = per_block_keys ? chunk : 1;
    DevBuf dpt, dkeys, dx, dy, dz, dct, dw, dkx, dky, dkz, drep;
    HIP_TRY(ctx, dpt.alloc(chunk * 16));
    HIP_TRY(ctx, dkeys.alloc(nk * 16));
    HIP_TRY(ctx, dx.alloc(chunk * cg.sx)); HIP_TRY(ctx, dy.alloc(chunk * cg.sy)); HIP_TRY(ctx, dz.alloc(chunk * cg.sz));
    HIP_TRY(ctx, dct.alloc(chunk * 16));
    HIP_TRY(ctx, dw.alloc(nk * WORDS_ROWS)); HIP_TRY(ctx, dkx.alloc(nk * cg.kxs)); HIP_TRY(ctx, dky.alloc(nk * cg.kys)); HIP_TRY(ctx, dkz.alloc(nk * cg.kzs));
    HIP_TRY(ctx, drep.alloc(sizeof(aesw_check_report)));
    const aesw_key_slab dks{dw.p, dkx.p, dky.p, dkz.p};
    if (!per_block_keys) {  // the one key slab of the batch travels once
        if (keys) HIP_TRY(ctx, hipMemcpy(dkeys.p, keys, 16, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dw.p, ks->w, WORDS_ROWS, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dkx.p, ks->kx, cg.kxs, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dky.p, ks->ky, cg.kys, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dkz.p, ks->kz, cg.kzs, hipMemcpyHostToDevice));
    }
    for (uint64_t lo = 0; lo < n; lo += chunk) {
        const uint64_t m = n - lo < chunk ? n - lo : chunk;
        HIP_TRY(ctx, hipMemcpy(dpt.p, pt + lo * 16, m * 16, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dx.p, x + lo * cg.sx, m * cg.sx, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dy.p, y + lo * cg.sy, m * cg.sy, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dz.p, z + lo * cg.sz, m * cg.sz, hipMemcpyHostToDevice));
        if (ct) HIP_TRY(ctx, hipMemcpy(dct.p, ct + lo * 16, m * 16, hipMemcpyHostToDevice));
        if (per_block_keys) {
            HIP_TRY(ctx, hipMemcpy(dkeys.p, keys + lo * 16, m * 16, hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(dw.p, ks->w + lo * WORDS_ROWS, m * WORDS_ROWS, hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(dkx.p, ks->kx + lo * cg.kxs, m * cg.kxs, hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(dky.p, ks->ky + lo * cg.kys, m * cg.kys, hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(dkz.p, ks->kz + lo * cg.kzs, m * cg.kzs, hipMemcpyHostToDevice));
        }
        const int rc = check_witness_impl(ctx, dpt.p, (per_block_keys || keys) ? dkeys.p : nullptr, per_block_keys, m, layout, dx.p, dy.p, dz.p,
                                          ct ? dct.p : nullptr, &dks, reinterpret_cast<aesw_check_report *>(drep.p), nullptr,
                                          /* skip the shared key slab */ !per_block_keys && lo != 0);
        if (rc != AESW_OK) return rc;
        aesw_check_report r;
        HIP_TRY(ctx, hipMemcpy(&r, drep.p, sizeof r, hipMemcpyDeviceToHost));  // (synchronises with the null stream's launch)
        report->blocks += r.blocks; report->keys += r.keys;
        report->lookup_failures += r.lookup_failures; report->copy_failures += r.copy_failures;
        report->gate_failures += r.gate_failures; report->input_failures += r.input_failures;
        if (r.first != AESW_CHECK_NONE) {
            // units of a stage count from its first block (the shared key slab is unit 0 of the batch as well)
            const uint64_t unit = (r.first >> 20) + ((!per_block_keys && ((r.first >> 19) & 1)) ? 0 : lo);
            const uint64_t f = unit << 20 | (r.first & 0xfffffu);
            if (f < report->first) report->first = f;
        }
    }
    return AESW_OK;
}

int aesw_schedule_key(aesw_ctx *ctx, const uint8_t key[16], int layout, const aesw_key_slab *ks) {
    if (!ctx || !valid_layout(layout) || !key) return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    const size_t kxs = aesw_key_column_stride(layout, 0), kys = aesw_key_column_stride(layout, 1),
                 kzs = aesw_key_column_stride(layout, 2);
    DevBuf dk, dw, dkx, dky, dkz;
    HIP_TRY(ctx, dk.alloc(16));
    HIP_TRY(ctx, dw.alloc(WORDS_ROWS)); HIP_TRY(ctx, dkx.alloc(kxs)); HIP_TRY(ctx, dky.alloc(kys)); HIP_TRY(ctx, dkz.alloc(kzs));
    HIP_TRY(ctx, hipMemcpy(dk.p, key, 16, hipMemcpyHostToDevice));
    aesw_key_slab dks{dw.p, dkx.p, dky.p, dkz.p};
    int rc = aesw_schedule_key_device(ctx, dk.p, layout, ks ? &dks : nullptr, nullptr);
    if (rc != AESW_OK) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (ks) {
        if (ks->w) HIP_TRY(ctx, hipMemcpy(ks->w, dw.p, WORDS_ROWS, hipMemcpyDeviceToHost));
        if (ks->kx) HIP_TRY(ctx, hipMemcpy(ks->kx, dkx.p, kxs, hipMemcpyDeviceToHost));
        if (ks->ky) HIP_TRY(ctx, hipMemcpy(ks->ky, dky.p, kys, hipMemcpyDeviceToHost));
        if (ks->kz) HIP_TRY(ctx, hipMemcpy(ks->kz, dkz.p, kzs, hipMemcpyDeviceToHost));
    }
    return AESW_OK;
}

void *aesw_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void aesw_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int aesw_lookup_table(aesw_ctx *ctx, uint8_t *t0, uint8_t *t1, uint8_t *t2, uint8_t *t3) {
    if (!ctx || !t0 || !t1 || !t2 || !t3) return AESW_ERR_INVALID_ARG;
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    DevBuf d;
    HIP_TRY(ctx, d.alloc(4 * (size_t)AESW_TABLE_ROWS));
    uint8_t *p = d.p;
    int rc = aesw_lookup_table_device(ctx, p, p + AESW_TABLE_ROWS, p + 2 * (size_t)AESW_TABLE_ROWS,
                                      p + 3 * (size_t)AESW_TABLE_ROWS, nullptr);
    if (rc != AESW_OK) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    uint8_t *outs[4] = {t0, t1, t2, t3};
    for (int i = 0; i < 4; ++i)
        HIP_TRY(ctx, hipMemcpy(outs[i], p + i * (size_t)AESW_TABLE_ROWS, AESW_TABLE_ROWS, hipMemcpyDeviceToHost));
    return AESW_OK;
}

}  // extern "C"
