// aesw_ctx.h -- the opaque context of include/aesw.h and the small helpers every translation unit of the C ABI uses
// (aesw_api.cpp: entry points; aesw_arena.cpp: the probed column arena).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/aesw.h"

struct aesw_ctx {
    int device = -1;
    uint8_t *d_tables = nullptr;  // 768 B
    uint8_t *d_fr_lut = nullptr;  // 256 x 32 B
    uint32_t *d_ftab[3] = {nullptr, nullptr, nullptr};  // flush descriptors per layout (aesw_layout.h "scheduled flush")
    uint32_t *d_chktab[2] = {nullptr, nullptr};         // check tables of the DENSE / PACKED layout (aesw_check.h), uploaded by aesw_create
    // The scheduled key (FixedAes128Config::schedule_key, src/aes128.rs:143-152: `self.keys = Some(..)` replaces the key between
    // encrypt calls).  Round keys live in SLOTS of 256 B (176 used); every aesw_schedule_key_device takes the next slot of a small
    // ring and every scheduled-key launch bakes the pointer of the slot that is current when it is ENQUEUED, so a launch never sees
    // a later key.  A slot is rewritten only behind every launch that reads it: one event per distinct reader stream (re-recorded
    // by that stream's later launches, which are ordered behind its earlier ones), all of them waited on by the schedule that
    // reuses the slot.  Slots a hipGraph capture has touched (a captured schedule writes one, a captured launch reads one on every
    // replay) are PINNED: the ring never hands them out again.
    struct KeyReader { hipStream_t s; hipEvent_t e; };
    struct KeySlot {
        uint8_t *d = nullptr;
        bool pinned = false;
        hipEvent_t ready = nullptr;  // recorded behind the key launch that wrote the slot: launches on other streams wait on it
        hipStream_t writer = nullptr;
        std::vector<KeyReader> readers;  // launches that may still be reading the slot
    };
    std::vector<KeySlot> key_slots;
    std::vector<uint8_t *> key_chunks;   // hipMalloc'ed backing of the slots (KEY_CHUNK_SLOTS each)
    std::vector<hipEvent_t> event_pool;  // reader events not in use
    std::vector<int> key_ring_slots;     // the ring: indices into key_slots, at most key_ring of them
    std::vector<int> key_spare;          // slots taken out of the ring when "key_slots" shrank (their readers are still tracked)
    int key_pos = 0;        // ring position of the slot the last eager schedule wrote
    int key_cur = -1;       // slot of the current key (-1: none scheduled)
    int key_ring = 4;       // option "key_slots": un-pinned slots the ring cycles through (1 = every schedule waits for all readers)
    uint64_t key_waits = 0;  // statistics: reader events a schedule had to wait on (option "key_reader_waits", read-only)
    bool have_key = false;
    bool xt = false;
    int waves_shared = 0;  // waves per group, shared-key kernels (0 = auto)
    int waves_pbk = 0;     // per-block-key and key kernels (0 = auto)
    int nt = 1;  // store flavour: 0 plain, 1 nontemporal (default since round 3), 2 write-through (sc1).  With all three flavours compiled to the
                 // same code (round 3: they used to differ by 60 VGPRs, i.e. in residency) nontemporal stores are 1-3 % ahead at 2^20 blocks on
                 // well-placed columns and within +-2 % of sc1 elsewhere (profiles/r03_study/README.md)
    int key_nt = 1;  // store flavour of key_kernel (one contiguous flush per column at the end): nontemporal 4-9 % ahead of sc1 (tools/keysweep.py)
    int fr_geo = 1;  // geometry of expand_fr: 1 = one-shot 4 KiB workgroups, LUT gathered from global memory: 7.3 TB/s with nontemporal stores
                     // against 5.2 for 0 = striding workgroups + LDS LUT and 5.9 for 2 = one-shot 16 KiB + LDS LUT (tools/frsweep.py)
    int asm_geo = 4;  // geometry of the Fr form of assemble: 0 striding workgroups, 1 one-shot (chunk, segment, column) grid, 2 / 3 / 4 one-shot workgroups
                      // on aligned output chunks: 256 threads x 1 piece, 256 x 2, 128 x 2 (4 = default: 6.9 TB/s for K = 20, N = 5 against 5.3 striding)
                      // workgroups on a (chunk, segment, column) grid (round 3: byte-exact, 5.2 TB/s -- a piece is a chain of three dependent loads
                      // (index table, slab byte, LUT) and a one-shot workgroup has nothing else in flight: latency x residency bounds it, not divisions)
    int fr_nt = 1;  // store flavour of the Fr-expanding kernels: nontemporal measured 19 % ahead of plain and sc1 there (tools/frsweep.py)
    int64_t grid_cap = 0;  // max workgroups per launch (0 = one per block group)
    uint32_t xcd_remap = 1;  // xcd_group() mode: 0 dispatch order, 1 one contiguous eighth of the groups per XCD (+3-4 % at 2^20 blocks over 0, tools/sweep.py xcd), C >= 2 turns of C groups
    int64_t lds_pad = 0;  // diagnostic (tools/occ.py): extra dynamic LDS per workgroup, lowers residency
    int arena_align_log2 = 0;  // aesw_columns_alloc: column alignment (0 = auto: 2 MiB)
    int arena_probe = -1;      // candidate backings aesw_columns_alloc measures per unit (-1 = auto, 0 = none: one hipMalloc)
    double best_fill_us_per_gb = 0;  // fastest linear fill any arena search of this context has seen (us per 10^9 bytes): the probe's yardstick
    int arena_unit = 2;        // what a candidate is: 0 = the whole set of columns in one range, 1 = one column (greedy, largest first),
                               // 2 = whole sets first, columns if no set candidate runs the pattern as fast as its fill (default)
    struct ArenaRange { void *p; size_t bytes; bool vmm; };  // vmm: built with the virtual-memory API (freed by unmap), else hipMalloc
    // A probed arena: its ranges, and what it was placed for (the shape decides whether a later request may take it over)
    struct ArenaRec {
        void *key;
        std::vector<ArenaRange> ranges;
        uint64_t n = 0;
        int layout = 0, with_key_slab = 0, with_ct = 0;
        uint32_t xcd = 0;          // the "xcd_remap" the store pattern was probed with
        aesw_columns cols = {};    // the column pointers and probe results handed to the caller
        uint64_t stamp = 0;        // when it was freed (cache order)
    };
    std::vector<ArenaRec> vmm_arenas;  // arenas built with the virtual-memory API (one range per column; freed by unmap, not hipFree)
    // Placement cache: a probed arena that is freed keeps its backing (physical placement is what the search paid for); the next
    // aesw_columns_alloc of the same shape takes it over without a search.  Bounded by arena_cache_max_bytes, oldest out first;
    // flushed when a search runs short of memory, by option "arena_cache" = 0 and by aesw_destroy.
    std::vector<ArenaRec> arena_cache;
    int arena_cache_on = 1;
    uint64_t arena_cache_max_bytes = (uint64_t)64 << 30;
    uint64_t arena_stamp = 0;
    uint64_t arena_cache_hits = 0;
    int64_t arena_probe_budget_ms = 3000;  // a search stops building candidates once it has run this long (0 = no limit); it always keeps the best so far
#ifdef AESW_TRACE
    uint64_t *trace = nullptr;
#endif
    int64_t chunk_blocks = 1 << 15;  // host-pointer path: blocks per pipeline stage
    int copy_threads = -1;           // host threads that move a stage from the page-locked bounce buffer into a pageable destination (-1 = auto)
    std::string last_error;
    hipStream_t s_compute = nullptr, s_copy = nullptr;
    int split_small = 0;    // experiment of round 4 (profiles/r04_study/split_small.md): a LONE shared / scheduled-key launch of 2^15 .. 2^17 blocks
                            // dealt as this many line-aligned sub-ranges onto the internal streams (0 / 1 = off)
    bool in_split = false;  // re-entrancy guard of the above
    int batch_streams = 3;  // aesw_encrypt_witness_batches_device: internal streams the batches are dealt onto
    hipStream_t s_batch[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint8_t *bounce[2] = {nullptr, nullptr};  // page-locked staging for pageable destinations
    size_t bounce_bytes = 0;
    uint8_t *scratch = nullptr;  // device buffers of the host-pointer path (grow-only)
    size_t scratch_bytes = 0;
    aesw_stream_stats stats = {};  // of the last streaming call
    int stream_check = 0;          // option: aesw_encrypt_witness_stream checks every chunk on the device before it travels (aesw_check.h)
    int64_t stream_poison = 0;     // diagnostic (tests): block index + 1 whose y / z cells the stream overwrites before its chunk is checked and shipped
    aesw_check_report stream_report = {0, 0, 0, 0, 0, 0, ~0ull};  // of the last streaming call with "stream_check" on
};

inline int fail_hip(aesw_ctx *ctx, hipError_t e, const char *what) {
    if (ctx) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        ctx->last_error = buf;
    }
    return e == hipErrorOutOfMemory ? AESW_ERR_NOMEM : AESW_ERR_HIP;
}

#define HIP_TRY(ctx, expr)                                  \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #expr); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

void aesw_arena_cache_trim(aesw_ctx *ctx, uint64_t keep_bytes);  // aesw_arena.cpp: release cached arenas, oldest first, until keep_bytes stay

inline bool aesw_valid_layout(int l) { return l == AESW_LAYOUT_DENSE || l == AESW_LAYOUT_PACKED || l == AESW_LAYOUT_VALUES; }
