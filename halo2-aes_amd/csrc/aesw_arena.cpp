// aesw_arena.cpp -- aesw_columns_alloc / aesw_columns_free: the device home of a batch's output columns, with its
// physical backing chosen by measurement (include/aesw.h "Placement probing"; the study behind it:
// profiles/r03_study/README.md, tools/allocbench.hip, tools/probebench.hip).  Host code only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "../../include/aesw.h"
#include "aesw_ctx.h"
#include "aesw_internal.h"
#include "aesw_layout.h"

using namespace aesw;

namespace {
bool valid_layout(int l) { return aesw_valid_layout(l); }
}  // namespace

extern "C" {

namespace {

// A virtual range of `total` bytes backed by physical chunks of `chunk` bytes each (the last one shorter).
int vmm_build(aesw_ctx *ctx, size_t total, size_t chunk, void **out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *va = nullptr;
    HIP_TRY(ctx, hipMemAddressReserve(&va, total, 0, nullptr, 0));
    size_t mapped = 0;
    hipError_t e = hipSuccess;
    while (mapped < total && e == hipSuccess) {
        const size_t sz = total - mapped < chunk ? total - mapped : chunk;
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, sz, &prop, 0);
        if (e != hipSuccess) break;
        e = hipMemMap(reinterpret_cast<uint8_t *>(va) + mapped, sz, 0, h, 0);
        (void)hipMemRelease(h);  // the mapping keeps the chunk alive; an unmapped chunk is gone with this
        if (e == hipSuccess) mapped += sz;
    }
    if (e == hipSuccess) e = hipMemSetAccess(va, total, &acc, 1);
    if (e != hipSuccess) {
        if (mapped) (void)hipMemUnmap(va, mapped);
        (void)hipMemAddressFree(va, total);
        return fail_hip(ctx, e, "virtual-memory arena (hipMemCreate / hipMemMap / hipMemSetAccess)");
    }
    *out = va;
    return AESW_OK;
}

void vmm_release(void *va, size_t total) {
    (void)hipMemUnmap(va, total);
    (void)hipMemAddressFree(va, total);
}

void release_ranges(const std::vector<aesw_ctx::ArenaRange> &ranges) {
    for (const auto &r : ranges) { if (r.vmm) vmm_release(r.p, r.bytes); else (void)hipFree(r.p); }
}

uint64_t cache_bytes(const aesw_ctx *ctx) {
    uint64_t b = 0;
    for (const auto &a : ctx->arena_cache) b += a.cols.bytes;
    return b;
}

// oldest entries out until at most `keep` bytes stay cached
void cache_trim(aesw_ctx *ctx, uint64_t keep) {
    while (!ctx->arena_cache.empty() && cache_bytes(ctx) > keep) {
        size_t oldest = 0;
        for (size_t i = 1; i < ctx->arena_cache.size(); ++i)
            if (ctx->arena_cache[i].stamp < ctx->arena_cache[oldest].stamp) oldest = i;
        release_ranges(ctx->arena_cache[oldest].ranges);
        ctx->arena_cache.erase(ctx->arena_cache.begin() + (long)oldest);
    }
}

}  // namespace

int aesw_columns_alloc(aesw_ctx *ctx, uint64_t n, int layout, int with_key_slab, int with_ct, aesw_columns *out) {
    if (!ctx || !out || !valid_layout(layout) || n == 0 || n > ((uint64_t)1 << 40)) return AESW_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    // with_key_slab 2: the key-schedule witness alone (aesw_key_schedule_witness_device): no encrypt columns
    const bool key_only = with_key_slab == 2;
    const uint64_t sx = key_only ? 0 : aesw_column_stride(layout, 0), sy = key_only ? 0 : aesw_column_stride(layout, 1),
                   sz = key_only ? 0 : aesw_column_stride(layout, 2);
    const uint64_t align = ctx->arena_align_log2 ? (uint64_t)1 << ctx->arena_align_log2 : (uint64_t)2 << 20;
    // sizes in the order the columns are laid out; a column of size 0 takes no room
    const uint64_t size[8] = {n * sx, n * sy, n * sz, with_ct ? n * 16 : 0,
                              with_key_slab ? n * WORDS_ROWS : 0, with_key_slab ? n * aesw_key_column_stride(layout, 0) : 0,
                              with_key_slab ? n * aesw_key_column_stride(layout, 1) : 0, with_key_slab ? n * aesw_key_column_stride(layout, 2) : 0};
    uint64_t off[8], end = 0;
    for (int i = 0; i < 8; ++i) {
        end = (end + align - 1) / align * align;
        off[i] = end;
        end += size[i];
    }
    DeviceGuard g(ctx->device);
    if (!g.ok) return AESW_ERR_NO_DEVICE;
    auto fill_out = [&](uint8_t *b) {
        auto at = [&](int i) -> uint8_t * { return size[i] ? b + off[i] : nullptr; };
        out->x = at(0); out->y = at(1); out->z = at(2); out->ct = at(3);
        out->key.w = at(4); out->key.kx = at(5); out->key.ky = at(6); out->key.kz = at(7);
    };
    int probe = ctx->arena_probe < 0 ? (n >= ((uint64_t)1 << 16) ? 8 : 0) : ctx->arena_probe;
    if (probe == 0) {
        // hipMalloc returns memory aligned to the allocation granule only: over-allocate by one alignment unit
        uint8_t *raw = nullptr;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&raw), end + align));
        out->base = raw;
        out->bytes = end + align;
        fill_out(raw + (align - reinterpret_cast<uintptr_t>(raw) % align) % align);
        return AESW_OK;
    }
    // Placement cache: this context has placed (and the caller has freed) an arena of exactly this shape: take it over.  The
    // backing is the one the earlier search chose, still mapped, so the pattern runs at the rate measured then.
    if (ctx->arena_cache_on) {
        for (size_t i = ctx->arena_cache.size(); i-- > 0;) {
            aesw_ctx::ArenaRec &c = ctx->arena_cache[i];
            if (c.n == n && c.layout == layout && c.with_key_slab == with_key_slab && c.with_ct == (with_ct ? 1 : 0) && c.xcd == ctx->xcd_remap) {
                *out = c.cols;
                out->candidates = 0;  // nothing was built or timed for this call; probe_us / fill_us are the earlier search's
                ctx->vmm_arenas.push_back(std::move(c));
                ctx->arena_cache.erase(ctx->arena_cache.begin() + (long)i);
                ++ctx->arena_cache_hits;
                return AESW_OK;
            }
        }
        // a new shape: what is cached must not starve the search (it holds its losing candidates until it ends)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        if (free_b < 2 * end + ((size_t)16 << 30)) cache_trim(ctx, 0);
    }
    const auto t_search = std::chrono::steady_clock::now();
    auto over_budget = [&]() {
        if (ctx->arena_probe_budget_ms <= 0) return false;
        return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_search).count() > ctx->arena_probe_budget_ms;
    };
    // Search by measurement.  A UNIT is what one candidate backs: the whole set of columns in one range ("arena_unit" 0), or
    // one column ("arena_unit" 1: greedy, largest column first).  For every unit up to `probe` candidates are built -- a plain
    // hipMalloc, then virtual ranges over physical chunks of 8 / 2 / 32 / 4 MiB, and round again --, the store-pattern
    // emulation and a linear fill are timed over the units chosen so far PLUS the candidate, the candidate whose pattern
    // runs fastest is kept (the search of a unit stops at the first candidate whose pattern runs as fast as its fill).  Candidates
    // that lose are HELD until the whole search is over (otherwise the driver hands the same memory out again), then released.
    using Range = aesw_ctx::ArenaRange;
    const uint32_t strides7[7] = {(uint32_t)sx, (uint32_t)sy, (uint32_t)sz, WORDS_ROWS, aesw_key_column_stride(layout, 0),
                                  aesw_key_column_stride(layout, 1), aesw_key_column_stride(layout, 2)};
    const int size_of7[7] = {0, 1, 2, 4, 5, 6, 7};  // probe column c (x y z w kx ky kz) -> index into size[] / off[]
    const size_t MiB2 = (size_t)2 << 20;
    auto round2m = [&](uint64_t v) { return (size_t)((v + MiB2 - 1) / MiB2 * MiB2); };
    auto build = [&](int kind, size_t bytes, Range *r) -> int {
        static const size_t chunk_of[4] = {(size_t)8 << 20, (size_t)2 << 20, (size_t)32 << 20, (size_t)4 << 20};
        if (kind % 5 == 0) {
            void *q = nullptr;
            HIP_TRY(ctx, hipMalloc(&q, bytes));  // 2 MiB aligned for allocations of this size
            *r = Range{q, bytes, false};
            return AESW_OK;
        }
        void *va = nullptr;
        const int rc = vmm_build(ctx, bytes, chunk_of[(kind % 5) - 1], &va);
        if (rc == AESW_OK) *r = Range{va, bytes, true};
        return rc;
    };
    aesw_ctx::ArenaRec rec{nullptr, {}};
    std::vector<Range> losers;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    bool done = false;
    struct Cleanup {
        aesw_ctx::ArenaRec &rec; std::vector<Range> &losers; bool &done; hipEvent_t &a, &b, &d, &e;
        ~Cleanup() {
            for (auto &r : losers) { if (r.vmm) vmm_release(r.p, r.bytes); else (void)hipFree(r.p); }
            if (!done)
                for (auto &r : rec.ranges) { if (r.vmm) vmm_release(r.p, r.bytes); else (void)hipFree(r.p); }
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
            if (d) (void)hipEventDestroy(d);
            if (e) (void)hipEventDestroy(e);
        }
    } cleanup{rec, losers, done, e0, e1, e2, e3};
    HIP_TRY(ctx, hipEventCreate(&e3));
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventCreate(&e2));
    ProbeParams pp = {};
    for (int c = 0; c < 7; ++c) pp.stride[c] = strides7[c];
    pp.n = n;
    pp.xcd_mode = ctx->xcd_remap;
    // two timed passes of a 2^20-block set; proportionally more for smaller batches (short launches time noisily)
    const int passes = n >= ((uint64_t)1 << 20) ? 2 : (int)std::min<uint64_t>(32, (((uint64_t)1 << 21) + n - 1) / n);
    // "as fast as a linear fill": the two levels are 0.98 - 1.02 and >= 1.05 of the fill at 2^18 blocks and more; a shorter launch
    // carries its ramp and tail (2^16 blocks: 1.04 - 1.06 at best), so the bar is lower there and the per-column search is not run
    const float accept = n >= ((uint64_t)1 << 18) ? 1.025f : 1.065f;
    uint32_t total_cands = 0;
    struct Placement {
        std::vector<Range> ranges;
        uint8_t *col[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        uint8_t *ct = nullptr;
        float probe_us = 0.f, fill_us = 1.f;
        int rc = AESW_OK;
    };
    // one search: unit_mode 0 = whole-set candidates, 1 = one column at a time (greedy, largest first)
    auto search = [&](int unit_mode) -> Placement {
        Placement pl;
        ProbeParams q = pp;
        for (int c = 0; c < 7; ++c) q.col[c] = nullptr;
        std::vector<std::vector<int>> units;
        if (unit_mode == 0) {
            units.push_back({0, 1, 2, 3, 4, 5, 6});
        } else {
            int order[7] = {0, 1, 2, 3, 4, 5, 6};
            std::sort(order, order + 7, [&](int a, int b) { return size[size_of7[a]] > size[size_of7[b]]; });
            for (int c : order)
                if (size[size_of7[c]]) units.push_back({c});
        }
        auto fail = [&](int rc) { for (auto &r : pl.ranges) losers.push_back(r); pl.ranges.clear(); pl.rc = rc; return pl; };
        auto hip = [&](hipError_t e, const char *what) { return e == hipSuccess ? AESW_OK : fail_hip(ctx, e, what); };
        for (size_t ui = 0; ui < units.size(); ++ui) {
            const std::vector<int> &cols = units[ui];
            // layout of the unit: its columns one after the other on 2 MiB boundaries (the ciphertext rides with a whole-set unit)
            size_t uoff[8], ubytes = 0;
            for (int c : cols) { uoff[c] = ubytes; ubytes += round2m(size[size_of7[c]]); }
            const bool with_ct_here = unit_mode == 0 && size[3];
            if (with_ct_here) { uoff[7] = ubytes; ubytes += round2m(size[3]); }
            if (!ubytes) continue;
            Range best{nullptr, 0, false};
            float best_probe = 0.f, best_fill = 1.f;
            float ref_fill = 0.f;  // the fastest linear fill seen for this unit: one slow fill sample must not make a candidate look good
            for (int k = 0; k < probe; ++k) {
                if (best.p && over_budget()) break;  // "arena_probe_budget_ms": keep the best candidate so far
                if (best.p) {  // the losers are held until the search ends: never let them take the device's last memory
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
                    if (free_b < ubytes + ((size_t)16 << 30)) break;
                }
                Range r{nullptr, 0, false};
                int rc = build(k + (int)ui, ubytes, &r);
                if (rc != AESW_OK) {
                    (void)hipGetLastError();       // an out-of-memory here must not surface from the next launch's error check
                    if (!best.p) return fail(rc);  // not even one candidate for this unit fits
                    break;                         // memory is getting short: choose among what we have
                }
                losers.push_back(r);  // owned by ~Cleanup unless chosen below
                ++total_cands;
                for (int c : cols) q.col[c] = size[size_of7[c]] ? reinterpret_cast<uint8_t *>(r.p) + uoff[c] : nullptr;
                rc = hip(launch_probe(q, false, nullptr), "probe launch");  // first touch + warm-up, untimed
                if (rc == AESW_OK) rc = hip(hipEventRecord(e0, nullptr), "hipEventRecord");
                for (int i = 0; i < passes && rc == AESW_OK; ++i) rc = hip(launch_probe(q, false, nullptr), "probe launch");
                if (rc == AESW_OK) rc = hip(hipEventRecord(e1, nullptr), "hipEventRecord");
                if (rc == AESW_OK) rc = hip(launch_probe(q, true, nullptr), "probe launch");  // one untimed fill: the first one after a search's releases runs slow
                if (rc == AESW_OK) rc = hip(hipEventRecord(e3, nullptr), "hipEventRecord");
                for (int i = 0; i < passes && rc == AESW_OK; ++i) rc = hip(launch_probe(q, true, nullptr), "probe launch");
                if (rc == AESW_OK) rc = hip(hipEventRecord(e2, nullptr), "hipEventRecord");
                if (rc == AESW_OK) rc = hip(hipEventSynchronize(e2), "hipEventSynchronize");
                float f = 0, l = 0;
                if (rc == AESW_OK) rc = hip(hipEventElapsedTime(&f, e0, e1), "hipEventElapsedTime");
                if (rc == AESW_OK) rc = hip(hipEventElapsedTime(&l, e3, e2), "hipEventElapsedTime");
                if (rc != AESW_OK) return fail(rc);
                {   // the yardstick: the fastest fill seen for this unit, and no slower than the fastest fill per byte this context ever saw
                    double probed_bytes = 0;
                    for (int c = 0; c < 7; ++c) if (q.col[c]) probed_bytes += (double)n * q.stride[c];
                    if (probed_bytes >= 1e9) {  // launches long enough that time per byte is a rate, not ramp and tail
                        const double per_gb = (double)l * 1e3 / passes / (probed_bytes * 1e-9);
                        if (ctx->best_fill_us_per_gb == 0 || per_gb < ctx->best_fill_us_per_gb) ctx->best_fill_us_per_gb = per_gb;
                        const float floor_ms = (float)(ctx->best_fill_us_per_gb * probed_bytes * 1e-9 * passes * 1e-3);
                        if (l > 1.03f * floor_ms) l = 1.03f * floor_ms;
                    }
                }
                if (ref_fill == 0.f || l < ref_fill) ref_fill = l;
                const float ratio = f / ref_fill;
                if (!best.p || f < best_probe * passes * 1e-3f) { best = r; best_probe = f * 1e3f / passes; }
                best_fill = ref_fill * 1e3f / passes;
                if (ratio <= accept) break;  // the many-front pattern as fast as a linear fill: as good as it gets
            }
            for (size_t i = 0; i < losers.size(); ++i)
                if (losers[i].p == best.p) { losers.erase(losers.begin() + (long)i); break; }
            pl.ranges.push_back(best);
            for (int c : cols) q.col[c] = size[size_of7[c]] ? reinterpret_cast<uint8_t *>(best.p) + uoff[c] : nullptr;
            if (with_ct_here) pl.ct = reinterpret_cast<uint8_t *>(best.p) + uoff[7];
            pl.probe_us = best_probe;  // after the last unit: the whole set as finally placed
            pl.fill_us = best_fill;
        }
        if (size[3] && !pl.ct) {  // column units: the ciphertext is one small linear stream outside the pattern: any backing
            Range r{nullptr, 0, false};
            const int rc = build(1, round2m(size[3]), &r);
            if (rc != AESW_OK) return fail(rc);
            pl.ranges.push_back(r);
            pl.ct = reinterpret_cast<uint8_t *>(r.p);
        }
        for (int c = 0; c < 7; ++c) pl.col[c] = q.col[c];
        return pl;
    };
    // "arena_unit" 2 (default): whole-set candidates first; when none of them runs the pattern as fast as its fill, a second
    // search places the columns one at a time (the whole-set losers stay held meanwhile) and the better of the two is kept
    Placement pl = search(ctx->arena_unit == 1 ? 1 : 0);
    if (pl.rc != AESW_OK) return pl.rc;
    if (ctx->arena_unit == 2 && n >= ((uint64_t)1 << 18) && pl.probe_us > accept * pl.fill_us && !over_budget()) {
        Placement alt = search(1);
        if (alt.rc == AESW_OK && alt.probe_us / alt.fill_us < pl.probe_us / pl.fill_us) std::swap(pl, alt);
        for (auto &r : alt.ranges) losers.push_back(r);  // the search that lost (or failed half-way: already handed over)
    }
    rec.ranges = pl.ranges;
    out->x = pl.col[0]; out->y = pl.col[1]; out->z = pl.col[2]; out->ct = pl.ct;
    out->key.w = pl.col[3]; out->key.kx = pl.col[4]; out->key.ky = pl.col[5]; out->key.kz = pl.col[6];
    out->base = reinterpret_cast<uint8_t *>(rec.ranges[0].p);  // the handle aesw_columns_free looks the arena up by
    for (auto &r : rec.ranges) out->bytes += r.bytes;
    out->candidates = total_cands;
    out->chosen = 0;
    out->probe_us = pl.probe_us;
    out->fill_us = pl.fill_us;
    rec.key = out->base;
    rec.n = n; rec.layout = layout; rec.with_key_slab = with_key_slab; rec.with_ct = with_ct ? 1 : 0; rec.xcd = ctx->xcd_remap;
    rec.cols = *out;
    ctx->vmm_arenas.push_back(rec);
    done = true;
    return AESW_OK;  // ~Cleanup releases the candidates that were not chosen
}

}  // extern "C"

// used by aesw_set_option("arena_cache" / "arena_cache_max_mb") and aesw_destroy (aesw_api.cpp)
void aesw_arena_cache_trim(aesw_ctx *ctx, uint64_t keep_bytes) {
    DeviceGuard g(ctx->device);
    if (g.ok) cache_trim(ctx, keep_bytes);
}

extern "C" {

int aesw_columns_free(aesw_ctx *ctx, aesw_columns *cols) {
    if (!ctx || !cols) return AESW_ERR_INVALID_ARG;
    if (cols->base) {
        DeviceGuard g(ctx->device);
        if (!g.ok) return AESW_ERR_NO_DEVICE;
        bool vmm = false;
        for (size_t i = 0; i < ctx->vmm_arenas.size(); ++i)
            if (ctx->vmm_arenas[i].key == cols->base) {
                if (ctx->arena_cache_on && ctx->vmm_arenas[i].cols.bytes <= ctx->arena_cache_max_bytes) {
                    // keep the placement: the next arena of this shape takes it over (freed memory would be handed out again
                    // in some other combination, and the search would start over)
                    ctx->vmm_arenas[i].stamp = ++ctx->arena_stamp;
                    ctx->arena_cache.push_back(std::move(ctx->vmm_arenas[i]));
                    ctx->vmm_arenas.erase(ctx->vmm_arenas.begin() + (long)i);
                    cache_trim(ctx, ctx->arena_cache_max_bytes);
                } else {
                    release_ranges(ctx->vmm_arenas[i].ranges);
                    ctx->vmm_arenas.erase(ctx->vmm_arenas.begin() + (long)i);
                }
                vmm = true;
                break;
            }
        if (!vmm) HIP_TRY(ctx, hipFree(cols->base));
    }
    std::memset(cols, 0, sizeof *cols);
    return AESW_OK;
}

}  // extern "C"
