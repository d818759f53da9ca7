// lane_model.cpp -- CPU execution of the device program's shared source:
// the per-lane code (halo2-aes_amd/csrc/aesw_lane.h), the staging windows and
// the scheduled whole-line flush (aesw_layout.h: descriptor table), wave by wave (16 blocks),
// with the cross-lane steps (DPP quad permutes) and LDS replaced by arrays.
// TEST INFRASTRUCTURE: lets `-m "not gpu"` tests compare the device program
// with the oracle before any GPU run.  Not reachable from the product library.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../halo2-aes_amd/csrc/aesw_lane.h"
#include "../../halo2-aes_amd/csrc/aesw_check.h"

using namespace aesw;

namespace {

constexpr int BPW = 16;

template <int L>
struct WinSink {  // writes into the block's three staging windows, like DevSink
    uint8_t *win[3];
    int w;
    static void put(uint8_t *p, uint32_t v) { std::memcpy(p, &v, 4); }
    template <int C> void plain(int off, uint32_t v) { put(win[C] + off + 4 * w, v); }
    template <int C> void mix(int off, int k, uint32_t v) {
        constexpr int mw = C == 0 ? Geo<L>::X_MIXW : C == 1 ? Geo<L>::Y_MIXW : Geo<L>::Z_MIXW;
        put(win[C] + off + mw * w + 4 * k, v);
    }
};

struct HostKSink {
    uint8_t *x, *y, *z, *wd;
    static void put(uint8_t *p, uint32_t v) { if (p) std::memcpy(p, &v, 4); }
    void kx(int off, uint32_t v) { put(x ? x + off : nullptr, v); }
    void ky(int off, uint32_t v) { put(y ? y + off : nullptr, v); }
    void kz(int off, uint32_t v) { put(z ? z + off : nullptr, v); }
    void words(int off, uint32_t v) { put(wd ? wd + off : nullptr, v); }
};

uint32_t ld32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

// One column's flush after round R for the whole wave, as the kernel runs it: the host-built descriptor table
// (build_flush_table), instruction by instruction, every lane storing its 16-byte piece when "offset < nvalid*stride".
template <class W>
struct ColTable {
    std::vector<uint32_t> t;
    ColTable() : t((size_t)sched_first<W>(10) * 64) { build_flush_table<W>(t.data()); }
};

template <class W>
void flush_col(int R, const std::vector<uint8_t> &stage, uint8_t *g, const ColTable<W> &tab, int nvalid) {
    const uint32_t limit = (uint32_t)nvalid * W::GSTRIDE;
    for (int i = sched_first<W>(R); i < sched_first<W>(R + 1); ++i)
        for (int lane = 0; lane < 64; ++lane) {
            const uint32_t d = tab.t[(size_t)i * 64 + lane], off = d >> 16;
            if (off < limit) std::memcpy(g + off, stage.data() + (d & 0xffffu), 16);
        }
}

// The scheduled flush against flush_piece (the closed-form specification) for every round and number of valid
// blocks: the same pieces leave in the same round from the same LDS bytes.  Returns the number of disagreements.
template <class W>
int flush_forms_disagree() {
    int bad = 0;
    const ColTable<W> tab;
    const int npieces = BPW * W::GSTRIDE / 16;
    for (int nvalid = 1; nvalid <= BPW; ++nvalid) {
        std::vector<int> spec(npieces, -1), got(npieces, -1);  // piece -> lds_off | round << 20
        for (int R = 1; R <= 9; ++R)
            for (int b = 0; b < BPW; ++b)
                for (int sub = 0; sub < 8; ++sub)
                    for (int t = 0; t < flush_maxc<W>(R); ++t) {
                        const FlushPiece fp = flush_piece<W>(R, b, sub, t, nvalid);
                        if (fp.ok) spec[fp.P / 16] = fp.lds_off | (R << 20);
                    }
        for (int R = 1; R <= 9; ++R)
            for (int i = sched_first<W>(R); i < sched_first<W>(R + 1); ++i)
                for (int lane = 0; lane < 64; ++lane) {
                    const uint32_t d = tab.t[(size_t)i * 64 + lane], off = d >> 16;
                    if (off < (uint32_t)nvalid * W::GSTRIDE) got[off / 16] = (int)(d & 0xffffu) | (R << 20);
                    // eight lanes = one whole 128-byte line; unused slots only in a round's last instruction, and only
                    // when the round's line count is not a multiple of 8 (the kernel predicates just that instruction)
                    if ((d >> 16) != SCHED_INVALID_P && ((d >> 16) & 127) != 16u * (lane & 7)) bad++;
                    if ((d >> 16) != SCHED_INVALID_P && (d >> 16) != (tab.t[(size_t)i * 64 + (lane & ~7)] >> 16) + 16u * (lane & 7)) bad++;
                    if ((d >> 16) == SCHED_INVALID_P && (i != sched_first<W>(R + 1) - 1 || sched_nlines<W>(R) % 8 == 0)) bad++;
                }
        for (int p = 0; p < npieces; ++p) bad += spec[p] != got[p];
    }
    return bad;
}

// Every 16-byte piece of a wave's column range is stored exactly once, for every number of valid blocks
// (a line flushed twice is invisible in the output -- same bytes -- but costs bandwidth and leans on store
// ordering).  Returns the number of pieces stored more or less than once.
template <class W>
int flush_not_exactly_once() {
    int bad = 0;
    const ColTable<W> tab;
    for (int nvalid = 1; nvalid <= BPW; ++nvalid) {
        std::vector<int> seen((size_t)BPW * W::GSTRIDE / 16, 0);
        for (uint32_t d : tab.t)
            if ((d >> 16) < (uint32_t)nvalid * W::GSTRIDE) seen[(d >> 16) / 16]++;
        for (size_t i = 0; i < seen.size(); ++i) bad += seen[i] != ((int)i < nvalid * W::GSTRIDE / 16 ? 1 : 0);
    }
    return bad;
}

template <int L, bool XT>
void run(const uint8_t *tab, const uint8_t *pt, const uint8_t *keys, int per_block_keys, int key_only, uint64_t n,
         uint8_t *x, uint8_t *y, uint8_t *z, uint8_t *ct, uint8_t *wd, uint8_t *kx, uint8_t *ky, uint8_t *kz,
         uint8_t *rk_out) {
    using G = Geo<L>;
    using WX = WinX<L>;
    using WY = WinY<L>;
    using WZ = WinZ<L>;
    Tables<XT> T{tab};
    std::vector<uint32_t> rk((size_t)BPW * 44);
    bool have_shared = false;
    uint32_t rks[44];
    for (uint64_t blk0 = 0; blk0 < n; blk0 += BPW) {
        const int nvalid = n - blk0 >= (uint64_t)BPW ? BPW : (int)(n - blk0);
        // ---- key phase
        for (int b = 0; b < BPW; ++b) {
            const bool live = b < nvalid;
            const bool own_key = per_block_keys || key_only;
            if (!own_key && have_shared) { std::memcpy(&rk[b * 44], rks, sizeof rks); continue; }
            uint8_t zero[16] = {0};
            const uint8_t *key = own_key ? (live ? keys + 16 * (blk0 + b) : zero) : keys;
            const bool emit = own_key && live;
            uint32_t *r = &rk[b * 44];
            for (int w = 0; w < 4; ++w) r[w] = ld32(key + 4 * w);
            const uint64_t gb = blk0 + b;
            HostKSink ks{emit && kx ? kx + (size_t)G::KXS * gb : nullptr, emit && ky ? ky + (size_t)G::KYS * gb : nullptr,
                         emit && kz ? kz + (size_t)G::KZS * gb : nullptr, emit && wd ? wd + (size_t)WORDS_ROWS * gb : nullptr};
            for (int w = 0; w < 4; ++w) ks.words(4 * w, r[w]);
            for (int rho = 1; rho <= 10; ++rho)
                for (int w = 0; w < 4; ++w)
                    r[4 * rho + w] = emit_key_round<L>(ks, rho, w, r[4 * rho - 4], r[4 * rho - 3], r[4 * rho - 2],
                                                       r[4 * rho - 1], rcon(rho - 1), T);
            if (emit && rk_out) std::memcpy(rk_out + (size_t)RK_BYTES * gb, r, RK_BYTES);
            if (!own_key) { std::memcpy(rks, r, sizeof rks); have_shared = true; }
        }
        if (key_only) continue;
        // ---- encrypt phase: one wave = 16 block windows per column
        std::vector<uint8_t> sx((size_t)BPW * WX::BYTES, 0xEE), sy((size_t)BPW * WY::BYTES, 0xEE), sz((size_t)BPW * WZ::BYTES, 0xEE);
        uint32_t st[BPW][4], sub[4], sh[4];
        auto sink = [&](int b, int w) {
            return WinSink<L>{{sx.data() + b * WX::BYTES, sy.data() + b * WY::BYTES, sz.data() + b * WZ::BYTES}, w};
        };
        for (int b = 0; b < BPW; ++b)
            for (int w = 0; w < 4; ++w) {
                auto s = sink(b, w);
                const uint32_t ptw = b < nvalid ? ld32(pt + 16 * (blk0 + b) + 4 * w) : 0u;
                st[b][w] = emit_head<L>(s, ptw, rk[b * 44 + w]);
            }
        uint8_t *gx = x + (size_t)G::XS * blk0, *gy = y + (size_t)G::YS * blk0, *gz = z + (size_t)G::ZS * blk0;
        static const ColTable<WX> fx;
        static const ColTable<WY> fy;
        static const ColTable<WZ> fz;
        for (int R = 1; R <= 9; ++R) {
            for (int b = 0; b < BPW; ++b) {
                for (int w = 0; w < 4; ++w) { auto s = sink(b, w); sub[w] = emit_sbox<L>(s, WX::woff(R), WY::woff(R), WZ::woff(R), st[b][w], T); }
                for (int w = 0; w < 4; ++w) sh[w] = shift_rows(sub[w], sub[(w + 1) & 3], sub[(w + 2) & 3], sub[(w + 3) & 3]);
                for (int w = 0; w < 4; ++w) { auto s = sink(b, w); st[b][w] = emit_mix_ark<L>(s, WX::woff(R), WY::woff(R), WZ::woff(R), sh[w], rk[b * 44 + 4 * R + w], T); }
                if (R == 9) {
                    for (int w = 0; w < 4; ++w) { auto s = sink(b, w); sub[w] = emit_sbox<L>(s, WX::woff(10), WY::woff(10), WZ::woff(10), st[b][w], T); }
                    for (int w = 0; w < 4; ++w) sh[w] = shift_rows(sub[w], sub[(w + 1) & 3], sub[(w + 2) & 3], sub[(w + 3) & 3]);
                    for (int w = 0; w < 4; ++w) {
                        auto s = sink(b, w);
                        st[b][w] = emit_final_ark<L>(s, WX::woff(10), WY::woff(10), WZ::woff(10), sh[w], rk[b * 44 + 40 + w]);
                        if (ct && b < nvalid) std::memcpy(ct + 16 * (blk0 + b) + 4 * w, &st[b][w], 4);
                    }
                }
            }
            if (G::HAS_X) flush_col<WX>(R, sx, gx, fx, nvalid);
            flush_col<WY>(R, sy, gy, fy, nvalid);
            flush_col<WZ>(R, sz, gz, fz, nvalid);
        }
    }
}

}  // namespace

extern "C" int lane_model_run(const uint8_t *tab768, const uint8_t *pt, const uint8_t *keys, int per_block_keys,
                              int key_only, uint64_t n, int layout, int xt_arith, uint8_t *x, uint8_t *y, uint8_t *z,
                              uint8_t *ct, uint8_t *wd, uint8_t *kx, uint8_t *ky, uint8_t *kz, uint8_t *rk) {
#define GO(L, XT) run<L, XT>(tab768, pt, keys, per_block_keys, key_only, n, x, y, z, ct, wd, kx, ky, kz, rk)
    if (layout == DENSE) { if (xt_arith) GO(DENSE, true); else GO(DENSE, false); }
    else if (layout == PACKED) { if (xt_arith) GO(PACKED, true); else GO(PACKED, false); }
    else if (layout == VALUES) { if (xt_arith) GO(VALUES, true); else GO(VALUES, false); }
    else return 1;
    return 0;
#undef GO
}

extern "C" void lane_model_values_mask(int col, uint8_t *enc_mask) { encrypt_values_mask(col, enc_mask); }

extern "C" void lane_model_masks(int col, uint8_t *enc_mask, uint8_t *key_mask) {
    encrypt_assigned_mask(col, enc_mask);
    key_assigned_mask(col, key_mask);
}

extern "C" int lane_model_packed_index(int key, int col, int row) {
    return key ? packed_index_key(col, row) : packed_index_enc(col, row);
}

// The device checker's own source (aesw_check.h) on the CPU: n units, images gathered exactly as check_kernel gathers them,
// 64 "lanes" one after the other.  report: the seven u64 of aesw_check_report.
extern "C" int lane_model_check(const uint8_t *tab768, int layout, const uint8_t *pt, const uint8_t *keys, int per_block_keys, uint64_t n,
                                const uint8_t *x, const uint8_t *y, const uint8_t *z, const uint8_t *ct, const uint8_t *kw, const uint8_t *kx,
                                const uint8_t *ky, const uint8_t *kz, uint64_t *report) {
    if (layout != DENSE && layout != PACKED) return 1;
    std::vector<uint32_t> t(CHK_WORDS);
    build_check_table(layout, t.data());
    const CheckGeo g = check_geo(layout);
    std::vector<uint8_t> img(g.bi + g.ki);
    CheckAcc acc;
    auto load_key = [&](uint64_t k) {
        uint8_t *ki = img.data() + g.bi;
        std::memcpy(ki, kx + k * g.kxs, g.kxs);
        std::memcpy(ki + g.kxs, ky + k * g.kys, g.kys);
        std::memcpy(ki + g.kxs + g.kys, kz + k * g.kzs, g.kzs);
        std::memcpy(ki + g.kxs + g.kys + g.kzs, kw + k * WORDS_ROWS, WORDS_ROWS);
    };
    if (!per_block_keys) {
        load_key(0);
        for (uint32_t lane = 0; lane < 64; ++lane) check_key(img.data(), t.data(), tab768, keys, 0, lane, 64, acc);
    }
    for (uint64_t b = 0; b < n; ++b) {
        std::memcpy(img.data(), x + b * g.sx, g.sx);
        std::memcpy(img.data() + g.sx, y + b * g.sy, g.sy);
        std::memcpy(img.data() + g.sx + g.sy, z + b * g.sz, g.sz);
        if (per_block_keys) load_key(b);
        for (uint32_t lane = 0; lane < 64; ++lane) {
            check_block(img.data(), t.data(), tab768, pt + 16 * b, ct ? ct + 16 * b : nullptr, b, lane, 64, acc);
            if (per_block_keys) check_key(img.data(), t.data(), tab768, keys ? keys + 16 * b : nullptr, b, lane, 64, acc);
        }
    }
    report[0] = n; report[1] = per_block_keys ? n : 1;
    report[2] = acc.lookup; report[3] = acc.copy; report[4] = acc.gate; report[5] = acc.input; report[6] = acc.first;
    return 0;
}

extern "C" int lane_model_assemble_kernel_choice(int as_fr, int geometry, uint32_t k, uint32_t col_count) {
    return assemble_kernel_choice(as_fr != 0, geometry, k, col_count);
}

// window geometry, for the tests
extern "C" void lane_model_window(int layout, int col, int out[6]) {
    auto fill = [&](auto w) {
        using W = decltype(w);
        out[0] = W::PERM_R; out[1] = W::NSLOT; out[2] = W::PERM_END; out[3] = W::TAIL0; out[4] = W::RAW; out[5] = W::BYTES;
    };
    if (layout == DENSE) { if (col == 0) fill(WinX<DENSE>{}); else if (col == 1) fill(WinY<DENSE>{}); else fill(WinZ<DENSE>{}); }
    else if (layout == VALUES) { if (col == 0) fill(WinX<VALUES>{}); else if (col == 1) fill(WinY<VALUES>{}); else fill(WinZ<VALUES>{}); }
    else { if (col == 0) fill(WinX<PACKED>{}); else if (col == 1) fill(WinY<PACKED>{}); else fill(WinZ<PACKED>{}); }
}

extern "C" int lane_model_flush_not_exactly_once(void) {
    return flush_not_exactly_once<WinX<DENSE>>() + flush_not_exactly_once<WinY<DENSE>>() + flush_not_exactly_once<WinZ<DENSE>>() +
           flush_not_exactly_once<WinX<PACKED>>() + flush_not_exactly_once<WinY<PACKED>>() + flush_not_exactly_once<WinZ<PACKED>>() +
           flush_not_exactly_once<WinY<VALUES>>() + flush_not_exactly_once<WinZ<VALUES>>();
}

// LDS bank-conflict cost (extra cycles per wave) of the flush reads: the table as built, and the same lines in plain
// address order, for the packed layout's three columns.
template <class W>
static void conflict_costs(int *built, int *sorted) {
    const ColTable<W> tab;
    *built += flush_table_conflict_cost<W>(tab.t.data());
    std::vector<uint32_t> plain(tab.t.size());
    int idx = 0;
    for (int R = 1; R <= 9; ++R) {
        std::vector<uint32_t> entries;  // the round's line slots in address order
        for (int i = sched_first<W>(R); i < sched_first<W>(R + 1); ++i)
            for (int slot = 0; slot < 8; ++slot)
                if ((tab.t[(size_t)i * 64 + 8 * slot] >> 16) != SCHED_INVALID_P) entries.push_back((uint32_t)(i * 8 + slot));
        std::sort(entries.begin(), entries.end(), [&](uint32_t a, uint32_t b) { return (tab.t[(size_t)a * 8] >> 16) < (tab.t[(size_t)b * 8] >> 16); });
        for (int i = sched_first<W>(R); i < sched_first<W>(R + 1); ++i, ++idx)
            for (int lane = 0; lane < 64; ++lane) {
                const size_t slot = (size_t)(i - sched_first<W>(R)) * 8 + (lane >> 3);
                plain[(size_t)i * 64 + lane] = slot < entries.size() ? tab.t[(size_t)entries[slot] * 8 + (lane & 7)] : (SCHED_INVALID_P << 16);
            }
    }
    *sorted += flush_table_conflict_cost<W>(plain.data());
}
extern "C" void lane_model_flush_conflict_costs(int layout, int *built, int *sorted) {
    *built = *sorted = 0;
    if (layout == PACKED) { conflict_costs<WinX<PACKED>>(built, sorted); conflict_costs<WinY<PACKED>>(built, sorted); conflict_costs<WinZ<PACKED>>(built, sorted); }
    else if (layout == DENSE) { conflict_costs<WinX<DENSE>>(built, sorted); conflict_costs<WinY<DENSE>>(built, sorted); conflict_costs<WinZ<DENSE>>(built, sorted); }
    else { conflict_costs<WinY<VALUES>>(built, sorted); conflict_costs<WinZ<VALUES>>(built, sorted); }
}

extern "C" int lane_model_flush_forms_disagree(void) {
    return flush_forms_disagree<WinX<DENSE>>() + flush_forms_disagree<WinY<DENSE>>() + flush_forms_disagree<WinZ<DENSE>>() +
           flush_forms_disagree<WinX<PACKED>>() + flush_forms_disagree<WinY<PACKED>>() + flush_forms_disagree<WinZ<PACKED>>() +
           flush_forms_disagree<WinY<VALUES>>() + flush_forms_disagree<WinZ<VALUES>>();
}
