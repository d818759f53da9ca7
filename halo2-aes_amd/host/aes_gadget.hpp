// aes_gadget.hpp -- host-side mirror of the reference's interface for the hot
// path, in C++ (no Rust toolchain in this image): same type and method names,
// argument meaning and error behaviour as
//   src/chips/{u8_xor,sbox,gf_mul,u8_range_check}_chip.rs
//   src/key_schedule.rs  (Aes128KeyScheduleConfig)
//   src/aes128.rs        (FixedAes128Config<K,N>)
//   src/table.rs         (load_enc_full_table)
// with ONE difference, the point of the whole exercise: every value closure is a
// pure read of a witness buffer that the device filled through the C ABI
// (include/aesw.h), instead of recomputing xor_bytes / sub_byte / MUL_BY_n per
// row.  configure(), selectors, lookups, region structure, copy constraints and
// the aes_callable() bookkeeping are the reference's.
#pragma once
#include <algorithm>
#include <array>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <string>

#include "../../include/aesw.h"
#include "halo2_lite.hpp"

namespace aesw {
namespace host {

// src/table.rs:10-16
enum class Tag : Fp { U8 = 1, Xor = 2, Sbox = 3, GfMul2 = 4, GfMul3 = 5 };

constexpr uint64_t AES_ROWS_ = AESW_AES_ROWS;                    // src/constant.rs:114
constexpr uint64_t KEY_SCHEDULE_ROWS_ = AESW_KEY_SCHEDULE_ROWS;  // src/constant.rs:113

// The device-computed witness of one circuit: key slab + n encrypt slabs.  The
// default layout is AESW_LAYOUT_PACKED (assigned cells only: nothing but live
// cells crosses PCIe; aesw_layout_index maps a slab row to its byte);
// AESW_LAYOUT_DENSE remains an option, AESW_LAYOUT_VALUES carries only what the
// chips' value closures read.  Computed once, read by every synthesize() pass
// (keygen_vk, keygen_pk, create_proof all re-run synthesize: SURVEY 3.1).
struct AesWitness {
    uint64_t n = 0;
    uint8_t key[16] = {0};
    std::vector<uint8_t> pt;                        // n*16
    std::vector<uint8_t> x, y, z;                   // packed: n*1360 / n*1056 / n*608; dense: n*1360 each; values only: none / n*448 / n*608
    std::vector<uint8_t> key_w, key_x, key_y, key_z;  // 96, then the key slab in the packed form (400 / 240 / 200) or dense (400 each)
    int layout = AESW_LAYOUT_PACKED;
    bool values_only = false;                       // AESW_LAYOUT_VALUES: only what the chips' value closures read
    size_t sx = AES_ROWS_, sy = AES_ROWS_, sz = AES_ROWS_;  // bytes per block
    int32_t iy[AESW_AES_ROWS], iz[AESW_AES_ROWS];   // slab row -> index in y / z (-1: the layout leaves the cell out)
    int32_t kiy[AESW_KEY_ROWS], kiz[AESW_KEY_ROWS]; // the same for the key slab
    // The blocks currently readable: the whole batch, or -- streaming -- the chunk the device has just handed over
    // (aesw_encrypt_witness_stream: the next chunk's kernel and D2H are in flight while this one is assigned).
    mutable const uint8_t *vx = nullptr, *vy = nullptr, *vz = nullptr;
    mutable uint64_t vfirst = 0, vcount = 0;

    // aesw_schedule_key + aesw_encrypt_witness (host-pointer entry points).
    // key slab of `layout` (VALUES keeps the packed key slab) + its row -> index maps
    void schedule(aesw_ctx *ctx) {
        const int kl = layout == AESW_LAYOUT_DENSE ? AESW_LAYOUT_DENSE : AESW_LAYOUT_PACKED;
        if (kl == AESW_LAYOUT_DENSE) {
            for (uint32_t r = 0; r < AESW_KEY_ROWS; ++r) kiy[r] = kiz[r] = (int32_t)r;
        } else if (aesw_key_packed_index(1, kiy) != AESW_OK || aesw_key_packed_index(2, kiz) != AESW_OK) {
            throw Error(Error::Synthesis, "aesw_key_packed_index failed");
        }
        key_w.resize(AESW_WORDS_ROWS);
        key_x.resize(aesw_key_column_stride(kl, 0)); key_y.resize(aesw_key_column_stride(kl, 1)); key_z.resize(aesw_key_column_stride(kl, 2));
        aesw_key_slab ks{key_w.data(), key_x.data(), key_y.data(), key_z.data()};
        const int rc = aesw_schedule_key(ctx, key, kl, &ks);
        if (rc != AESW_OK) throw Error(Error::Synthesis, std::string("aesw_schedule_key: ") + aesw_strerror(rc));
    }

    static std::shared_ptr<const AesWitness> generate(aesw_ctx *ctx, const uint8_t key[16], const uint8_t *pts, uint64_t n,
                                                      int layout = AESW_LAYOUT_PACKED) {
        auto w = std::make_shared<AesWitness>();
        w->n = n;
        w->layout = layout;
        w->values_only = layout == AESW_LAYOUT_VALUES;
        std::memcpy(w->key, key, 16);
        w->pt.assign(pts, pts + 16 * n);
        w->sx = aesw_column_stride(layout, 0); w->sy = aesw_column_stride(layout, 1); w->sz = aesw_column_stride(layout, 2);
        if (!w->sy || !w->sz) throw Error(Error::Synthesis, "unknown witness layout");
        w->x.resize(n * w->sx); w->y.resize(n * w->sy); w->z.resize(n * w->sz);
        if (aesw_layout_index(layout, 1, w->iy) != AESW_OK || aesw_layout_index(layout, 2, w->iz) != AESW_OK)
            throw Error(Error::Synthesis, "aesw_layout_index failed");
        w->schedule(ctx);
        int rc = AESW_OK;
        if (n)
            rc = aesw_encrypt_witness(ctx, pts, nullptr, 0, n, layout, w->sx ? w->x.data() : nullptr, w->y.data(), w->z.data(), nullptr, nullptr);
        if (rc != AESW_OK) throw Error(Error::Synthesis, std::string("device witness generation failed: ") + aesw_strerror(rc));
        w->vx = w->sx ? w->x.data() : nullptr; w->vy = w->y.data(); w->vz = w->z.data();
        w->vfirst = 0; w->vcount = n;
        return w;
    }

    // Streaming (BASELINE configs[4]): nothing is fetched up front except the key slab; stream() then runs
    // `assign(first, count)` once per chunk, in block order, while the device produces and copies the next one.
    static std::shared_ptr<const AesWitness> prepare_stream(aesw_ctx *ctx, const uint8_t key[16], const uint8_t *pts, uint64_t n) {
        auto w = std::make_shared<AesWitness>();
        w->n = n;
        w->layout = AESW_LAYOUT_VALUES;
        w->values_only = true;
        std::memcpy(w->key, key, 16);
        w->pt.assign(pts, pts + 16 * n);
        w->sx = aesw_column_stride(AESW_LAYOUT_VALUES, 0); w->sy = aesw_column_stride(AESW_LAYOUT_VALUES, 1); w->sz = aesw_column_stride(AESW_LAYOUT_VALUES, 2);
        if (aesw_layout_index(AESW_LAYOUT_VALUES, 1, w->iy) != AESW_OK || aesw_layout_index(AESW_LAYOUT_VALUES, 2, w->iz) != AESW_OK)
            throw Error(Error::Synthesis, "aesw_layout_index failed");
        w->schedule(ctx);
        return w;
    }
    void stream(aesw_ctx *ctx, const std::function<void(uint64_t, uint64_t)> &assign) const {
        struct State { const AesWitness *w; const std::function<void(uint64_t, uint64_t)> *assign; std::exception_ptr err; } st{this, &assign, nullptr};
        auto trampoline = [](void *user, uint64_t first, uint64_t count, const uint8_t *x, const uint8_t *y, const uint8_t *z) -> int {
            State *s = static_cast<State *>(user);
            s->w->vx = x; s->w->vy = y; s->w->vz = z; s->w->vfirst = first; s->w->vcount = count;
            try {
                (*s->assign)(first, count);
            } catch (...) {  // nothing may unwind through the C ABI
                s->err = std::current_exception();
                return 1;
            }
            return 0;
        };
        const int rc = n ? aesw_encrypt_witness_stream(ctx, pt.data(), nullptr, 0, n, AESW_LAYOUT_VALUES, trampoline, &st) : AESW_OK;
        vx = vy = vz = nullptr; vfirst = vcount = 0;  // the chunk buffers are gone
        if (st.err) std::rethrow_exception(st.err);
        if (rc != AESW_OK) throw Error(Error::Synthesis, std::string("aesw_encrypt_witness_stream: ") + aesw_strerror(rc));
    }
};

// Where the gadget currently is inside a slab: three column arrays, the slab-row -> index maps of the
// layout (column x has every row in the layouts that carry it) and a row.  With the values-only witness x is
// absent and y holds only S-box / mul outputs: the copied cells have nothing to be checked against, exactly
// as in the reference, where copy_advice() takes the source cell's value.
struct WitnessCursor {
    const uint8_t *x = nullptr, *y = nullptr, *z = nullptr;
    uint64_t row = 0, rows = 0;
    const int32_t *iy = nullptr, *iz = nullptr;
    const uint8_t *pt = nullptr;  // the block's plaintext literal (src/aes128.rs:187), used when x is absent
    uint8_t yv(uint64_t r) const { return y[iy[r]]; }
    uint8_t zv(uint64_t r) const { return z[iz[r]]; }
    void expect_x(uint64_t r, Fp v, const char *what) const {
        // copy_advice carries the source cell's value; the device's x/y byte for the same cell must agree
        if (!x) return;
        if (r >= rows || x[r] != v) mismatch(what, r, r < rows ? x[r] : -1, v);
    }
    void expect_y(uint64_t r, Fp v, const char *what) const {
        if (r >= rows) mismatch(what, r, -1, v);
        if (iy[r] < 0) return;  // values-only: the y cell of an xor row is not shipped
        if (y[iy[r]] != v) mismatch(what, r, y[iy[r]], v);
    }
    [[noreturn]] void mismatch(const char *what, uint64_t r, int dev, Fp host) const {
        throw Error(Error::Mismatch, std::string("device witness disagrees with copied value: ") + what + " (slab row " + std::to_string(r) + " of " +
                                         std::to_string(rows) + ": device " + std::to_string(dev) + ", host " + std::to_string((unsigned long long)host) + ")");
    }
};

// ---- src/chips/u8_range_check_chip.rs ---------------------------------------
struct U8RangeCheckConfig { Column x; Selector q; };
class U8RangeCheckChip {
public:
    static U8RangeCheckChip construct(U8RangeCheckConfig c, WitnessCursor *cur) { return U8RangeCheckChip{c, cur}; }
    static U8RangeCheckConfig configure(ConstraintSystem &meta, Column x_col, Selector selector, TableColumn tag_tab, TableColumn value_tab) {
        meta.lookup(Lookup{"Range check u8 value", selector, (Fp)Tag::U8, {x_col}, {tag_tab, value_tab}});  // :34-42
        return U8RangeCheckConfig{x_col, selector};
    }
    void range_check(Layouter &layouter, const AssignedCell &x) const {  // :51-70
        const uint64_t r = cur->row;
        layouter.assign_region<int>("", [&](Region &region) {
            region.enable_selector(config.q, 0);
            x.copy_advice(region, config.x, 0);
            return 0;
        });
        if (x.val.known) cur->expect_x(r, x.val.v, "range_check x");
        cur->row++;
    }
    U8RangeCheckConfig config;
    WitnessCursor *cur;
};

// ---- src/chips/u8_xor_chip.rs ------------------------------------------------
struct U8XorConfig { Column x, y, z; Selector q; };
class U8XorChip {
public:
    static U8XorChip construct(U8XorConfig c, WitnessCursor *cur) { return U8XorChip{c, cur}; }
    static U8XorConfig configure(ConstraintSystem &meta, Column x_col, Column y_col, Column z_col, Selector selector,
                                 TableColumn tag_tab, TableColumn x_tab, TableColumn y_tab, TableColumn z_tab) {
        meta.lookup(Lookup{"Check correct XOR of u8 values", selector, (Fp)Tag::Xor, {x_col, y_col, z_col}, {tag_tab, x_tab, y_tab, z_tab}});  // :41-53
        return U8XorConfig{x_col, y_col, z_col, selector};
    }
    AssignedCell xor_(Layouter &layouter, const AssignedCell &x, const AssignedCell &y) const {  // :63-100
        const uint64_t r = cur->row;
        const WitnessCursor *c = cur;
        AssignedCell z = layouter.assign_region<AssignedCell>("", [&](Region &region) {
            region.enable_selector(config.q, 0);
            x.copy_advice(region, config.x, 0);
            y.copy_advice(region, config.y, 0);
            // was: xor_bytes(x_copied.value, y_copied.value) (:85-95) -- now the device's byte
            return region.assign_advice(config.z, 0, [c, r] { return Value::of(c->zv(r)); });
        });
        if (x.val.known) cur->expect_x(r, x.val.v, "xor x");
        if (y.val.known) cur->expect_y(r, y.val.v, "xor y");
        cur->row++;
        return z;
    }
    U8XorConfig config;
    WitnessCursor *cur;
};

// ---- src/chips/sbox_chip.rs ---------------------------------------------------
struct SboxConfig { Column x, y; Selector q; };
class SboxChip {
public:
    static SboxChip construct(SboxConfig c, WitnessCursor *cur) { return SboxChip{c, cur}; }
    static SboxConfig configure(ConstraintSystem &meta, Column x_col, Column y_col, Selector selector, TableColumn tag_tab,
                                TableColumn x_tab, TableColumn y_tab) {
        meta.lookup(Lookup{"Check correct Sbox substitution", selector, (Fp)Tag::Sbox, {x_col, y_col}, {tag_tab, x_tab, y_tab}});  // :38-48
        return SboxConfig{x_col, y_col, selector};
    }
    AssignedCell substitute(Layouter &layouter, const AssignedCell &x) const {  // :57-83
        const uint64_t r = cur->row;
        const WitnessCursor *c = cur;
        AssignedCell y = layouter.assign_region<AssignedCell>("", [&](Region &region) {
            region.enable_selector(config.q, 0);
            x.copy_advice(region, config.x, 0);
            // was: sub_byte(x_copied.value) (:73-78)
            return region.assign_advice(config.y, 0, [c, r] { return Value::of(c->yv(r)); });
        });
        if (x.val.known) cur->expect_x(r, x.val.v, "sbox x");
        cur->row++;
        return y;
    }
    SboxConfig config;
    WitnessCursor *cur;
};

// ---- src/chips/gf_mul_chip.rs (macro instantiated for 2 and 3) -----------------
struct MulConfig { Column x, y; Selector q; };
template <int N>
class MulByChip {
public:
    static MulByChip construct(MulConfig c, WitnessCursor *cur) { return MulByChip{c, cur}; }
    static MulConfig configure(ConstraintSystem &meta, Column x_col, Column y_col, Selector selector, TableColumn tag_tab,
                               TableColumn x_tab, TableColumn y_tab) {
        meta.lookup(Lookup{N == 2 ? "Check correct gf mul by 2" : "Check correct gf mul by 3", selector,
                           (Fp)(N == 2 ? Tag::GfMul2 : Tag::GfMul3), {x_col, y_col}, {tag_tab, x_tab, y_tab}});  // :38-48
        return MulConfig{x_col, y_col, selector};
    }
    AssignedCell mul(Layouter &layouter, const AssignedCell &x) const {  // :59-89
        const uint64_t r = cur->row;
        const WitnessCursor *c = cur;
        AssignedCell y = layouter.assign_region<AssignedCell>("", [&](Region &region) {
            region.enable_selector(config.q, 0);
            x.copy_advice(region, config.x, 0);
            // was: Fp::from($dict[x]) (:75-84)
            return region.assign_advice(config.y, 0, [c, r] { return Value::of(c->yv(r)); });
        });
        if (x.val.known) cur->expect_x(r, x.val.v, "gf mul x");
        cur->row++;
        return y;
    }
    MulConfig config;
    WitnessCursor *cur;
};
using MulBy2Chip = MulByChip<2>;
using MulBy3Chip = MulByChip<3>;

// ---- src/table.rs:18-192 --------------------------------------------------------
// The four table columns come from the device (aesw_lookup_table), 66561 rows.
inline void load_enc_full_table(Layouter &layouter, const TableColumn tables[4], aesw_ctx *ctx) {
    std::vector<uint8_t> t[4];
    for (auto &c : t) c.resize(AESW_TABLE_ROWS);
    const int rc = aesw_lookup_table(ctx, t[0].data(), t[1].data(), t[2].data(), t[3].data());
    if (rc != AESW_OK) throw Error(Error::Synthesis, std::string("aesw_lookup_table: ") + aesw_strerror(rc));
    layouter.assign_table("Assign full table", [&](Table &table) {
        for (int c = 0; c < 4; ++c)
            for (uint64_t row = 0; row < AESW_TABLE_ROWS; ++row) table.assign_cell(tables[c], row, t[c][row]);
    });
}

// ---- src/key_schedule.rs ----------------------------------------------------------
class Aes128KeyScheduleConfig {
public:
    // :40-77
    static Aes128KeyScheduleConfig configure(ConstraintSystem &meta, const Column advices[3], U8XorConfig u8_xor_config,
                                             SboxConfig sbox_config, U8RangeCheckConfig u8_range_check_config) {
        Aes128KeyScheduleConfig c;
        c.words_column = meta.advice_column();
        c.round_constants = meta.fixed_column();
        c.q_eq_rcon = meta.selector();
        for (int i = 0; i < 3; ++i) meta.enable_equality(advices[i]);
        meta.enable_equality(c.words_column);
        meta.enable_constant(c.round_constants);
        meta.create_gate(Gate{"Equality RC", c.q_eq_rcon, c.words_column, c.round_constants});  // q * (x - c), :59-64
        c.u8_range_check_config = u8_range_check_config;
        c.u8_xor_config = u8_xor_config;
        c.sbox_config = sbox_config;
        return c;
    }

    void attach_witness(std::shared_ptr<const AesWitness> w) { wit = std::move(w); }

    // :80-96
    std::vector<std::vector<AssignedCell>> schedule_keys(Layouter &layouter, const uint8_t key[16]) {
        if (!wit) throw Error(Error::Synthesis, "no device witness attached");
        if (std::memcmp(key, wit->key, 16) != 0) throw Error(Error::Mismatch, "key differs from the key the device witness was generated for");
        chip_cur = WitnessCursor{wit->key_x.data(), wit->key_y.data(), wit->key_z.data(), 0, AESW_KEY_ROWS, wit->kiy, wit->kiz};
        words_row = 0;
        std::vector<std::vector<AssignedCell>> words;
        std::vector<AssignedCell> round = assign_first_round(layouter);
        words.push_back(round);
        for (uint32_t i = 1; i <= 10; ++i) {
            round = assign_round(layouter, i, round);
            words.push_back(round);
        }
        return words;
    }

    Column words_column, round_constants;
    Selector q_eq_rcon;
    U8RangeCheckConfig u8_range_check_config;
    U8XorConfig u8_xor_config;
    SboxConfig sbox_config;

private:
    // :98-118 -- was Fp::from(byte as u64) of the key literal (:112), now words_column's device bytes
    std::vector<AssignedCell> assign_first_round(Layouter &layouter) {
        const uint8_t *w = wit->key_w.data();
        auto out = layouter.assign_region<std::vector<AssignedCell>>("Assign first four words", [&](Region &region) {
            std::vector<AssignedCell> words;
            for (uint64_t i = 0; i < 16; ++i) words.push_back(region.assign_advice(words_column, i, [w, i] { return Value::of(w[i]); }));
            return words;
        });
        words_row = 16;
        return out;
    }

    // :122-224
    std::vector<AssignedCell> assign_round(Layouter &layouter, uint32_t round, const std::vector<AssignedCell> &prev_round_bytes) {
        const U8XorChip xor_chip = U8XorChip::construct(u8_xor_config, &chip_cur);
        const SboxChip sbox_chip = SboxChip::construct(sbox_config, &chip_cur);
        const U8RangeCheckChip range_chip = U8RangeCheckChip::construct(u8_range_check_config, &chip_cur);
        const uint8_t *w = wit->key_w.data();
        // :141-154 shift previous round: copy bytes 13,14,15,12 into words_column
        const uint64_t r0 = words_row;
        auto shifted = layouter.assign_region<std::vector<AssignedCell>>("shift previous round", [&](Region &region) {
            static const size_t rot[4] = {13, 14, 15, 12};
            std::vector<AssignedCell> v;
            for (uint64_t i = 0; i < 4; ++i) v.push_back(prev_round_bytes[rot[i]].copy_advice(region, words_column, i));
            return v;
        });
        for (int i = 0; i < 4; ++i)
            if (shifted[i].val.known && w[r0 + i] != shifted[i].val.v) throw Error(Error::Mismatch, "words_column shift bytes");
        words_row += 4;
        // :156-159
        std::vector<AssignedCell> subbed;
        for (const auto &b : shifted) subbed.push_back(sbox_chip.substitute(layouter, b));
        // :161-187 "Assign rc": the rcon and three zero pads, read from the device's words_column bytes
        const uint64_t r1 = words_row;
        const Fp rc = ROUND_CONSTANT(round - 1);  // fixed column value stays a host constant (src/utils.rs:28)
        auto rc_assigned = layouter.assign_region<std::vector<AssignedCell>>("Assign rc", [&](Region &region) {
            std::vector<AssignedCell> res;
            region.enable_selector(q_eq_rcon, 0);
            region.assign_fixed(round_constants, 0, [rc] { return Value::of(rc); });
            for (uint64_t i = 0; i < 4; ++i) res.push_back(region.assign_advice(words_column, i, [w, r1, i] { return Value::of(w[r1 + i]); }));
            return res;
        });
        words_row += 4;
        // :189-194
        std::vector<AssignedCell> rconned;
        for (int i = 0; i < 4; ++i) rconned.push_back(xor_chip.xor_(layouter, subbed[i], rc_assigned[i]));
        // :197-204
        std::vector<AssignedCell> next_word, words;
        for (int i = 0; i < 4; ++i) next_word.push_back(xor_chip.xor_(layouter, prev_round_bytes[i], rconned[i]));
        words = next_word;
        // :207-216
        for (int i = 1; i < 4; ++i) {
            std::vector<AssignedCell> nw;
            for (int j = 0; j < 4; ++j) nw.push_back(xor_chip.xor_(layouter, prev_round_bytes[i * 4 + j], next_word[j]));
            next_word = nw;
            words.insert(words.end(), nw.begin(), nw.end());
        }
        // :218-221
        for (const auto &b : words) range_chip.range_check(layouter, b);
        return words;
    }

    static Fp ROUND_CONSTANT(uint32_t i) {  // src/utils.rs:28
        static const Fp rc[10] = {1, 2, 4, 8, 16, 32, 64, 128, 27, 54};
        return rc[i];
    }

    std::shared_ptr<const AesWitness> wit;
    WitnessCursor chip_cur;
    uint64_t words_row = 0;
};

// ---- src/aes128.rs -------------------------------------------------------------------
// FixedAes128Config<K, N>: K and N are runtime members here (const generics in the reference).
class FixedAes128Config {
public:
    // :46-141
    static FixedAes128Config configure(ConstraintSystem &meta, uint32_t K, uint32_t N) {
        FixedAes128Config c;
        c.K = K;
        c.N = N;
        for (int i = 0; i < 4; ++i) c.tables[i] = meta.lookup_table_column();  // first is the tag column
        c.advices.resize(N);
        for (uint32_t i = 0; i < N; ++i)
            for (int j = 0; j < 3; ++j) c.advices[i][j] = meta.advice_column();
        for (uint32_t i = 0; i < N; ++i) {
            const Selector q_u8_range_check = meta.complex_selector();
            const Selector q_u8_xor = meta.complex_selector();
            const Selector q_sbox = meta.complex_selector();
            const Selector q_mul_by_2 = meta.complex_selector();
            const Selector q_mul_by_3 = meta.complex_selector();
            const auto &a = c.advices[i];
            c.range_configs.push_back(U8RangeCheckChip::configure(meta, a[0], q_u8_range_check, c.tables[0], c.tables[1]));
            c.xor_configs.push_back(U8XorChip::configure(meta, a[0], a[1], a[2], q_u8_xor, c.tables[0], c.tables[1], c.tables[2], c.tables[3]));
            c.sbox_configs.push_back(SboxChip::configure(meta, a[0], a[1], q_sbox, c.tables[0], c.tables[1], c.tables[2]));
            c.mul2_configs.push_back(MulBy2Chip::configure(meta, a[0], a[1], q_mul_by_2, c.tables[0], c.tables[1], c.tables[2]));
            c.mul3_configs.push_back(MulBy3Chip::configure(meta, a[0], a[1], q_mul_by_3, c.tables[0], c.tables[1], c.tables[2]));
        }
        c.key_schedule_config = Aes128KeyScheduleConfig::configure(meta, c.advices[0].data(), c.xor_configs[0], c.sbox_configs[0], c.range_configs[0]);
        for (const auto &v : c.advices)
            for (const auto &col : v) meta.enable_equality(col);
        return c;
    }

    void attach_witness(std::shared_ptr<const AesWitness> w) {
        wit = w;
        key_schedule_config.attach_witness(std::move(w));
    }

    // :143-152
    void schedule_key(Layouter &layouter, const uint8_t key[16]) {
        keys = key_schedule_config.schedule_keys(layouter, key);
        have_keys = true;
    }

    // Bulk assignment: after the first block has gone through the reference's
    // 1 360 one-row regions, every later block is ONE region of 1 360 rows whose cells are the device
    // bytes, whose selectors come from aesw_selector_tags() and whose copy constraints replay the
    // first block's (block-relative) copy graph.  Same cells, selectors and equality constraints
    // as the per-region path (tests compare them), without 1 360 shape+assign double passes per block.
    bool bulk_assign = false;

    // :154-265
    std::vector<AssignedCell> encrypt(Layouter &layouter, const uint8_t plaintext[16]) {
        if (!aes_callable()) throw Panic(Panic::Capacity, "AES calls too many. doesn't fit in the rows");  // :159-162
        count += 1;
        if (!have_keys) throw Panic(Panic::NoKey, "Keys should be scheduled");  // :170
        if (!wit) throw Error(Error::Synthesis, "no device witness attached");
        const uint64_t b = total++;  // the b-th encrypt() call reads slab b
        if (b >= wit->n) throw Error(Error::Mismatch, "more encrypt() calls than blocks in the device witness");
        if (std::memcmp(plaintext, wit->pt.data() + 16 * b, 16) != 0) throw Error(Error::Mismatch, "plaintext differs from the device witness's block");
        if (b < wit->vfirst || b >= wit->vfirst + wit->vcount) throw Error(Error::Mismatch, "encrypt() call outside the blocks the device has handed over");
        const uint64_t vb = b - wit->vfirst;
        cur = WitnessCursor{wit->vx ? wit->vx + vb * wit->sx : nullptr, wit->vy + vb * wit->sy, wit->vz + vb * wit->sz, 0, AES_ROWS_,
                            wit->iy, wit->iz, wit->pt.data() + 16 * b};
        if (bulk_assign && wit->values_only) throw Error(Error::Synthesis, "bulk assignment needs whole columns, not the values-only witness");
        if (bulk_assign && graph_ready) return encrypt_bulk(layouter);
        const size_t copies_before = layouter.copies().size();
        const uint64_t block_start = layouter.column_height(get_advices()[0]);
        std::vector<AssignedCell> out = encrypt_regions(layouter);
        if (bulk_assign) record_copy_graph(layouter, copies_before, block_start);
        return out;
    }

private:
    struct CopyEdge {
        uint8_t dst_col;   // 0..2 within the set
        uint16_t dst_row;  // block-relative
        bool from_key;     // source is a round-key cell (round, idx) instead of a block cell
        uint8_t src_col;
        uint16_t src_row;
        uint8_t key_round, key_idx;
    };
    std::vector<CopyEdge> graph;
    bool graph_ready = false;

    void record_copy_graph(const Layouter &layouter, size_t first, uint64_t block_start) {
        const Column *adv = get_advices();
        auto rel_col = [&](const Cell &c) -> int {
            for (int j = 0; j < 3; ++j)
                if (c.column.type == Any::Advice && c.column.index == adv[j].index) return j;
            return -1;
        };
        graph.clear();
        const auto &copies = layouter.copies();
        for (size_t i = first; i < copies.size(); ++i) {
            const Cell &dst = copies[i].first, &src = copies[i].second;
            CopyEdge e{};
            const int dc = rel_col(dst);
            if (dc < 0 || dst.row < block_start || dst.row >= block_start + AES_ROWS_) throw Error(Error::Synthesis, "copy target outside the block");
            e.dst_col = (uint8_t)dc;
            e.dst_row = (uint16_t)(dst.row - block_start);
            const int sc = rel_col(src);
            if (sc >= 0 && src.row >= block_start && src.row < block_start + AES_ROWS_) {
                e.from_key = false;
                e.src_col = (uint8_t)sc;
                e.src_row = (uint16_t)(src.row - block_start);
            } else {
                bool found = false;
                for (size_t r = 0; r < keys.size() && !found; ++r)
                    for (size_t j = 0; j < keys[r].size() && !found; ++j) {
                        const Cell &kc = keys[r][j].cell;
                        if (kc.column.type == src.column.type && kc.column.index == src.column.index && kc.row == src.row) {
                            e.from_key = true;
                            e.key_round = (uint8_t)r;
                            e.key_idx = (uint8_t)j;
                            found = true;
                        }
                    }
                if (!found) throw Error(Error::Synthesis, "copy source is neither a block cell nor a round-key cell");
            }
            graph.push_back(e);
        }
        // fixed per-row data of a block: which cells exist, which selector is on
        for (int c = 0; c < 3; ++c) aesw_packed_index(c, pidx[c]);
        aesw_selector_tags(tags, nullptr, nullptr, nullptr);
        graph_ready = true;
    }

    std::vector<AssignedCell> encrypt_bulk(Layouter &layouter) {
        const Column *adv = get_advices();
        const WitnessCursor *c = &cur;
        const Selector sel_of_tag[6] = {Selector{}, range_configs.at(current).q, xor_config().q, sbox_config().q,
                                        mul2_config().q, mul3_config().q};
        std::vector<AssignedCell> out = layouter.assign_region<std::vector<AssignedCell>>("AES block (bulk)", [&](Region &region) {
            std::vector<Cell> cells[3];
            for (auto &v : cells) v.resize(AES_ROWS_);
            std::vector<AssignedCell> ct;
            for (uint64_t r = 0; r < AES_ROWS_; ++r) {
                if (tags[r]) region.enable_selector(sel_of_tag[tags[r]], r);
                cells[0][r] = region.assign_advice(adv[0], r, [c, r] { return Value::of(c->x[r]); }).cell;
                if (pidx[1][r] >= 0) cells[1][r] = region.assign_advice(adv[1], r, [c, r] { return Value::of(c->yv(r)); }).cell;
                if (pidx[2][r] >= 0) {
                    AssignedCell z = region.assign_advice(adv[2], r, [c, r] { return Value::of(c->zv(r)); });
                    cells[2][r] = z.cell;
                    if (r >= AES_ROWS_ - 16) ct.push_back(z);
                }
            }
            for (const CopyEdge &e : graph) {
                const Cell &dst = cells[e.dst_col][e.dst_row];
                const Cell &src = e.from_key ? keys[e.key_round][e.key_idx].cell : cells[e.src_col][e.src_row];
                region.constrain_equal(dst, src);
            }
            return ct;
        });
        cur.row = AES_ROWS_;
        return out;
    }

    int32_t pidx[3][AESW_AES_ROWS];
    uint8_t tags[AESW_AES_ROWS];

    // the reference's body of encrypt(): :176-265
    std::vector<AssignedCell> encrypt_regions(Layouter &layouter) {

        const U8XorChip xor_chip = U8XorChip::construct(xor_config(), &cur);
        const SboxChip sbox_chip = SboxChip::construct(sbox_config(), &cur);
        const Column *adv = get_advices();
        const WitnessCursor *c = &cur;

        // :176-192 "Assign plaintext" -- was Fp::from(p as u64) of the literal (:187)
        auto assigned_plaintext = layouter.assign_region<std::vector<AssignedCell>>("Assign plaintext", [&](Region &region) {
            std::vector<AssignedCell> v;
            for (uint64_t i = 0; i < 16; ++i) v.push_back(region.assign_advice(adv[0], i, [c, i] { return Value::of(c->x ? c->x[i] : c->pt[i]); }));
            return v;
        });
        cur.row = 16;
        // :194-198
        std::vector<AssignedCell> prev_round;
        for (int i = 0; i < 16; ++i) prev_round.push_back(xor_chip.xor_(layouter, assigned_plaintext[i], keys[0][i]));

        static const uint32_t matrix[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};  // :228-233
        for (int no_round = 1; no_round < 11; ++no_round) {
            std::vector<std::vector<AssignedCell>> subbed(4);  // :203-209
            for (int i = 0; i < 16; ++i) subbed[i / 4].push_back(sbox_chip.substitute(layouter, prev_round[i]));
            std::vector<std::vector<AssignedCell>> shifted(4);  // :216-223
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) shifted[i].push_back(subbed[(i + j) % 4][j]);
            std::vector<std::vector<AssignedCell>> mixed;  // :236-248
            if (no_round == 10) {
                mixed = shifted;
            } else {
                for (const auto &word : shifted) {
                    std::vector<AssignedCell> col;
                    for (const auto &coeffs : matrix) col.push_back(lcon(layouter, word, coeffs));
                    mixed.push_back(col);
                }
            }
            std::vector<AssignedCell> next;  // :250-261
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) next.push_back(xor_chip.xor_(layouter, mixed[i][j], keys[no_round][i * 4 + j]));
            prev_round = next;
        }
        if (cur.row != AES_ROWS_) throw Error(Error::Synthesis, "a block must use exactly AES_ROWS rows");
        return prev_round;
    }

public:
    Aes128KeyScheduleConfig key_schedule_config;
    std::vector<std::array<Column, 3>> advices;
    TableColumn tables[4];
    uint32_t K = 0, N = 0;

private:
    // :268-301
    AssignedCell lcon(Layouter &layouter, const std::vector<AssignedCell> &word, const uint32_t coeffs[4]) {
        const U8XorChip xor_chip = U8XorChip::construct(xor_config(), &cur);
        const MulBy2Chip mul2_chip = MulBy2Chip::construct(mul2_config(), &cur);
        const MulBy3Chip mul3_chip = MulBy3Chip::construct(mul3_config(), &cur);
        const Column *adv = get_advices();
        std::vector<AssignedCell> tmp;
        for (int t = 0; t < 4; ++t) {
            switch (coeffs[t]) {
            case 1: {  // :279-288 just copy advice from word
                const uint64_t r = cur.row;
                tmp.push_back(layouter.assign_region<AssignedCell>("", [&](Region &region) { return word[t].copy_advice(region, adv[0], 0); }));
                if (word[t].val.known) cur.expect_x(r, word[t].val.v, "lcon copy");
                cur.row++;
                break;
            }
            case 2: tmp.push_back(mul2_chip.mul(layouter, word[t])); break;
            case 3: tmp.push_back(mul3_chip.mul(layouter, word[t])); break;
            default: throw Panic(Panic::Other, "col should be 1, 2, or 3.");  // :294
            }
        }
        const AssignedCell inter_1 = xor_chip.xor_(layouter, tmp[0], tmp[1]);
        const AssignedCell inter_2 = xor_chip.xor_(layouter, tmp[2], tmp[3]);
        return xor_chip.xor_(layouter, inter_1, inter_2);
    }

    // :303-325
    bool aes_callable() {
        uint64_t max_row = (uint64_t)1 << K;
        if (current == 0) max_row -= KEY_SCHEDULE_ROWS_;
        if (max_row >= count * AES_ROWS_ + AES_ROWS_) return true;
        if (current < N - 1) {
            current += 1;
            count = 0;
            return true;
        }
        return false;
    }
    // :328-356 config getters
    U8XorConfig xor_config() const { return xor_configs.at(current); }
    SboxConfig sbox_config() const { return sbox_configs.at(current); }
    MulConfig mul2_config() const { return mul2_configs.at(current); }
    MulConfig mul3_config() const { return mul3_configs.at(current); }
    const Column *get_advices() const { return advices.at(current).data(); }

    std::vector<U8RangeCheckConfig> range_configs;
    std::vector<U8XorConfig> xor_configs;
    std::vector<SboxConfig> sbox_configs;
    std::vector<MulConfig> mul2_configs, mul3_configs;
    std::vector<std::vector<AssignedCell>> keys;
    bool have_keys = false;
    uint32_t current = 0;  // which column set is in use
    uint64_t count = 0;    // AES calls in the current set
    uint64_t total = 0;    // AES calls overall = slab index
    std::shared_ptr<const AesWitness> wit;
    WitnessCursor cur;
};

}  // namespace host
}  // namespace aesw
