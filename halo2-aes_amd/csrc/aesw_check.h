// aesw_check.h -- MockProver::assert_satisfied for witness SLABS: what the reference's only executable correctness checks
// verify (src/aes128.rs:409-418, src/key_schedule.rs:385-392: every enabled lookup has a table row, the "Equality RC" gate
// holds, every copy_advice() pair is equal), restated per block slab and per key slab so that a whole batch can be checked
// where it lies.  Shared by the device kernel (aesw_kernels.hip: check_kernel) and the CPU lane model (tests/lane_model/),
// which runs exactly this code against the oracle's verifier.
//
// A UNIT is one block (with the key slab its AddRoundKey rows copy from) or one key slab.  Its cells are gathered into one
// byte IMAGE -- block columns x | y | z, then key columns kx | ky | kz | words_column -- and every check is a table entry
// holding 16-bit offsets into that image, built once per layout on the host (build_check_table):
//   row entry   (2 words)  ox | oy << 16,  oz | tag << 16        tag = src/table.rs Tag of the lookup enabled on the row
//   edge entry  (1 word)   dst | src << 16                        one copy_advice(): aesw_layout.h block_copy_graph / key_copy_graph
//   gate entry  (1 word)   ow | rcon << 16 | enabled << 24        q_eq_rcon * (words - fixed), src/key_schedule.rs:59-64
#pragma once
#include "aesw_layout.h"

namespace aesw {

constexpr uint32_t CHECK_NONE = 0xffffu;  // offset of a cell the layout does not hold (never referenced by an enabled check)
// table layout in 32-bit words
constexpr int CHK_ROWS = 0;                               // AES_ROWS x 2
constexpr int CHK_EDGES = CHK_ROWS + 2 * AES_ROWS;        // BLOCK_COPIES
constexpr int CHK_KROWS = CHK_EDGES + BLOCK_COPIES;       // KEY_ROWS x 2
constexpr int CHK_KEDGES = CHK_KROWS + 2 * KEY_ROWS;      // KEY_COPIES
constexpr int CHK_GATES = CHK_KEDGES + KEY_COPIES;        // WORDS_ROWS
constexpr int CHK_WORDS = CHK_GATES + WORDS_ROWS;         // 6208 words = 24.25 KiB
// failure kinds
enum { CHK_LOOKUP = 1, CHK_COPY = 2, CHK_GATE = 3, CHK_INPUT = 4 };

struct CheckGeo {
    uint32_t sx, sy, sz, kxs, kys, kzs;  // bytes per block / per key of the layout
    uint32_t bi, ki;                     // block image bytes, key image bytes (kx | ky | kz | words)
};

inline CheckGeo check_geo(int layout) {
    CheckGeo g;
    if (layout == DENSE) { g.sx = g.sy = g.sz = AES_ROWS; g.kxs = g.kys = g.kzs = KEY_ROWS; }
    else { g.sx = Geo<PACKED>::XS; g.sy = Geo<PACKED>::YS; g.sz = Geo<PACKED>::ZS; g.kxs = Geo<PACKED>::KXS; g.kys = Geo<PACKED>::KYS; g.kzs = Geo<PACKED>::KZS; }
    g.bi = g.sx + g.sy + g.sz;
    g.ki = g.kxs + g.kys + g.kzs + WORDS_ROWS;
    return g;
}

// The check table of a layout (DENSE or PACKED), CHK_WORDS words.
inline void build_check_table(int layout, uint32_t *t) {
    const CheckGeo g = check_geo(layout);
    const bool packed = layout != DENSE;
    auto off = [&](int space, int col, int row) -> uint32_t {  // image offset of a cell, CHECK_NONE if the layout leaves it out
        if (space == 0) {
            const int i = packed ? packed_index_enc(col, row) : row;
            if (i < 0) return CHECK_NONE;
            return (col == 0 ? 0 : col == 1 ? g.sx : g.sx + g.sy) + (uint32_t)i;
        }
        if (space == 1) {
            const int i = packed ? packed_index_key(col, row) : row;
            if (i < 0) return CHECK_NONE;
            return g.bi + (col == 0 ? 0 : col == 1 ? g.kxs : g.kxs + g.kys) + (uint32_t)i;
        }
        return g.bi + g.kxs + g.kys + g.kzs + (uint32_t)row;
    };
    uint8_t etag[AES_ROWS], ktag[KEY_ROWS], q[WORDS_ROWS], rc[WORDS_ROWS];
    encrypt_selector_tags(etag);
    key_selector_tags(ktag, q, rc);
    for (int r = 0; r < AES_ROWS; ++r) {
        t[CHK_ROWS + 2 * r] = off(0, 0, r) | off(0, 1, r) << 16;
        t[CHK_ROWS + 2 * r + 1] = off(0, 2, r) | (uint32_t)etag[r] << 16;
    }
    CopyEdge be[BLOCK_COPIES], ke[KEY_COPIES];
    block_copy_graph(be);
    key_copy_graph(ke);
    for (int i = 0; i < BLOCK_COPIES; ++i)
        t[CHK_EDGES + i] = off(be[i].dst_space, be[i].dst_col, be[i].dst_row) | off(be[i].src_space, be[i].src_col, be[i].src_row) << 16;
    for (int r = 0; r < KEY_ROWS; ++r) {
        t[CHK_KROWS + 2 * r] = off(1, 0, r) | off(1, 1, r) << 16;
        t[CHK_KROWS + 2 * r + 1] = off(1, 2, r) | (uint32_t)ktag[r] << 16;
    }
    for (int i = 0; i < KEY_COPIES; ++i)
        t[CHK_KEDGES + i] = off(ke[i].dst_space, ke[i].dst_col, ke[i].dst_row) | off(ke[i].src_space, ke[i].src_col, ke[i].src_row) << 16;
    for (int r = 0; r < WORDS_ROWS; ++r) t[CHK_GATES + r] = off(2, 0, r) | (uint32_t)rc[r] << 16 | (uint32_t)q[r] << 24;
}

// What one lane found: counts per kind and the smallest failure key it saw.
// key = unit << 20 | is_key_slab << 19 | kind << 16 | index (row, or edge number): the report's `first` is the minimum.
struct CheckAcc {
    uint32_t lookup = 0, copy = 0, gate = 0, input = 0;
    uint64_t first = ~0ull;
    AESW_HD void hit(int kind, uint64_t unit, int is_key, uint32_t index) {
        if (kind == CHK_LOOKUP) ++lookup; else if (kind == CHK_COPY) ++copy; else if (kind == CHK_GATE) ++gate; else ++input;
        const uint64_t k = unit << 20 | (uint64_t)is_key << 19 | (uint64_t)kind << 16 | index;
        if (k < first) first = k;
    }
};

// One lookup row: is (tag, x, y, z) a row of the table load_enc_full_table() builds (src/table.rs:27-187)?  tab768 = S_BOX | MUL_BY_2 | MUL_BY_3.
AESW_HD bool check_row_ok(const uint8_t *img, const uint8_t *tab768, uint32_t w0, uint32_t w1) {
    const uint32_t tag = w1 >> 16;
    if (tag < 2) return true;  // 0: no lookup enabled; 1: U8 range -- every byte is in range
    const uint32_t x = img[w0 & 0xffffu], y = img[w0 >> 16];
    if (tag == 2) return img[w1 & 0xffffu] == (x ^ y);
    return y == tab768[(tag - 3) * 256 + x];  // 3 Sbox, 4 GfMul2, 5 GfMul3
}

// The checks of one block (lanes lane, lane + nlanes, ... of each loop).  pt / ct: the block's 16 input / output bytes (ct may be null).
AESW_HD void check_block(const uint8_t *img, const uint32_t *t, const uint8_t *tab768, const uint8_t *pt, const uint8_t *ct, uint64_t unit,
                         uint32_t lane, uint32_t nlanes, CheckAcc &acc) {
    for (uint32_t r = lane; r < (uint32_t)AES_ROWS; r += nlanes)
        if (!check_row_ok(img, tab768, t[CHK_ROWS + 2 * r], t[CHK_ROWS + 2 * r + 1])) acc.hit(CHK_LOOKUP, unit, 0, r);
    for (uint32_t e = lane; e < (uint32_t)BLOCK_COPIES; e += nlanes) {
        const uint32_t d = t[CHK_EDGES + e];
        if (img[d & 0xffffu] != img[d >> 16]) acc.hit(CHK_COPY, unit, 0, e);
    }
    for (uint32_t i = lane; i < 16; i += nlanes) {  // the literals: plaintext rows (src/aes128.rs:176-192) and, when given, the ciphertext
        if (img[t[CHK_ROWS + 2 * i] & 0xffffu] != pt[i]) acc.hit(CHK_INPUT, unit, 0, i);
        if (ct && img[t[CHK_ROWS + 2 * (1344 + i) + 1] & 0xffffu] != ct[i]) acc.hit(CHK_INPUT, unit, 0, 1344 + i);
    }
}

// The checks of one key slab.  key: its 16 key bytes (may be null: the literal rows of words_column are then not compared).
AESW_HD void check_key(const uint8_t *img, const uint32_t *t, const uint8_t *tab768, const uint8_t *key, uint64_t unit, uint32_t lane,
                       uint32_t nlanes, CheckAcc &acc) {
    for (uint32_t r = lane; r < (uint32_t)KEY_ROWS; r += nlanes)
        if (!check_row_ok(img, tab768, t[CHK_KROWS + 2 * r], t[CHK_KROWS + 2 * r + 1])) acc.hit(CHK_LOOKUP, unit, 1, r);
    for (uint32_t e = lane; e < (uint32_t)KEY_COPIES; e += nlanes) {
        const uint32_t d = t[CHK_KEDGES + e];
        if (img[d & 0xffffu] != img[d >> 16]) acc.hit(CHK_COPY, unit, 1, e);
    }
    for (uint32_t r = lane; r < (uint32_t)WORDS_ROWS; r += nlanes) {
        const uint32_t gte = t[CHK_GATES + r];
        if ((gte >> 24) && img[gte & 0xffffu] != ((gte >> 16) & 0xffu)) acc.hit(CHK_GATE, unit, 1, r);
        if (key && r < 16 && img[gte & 0xffffu] != key[r]) acc.hit(CHK_INPUT, unit, 1, r);  // src/key_schedule.rs:107-114
    }
}

}  // namespace aesw
