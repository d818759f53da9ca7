/*
 * aesw_oracle.c -- CPU restatement of the tkmct/halo2-aes witness path.
 *
 * TEST INFRASTRUCTURE ONLY (see aesw_oracle.h).  Plain C, no GPU, no torch.
 * Every function cites the reference file:line it follows.  The reference's
 * halo2 front end is restated as a tiny single-pass layouter: a region is
 * placed at the first row at which none of its columns is in use
 * ([upstream] halo2_proofs v0.3.0 SimpleFloorPlanner / SingleChipLayouter),
 * so that the row each value lands on is *derived from the reference's call
 * order*, not hard-coded.
 */
#include "aesw_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* constants: src/constant.rs:1-47, src/utils.rs:28                           */
/* ------------------------------------------------------------------------- */

static uint8_t gf_xtime(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); }

static uint8_t gf_mul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    while (b) {
        if (b & 1) r ^= a;
        a = gf_xtime(a);
        b >>= 1;
    }
    return r;
}

static void build_tables(aesw_o_tables *t, int reference_typo) {
    /* FIPS-197 S-box: multiplicative inverse in GF(2^8) then the affine map. */
    for (int x = 0; x < 256; ++x) {
        uint8_t inv = 0;
        if (x) {
            for (int y = 1; y < 256; ++y)
                if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; ++i) {
            r = (uint8_t)((r << 1) | (r >> 7));
            s ^= r;
        }
        t->sbox[x] = (uint8_t)(s ^ 0x63);
        t->mul2[x] = gf_xtime((uint8_t)x);                       /* src/constant.rs:17-31 */
        t->mul3[x] = (uint8_t)(gf_xtime((uint8_t)x) ^ x);        /* src/constant.rs:33-47 */
    }
    /* src/constant.rs:14: the reference's last entry is 23, FIPS-197 has 22. */
    if (reference_typo) t->sbox[255] = 23;
}

void aesw_o_reference_tables(aesw_o_tables *t) { build_tables(t, 1); }
void aesw_o_fips_tables(aesw_o_tables *t) { build_tables(t, 0); }

/* src/utils.rs:28 */
static const uint64_t ROUND_CONSTANT[10] = {1, 2, 4, 8, 16, 32, 64, 128, 27, 54};

uint64_t aesw_o_round_constant(uint32_t round) { return ROUND_CONSTANT[round]; }

/* Fp::to_bytes(): canonical 32-byte little-endian repr. Every value on this
 * path is < 2^64 << p, so the repr is the integer itself. */
static void fp_to_bytes(uint64_t v, uint8_t out[32]) {
    memset(out, 0, 32);
    for (int i = 0; i < 8; ++i) out[i] = (uint8_t)(v >> (8 * i));
}

/* Fp::from_bytes(): rejects non-canonical reprs; anything with a byte set
 * above index 7 cannot be produced here, report it instead of wrapping. */
static int fp_from_bytes(const uint8_t in[32], uint64_t *v) {
    for (int i = 8; i < 32; ++i)
        if (in[i]) return AESW_O_ERR_ARG;
    uint64_t r = 0;
    for (int i = 0; i < 8; ++i) r |= (uint64_t)in[i] << (8 * i);
    *v = r;
    return AESW_O_OK;
}

/* src/utils.rs:8-19 */
int aesw_o_xor_bytes(uint64_t x, uint64_t y, uint64_t *z) {
    uint8_t xb[32], yb[32], zb[32];
    fp_to_bytes(x, xb);
    fp_to_bytes(y, yb);
    for (int i = 0; i < 32; ++i) zb[i] = xb[i] ^ yb[i];
    return fp_from_bytes(zb, z);
}

/* src/utils.rs:22-24 */
uint8_t aesw_o_sub_byte(const aesw_o_tables *t, uint64_t x) {
    uint8_t xb[32];
    fp_to_bytes(x, xb);
    return t->sbox[xb[0]];
}

/* ------------------------------------------------------------------------- */
/* mini layouter                                                              */
/* ------------------------------------------------------------------------- */

typedef struct { uint32_t col; uint64_t row; } cell_t; /* AssignedCell<Fp,Fp> */
typedef struct { cell_t a, b; } copy_t;

enum { RC_ADVICE = 0, RC_FIXED = 1, RC_SELECTOR = 2 };
typedef struct { uint8_t kind; uint32_t idx; } regcol_t;

/* src/table.rs:10-16 */
enum { TAG_U8 = 1, TAG_XOR = 2, TAG_SBOX = 3, TAG_GFMUL2 = 4, TAG_GFMUL3 = 5 };

typedef struct { uint32_t x, q; } range_cfg_t;          /* u8_range_check_chip.rs:8-11 */
typedef struct { uint32_t x, y, z, q; } xor_cfg_t;      /* u8_xor_chip.rs:13-18 */
typedef struct { uint32_t x, y, q; } sbox_cfg_t;        /* sbox_chip.rs:13-17 */
typedef struct { uint32_t x, y, q; } mul_cfg_t;         /* gf_mul_chip.rs:14-18 */

typedef struct {
    uint32_t words_column;  /* advice */
    uint32_t q_eq_rcon;     /* selector */
    range_cfg_t range;
    xor_cfg_t xor_;
    sbox_cfg_t sbox;
} keysched_cfg_t; /* src/key_schedule.rs:26-36 */

#define MAX_SETS 64

struct aesw_o_circuit {
    int status;
    uint32_t k, n_sets;
    uint64_t n_rows;
    uint32_t n_advice, n_selectors;
    uint8_t *advice, *assigned; /* [n_advice][n_rows] */
    uint8_t *selectors;         /* [n_selectors][n_rows] */
    uint8_t *fixed, *fixed_assigned;
    uint64_t *h_adv, *h_sel;
    uint64_t h_fixed;
    copy_t *copies;
    uint64_t n_copies, cap_copies;
    int record_copies;
    uint64_t n_regions;
    aesw_o_tables t;
    uint8_t *tab[4]; /* lookup table columns, NULL when not loaded */
    /* FixedAes128Config state, src/aes128.rs:27-43 */
    range_cfg_t c_range[MAX_SETS];
    xor_cfg_t c_xor[MAX_SETS];
    sbox_cfg_t c_sbox[MAX_SETS];
    mul_cfg_t c_mul2[MAX_SETS], c_mul3[MAX_SETS];
    keysched_cfg_t ks;
    int have_keys;
    cell_t keys[11][16];
    uint32_t current;
    uint64_t count;
    /* per-block bookkeeping for the tests */
    uint64_t n_blocks, cap_blocks;
    uint32_t *blk_set;
    uint64_t *blk_row;
    cell_t *blk_ct; /* 16 per block */
};

static uint8_t *adv_col(const aesw_o_circuit *c, uint32_t col) { return c->advice + (size_t)col * c->n_rows; }
static uint8_t *asg_col(const aesw_o_circuit *c, uint32_t col) { return c->assigned + (size_t)col * c->n_rows; }
static uint8_t *sel_col(const aesw_o_circuit *c, uint32_t s) { return c->selectors + (size_t)s * c->n_rows; }

/* SingleChipLayouter::assign_region: "position the region starting at the
 * earliest row for which none of the columns are in use" [upstream]. */
static uint64_t region_place(aesw_o_circuit *c, const regcol_t *cols, int ncols, uint64_t row_count) {
    uint64_t start = 0;
    for (int i = 0; i < ncols; ++i) {
        uint64_t h = cols[i].kind == RC_ADVICE ? c->h_adv[cols[i].idx]
                   : cols[i].kind == RC_SELECTOR ? c->h_sel[cols[i].idx] : c->h_fixed;
        if (h > start) start = h;
    }
    for (int i = 0; i < ncols; ++i) {
        if (cols[i].kind == RC_ADVICE) c->h_adv[cols[i].idx] = start + row_count;
        else if (cols[i].kind == RC_SELECTOR) c->h_sel[cols[i].idx] = start + row_count;
        else c->h_fixed = start + row_count;
    }
    c->n_regions++;
    if (start + row_count > c->n_rows && c->status == AESW_O_OK) c->status = AESW_O_ERR_ROWS;
    return start;
}

static cell_t assign_advice(aesw_o_circuit *c, uint32_t col, uint64_t row, uint64_t v) {
    cell_t cell = {col, row};
    if (row >= c->n_rows || v > 255) {
        if (c->status == AESW_O_OK) c->status = row >= c->n_rows ? AESW_O_ERR_ROWS : AESW_O_ERR_ARG;
        return cell;
    }
    adv_col(c, col)[row] = (uint8_t)v;
    asg_col(c, col)[row] = 1;
    return cell;
}

static uint64_t cell_value(const aesw_o_circuit *c, cell_t cell) {
    if (cell.row >= c->n_rows) return 0;
    return adv_col(c, cell.col)[cell.row];
}

/* AssignedCell::copy_advice: assign the same value in the target column and
 * constrain the two cells equal [upstream]. */
static cell_t copy_advice(aesw_o_circuit *c, cell_t src, uint32_t col, uint64_t row) {
    cell_t dst = assign_advice(c, col, row, cell_value(c, src));
    if (c->record_copies) {
        if (c->n_copies == c->cap_copies) {
            uint64_t ncap = c->cap_copies ? c->cap_copies * 2 : 4096;
            copy_t *p = (copy_t *)realloc(c->copies, ncap * sizeof(copy_t));
            if (!p) { c->status = AESW_O_ERR_NOMEM; return dst; }
            c->copies = p;
            c->cap_copies = ncap;
        }
        c->copies[c->n_copies].a = src;
        c->copies[c->n_copies].b = dst;
        c->n_copies++;
    }
    return dst;
}

static void enable_selector(aesw_o_circuit *c, uint32_t sel, uint64_t row) {
    if (row < c->n_rows) sel_col(c, sel)[row] = 1;
}

/* ------------------------------------------------------------------------- */
/* chips                                                                      */
/* ------------------------------------------------------------------------- */

/* src/chips/u8_xor_chip.rs:63-100 */
static cell_t chip_xor(aesw_o_circuit *c, const xor_cfg_t *cfg, cell_t x, cell_t y) {
    regcol_t cols[4] = {{RC_SELECTOR, cfg->q}, {RC_ADVICE, cfg->x}, {RC_ADVICE, cfg->y}, {RC_ADVICE, cfg->z}};
    uint64_t r = region_place(c, cols, 4, 1);
    enable_selector(c, cfg->q, r);
    cell_t xc = copy_advice(c, x, cfg->x, r);
    cell_t yc = copy_advice(c, y, cfg->y, r);
    uint64_t z = 0;
    if (aesw_o_xor_bytes(cell_value(c, xc), cell_value(c, yc), &z) != AESW_O_OK && c->status == AESW_O_OK)
        c->status = AESW_O_ERR_ARG;
    return assign_advice(c, cfg->z, r, z);
}

/* src/chips/sbox_chip.rs:57-83 */
static cell_t chip_sbox(aesw_o_circuit *c, const sbox_cfg_t *cfg, cell_t x) {
    regcol_t cols[3] = {{RC_SELECTOR, cfg->q}, {RC_ADVICE, cfg->x}, {RC_ADVICE, cfg->y}};
    uint64_t r = region_place(c, cols, 3, 1);
    enable_selector(c, cfg->q, r);
    cell_t xc = copy_advice(c, x, cfg->x, r);
    return assign_advice(c, cfg->y, r, aesw_o_sub_byte(&c->t, cell_value(c, xc)));
}

/* src/chips/gf_mul_chip.rs:59-89 (value read from the SOURCE cell, :80) */
static cell_t chip_mul(aesw_o_circuit *c, const mul_cfg_t *cfg, const uint8_t dict[256], cell_t x) {
    regcol_t cols[3] = {{RC_SELECTOR, cfg->q}, {RC_ADVICE, cfg->x}, {RC_ADVICE, cfg->y}};
    uint64_t r = region_place(c, cols, 3, 1);
    enable_selector(c, cfg->q, r);
    copy_advice(c, x, cfg->x, r);
    uint8_t xb[32];
    fp_to_bytes(cell_value(c, x), xb);
    return assign_advice(c, cfg->y, r, dict[xb[0]]);
}

/* src/chips/u8_range_check_chip.rs:51-70 */
static void chip_range(aesw_o_circuit *c, const range_cfg_t *cfg, cell_t x) {
    regcol_t cols[2] = {{RC_SELECTOR, cfg->q}, {RC_ADVICE, cfg->x}};
    uint64_t r = region_place(c, cols, 2, 1);
    enable_selector(c, cfg->q, r);
    copy_advice(c, x, cfg->x, r);
}

/* ------------------------------------------------------------------------- */
/* key schedule: src/key_schedule.rs                                          */
/* ------------------------------------------------------------------------- */

/* src/key_schedule.rs:98-118 */
static void ks_assign_first_round(aesw_o_circuit *c, const uint8_t key[16], cell_t out[16]) {
    regcol_t cols[1] = {{RC_ADVICE, c->ks.words_column}};
    uint64_t r = region_place(c, cols, 1, 16);
    for (int i = 0; i < 16; ++i) out[i] = assign_advice(c, c->ks.words_column, r + i, key[i]);
}

/* src/key_schedule.rs:122-224 */
static void ks_assign_round(aesw_o_circuit *c, uint32_t round, const cell_t prev[16], cell_t words[16]) {
    const keysched_cfg_t *ks = &c->ks;
    /* :141-154 "shift previous round": copy bytes 13,14,15,12 to words_column */
    static const int rot[4] = {13, 14, 15, 12};
    cell_t shifted[4], subbed[4], rc_assigned[4], rconned[4], next_word[4];
    {
        regcol_t cols[1] = {{RC_ADVICE, ks->words_column}};
        uint64_t r = region_place(c, cols, 1, 4);
        for (int i = 0; i < 4; ++i) shifted[i] = copy_advice(c, prev[rot[i]], ks->words_column, r + i);
    }
    /* :156-159 */
    for (int i = 0; i < 4; ++i) subbed[i] = chip_sbox(c, &ks->sbox, shifted[i]);
    /* :161-187 "Assign rc": selector + fixed + advice rc, then three zero pads */
    {
        uint64_t rc = aesw_o_round_constant(round - 1);
        regcol_t cols[3] = {{RC_SELECTOR, ks->q_eq_rcon}, {RC_FIXED, 0}, {RC_ADVICE, ks->words_column}};
        uint64_t r = region_place(c, cols, 3, 4);
        enable_selector(c, ks->q_eq_rcon, r);
        if (r < c->n_rows) { c->fixed[r] = (uint8_t)rc; c->fixed_assigned[r] = 1; }
        rc_assigned[0] = assign_advice(c, ks->words_column, r, rc);
        for (int i = 0; i < 3; ++i) rc_assigned[i + 1] = assign_advice(c, ks->words_column, r + i + 1, 0);
    }
    /* :189-194 */
    for (int i = 0; i < 4; ++i) rconned[i] = chip_xor(c, &ks->xor_, subbed[i], rc_assigned[i]);
    /* :197-204 */
    for (int i = 0; i < 4; ++i) next_word[i] = chip_xor(c, &ks->xor_, prev[i], rconned[i]);
    for (int i = 0; i < 4; ++i) words[i] = next_word[i];
    /* :207-216 */
    for (int i = 1; i < 4; ++i) {
        cell_t nw[4];
        for (int j = 0; j < 4; ++j) nw[j] = chip_xor(c, &ks->xor_, prev[i * 4 + j], next_word[j]);
        for (int j = 0; j < 4; ++j) { next_word[j] = nw[j]; words[i * 4 + j] = nw[j]; }
    }
    /* :218-221 */
    for (int i = 0; i < 16; ++i) chip_range(c, &ks->range, words[i]);
}

/* src/key_schedule.rs:80-96 */
static void ks_schedule_keys(aesw_o_circuit *c, const uint8_t key[16], cell_t words[11][16]) {
    ks_assign_first_round(c, key, words[0]);
    for (uint32_t i = 1; i <= 10; ++i) ks_assign_round(c, i, words[i - 1], words[i]);
}

/* ------------------------------------------------------------------------- */
/* AES-128 gadget: src/aes128.rs                                              */
/* ------------------------------------------------------------------------- */

/* src/aes128.rs:143-152 */
static void aes_schedule_key(aesw_o_circuit *c, const uint8_t key[16]) {
    ks_schedule_keys(c, key, c->keys);
    c->have_keys = 1;
}

/* src/aes128.rs:303-325 */
static int aes_callable(aesw_o_circuit *c) {
    uint64_t max_row = (uint64_t)1 << c->k;
    if (c->current == 0) max_row -= AESW_O_KEY_SCHEDULE_ROWS;
    if (max_row >= c->count * AESW_O_AES_ROWS + AESW_O_AES_ROWS) return 1;
    if (c->current < c->n_sets - 1) {
        c->current += 1;
        c->count = 0;
        return 1;
    }
    return 0;
}

/* src/aes128.rs:268-301 */
static cell_t aes_lcon(aesw_o_circuit *c, const cell_t word[4], const uint32_t coeffs[4]) {
    const uint32_t cur = c->current;
    cell_t tmp[4];
    for (int t = 0; t < 4; ++t) {
        switch (coeffs[t]) {
        case 1: { /* :279-288 one-row region, copy into advices[0] only */
            regcol_t cols[1] = {{RC_ADVICE, c->c_xor[cur].x}};
            uint64_t r = region_place(c, cols, 1, 1);
            tmp[t] = copy_advice(c, word[t], c->c_xor[cur].x, r);
            break;
        }
        case 2: tmp[t] = chip_mul(c, &c->c_mul2[cur], c->t.mul2, word[t]); break;
        case 3: tmp[t] = chip_mul(c, &c->c_mul3[cur], c->t.mul3, word[t]); break;
        default: c->status = AESW_O_ERR_ARG; tmp[t] = word[t]; break; /* panic :294 */
        }
    }
    cell_t i1 = chip_xor(c, &c->c_xor[cur], tmp[0], tmp[1]);
    cell_t i2 = chip_xor(c, &c->c_xor[cur], tmp[2], tmp[3]);
    return chip_xor(c, &c->c_xor[cur], i1, i2);
}

/* src/aes128.rs:154-265 */
static int aes_encrypt(aesw_o_circuit *c, const uint8_t plaintext[16], cell_t out[16]) {
    if (!aes_callable(c)) return AESW_O_ERR_CAPACITY; /* :159-162 */
    c->count += 1;
    if (!c->have_keys) return AESW_O_ERR_NO_KEY; /* :170 */
    const uint32_t cur = c->current;
    const xor_cfg_t *xc = &c->c_xor[cur];
    const sbox_cfg_t *sc = &c->c_sbox[cur];
    const uint32_t adv0 = xc->x;

    /* :176-192 "Assign plaintext": one 16-row region in advices[0] */
    cell_t prev[16];
    uint64_t first_row;
    {
        regcol_t cols[1] = {{RC_ADVICE, adv0}};
        uint64_t r = region_place(c, cols, 1, 16);
        first_row = r;
        for (int i = 0; i < 16; ++i) prev[i] = assign_advice(c, adv0, r + i, plaintext[i]);
    }
    /* :194-198 */
    {
        cell_t nx[16];
        for (int i = 0; i < 16; ++i) nx[i] = chip_xor(c, xc, prev[i], c->keys[0][i]);
        memcpy(prev, nx, sizeof nx);
    }
    static const uint32_t matrix[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}}; /* :228-233 */
    for (int no_round = 1; no_round < 11; ++no_round) {
        /* :203-209 */
        cell_t subbed[4][4];
        for (int i = 0; i < 16; ++i) subbed[i / 4][i % 4] = chip_sbox(c, sc, prev[i]);
        /* :216-223 */
        cell_t shifted[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) shifted[i][j] = subbed[(i + j) % 4][j];
        /* :236-248 */
        cell_t mixed[4][4];
        if (no_round == 10) {
            memcpy(mixed, shifted, sizeof mixed);
        } else {
            for (int w = 0; w < 4; ++w)
                for (int m = 0; m < 4; ++m) mixed[w][m] = aes_lcon(c, shifted[w], matrix[m]);
        }
        /* :250-261 */
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                prev[i * 4 + j] = chip_xor(c, xc, mixed[i][j], c->keys[no_round][i * 4 + j]);
    }
    memcpy(out, prev, 16 * sizeof(cell_t));
    /* bookkeeping */
    if (c->blk_set) {
        if (c->n_blocks < c->cap_blocks) {
            c->blk_set[c->n_blocks] = cur;
            c->blk_row[c->n_blocks] = first_row;
            memcpy(c->blk_ct + 16 * c->n_blocks, prev, 16 * sizeof(cell_t));
        }
    }
    c->n_blocks++;
    return c->status;
}

/* ------------------------------------------------------------------------- */
/* lookup table: src/table.rs:18-192                                          */
/* ------------------------------------------------------------------------- */

int aesw_o_lookup_table(const aesw_o_tables *t, uint8_t *t0, uint8_t *t1, uint8_t *t2, uint8_t *t3) {
    if (!t || !t0 || !t1 || !t2 || !t3) return AESW_O_ERR_ARG;
    size_t offset = 0;
    for (int i = 0; i < 256; ++i) { /* :27-53 u8 range */
        size_t pos = (size_t)i + offset;
        t0[pos] = TAG_U8; t1[pos] = (uint8_t)i; t2[pos] = 0; t3[pos] = 0;
    }
    offset += 256;
    for (int i = 0; i < 256; ++i) { /* :57-83 sbox */
        size_t pos = offset + i;
        t0[pos] = TAG_SBOX; t1[pos] = (uint8_t)i; t2[pos] = t->sbox[i]; t3[pos] = 0;
    }
    offset += 256;
    size_t l = offset;
    for (int i = 0; i < 256; ++i) /* :88-116 xor */
        for (int j = 0; j < 256; ++j) {
            t0[l] = TAG_XOR; t1[l] = (uint8_t)i; t2[l] = (uint8_t)j; t3[l] = (uint8_t)(i ^ j);
            l++;
        }
    offset += 65536;
    for (int i = 0; i < 256; ++i) { /* :120-145 mul2 */
        t0[offset + i] = TAG_GFMUL2; t1[offset + i] = (uint8_t)i; t2[offset + i] = t->mul2[i]; t3[offset + i] = 0;
    }
    offset += 256;
    for (int i = 0; i < 256; ++i) { /* :149-174 mul3 */
        t0[offset + i] = TAG_GFMUL3; t1[offset + i] = (uint8_t)i; t2[offset + i] = t->mul3[i]; t3[offset + i] = 0;
    }
    offset += 256;
    t0[offset] = t1[offset] = t2[offset] = t3[offset] = 0; /* :178-187 empty row */
    return AESW_O_OK;
}

/* ------------------------------------------------------------------------- */
/* circuit construction: configure()                                          */
/* ------------------------------------------------------------------------- */

static aesw_o_circuit *circuit_alloc(uint32_t k, uint64_t n_rows, uint32_t n_sets, uint32_t n_advice,
                                     uint32_t n_selectors, const aesw_o_tables *t, int record_copies,
                                     uint64_t cap_blocks, int with_table) {
    aesw_o_circuit *c = (aesw_o_circuit *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->k = k;
    c->n_rows = n_rows;
    c->n_sets = n_sets;
    c->n_advice = n_advice;
    c->n_selectors = n_selectors;
    c->t = *t;
    c->record_copies = record_copies;
    c->advice = (uint8_t *)calloc((size_t)n_advice * n_rows, 1);
    c->assigned = (uint8_t *)calloc((size_t)n_advice * n_rows, 1);
    c->selectors = (uint8_t *)calloc((size_t)n_selectors * n_rows, 1);
    c->fixed = (uint8_t *)calloc(n_rows, 1);
    c->fixed_assigned = (uint8_t *)calloc(n_rows, 1);
    c->h_adv = (uint64_t *)calloc(n_advice, sizeof(uint64_t));
    c->h_sel = (uint64_t *)calloc(n_selectors, sizeof(uint64_t));
    int ok = c->advice && c->assigned && c->selectors && c->fixed && c->fixed_assigned && c->h_adv && c->h_sel;
    if (ok && cap_blocks) {
        c->cap_blocks = cap_blocks;
        c->blk_set = (uint32_t *)calloc(cap_blocks, sizeof(uint32_t));
        c->blk_row = (uint64_t *)calloc(cap_blocks, sizeof(uint64_t));
        c->blk_ct = (cell_t *)calloc(cap_blocks * 16, sizeof(cell_t));
        ok = c->blk_set && c->blk_row && c->blk_ct;
    }
    if (ok && with_table) {
        for (int i = 0; i < 4 && ok; ++i) {
            c->tab[i] = (uint8_t *)calloc(AESW_O_TABLE_ROWS, 1);
            ok = c->tab[i] != NULL;
        }
        if (ok) aesw_o_lookup_table(t, c->tab[0], c->tab[1], c->tab[2], c->tab[3]); /* load_enc_full_table */
    }
    if (!ok) { aesw_o_circuit_free(c); return NULL; }
    return c;
}

/* FixedAes128Config::configure, src/aes128.rs:46-141: advice set i = columns
 * 3i..3i+2 (:54-60); per set selectors in the order range, xor, sbox, mul2,
 * mul3 (:64-68); key schedule built on set 0 (:118-124) which allocates
 * words_column next (src/key_schedule.rs:48) and selector q_eq_rcon (:50). */
static void configure_aes(aesw_o_circuit *c) {
    for (uint32_t i = 0; i < c->n_sets; ++i) {
        uint32_t a0 = 3 * i, a1 = 3 * i + 1, a2 = 3 * i + 2, s = 5 * i;
        c->c_range[i] = (range_cfg_t){a0, s + 0};
        c->c_xor[i] = (xor_cfg_t){a0, a1, a2, s + 1};
        c->c_sbox[i] = (sbox_cfg_t){a0, a1, s + 2};
        c->c_mul2[i] = (mul_cfg_t){a0, a1, s + 3};
        c->c_mul3[i] = (mul_cfg_t){a0, a1, s + 4};
    }
    c->ks.words_column = 3 * c->n_sets;
    c->ks.q_eq_rcon = 5 * c->n_sets;
    c->ks.range = c->c_range[0];
    c->ks.xor_ = c->c_xor[0];
    c->ks.sbox = c->c_sbox[0];
}

aesw_o_circuit *aesw_o_circuit_synthesize(uint32_t k, uint32_t n_sets, const aesw_o_tables *t,
                                          const uint8_t key[16], const uint8_t *pts, uint64_t n_blocks,
                                          int record_copies) {
    if (!t || !key || (n_blocks && !pts) || n_sets == 0 || n_sets > MAX_SETS || k < 11 || k > 30) return NULL;
    aesw_o_circuit *c = circuit_alloc(k, (uint64_t)1 << k, n_sets, 3 * n_sets + 1, 5 * n_sets + 1, t,
                                      record_copies, n_blocks ? n_blocks : 1, 1);
    if (!c) return NULL;
    configure_aes(c);
    aes_schedule_key(c, key); /* benches/aes128.rs:50 */
    for (uint64_t b = 0; b < n_blocks && c->status == AESW_O_OK; ++b) {
        cell_t out[16];
        int rc = aes_encrypt(c, pts + 16 * b, out); /* benches/aes128.rs:51-53 */
        if (rc != AESW_O_OK && c->status == AESW_O_OK) c->status = rc;
    }
    return c;
}

/* src/key_schedule.rs:245-320 TestCircuit: advice 0..2, then words_column 3;
 * selectors range 0, xor 1, sbox 2, q_eq_rcon 3. */
aesw_o_circuit *aesw_o_key_circuit_synthesize(uint32_t k, const aesw_o_tables *t, const uint8_t key[16],
                                              int record_copies) {
    if (!t || !key || k < 9 || k > 30) return NULL;
    aesw_o_circuit *c = circuit_alloc(k, (uint64_t)1 << k, 1, 4, 4, t, record_copies, 0, 1);
    if (!c) return NULL;
    c->c_range[0] = (range_cfg_t){0, 0};
    c->c_xor[0] = (xor_cfg_t){0, 1, 2, 1};
    c->c_sbox[0] = (sbox_cfg_t){0, 1, 2};
    c->ks.words_column = 3;
    c->ks.q_eq_rcon = 3;
    c->ks.range = c->c_range[0];
    c->ks.xor_ = c->c_xor[0];
    c->ks.sbox = c->c_sbox[0];
    ks_schedule_keys(c, key, c->keys);
    c->have_keys = 1;
    return c;
}

void aesw_o_circuit_free(aesw_o_circuit *c) {
    if (!c) return;
    free(c->advice); free(c->assigned); free(c->selectors); free(c->fixed); free(c->fixed_assigned);
    free(c->h_adv); free(c->h_sel); free(c->copies); free(c->blk_set); free(c->blk_row); free(c->blk_ct);
    for (int i = 0; i < 4; ++i) free(c->tab[i]);
    free(c);
}

int aesw_o_circuit_status(const aesw_o_circuit *c) { return c ? c->status : AESW_O_ERR_ARG; }
uint32_t aesw_o_circuit_num_advice(const aesw_o_circuit *c) { return c->n_advice; }
uint32_t aesw_o_circuit_num_selectors(const aesw_o_circuit *c) { return c->n_selectors; }
uint64_t aesw_o_circuit_num_rows(const aesw_o_circuit *c) { return c->n_rows; }
uint64_t aesw_o_circuit_num_regions(const aesw_o_circuit *c) { return c->n_regions; }
uint64_t aesw_o_circuit_num_copies(const aesw_o_circuit *c) { return c->n_copies; }
int aesw_o_circuit_copies(const aesw_o_circuit *c, uint64_t *out) {
    if (!c || !out || !c->record_copies) return AESW_O_ERR_ARG;
    for (uint64_t i = 0; i < c->n_copies; ++i) {
        out[4 * i + 0] = c->copies[i].a.col; out[4 * i + 1] = c->copies[i].a.row;
        out[4 * i + 2] = c->copies[i].b.col; out[4 * i + 3] = c->copies[i].b.row;
    }
    return AESW_O_OK;
}
uint64_t aesw_o_circuit_column_height(const aesw_o_circuit *c, uint32_t col) {
    return col < c->n_advice ? c->h_adv[col] : 0;
}
const uint8_t *aesw_o_circuit_advice(const aesw_o_circuit *c, uint32_t col) {
    return col < c->n_advice ? adv_col(c, col) : NULL;
}
const uint8_t *aesw_o_circuit_advice_assigned(const aesw_o_circuit *c, uint32_t col) {
    return col < c->n_advice ? asg_col(c, col) : NULL;
}
const uint8_t *aesw_o_circuit_selector(const aesw_o_circuit *c, uint32_t s) {
    return s < c->n_selectors ? sel_col(c, s) : NULL;
}
const uint8_t *aesw_o_circuit_fixed(const aesw_o_circuit *c) { return c->fixed; }

int aesw_o_circuit_round_keys(const aesw_o_circuit *c, uint8_t rk[176]) {
    if (!c || !c->have_keys) return AESW_O_ERR_NO_KEY;
    for (int r = 0; r < 11; ++r)
        for (int i = 0; i < 16; ++i) rk[16 * r + i] = (uint8_t)cell_value(c, c->keys[r][i]);
    return AESW_O_OK;
}

int aesw_o_circuit_round_key_cells(const aesw_o_circuit *c, uint32_t col[176], uint64_t row[176]) {
    if (!c || !c->have_keys) return AESW_O_ERR_NO_KEY;
    for (int r = 0; r < 11; ++r)
        for (int i = 0; i < 16; ++i) {
            col[16 * r + i] = c->keys[r][i].col;
            row[16 * r + i] = c->keys[r][i].row;
        }
    return AESW_O_OK;
}

int aesw_o_circuit_ciphertext(const aesw_o_circuit *c, uint64_t b, uint8_t ct[16]) {
    if (!c || !c->blk_ct || b >= c->n_blocks || b >= c->cap_blocks) return AESW_O_ERR_ARG;
    for (int i = 0; i < 16; ++i) ct[i] = (uint8_t)cell_value(c, c->blk_ct[16 * b + i]);
    return AESW_O_OK;
}

int aesw_o_circuit_block_placement(const aesw_o_circuit *c, uint64_t b, uint32_t *set, uint64_t *row) {
    if (!c || !c->blk_set || b >= c->n_blocks || b >= c->cap_blocks) return AESW_O_ERR_ARG;
    *set = c->blk_set[b];
    *row = c->blk_row[b];
    return AESW_O_OK;
}

/* ------------------------------------------------------------------------- */
/* MockProver::assert_satisfied restated                                      */
/* ------------------------------------------------------------------------- */

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

static int in_set(const uint32_t *set, size_t n, uint32_t key) {
    return bsearch(&key, set, n, sizeof(uint32_t), cmp_u32) != NULL;
}

int aesw_o_circuit_verify(const aesw_o_circuit *c, char *msg, size_t msg_len) {
#define FAIL(...) do { if (msg && msg_len) snprintf(msg, msg_len, __VA_ARGS__); rc = AESW_O_ERR_UNSATISFIED; goto done; } while (0)
    int rc = AESW_O_OK;
    if (msg && msg_len) msg[0] = 0;
    if (!c || !c->tab[0]) return AESW_O_ERR_ARG;
    if (c->status != AESW_O_OK) {
        if (msg && msg_len) snprintf(msg, msg_len, "synthesis failed with status %d", c->status);
        return AESW_O_ERR_UNSATISFIED;
    }
    const size_t T = AESW_O_TABLE_ROWS;
    uint32_t *p2 = (uint32_t *)malloc(T * 4), *p3 = (uint32_t *)malloc(T * 4), *p4 = (uint32_t *)malloc(T * 4);
    if (!p2 || !p3 || !p4) { free(p2); free(p3); free(p4); return AESW_O_ERR_NOMEM; }
    for (size_t i = 0; i < T; ++i) {
        uint32_t a = c->tab[0][i], b = c->tab[1][i], d = c->tab[2][i], e = c->tab[3][i];
        p2[i] = a << 8 | b;
        p3[i] = a << 16 | b << 8 | d;
        p4[i] = a << 24 | b << 16 | d << 8 | e;
    }
    qsort(p2, T, 4, cmp_u32); qsort(p3, T, 4, cmp_u32); qsort(p4, T, 4, cmp_u32);
    /* With the selector off every lookup input is (0,..,0): must be a table row. */
    if (!in_set(p2, T, 0) || !in_set(p3, T, 0) || !in_set(p4, T, 0)) FAIL("zero row missing from table");

    for (uint32_t i = 0; i < c->n_sets; ++i) {
        const uint8_t *x = adv_col(c, c->c_xor[i].x), *y = adv_col(c, c->c_xor[i].y), *z = adv_col(c, c->c_xor[i].z);
        const uint8_t *ax = asg_col(c, c->c_xor[i].x), *ay = asg_col(c, c->c_xor[i].y), *az = asg_col(c, c->c_xor[i].z);
        const uint8_t *q_range = sel_col(c, c->c_range[i].q), *q_xor = sel_col(c, c->c_xor[i].q);
        const uint8_t *q_sbox = sel_col(c, c->c_sbox[i].q);
        const int has_mul = c->n_selectors > 4; /* the key-schedule test circuit has no mul chips */
        const uint8_t *q_m2 = has_mul ? sel_col(c, c->c_mul2[i].q) : NULL;
        const uint8_t *q_m3 = has_mul ? sel_col(c, c->c_mul3[i].q) : NULL;
        for (uint64_t r = 0; r < c->n_rows; ++r) {
            if (q_range[r]) { /* u8_range_check_chip.rs:34-42 */
                if (!ax[r]) FAIL("set %u row %llu: range x unassigned", i, (unsigned long long)r);
                if (!in_set(p2, T, (uint32_t)TAG_U8 << 8 | x[r])) FAIL("set %u row %llu: range lookup", i, (unsigned long long)r);
            }
            if (q_xor[r]) { /* u8_xor_chip.rs:41-53 */
                if (!ax[r] || !ay[r] || !az[r]) FAIL("set %u row %llu: xor cell unassigned", i, (unsigned long long)r);
                if (!in_set(p4, T, (uint32_t)TAG_XOR << 24 | (uint32_t)x[r] << 16 | (uint32_t)y[r] << 8 | z[r]))
                    FAIL("set %u row %llu: xor lookup (%u,%u,%u)", i, (unsigned long long)r, x[r], y[r], z[r]);
            }
            if (q_sbox[r]) { /* sbox_chip.rs:38-48 */
                if (!ax[r] || !ay[r]) FAIL("set %u row %llu: sbox cell unassigned", i, (unsigned long long)r);
                if (!in_set(p3, T, (uint32_t)TAG_SBOX << 16 | (uint32_t)x[r] << 8 | y[r]))
                    FAIL("set %u row %llu: sbox lookup (%u,%u)", i, (unsigned long long)r, x[r], y[r]);
            }
            if (has_mul && q_m2[r]) { /* gf_mul_chip.rs:38-48 */
                if (!ax[r] || !ay[r]) FAIL("set %u row %llu: mul2 cell unassigned", i, (unsigned long long)r);
                if (!in_set(p3, T, (uint32_t)TAG_GFMUL2 << 16 | (uint32_t)x[r] << 8 | y[r]))
                    FAIL("set %u row %llu: mul2 lookup", i, (unsigned long long)r);
            }
            if (has_mul && q_m3[r]) {
                if (!ax[r] || !ay[r]) FAIL("set %u row %llu: mul3 cell unassigned", i, (unsigned long long)r);
                if (!in_set(p3, T, (uint32_t)TAG_GFMUL3 << 16 | (uint32_t)x[r] << 8 | y[r]))
                    FAIL("set %u row %llu: mul3 lookup", i, (unsigned long long)r);
            }
        }
    }
    { /* gate "Equality RC": q*(x-c), src/key_schedule.rs:59-64 */
        const uint8_t *q = sel_col(c, c->ks.q_eq_rcon), *w = adv_col(c, c->ks.words_column);
        const uint8_t *aw = asg_col(c, c->ks.words_column);
        for (uint64_t r = 0; r < c->n_rows; ++r)
            if (q[r]) {
                if (!aw[r] || !c->fixed_assigned[r]) FAIL("row %llu: rcon gate cell unassigned", (unsigned long long)r);
                if (w[r] != c->fixed[r]) FAIL("row %llu: rcon gate", (unsigned long long)r);
            }
    }
    if (c->record_copies) { /* permutation argument */
        for (uint64_t i = 0; i < c->n_copies; ++i) {
            cell_t a = c->copies[i].a, b = c->copies[i].b;
            if (!asg_col(c, a.col)[a.row] || !asg_col(c, b.col)[b.row]) FAIL("copy %llu: unassigned", (unsigned long long)i);
            if (cell_value(c, a) != cell_value(c, b)) FAIL("copy %llu: values differ", (unsigned long long)i);
        }
    }
done:
    free(p2); free(p3); free(p4);
    return rc;
#undef FAIL
}

int aesw_o_circuit_poke(aesw_o_circuit *c, uint32_t col, uint64_t row, uint8_t value) {
    if (!c || col >= c->n_advice || row >= c->n_rows) return AESW_O_ERR_ARG;
    adv_col(c, col)[row] = value;
    return AESW_O_OK;
}

/* ------------------------------------------------------------------------- */
/* slab level                                                                 */
/* ------------------------------------------------------------------------- */

#define WORKER_K 12u /* 2^12 - 1760 >= 1360: one block fits in set 0 after the key */

static aesw_o_circuit *worker_new(const aesw_o_tables *t) {
    aesw_o_circuit *c = circuit_alloc(WORKER_K, (uint64_t)1 << WORKER_K, 1, 4, 6, t, 0, 0, 0);
    if (c) configure_aes(c);
    return c;
}

/* Forget everything assigned at or after `row` in the three set-0 columns and
 * their selectors, so the next encrypt() lands on the same rows again. */
static void worker_rewind(aesw_o_circuit *c, uint64_t row) {
    for (uint32_t col = 0; col < 3; ++col) {
        memset(adv_col(c, col) + row, 0, c->n_rows - row);
        memset(asg_col(c, col) + row, 0, c->n_rows - row);
        c->h_adv[col] = row;
    }
    for (uint32_t s = 0; s < 5; ++s) {
        memset(sel_col(c, s) + row, 0, c->n_rows - row);
        if (c->h_sel[s] > row) c->h_sel[s] = row;
    }
    c->current = 0;
    c->count = 0;
    c->n_blocks = 0;
}

static void worker_reset(aesw_o_circuit *c) {
    memset(c->advice, 0, (size_t)c->n_advice * c->n_rows);
    memset(c->assigned, 0, (size_t)c->n_advice * c->n_rows);
    memset(c->selectors, 0, (size_t)c->n_selectors * c->n_rows);
    memset(c->fixed, 0, c->n_rows);
    memset(c->fixed_assigned, 0, c->n_rows);
    memset(c->h_adv, 0, c->n_advice * sizeof(uint64_t));
    memset(c->h_sel, 0, c->n_selectors * sizeof(uint64_t));
    c->h_fixed = 0;
    c->have_keys = 0;
    c->current = 0;
    c->count = 0;
    c->n_blocks = 0;
    c->n_regions = 0;
    c->status = AESW_O_OK;
}

/* masks / packed maps, computed once from a real run */
static pthread_once_t g_mask_once = PTHREAD_ONCE_INIT;
static uint8_t g_enc_mask[3][AESW_O_AES_ROWS];
static uint8_t g_key_mask[3][AESW_O_KEY_ROWS];
static int32_t g_enc_idx[3][AESW_O_AES_ROWS];
static int32_t g_key_idx[3][AESW_O_KEY_ROWS];
static uint32_t g_enc_live[3], g_key_live[3];
static int g_mask_status = AESW_O_ERR_NOMEM;

static void mask_init(void) {
    aesw_o_tables t;
    aesw_o_reference_tables(&t);
    aesw_o_circuit *c = worker_new(&t);
    if (!c) return;
    uint8_t zero[16] = {0};
    cell_t out[16];
    aes_schedule_key(c, zero);
    if (c->h_adv[0] != AESW_O_KEY_ROWS || c->h_adv[3] != AESW_O_WORDS_ROWS) { g_mask_status = AESW_O_ERR_ROWS; aesw_o_circuit_free(c); return; }
    int rc = aes_encrypt(c, zero, out);
    if (rc != AESW_O_OK || c->h_adv[0] != AESW_O_KEY_ROWS + AESW_O_AES_ROWS) { g_mask_status = AESW_O_ERR_ROWS; aesw_o_circuit_free(c); return; }
    for (int col = 0; col < 3; ++col) {
        uint32_t n = 0;
        for (uint32_t r = 0; r < AESW_O_KEY_ROWS; ++r) {
            g_key_mask[col][r] = asg_col(c, col)[r];
            g_key_idx[col][r] = g_key_mask[col][r] ? (int32_t)n++ : -1;
        }
        g_key_live[col] = n;
        n = 0;
        for (uint32_t r = 0; r < AESW_O_AES_ROWS; ++r) {
            g_enc_mask[col][r] = asg_col(c, col)[AESW_O_KEY_ROWS + r];
            g_enc_idx[col][r] = g_enc_mask[col][r] ? (int32_t)n++ : -1;
        }
        g_enc_live[col] = n;
    }
    aesw_o_circuit_free(c);
    g_mask_status = AESW_O_OK;
}

int aesw_o_encrypt_assigned_mask(int col, uint8_t mask[AESW_O_AES_ROWS]) {
    pthread_once(&g_mask_once, mask_init);
    if (g_mask_status != AESW_O_OK) return g_mask_status;
    if (col < 0 || col > 2 || !mask) return AESW_O_ERR_ARG;
    memcpy(mask, g_enc_mask[col], AESW_O_AES_ROWS);
    return AESW_O_OK;
}

int aesw_o_key_assigned_mask(int col, uint8_t mask[AESW_O_KEY_ROWS]) {
    pthread_once(&g_mask_once, mask_init);
    if (g_mask_status != AESW_O_OK) return g_mask_status;
    if (col < 0 || col > 2 || !mask) return AESW_O_ERR_ARG;
    memcpy(mask, g_key_mask[col], AESW_O_KEY_ROWS);
    return AESW_O_OK;
}

int aesw_o_encrypt_packed_index(int col, int32_t idx[AESW_O_AES_ROWS], uint32_t *live) {
    pthread_once(&g_mask_once, mask_init);
    if (g_mask_status != AESW_O_OK) return g_mask_status;
    if (col < 0 || col > 2) return AESW_O_ERR_ARG;
    if (idx) memcpy(idx, g_enc_idx[col], sizeof g_enc_idx[col]);
    if (live) *live = g_enc_live[col];
    return AESW_O_OK;
}

int aesw_o_key_packed_index(int col, int32_t idx[AESW_O_KEY_ROWS], uint32_t *live) {
    pthread_once(&g_mask_once, mask_init);
    if (g_mask_status != AESW_O_OK) return g_mask_status;
    if (col < 0 || col > 2) return AESW_O_ERR_ARG;
    if (idx) memcpy(idx, g_key_idx[col], sizeof g_key_idx[col]);
    if (live) *live = g_key_live[col];
    return AESW_O_OK;
}

static void emit_column(const uint8_t *src, const uint8_t *mask, uint32_t rows, int layout, uint8_t *dst) {
    if (layout == AESW_O_LAYOUT_DENSE) {
        memcpy(dst, src, rows); /* never-assigned cells were zeroed by rewind/reset */
    } else {
        uint32_t n = 0;
        for (uint32_t r = 0; r < rows; ++r)
            if (mask[r]) dst[n++] = src[r];
    }
}

typedef struct {
    const aesw_o_tables *t;
    const uint8_t *pt, *keys;
    int per_block_keys, layout, key_only;
    uint64_t b0, b1;
    uint8_t *x, *y, *z, *ct;          /* encrypt slab outputs (may be NULL) */
    uint8_t *w, *kx, *ky, *kz, *rk;   /* key slab outputs (may be NULL) */
    int status;
} job_t;

static void *job_run(void *arg) {
    job_t *j = (job_t *)arg;
    aesw_o_circuit *c = worker_new(j->t);
    if (!c) { j->status = AESW_O_ERR_NOMEM; return NULL; }
    const uint32_t sx = j->layout == AESW_O_LAYOUT_DENSE ? AESW_O_AES_ROWS : g_enc_live[0];
    const uint32_t sy = j->layout == AESW_O_LAYOUT_DENSE ? AESW_O_AES_ROWS : g_enc_live[1];
    const uint32_t sz = j->layout == AESW_O_LAYOUT_DENSE ? AESW_O_AES_ROWS : g_enc_live[2];
    const uint32_t kxs = j->layout == AESW_O_LAYOUT_DENSE ? AESW_O_KEY_ROWS : g_key_live[0];
    const uint32_t kys = j->layout == AESW_O_LAYOUT_DENSE ? AESW_O_KEY_ROWS : g_key_live[1];
    const uint32_t kzs = j->layout == AESW_O_LAYOUT_DENSE ? AESW_O_KEY_ROWS : g_key_live[2];
    int keyed = 0;
    for (uint64_t b = j->b0; b < j->b1; ++b) {
        if (j->per_block_keys || j->key_only || !keyed) {
            worker_reset(c);
            aes_schedule_key(c, j->keys + (j->per_block_keys || j->key_only ? 16 * b : 0));
            keyed = 1;
            if (j->key_only || j->per_block_keys) {
                if (j->w) memcpy(j->w + (size_t)AESW_O_WORDS_ROWS * b, adv_col(c, 3), AESW_O_WORDS_ROWS);
                if (j->kx) emit_column(adv_col(c, 0), g_key_mask[0], AESW_O_KEY_ROWS, j->layout, j->kx + (size_t)kxs * b);
                if (j->ky) emit_column(adv_col(c, 1), g_key_mask[1], AESW_O_KEY_ROWS, j->layout, j->ky + (size_t)kys * b);
                if (j->kz) emit_column(adv_col(c, 2), g_key_mask[2], AESW_O_KEY_ROWS, j->layout, j->kz + (size_t)kzs * b);
                if (j->rk) aesw_o_circuit_round_keys(c, j->rk + (size_t)176 * b);
            }
        } else {
            worker_rewind(c, AESW_O_KEY_ROWS);
        }
        if (j->key_only) continue;
        cell_t out[16];
        int rc = aes_encrypt(c, j->pt + 16 * b, out);
        if (rc != AESW_O_OK) { j->status = rc; break; }
        if (j->x) emit_column(adv_col(c, 0) + AESW_O_KEY_ROWS, g_enc_mask[0], AESW_O_AES_ROWS, j->layout, j->x + (size_t)sx * b);
        if (j->y) emit_column(adv_col(c, 1) + AESW_O_KEY_ROWS, g_enc_mask[1], AESW_O_AES_ROWS, j->layout, j->y + (size_t)sy * b);
        if (j->z) emit_column(adv_col(c, 2) + AESW_O_KEY_ROWS, g_enc_mask[2], AESW_O_AES_ROWS, j->layout, j->z + (size_t)sz * b);
        if (j->ct)
            for (int i = 0; i < 16; ++i) j->ct[16 * b + i] = (uint8_t)cell_value(c, out[i]);
    }
    if (c->status != AESW_O_OK && j->status == AESW_O_OK) j->status = c->status;
    aesw_o_circuit_free(c);
    return NULL;
}

static int run_jobs(job_t *proto, uint64_t n, int nthreads) {
    pthread_once(&g_mask_once, mask_init);
    if (g_mask_status != AESW_O_OK) return g_mask_status;
    if (n == 0) return AESW_O_OK;
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > n) nthreads = (int)n;
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256];
    pthread_t th[256];
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = *proto;
        jobs[i].b0 = n * (uint64_t)i / (uint64_t)nthreads;
        jobs[i].b1 = n * (uint64_t)(i + 1) / (uint64_t)nthreads;
        jobs[i].status = AESW_O_OK;
    }
    if (nthreads == 1) {
        job_run(&jobs[0]);
        return jobs[0].status;
    }
    int started = 0;
    for (int i = 0; i < nthreads; ++i) {
        if (pthread_create(&th[i], NULL, job_run, &jobs[i]) != 0) break;
        started++;
    }
    for (int i = started; i < nthreads; ++i) job_run(&jobs[i]);
    int rc = AESW_O_OK;
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    for (int i = 0; i < nthreads; ++i)
        if (jobs[i].status != AESW_O_OK) rc = jobs[i].status;
    return rc;
}

int aesw_o_encrypt_witness(const aesw_o_tables *t, const uint8_t *pt, const uint8_t *keys, int per_block_keys,
                           uint64_t n, int layout, uint8_t *x, uint8_t *y, uint8_t *z, uint8_t *ct,
                           int nthreads) {
    if (!t || (n && (!pt || !keys)) || (layout != AESW_O_LAYOUT_DENSE && layout != AESW_O_LAYOUT_PACKED))
        return AESW_O_ERR_ARG;
    job_t j;
    memset(&j, 0, sizeof j);
    j.t = t; j.pt = pt; j.keys = keys; j.per_block_keys = per_block_keys; j.layout = layout;
    j.x = x; j.y = y; j.z = z; j.ct = ct;
    return run_jobs(&j, n, nthreads);
}

int aesw_o_key_schedule_witness(const aesw_o_tables *t, const uint8_t *keys, uint64_t n, int layout,
                                uint8_t *w, uint8_t *kx, uint8_t *ky, uint8_t *kz, uint8_t *rk,
                                int nthreads) {
    if (!t || (n && !keys) || (layout != AESW_O_LAYOUT_DENSE && layout != AESW_O_LAYOUT_PACKED))
        return AESW_O_ERR_ARG;
    job_t j;
    memset(&j, 0, sizeof j);
    j.t = t; j.keys = keys; j.key_only = 1; j.layout = layout;
    j.w = w; j.kx = kx; j.ky = ky; j.kz = kz; j.rk = rk;
    return run_jobs(&j, n, nthreads);
}
