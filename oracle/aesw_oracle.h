/*
 * aesw_oracle.h -- CPU restatement of the tkmct/halo2-aes witness path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under halo2-aes_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * The reference cannot be compiled here (Rust + un-vendored halo2_proofs
 * v0.3.0, Cargo.toml:14-18), so this oracle restates, region by region, what
 * the reference's synthesize() assigns:
 *   - a SimpleFloorPlanner-style layouter (region start = max height of the
 *     region's columns; [upstream halo2 v0.3.0, single_pass.rs]),
 *   - the four chips       (src/chips/{u8_xor,sbox,gf_mul,u8_range_check}_chip.rs),
 *   - the AES-128 gadget   (src/aes128.rs:143-325),
 *   - the key schedule     (src/key_schedule.rs:80-224),
 *   - the lookup table     (src/table.rs:18-192),
 *   - the byte helpers     (src/utils.rs:8-33) and tables (src/constant.rs:1-47).
 *
 * Parity pins (tests/test_oracle_pins.py): constant.rs tables (fixture),
 * test_xor_bytes 5^12=9 (src/utils.rs:40-47), EXPANDED zero-key round keys
 * (src/key_schedule.rs:337-345), AES_ROWS=1360 / KEY_SCHEDULE_ROWS
 * (src/constant.rs:113-114), MockProver-style satisfaction of every lookup,
 * gate and copy constraint (src/aes128.rs:409-418, src/key_schedule.rs:385-392),
 * FIPS-197 App. B / C.1 (valid because neither reaches S_BOX[0xff]).
 * Row ORDER inside a slab is derived from the reference's call order, not
 * executed: see DESIGN.md "parity pins".
 */
#ifndef AESW_ORACLE_H
#define AESW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AESW_O_AES_ROWS 1360u           /* src/constant.rs:114 */
#define AESW_O_KEY_SCHEDULE_ROWS 1760u  /* src/constant.rs:113 (capacity constant) */
#define AESW_O_KEY_ROWS 400u            /* rows the key schedule really uses in set 0 */
#define AESW_O_WORDS_ROWS 96u           /* rows used in words_column */
#define AESW_O_TABLE_ROWS 66561u        /* src/table.rs: 256+256+65536+256+256+1 */

enum {
    AESW_O_OK = 0,
    AESW_O_ERR_CAPACITY = 1,     /* panic!("AES calls too many...") src/aes128.rs:160-162 */
    AESW_O_ERR_NO_KEY = 2,       /* expect("Keys should be scheduled") src/aes128.rs:170 */
    AESW_O_ERR_ROWS = 3,         /* Error::NotEnoughRowsAvailable [upstream] */
    AESW_O_ERR_ARG = 4,
    AESW_O_ERR_NOMEM = 5,
    AESW_O_ERR_UNSATISFIED = 6
};

enum { AESW_O_LAYOUT_DENSE = 0, AESW_O_LAYOUT_PACKED = 1 };

typedef struct aesw_o_tables {
    uint8_t sbox[256];
    uint8_t mul2[256];
    uint8_t mul3[256];
} aesw_o_tables;

/* The reference's constants: S_BOX (with S_BOX[255]==23, src/constant.rs:14),
 * MUL_BY_2, MUL_BY_3 (src/constant.rs:17-47).  Generated arithmetically and
 * pinned against the text of constant.rs by tests (fixture in tests/golden). */
void aesw_o_reference_tables(aesw_o_tables *t);
/* FIPS-197 tables (S_BOX[255]==22), for the KAT cross-checks only. */
void aesw_o_fips_tables(aesw_o_tables *t);

/* src/utils.rs:8-19 on canonical 32-byte little-endian reprs of x and y. */
int aesw_o_xor_bytes(uint64_t x, uint64_t y, uint64_t *z);
/* src/utils.rs:22-24 */
uint8_t aesw_o_sub_byte(const aesw_o_tables *t, uint64_t x);
/* src/utils.rs:28,31-33 ; round in 0..9 */
uint64_t aesw_o_round_constant(uint32_t round);

/* ---- slab level (block-relative witness, what the HIP path emits) -------- */

/* Which rows of slab column col (0=x,1=y,2=z) are ever assigned by encrypt();
 * derived by running the layouter, not hard-coded. mask[r] in {0,1}. */
int aesw_o_encrypt_assigned_mask(int col, uint8_t mask[AESW_O_AES_ROWS]);
/* Same for the key slab columns (0=kx,1=ky,2=kz). */
int aesw_o_key_assigned_mask(int col, uint8_t mask[AESW_O_KEY_ROWS]);
/* packed index of dense row r in column col, or -1 when never assigned */
int aesw_o_encrypt_packed_index(int col, int32_t idx[AESW_O_AES_ROWS], uint32_t *live);
int aesw_o_key_packed_index(int col, int32_t idx[AESW_O_KEY_ROWS], uint32_t *live);

/* Batched witness: for block b, x/y/z hold its 1360-row slab (dense: stride
 * 1360 with never-assigned cells = 0; packed: stride = live cells of the
 * column).  keys: 16 B if !per_block_keys else n*16.  ct optional (n*16).
 * nthreads<=1 -> single thread (like the reference's synthesis). */
int aesw_o_encrypt_witness(const aesw_o_tables *t, const uint8_t *pt, const uint8_t *keys,
                           int per_block_keys, uint64_t n, int layout, uint8_t *x, uint8_t *y,
                           uint8_t *z, uint8_t *ct, int nthreads);

/* Key-schedule witness for n keys: w (n*96), kx/ky/kz (dense stride 400 or
 * packed stride = live), rk optional (n*176 round-key bytes). */
int aesw_o_key_schedule_witness(const aesw_o_tables *t, const uint8_t *keys, uint64_t n,
                                int layout, uint8_t *w, uint8_t *kx, uint8_t *ky, uint8_t *kz,
                                uint8_t *rk, int nthreads);

/* src/table.rs:18-192: 66561 rows x 4 table columns (values all < 256). */
int aesw_o_lookup_table(const aesw_o_tables *t, uint8_t *t0, uint8_t *t1, uint8_t *t2,
                        uint8_t *t3);

/* ---- circuit level: FixedAes128Config<K,N> synthesize() ------------------- */

typedef struct aesw_o_circuit aesw_o_circuit;

/* Mirrors Aes128BenchCircuit / TestAesCircuit::synthesize
 * (benches/aes128.rs:44-56, src/aes128.rs:389-402): load table, schedule_key,
 * n_blocks x encrypt(pts[b]).  rows = 2^K.  Returns NULL on allocation failure;
 * aesw_o_circuit_status() tells whether synthesis itself failed (capacity...). */
aesw_o_circuit *aesw_o_circuit_synthesize(uint32_t k, uint32_t n_sets, const aesw_o_tables *t,
                                          const uint8_t key[16], const uint8_t *pts,
                                          uint64_t n_blocks, int record_copies);
/* Mirrors key_schedule.rs TestCircuit (src/key_schedule.rs:245-320): 3 advice
 * columns + words_column, schedule_keys only. */
aesw_o_circuit *aesw_o_key_circuit_synthesize(uint32_t k, const aesw_o_tables *t,
                                              const uint8_t key[16], int record_copies);
void aesw_o_circuit_free(aesw_o_circuit *c);
int aesw_o_circuit_status(const aesw_o_circuit *c);
uint32_t aesw_o_circuit_num_advice(const aesw_o_circuit *c);   /* 3N+1 */
uint32_t aesw_o_circuit_num_selectors(const aesw_o_circuit *c);
uint64_t aesw_o_circuit_num_rows(const aesw_o_circuit *c);
uint64_t aesw_o_circuit_num_regions(const aesw_o_circuit *c);
uint64_t aesw_o_circuit_num_copies(const aesw_o_circuit *c);
/* the recorded copy_advice() calls, in call order: out[4*i..] = source column, source row, copy column, copy row */
int aesw_o_circuit_copies(const aesw_o_circuit *c, uint64_t *out);
uint64_t aesw_o_circuit_column_height(const aesw_o_circuit *c, uint32_t advice_col);
const uint8_t *aesw_o_circuit_advice(const aesw_o_circuit *c, uint32_t col);
const uint8_t *aesw_o_circuit_advice_assigned(const aesw_o_circuit *c, uint32_t col);
const uint8_t *aesw_o_circuit_selector(const aesw_o_circuit *c, uint32_t sel);
const uint8_t *aesw_o_circuit_fixed(const aesw_o_circuit *c);
/* round keys as returned by schedule_keys(): 176 bytes, and their cells */
int aesw_o_circuit_round_keys(const aesw_o_circuit *c, uint8_t rk[176]);
int aesw_o_circuit_round_key_cells(const aesw_o_circuit *c, uint32_t col[176], uint64_t row[176]);
/* ciphertext cells' values of block b (Vec<AssignedCell> returned by encrypt) */
int aesw_o_circuit_ciphertext(const aesw_o_circuit *c, uint64_t b, uint8_t ct[16]);
/* (set, first row) where block b was placed */
int aesw_o_circuit_block_placement(const aesw_o_circuit *c, uint64_t b, uint32_t *set,
                                   uint64_t *row);
/* MockProver::assert_satisfied restated: every enabled lookup row is in the
 * table, gate q*(x-c)=0 holds, every copy constraint joins equal values and
 * every queried cell is assigned.  Returns AESW_O_OK or ERR_UNSATISFIED and
 * writes a short description to msg. */
int aesw_o_circuit_verify(const aesw_o_circuit *c, char *msg, size_t msg_len);
/* Test hook: overwrite one advice cell (to show that verify() notices). */
int aesw_o_circuit_poke(aesw_o_circuit *c, uint32_t col, uint64_t row, uint8_t value);

#ifdef __cplusplus
}
#endif
#endif
