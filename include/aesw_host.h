/*
 * aesw_host.h -- C entry points of the host-side mirror of the reference's
 * interface (halo2-aes_amd/host/: FixedAes128Config, Aes128KeyScheduleConfig,
 * the four chips, load_enc_full_table on a minimal halo2-style front end).
 *
 * They run the reference's own circuits with every value closure reading the
 * device witness obtained through include/aesw.h, then expose what
 * synthesize() assigned, so the parity tests can read like the reference's:
 *   aesw_host_aes_circuit_run  = MockProver::run(K, &TestAesCircuit / Aes128BenchCircuit)
 *                                (src/aes128.rs:376-418, benches/aes128.rs:30-61)
 *   aesw_host_key_circuit_run  = MockProver::run(K, &TestCircuit)   (src/key_schedule.rs:245-320, :385-392)
 *   aesw_host_circuit_verify   = mock.assert_satisfied()
 * Status codes are aesw_status (aesw.h): AESW_ERR_CAPACITY / AESW_ERR_NO_KEY
 * where the reference panics, AESW_ERR_MISMATCH for plonk::Error::Synthesis
 * raised because a host value disagrees with the device witness.
 */
#ifndef AESW_HOST_H
#define AESW_HOST_H

#include "aesw.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aesw_host_circuit aesw_host_circuit;

/* load_enc_full_table, schedule_key(key), then encrypt(pts[b]) for b < n, in a
 * FixedAes128Config<K, n_sets> circuit of 2^k rows.  with_witnesses = 0 mimics
 * keygen (value closures are never evaluated).  skip_schedule_key = 1 omits
 * schedule_key() to exercise the reference's expect("Keys should be scheduled").
 * The device witness travels in AESW_LAYOUT_PACKED (assigned cells only;
 * aesw_layout_index places them) unless a mode below says otherwise.
 * assign_mode = 1 (bulk): the first block goes through the reference's 1 360
 * one-row regions, every later block is assigned as ONE 1 360-row region that
 * replays the first block's copy graph; cells, selectors and
 * equality constraints are the same, only the region count differs.
 * assign_mode = 2 (values only): the reference's regions, fed by the
 * AESW_LAYOUT_VALUES witness -- the device hands over only the S-box / mul / xor
 * outputs the value closures read; every other cell takes its value through
 * copy_advice(), as in the reference.
 * assign_mode = 3 (streaming, BASELINE configs[4]): as 2, but the witness arrives
 * through aesw_encrypt_witness_stream: encrypt() calls of chunk i run while the
 * device produces chunk i+1 and copies it to the host.
 * assign_mode = 4: as 0 with the AESW_LAYOUT_DENSE witness (an exact image of
 * the advice rows; 35 % of it is zeros for never-assigned cells). */
int aesw_host_aes_circuit_run(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, const uint8_t key[16],
                              const uint8_t *pts, uint64_t n, int with_witnesses,
                              int skip_schedule_key, int assign_mode, aesw_host_circuit **out);
/* The same circuit WITHOUT running a region: advice columns = the device witness placed by aesw_block_placement, selectors /
 * fixed column / table / equality constraints = the library's input-independent keygen data (aesw_assemble_selectors,
 * aesw_lookup_table, aesw_block_copy_graph, aesw_key_copy_graph).  What a host with bulk column access does; verify() then
 * checks every lookup, the rcon gate and all copy constraints on it.  num_regions is 0. */
int aesw_host_aes_circuit_columns(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, const uint8_t key[16],
                                  const uint8_t *pts, uint64_t n, aesw_host_circuit **out);
/* key_schedule.rs TestCircuit: 3 advice columns + words_column, schedule_keys only. */
int aesw_host_key_circuit_run(aesw_ctx *ctx, uint32_t k, const uint8_t key[16],
                              aesw_host_circuit **out);
void aesw_host_circuit_free(aesw_host_circuit *c);
/* AESW_OK or AESW_ERR_UNSATISFIED (+ first failure in msg). */
int aesw_host_circuit_verify(const aesw_host_circuit *c, char *msg, size_t msg_len);

uint32_t aesw_host_circuit_num_advice(const aesw_host_circuit *c);
uint32_t aesw_host_circuit_num_selectors(const aesw_host_circuit *c);
uint64_t aesw_host_circuit_num_rows(const aesw_host_circuit *c);
uint64_t aesw_host_circuit_num_regions(const aesw_host_circuit *c);
uint64_t aesw_host_circuit_num_copies(const aesw_host_circuit *c);
uint64_t aesw_host_circuit_closure_calls(const aesw_host_circuit *c);
/* equality constraints as (copy column, copy row, original column, original row), 4 x num_copies values */
int aesw_host_circuit_copies(const aesw_host_circuit *c, uint64_t *out);
const uint8_t *aesw_host_circuit_advice(const aesw_host_circuit *c, uint32_t col);
const uint8_t *aesw_host_circuit_advice_assigned(const aesw_host_circuit *c, uint32_t col);
const uint8_t *aesw_host_circuit_selector(const aesw_host_circuit *c, uint32_t sel);
const uint8_t *aesw_host_circuit_fixed(const aesw_host_circuit *c);
const uint8_t *aesw_host_circuit_table(const aesw_host_circuit *c, uint32_t col, uint64_t *rows);
/* values of the cells encrypt() returned for block b */
int aesw_host_circuit_ciphertext(const aesw_host_circuit *c, uint64_t b, uint8_t ct[16]);
/* test hook: overwrite one advice cell */
int aesw_host_circuit_poke(aesw_host_circuit *c, uint32_t col, uint64_t row, uint8_t value);
/* text of the failure of the last *_run call on this thread ("" if none) */
const char *aesw_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
