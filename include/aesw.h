/*
 * aesw.h -- C ABI of the MI355X batched witness generator for the
 * tkmct/halo2-aes AES-128 gadget.
 *
 * The reference has no FFI at all (no `extern`, no `unsafe`): this header IS
 * the seam a Rust host would bind (INTEGRATION.md shows the `extern "C"` block
 * and the forked chips).  Each entry point names the reference code whose
 * *values* it produces; the host keeps configure()/synthesize() and assigns
 * the bytes returned here verbatim:
 *
 *   value source replaced                         reference file:line
 *   ------------------------------------------   -------------------------------
 *   plaintext literal                             src/aes128.rs:187
 *   U8XorChip::xor      z = x ^ y                 src/chips/u8_xor_chip.rs:85-95
 *   SboxChip::substitute y = S_BOX[x]             src/chips/sbox_chip.rs:73-78
 *   MulBy{2,3}Chip::mul  y = MUL_BY_n[x]          src/chips/gf_mul_chip.rs:75-84
 *   key literal / rcon / zero pads                src/key_schedule.rs:112,161,174,182
 *   every copy_advice target (same value)         src/aes128.rs:285, chips
 *   load_enc_full_table                           src/table.rs:18-192
 *   aes_callable placement                        src/aes128.rs:303-325
 *
 * Conventions: plain pointers and sizes, caller-owned buffers, int status
 * (0 = AESW_OK), no exceptions across the boundary, no global mutable state
 * besides the opaque context.  A context is thread-compatible, not thread-safe.
 * Streams: the *_device entry points are asynchronous on the stream they are given, and independent batches may be
 * issued on SEVERAL streams of one context (from one thread) -- launches only read the context's tables.  That is how
 * a stream of small batches should be run: alone, a 2^16-block launch spends 6 of its 35 us ramping up and draining;
 * round-robin on two or three streams those phases overlap the neighbours' bodies and a launch costs 29 - 30 us, the
 * time of a linear fill of its bytes (bench.py "overlapped_batches", tests/test_gpu_round3.py).
 * The library needs a gfx950 device: there is NO CPU fallback -- every entry
 * point that computes returns AESW_ERR_NO_DEVICE / AESW_ERR_HIP instead.
 *
 * Witness layout ("slab"): for block b the encrypt witness is 1360 rows
 * (AES_ROWS, src/constant.rs:114) of the three advice columns x,y,z of one
 * column set (src/aes128.rs:54-60), in the order the reference's regions are
 * placed (DESIGN.md "slab map").  Column buffers are column-major: column c of
 * block b starts at c_buf + b * aesw_column_stride(layout, c).
 *   AESW_LAYOUT_DENSE : stride 1360 for x,y,z; a cell the reference never
 *                       assigns holds 0 (what the prover sees).
 *   AESW_LAYOUT_PACKED: only assigned cells, in row order: strides 1360/1056/608;
 *                       aesw_packed_index() gives dense row -> packed index.
 *   AESW_LAYOUT_VALUES: only the cells whose VALUE a chip closure computes: y of
 *                       the S-box and mul rows (src/chips/sbox_chip.rs:73-78,
 *                       gf_mul_chip.rs:75-84), z of the xor rows
 *                       (u8_xor_chip.rs:85-95), in row order: strides 0/448/608.
 *                       Column x and the y cells of xor rows are copy_advice() of
 *                       earlier cells in the reference, so a host that keeps the
 *                       chips never reads them: 2.9x fewer bytes
 *                       over PCIe.  x pointers are ignored; aesw_layout_index()
 *                       gives dense row -> index.  Key slabs use the PACKED form.
 * The key-schedule witness per key is words_column (96 rows,
 * src/key_schedule.rs:98-187) plus 400 rows of set 0's x,y,z
 * (dense 400/400/400, packed 400/240/200).
 */
#ifndef AESW_H
#define AESW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AESW_VERSION 102 /* 0.1.2: + aesw_check_witness_device, aesw_check_report (0.1.1: aesw_columns_alloc / _free, aesw_encrypt_witness_batches_device, aesw_batch) */

#define AESW_AES_ROWS 1360u          /* src/constant.rs:114 */
#define AESW_KEY_SCHEDULE_ROWS 1760u /* src/constant.rs:113 (capacity constant only) */
#define AESW_KEY_ROWS 400u           /* rows schedule_keys() really uses in set 0 */
#define AESW_WORDS_ROWS 96u          /* rows used in words_column */
#define AESW_TABLE_ROWS 66561u       /* src/table.rs:27-187 */
#define AESW_FR_BYTES 32u            /* one bn256::Fr cell, little-endian Montgomery */

enum aesw_status {
    AESW_OK = 0,
    AESW_ERR_INVALID_ARG = 1,
    AESW_ERR_NO_DEVICE = 2, /* no gfx950 GPU / device index out of range */
    AESW_ERR_HIP = 3,       /* a HIP runtime call failed; see aesw_last_error() */
    AESW_ERR_NOMEM = 4,
    AESW_ERR_CAPACITY = 5,  /* the reference panics: "AES calls too many" src/aes128.rs:160-162 */
    AESW_ERR_NO_KEY = 6,    /* the reference panics: "Keys should be scheduled" src/aes128.rs:170 */
    AESW_ERR_MISMATCH = 7,  /* host value disagrees with the device witness (Error::Synthesis) */
    AESW_ERR_UNSATISFIED = 8,
    AESW_ERR_COMM = 9       /* RCCL is missing or a collective call failed; see aesw_comm_last_error() */
};

enum aesw_layout { AESW_LAYOUT_DENSE = 0, AESW_LAYOUT_PACKED = 1, AESW_LAYOUT_VALUES = 2 };
enum aesw_column { AESW_COL_X = 0, AESW_COL_Y = 1, AESW_COL_Z = 2 };

typedef struct aesw_ctx aesw_ctx;

/* Optional key-schedule witness outputs of a per-block-key encrypt call. */
typedef struct aesw_key_slab {
    uint8_t *w;  /* n * 96 */
    uint8_t *kx; /* n * aesw_key_column_stride(layout, 0) */
    uint8_t *ky;
    uint8_t *kz;
} aesw_key_slab;

int aesw_version(void);
const char *aesw_strerror(int status);
/* Text of the last HIP failure seen by this context ("" if none). */
const char *aesw_last_error(const aesw_ctx *ctx);
int aesw_device_count(int *count);

/* Tables come IN from the host's constants (src/constant.rs:1-47) so the
 * device is bit-exact with whatever the host's lookup table holds, including
 * the reference's S_BOX[255]==23.  device = HIP device ordinal. */
int aesw_create(aesw_ctx **out, int device, const uint8_t sbox[256], const uint8_t mul2[256],
                const uint8_t mul3[256]);
void aesw_destroy(aesw_ctx *ctx);
int aesw_device(const aesw_ctx *ctx);

/* ---- geometry (pure host, no device needed) ------------------------------ */
uint32_t aesw_column_stride(int layout, int col);     /* encrypt slab, bytes per block */
uint32_t aesw_key_column_stride(int layout, int col); /* key slab, bytes per key */
/* dense row -> packed index (or -1 if the reference never assigns the cell) */
int aesw_packed_index(int col, int32_t idx[AESW_AES_ROWS]);
/* the same for any layout (DENSE: identity; VALUES: -1 for cells the layout leaves out) */
int aesw_layout_index(int layout, int col, int32_t idx[AESW_AES_ROWS]);
int aesw_key_packed_index(int col, int32_t idx[AESW_KEY_ROWS]);
/* FixedAes128Config::aes_callable (src/aes128.rs:303-325) as a pure function:
 * (set, first row) of the b-th encrypt() call in a K/N circuit; set 0 blocks
 * start behind the 400 key rows.  AESW_ERR_CAPACITY when the reference panics. */
int aesw_block_placement(uint32_t k, uint32_t n_sets, uint64_t b, uint32_t *set, uint64_t *row);
uint64_t aesw_block_capacity(uint32_t k, uint32_t n_sets);
/* Fixed (input-independent) selector data for keygen: per slab row the Tag
 * (src/table.rs:10-16: 1 U8, 2 Xor, 3 Sbox, 4 GfMul2, 5 GfMul3, 0 = no lookup)
 * of the chip whose selector the reference enables there; for words_column the
 * rows where q_eq_rcon is on and the round constant in the fixed column
 * (src/key_schedule.rs:161-175).  Any output may be NULL. */
int aesw_selector_tags(uint8_t enc_tag[AESW_AES_ROWS], uint8_t key_tag[AESW_KEY_ROWS],
                       uint8_t q_eq_rcon[AESW_WORDS_ROWS], uint8_t rcon_fixed[AESW_WORDS_ROWS]);

/* The selector columns and the fixed round-constant column of a whole
 * FixedAes128Config<K, n_sets> circuit holding n_blocks encrypt() calls, as
 * keygen lays them out (configure() order: per set range, xor, sbox, mul2, mul3
 * (src/aes128.rs:63-68), then q_eq_rcon (src/key_schedule.rs:50)): selectors is
 * (5*n_sets+1) columns of 2^k bytes (0/1), fixed is 2^k bytes.  Pure host,
 * input independent.  AESW_ERR_CAPACITY when n_blocks does not fit. */
int aesw_assemble_selectors(uint32_t k, uint32_t n_sets, uint64_t n_blocks, uint8_t *selectors,
                            uint8_t *fixed);

/* Equality constraints for keygen, input independent: every copy_advice() of one encrypt() call
 * (src/aes128.rs:154-301, AESW_BLOCK_COPIES edges) and of schedule_keys() (src/key_schedule.rs:80-224,
 * AESW_KEY_COPIES edges), in the reference's call order.  A cell is (space, col, row): space 0 = the block's
 * slab (columns x/y/z of its column set, block-relative row), 1 = the key slab (columns x/y/z of set 0,
 * rows 0..399), 2 = words_column (rows 0..95); col is 0..2 (0 for words_column). */
typedef struct aesw_copy_edge {
    uint8_t dst_space, dst_col;
    uint16_t dst_row;
    uint8_t src_space, src_col;
    uint16_t src_row;
} aesw_copy_edge;
#define AESW_BLOCK_COPIES 1952u
#define AESW_KEY_COPIES 640u
int aesw_block_copy_graph(aesw_copy_edge edges[AESW_BLOCK_COPIES]);
int aesw_key_copy_graph(aesw_copy_edge edges[AESW_KEY_COPIES]);

/* ---- device-pointer entry points (asynchronous on `stream`) -------------- */
/* stream is a hipStream_t passed as void* (NULL = the default stream).
 * All pointers are device pointers on aesw_device(ctx); column buffers must be
 * 16-byte aligned (128-byte aligned for full speed: stores leave as whole lines).
 *
 * FixedAes128Config::schedule_key (src/aes128.rs:143-152): expands one key on
 * the device, keeps its 11 round keys inside the context for later
 * aesw_encrypt_witness_device(..., d_keys = NULL, per_block_keys = 0) calls,
 * and optionally emits its key-schedule witness (one key slab).  This is the
 * reference's call shape: schedule_key once, encrypt many times
 * (benches/aes128.rs:50-53).
 *
 * What is guaranteed (the reference's `self.keys = Some(..)`, src/aes128.rs:143-152, replaces the key atomically between
 * encrypt calls; here "between" means HOST CALL ORDER on the context, whatever streams the calls name):
 *  - A scheduled-key launch (d_keys = NULL) uses the key of the last aesw_schedule_key* call that RETURNED before the launch
 *    was enqueued -- never a later one, even if the later schedule's kernel runs first or concurrently on another stream: round
 *    keys live in slots, a schedule takes the next slot of a ring ("key_slots", default 4) and the launch keeps the pointer of
 *    the slot that was current at enqueue time.
 *  - The launch is ordered behind the key kernel that fills its slot (same stream: stream order; another stream: an event, no
 *    host wait).
 *  - A slot is rewritten only behind EVERY launch that reads it, on any number of streams, including the internal streams of
 *    aesw_encrypt_witness_batches_device: one event per distinct reader stream since the slot was written, all of them waited on
 *    (on the scheduling stream, no host wait) by the schedule that takes the slot again.  More than 16 distinct reader streams
 *    per slot are folded (the 17th stream waits for the first one's launches AFTER its own launch).  Streams that carried
 *    scheduled-key launches may be destroyed at any time (hipStreamDestroy drains a stream in the ROCm runtime).
 *  - Under hipGraph capture nothing can be tracked per replay, so slots are frozen instead: a schedule captured into a graph
 *    writes a slot of its own, and a slot read by a captured launch is pinned -- in both cases the ring never hands that slot
 *    out again (256 B each, until aesw_destroy).  A captured launch therefore reads, on every replay, the key that was current
 *    when it was CAPTURED; re-scheduling on the context does not change what an existing graph encrypts with, and needs no
 *    synchronisation with its replays.  What remains the caller's: a graph that contains a schedule must be launched (and
 *    ordered, by the caller) before un-captured launches that are to read that key run.
 *  - A scheduled-key launch captured on a stream OTHER than the one its key was scheduled on cannot take a dependency on the
 *    key kernel: it is accepted when that kernel has already finished (synchronise first), or when its stream was forked from
 *    the capture of the key's own stream (same graph: the internal streams of aesw_encrypt_witness_batches_device are), and
 *    refused with AESW_ERR_INVALID_ARG otherwise.
 * All launch attributes (dynamic LDS sizes) are set by aesw_create(): launches never change function
 * attributes, so every *_device entry point may be captured into a hipGraph
 * (hipStreamBeginCapture on `stream`) and replayed. */
int aesw_schedule_key_device(aesw_ctx *ctx, const uint8_t *d_key, int layout,
                             const aesw_key_slab *d_key_slab, void *stream);
/* d_keys: n*16 B when per_block_keys; 16 B (one key, expanded inside the call)
 * or NULL (use the key of aesw_schedule_key*; AESW_ERR_NO_KEY if there is none,
 * where the reference panics "Keys should be scheduled") otherwise.
 * d_ct and d_key_slab are optional (NULL).
 * Alignment: every output pointer must be 16-byte aligned (AESW_ERR_INVALID_ARG otherwise) and SHOULD be 128-byte aligned:
 * a wave writes whole 128-byte lines of its 16-block range, and a base that is only 16-byte aligned makes every one of them
 * straddle two lines -- 76 us instead of 40 for a 2^16-block launch (64-byte aligned: 44; beyond 128 nothing more;
 * examples/aesw_batches.c with AESW_EXAMPLE_ALIGN).  hipMalloc, aesw_columns_alloc (2 MiB) and torch tensors are aligned;
 * sub-ranges of one buffer at b * n * stride offsets are not, unless n is a multiple of 8. */
int aesw_encrypt_witness_device(aesw_ctx *ctx, const uint8_t *d_pt, const uint8_t *d_keys,
                                int per_block_keys, uint64_t n, int layout, uint8_t *d_x,
                                uint8_t *d_y, uint8_t *d_z, uint8_t *d_ct,
                                const aesw_key_slab *d_key_slab, void *stream);
/* Many independent batches in one call: `count` batches (each its own inputs and output columns; the reference's
 * counterpart is one FixedAes128Config::encrypt loop per circuit, src/aes128.rs:154-265) are issued round-robin on the
 * context's "batch_streams" internal streams (default 3), ordered behind everything already on `stream`, and joined back
 * into `stream` before the call returns (asynchronously: events, no host wait).  The ramp and tail of one launch then
 * overlap the bodies of its neighbours: 29.5 us instead of 35 per 2^16-block batch, 588 instead of 602 per 2^20-block batch
 * with per-block keys -- the time of a linear fill of the bytes.  Semantics per batch are those of
 * aesw_encrypt_witness_device (d_keys NULL = the scheduled key, 16 B, or n*16 B when per_block_keys; d_ct, d_key_slab
 * optional).  Output ranges of different batches must not overlap.  Capturable like the other *_device calls (the
 * internal streams join the capture through the fork / join events), except with the scheduled key, which can only be
 * captured on its own stream. */
typedef struct aesw_batch {
    const uint8_t *d_pt;   /* n * 16 */
    const uint8_t *d_keys; /* NULL, 16 B or n * 16 */
    uint64_t n;
    uint8_t *d_x;
    uint8_t *d_y;
    uint8_t *d_z;
    uint8_t *d_ct;               /* n * 16, or NULL */
    const aesw_key_slab *d_key_slab; /* or NULL */
} aesw_batch;
int aesw_encrypt_witness_batches_device(aesw_ctx *ctx, const aesw_batch *batches, uint32_t count,
                                        int per_block_keys, int layout, void *stream);
/* src/key_schedule.rs:80-224 for n keys.  d_rk optional: n*176 round-key bytes. */
int aesw_key_schedule_witness_device(aesw_ctx *ctx, const uint8_t *d_keys, uint64_t n, int layout,
                                     uint8_t *d_w, uint8_t *d_kx, uint8_t *d_ky, uint8_t *d_kz,
                                     uint8_t *d_rk, void *stream);
/* src/table.rs:18-192: the four table columns as bytes (every value < 256). */
int aesw_lookup_table_device(aesw_ctx *ctx, uint8_t *d_t0, uint8_t *d_t1, uint8_t *d_t2,
                             uint8_t *d_t3, void *stream);
/* MockProver::assert_satisfied for a batch where it lies (the reference's only executable correctness checks,
 * src/aes128.rs:409-418 and src/key_schedule.rs:385-392, verify constraint SATISFACTION: every enabled lookup has a row in the
 * table of src/table.rs:18-192, the "Equality RC" gate q * (words - rcon) holds, src/key_schedule.rs:59-64, and the two cells
 * of every copy_advice() are equal).  Per block: its 1 360 rows against the chip enabled on each (aesw_selector_tags), its
 * 1 952 copies (aesw_block_copy_graph; the AddRoundKey rows copy from the key slab), rows 0..15 of x against d_pt and, when
 * d_ct is given, the last xor rows against it.  Per key slab: 400 rows, 640 copies (aesw_key_copy_graph), the round-constant
 * gate, and words_column rows 0..15 against the key when d_keys is given.  Nothing is recomputed: a witness that satisfies
 * all of it is one the reference's circuit accepts for these inputs.
 * d_keys: NULL, 16 B (one key) or n * 16 with per_block_keys; d_key_slab (REQUIRED): one key slab, or n with per_block_keys.
 * layout: DENSE or PACKED (VALUES holds no x column: AESW_ERR_INVALID_ARG).  d_report: device memory, written on `stream`
 * (zeroed by the call); read it back after synchronising.  Read-bound, ~1.5 ms per 2^20 blocks; capturable. */
typedef struct aesw_check_report {
    uint64_t blocks;          /* block slabs checked */
    uint64_t keys;            /* key slabs checked (1, or n with per-block keys) */
    uint64_t lookup_failures; /* rows whose enabled lookup has no table row */
    uint64_t copy_failures;   /* copy_advice() pairs whose cells differ */
    uint64_t gate_failures;   /* q_eq_rcon rows whose words_column cell is not the round constant */
    uint64_t input_failures;  /* plaintext / key / ciphertext literal rows that differ from the inputs given */
    uint64_t first;           /* AESW_CHECK_NONE, or the smallest failing check: decode with the macros below */
} aesw_check_report;
#define AESW_CHECK_NONE UINT64_MAX
#define AESW_CHECK_UNIT(f) ((f) >> 20)              /* block index (key index for a key slab) */
#define AESW_CHECK_IS_KEY_SLAB(f) (((f) >> 19) & 1u)
#define AESW_CHECK_KIND(f) (((f) >> 16) & 7u)       /* 1 lookup, 2 copy, 3 gate, 4 input */
#define AESW_CHECK_INDEX(f) ((f) & 0xffffu)         /* slab row; for a copy the edge number in aesw_block_copy_graph / aesw_key_copy_graph */
int aesw_check_witness_device(aesw_ctx *ctx, const uint8_t *d_pt, const uint8_t *d_keys, int per_block_keys,
                              uint64_t n, int layout, const uint8_t *d_x, const uint8_t *d_y,
                              const uint8_t *d_z, const uint8_t *d_ct, const aesw_key_slab *d_key_slab,
                              aesw_check_report *d_report, void *stream);

/* Byte cells -> bn256::Fr cells (what Fp::from(u64) builds, src/utils.rs:23,
 * src/aes128.rs:187): 32-byte little-endian Montgomery form, n_cells*32 B out. */
int aesw_expand_fr_device(aesw_ctx *ctx, const uint8_t *d_cells, uint64_t n_cells, uint8_t *d_fr,
                          void *stream);

/* The advice columns of a FixedAes128Config<K, n_sets> circuit exactly as the
 * prover holds them after synthesize(): 3*n_sets+1 columns of 2^k cells (set i
 * -> columns 3i, 3i+1, 3i+2; words_column last, src/aes128.rs:54-60,
 * src/key_schedule.rs:48), the key rows and block b placed where
 * aesw_block_placement() says, never-assigned cells 0.  Inputs: n_blocks slabs
 * (d_x, d_y, d_z in `layout`) and one key slab (optional).  as_fr = 0: one byte
 * per cell; as_fr = 1: 32-byte little-endian Montgomery bn256::Fr per cell, so
 * the host can bulk-copy a column into halo2's advice polynomial.
 * d_out holds (3*n_sets+1) << k cells, column after column.
 * AESW_ERR_CAPACITY when n_blocks does not fit (the reference panics). */
int aesw_assemble_advice_device(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, uint64_t n_blocks,
                                int layout, const uint8_t *d_x, const uint8_t *d_y,
                                const uint8_t *d_z, const aesw_key_slab *d_key_slab, int as_fr,
                                uint8_t *d_out, void *stream);

/* The device-side home of what FixedAes128Config::encrypt (src/aes128.rs:154-265) and schedule_keys
 * (src/key_schedule.rs:80-96) assign: every output column of a batch of n blocks -- x, y, z, optionally the
 * ciphertext and n key slabs (w, kx, ky, kz) -- allocated together ("arena").  The reference has no counterpart (its
 * advice cells live in halo2's WitnessCollection).
 *
 * Placement probing.  Measured on MI355X (profiles/r03_study/README.md): how fast the witness kernel's store pattern --
 * thousands of waves, each filling its own 16-block range of seven columns -- runs depends on WHICH physical memory
 * backs the columns: the same kernel, inputs and virtual layout took 600 us on one allocation and 770 us on the next,
 * stable for the life of an allocation, while a linear fill of the same buffers took 587 us on both.  The property
 * belongs to the combination of columns (256 MiB tiles that all look fast in isolation combine into slow sets) and no
 * virtual-address rule predicts it, so the arena is chosen by measurement.  A UNIT is what one candidate backs: the whole
 * set of columns in one range, or one column ("arena_unit", below).  Per unit up to "arena_probe" candidate backings
 * (default 8 for batches of at least 2^16 blocks; 0 = off: one hipMalloc, columns on 2^arena_align_log2-byte boundaries,
 * default 2 MiB) are built -- a plain hipMalloc, then virtual ranges over physical chunks of 8 / 2 / 32 / 4 MiB
 * (hipMemCreate / hipMemMap) --; a store-only emulation of the kernel's pattern over the units placed so far plus the
 * candidate is timed against a linear fill of the same bytes; the first candidate whose pattern runs within 2.5 % of the
 * fill is kept, otherwise the fastest one; the rest are held until the search is over (or the driver would hand the same
 * memory out again) and then go back.  Default ("arena_unit" 2): whole-set candidates first, and only when none of them
 * passes (batches of 2^18 blocks and more) a second search that places the columns one at a time, largest first; the
 * better of the two is kept.  probe_us / fill_us report the two times for the complete set as finally placed.
 * Probing writes garbage into the (uninitialised) columns, runs on the device's null stream (it synchronises with the
 * caller's blocking streams: allocate outside timed or latency-critical sections), never lets the held candidates take
 * the device's last 16 GB, and takes 0.2 - 1.5 s for a 2^20-block set (up to 5 s on a GPU where no candidate is fast).  Every column
 * starts on a 2 MiB boundary; with probing `base` is only the handle the arena is freed by and `bytes` the sum of its ranges.
 * "arena_probe_budget_ms" (default 3000, 0 = no limit) bounds a search: once it has run that long it builds no further
 * candidate and keeps the best one it has (every unit still gets its first candidate).
 * Placement cache ("arena_cache", default 1).  What the search pays for is a physical placement, and freed memory comes back
 * from the driver in some other combination, so aesw_columns_free of a PROBED arena keeps its ranges mapped inside the
 * context, and the next aesw_columns_alloc with the same n, layout, with_key_slab, with_ct (and "xcd_remap") takes them over:
 * no candidate is built or timed (candidates = 0; probe_us / fill_us are the earlier search's), the call returns in
 * microseconds and the kernel runs at the rate measured then.  This is the shape of the consumer: one synthesize() per proof,
 * three passes per proof (keygen_vk, keygen_pk, create_proof: benches/aes128.rs:80-107), batch after batch of one size.
 * At most "arena_cache_max_mb" (default 65536) stay cached, oldest out first; everything cached is released when a search
 * for a new shape finds less free memory than twice its size + 16 GB, by aesw_set_option("arena_cache", 0) and by
 * aesw_destroy.  Arenas allocated with probing off are plain hipMalloc / hipFree and never cached.
 * Unused members are NULL.  aesw_columns_free releases the arena (or hands it to the cache) and clears the struct; cols must
 * come from aesw_columns_alloc on the same context. */
typedef struct aesw_columns {
    uint8_t *base;  /* the allocation (arena_probe 0) / the handle of the arena */
    uint64_t bytes; /* device memory held */
    uint8_t *x;     /* n * aesw_column_stride(layout, 0); NULL for AESW_LAYOUT_VALUES */
    uint8_t *y;
    uint8_t *z;
    uint8_t *ct;    /* n * 16, or NULL */
    aesw_key_slab key; /* n key slabs, or NULLs */
    uint32_t candidates; /* candidate backings built and timed in all, over every unit (0: probing off, or taken from the placement cache) */
    uint32_t chosen;     /* reserved (0) */
    float probe_us;      /* store-pattern emulation over the set as placed, microseconds per pass */
    float fill_us;       /* a linear fill of the same bytes */
} aesw_columns;
/* with_key_slab: 0 = encrypt columns only, 1 = + n key slabs, 2 = the n key slabs alone (for aesw_key_schedule_witness_device) */
int aesw_columns_alloc(aesw_ctx *ctx, uint64_t n, int layout, int with_key_slab, int with_ct,
                       aesw_columns *out);
int aesw_columns_free(aesw_ctx *ctx, aesw_columns *cols);

/* ---- host-pointer entry points (synchronous) ----------------------------- */
/* Same contracts with host buffers.  Blocks are cut into chunks ("chunk_blocks"
 * option); chunk i's kernel runs while chunk i-1's columns travel D2H on a
 * second stream (config 5 of BASELINE.json).  Output buffers obtained from
 * aesw_host_alloc() are page-locked and receive the DMA directly; ordinary
 * (pageable) buffers go through page-locked bounce buffers owned by the context. */
void *aesw_host_alloc(size_t bytes);
void aesw_host_free(void *p);
int aesw_encrypt_witness(aesw_ctx *ctx, const uint8_t *pt, const uint8_t *keys, int per_block_keys,
                         uint64_t n, int layout, uint8_t *x, uint8_t *y, uint8_t *z, uint8_t *ct,
                         const aesw_key_slab *key_slab);
/* Streaming form (BASELINE.json configs[4]: D2H overlapped with the host's assign loop): the batch
 * is produced chunk by chunk; `consume` is called on the calling thread for each chunk, in block
 * order, with page-locked buffers holding the chunk's columns, WHILE the next chunk's kernel and
 * D2H are in flight.  The pointers are valid only during the call.  A non-zero return from
 * `consume` aborts the stream and is returned as AESW_ERR_MISMATCH.  keys as in
 * aesw_encrypt_witness (NULL = scheduled key). */
typedef int (*aesw_chunk_fn)(void *user, uint64_t first_block, uint64_t n_blocks, const uint8_t *x,
                             const uint8_t *y, const uint8_t *z);
int aesw_encrypt_witness_stream(aesw_ctx *ctx, const uint8_t *pt, const uint8_t *keys,
                                int per_block_keys, uint64_t n, int layout, aesw_chunk_fn consume,
                                void *user);
/* Where the time of the last streaming call on this context went (aesw_encrypt_witness_stream,
 * aesw_assemble_advice_stream): device time of the chunks' kernels and of their device-to-host
 * copies (HIP events, summed over chunks; the two overlap each other and the consumer), host time
 * spent inside `consume` and host time spent waiting for a chunk, and the call's wall time. */
typedef struct aesw_stream_stats {
    uint64_t chunks;
    uint64_t bytes_to_host;
    uint64_t kernel_ns;
    uint64_t d2h_ns;
    uint64_t consumer_ns;
    uint64_t wait_ns;
    uint64_t wall_ns;
} aesw_stream_stats;
int aesw_last_stream_stats(const aesw_ctx *ctx, aesw_stream_stats *out);
/* The advice columns of a FixedAes128Config<K, n_sets> circuit delivered to the HOST, column by
 * column (the host bulk-copies whole columns into halo2's advice polynomials
 * instead of one assign_advice per cell, src/utils.rs:23, src/aes128.rs:187).  Inputs as for
 * aesw_assemble_advice_device (device slabs, already complete: synchronise the stream that
 * produced them first).  `consume` is called on the calling thread for columns 0 .. 3*n_sets in
 * order with a page-locked buffer of 2^k cells (one byte each, or 32-byte little-endian
 * Montgomery bn256::Fr cells when as_fr), valid only during the call, WHILE the next column is
 * being assembled and copied.  A non-zero return aborts the stream (AESW_ERR_MISMATCH). */
typedef int (*aesw_column_fn)(void *user, uint32_t column, const uint8_t *cells, uint64_t n_cells);
int aesw_assemble_advice_stream(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, uint64_t n_blocks,
                                int layout, const uint8_t *d_x, const uint8_t *d_y,
                                const uint8_t *d_z, const aesw_key_slab *d_key_slab, int as_fr,
                                aesw_column_fn consume, void *user);
/* The same columns into ONE host buffer, column after column ((3*n_sets+1) << k cells): the DMA goes
 * straight into `out` when it is page-locked -- from aesw_host_alloc(), or the host's own advice
 * polynomial memory pinned with aesw_host_register() -- and through the context's bounce buffers
 * otherwise.  Synchronous; column j+1 is assembled while column j travels. */
int aesw_assemble_advice_host(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, uint64_t n_blocks,
                              int layout, const uint8_t *d_x, const uint8_t *d_y,
                              const uint8_t *d_z, const aesw_key_slab *d_key_slab, int as_fr,
                              uint8_t *out);
/* Page-lock / release memory the host already owns (hipHostRegister), so D2H lands in it directly. */
int aesw_host_register(void *p, size_t bytes);
int aesw_host_unregister(void *p);
int aesw_key_schedule_witness(aesw_ctx *ctx, const uint8_t *keys, uint64_t n, int layout,
                              uint8_t *w, uint8_t *kx, uint8_t *ky, uint8_t *kz, uint8_t *rk);
/* With option "stream_check" = 1, aesw_encrypt_witness_stream runs aesw_check_witness_device over every chunk on the device -- behind
 * the kernel that produced it, while the previous chunk travels to the host -- so the whole stream is certified against the
 * reference's constraints without the consumer doing anything (BASELINE configs[4]; DENSE and PACKED layouts; ~70 us per 2^15-block
 * chunk, hidden under its 1.8 ms of D2H).  The sum over the chunks, with batch-wide unit numbers in `first`, is read back here after
 * the stream call has returned (blocks = 0 when the option was off or the layout was VALUES). */
int aesw_last_stream_check(const aesw_ctx *ctx, aesw_check_report *out);
/* host-pointer aesw_check_witness_device: every pointer is a host buffer, *report is written on the host when the call returns.
 * The batch is uploaded and checked in stages of "chunk_blocks" blocks; units in report->first are batch-wide indices. */
int aesw_check_witness(aesw_ctx *ctx, const uint8_t *pt, const uint8_t *keys, int per_block_keys, uint64_t n,
                       int layout, const uint8_t *x, const uint8_t *y, const uint8_t *z, const uint8_t *ct,
                       const aesw_key_slab *key_slab, aesw_check_report *report);
/* host-pointer aesw_schedule_key_device: key is 16 host bytes, key_slab host buffers (optional) */
int aesw_schedule_key(aesw_ctx *ctx, const uint8_t key[16], int layout, const aesw_key_slab *key_slab);
int aesw_lookup_table(aesw_ctx *ctx, uint8_t *t0, uint8_t *t1, uint8_t *t2, uint8_t *t3);

/* ---- multi-GPU exchange (one process per GPU, RCCL over xGMI) -------------- */
/* Blocks shard by contiguous index range and need no collective to be generated
 * (SURVEY 8(e)); handing the whole witness to one consumer is ONE gather: every
 * rank's range of a column is a contiguous byte range of the gathered column
 * (BASELINE configs[3]).  The reference has no counterpart (single process).
 * aesw_comm_unique_id: called by one rank; the host carries the 128 bytes to
 * the others (MPI, a socket, ...).  aesw_comm_create is collective over the
 * nranks processes (ncclCommInitRank); rank r drives the device of its ctx.
 * A communicator with nranks = 1 needs no RCCL at all. */
#define AESW_COMM_ID_BYTES 128u
typedef struct aesw_comm aesw_comm;
int aesw_comm_unique_id(uint8_t id[AESW_COMM_ID_BYTES]);
int aesw_comm_create(aesw_ctx *ctx, int nranks, int rank, const uint8_t id[AESW_COMM_ID_BYTES],
                     aesw_comm **out);
void aesw_comm_destroy(aesw_comm *comm);
/* Largest single send/recv in bytes (default 1 GiB): a rank's range of one column travels in pieces. */
int aesw_comm_set_max_message(aesw_comm *comm, uint64_t bytes);
/* Text of the last communicator failure on this thread ("" if none). */
const char *aesw_comm_last_error(void);
/* Pure host: offsets[r] = blocks before rank r's range in the gathered columns, *total = all blocks. */
int aesw_gather_offsets(int nranks, const uint64_t *counts, uint64_t *offsets, uint64_t *total);
/* Gather n_cols columns on `root`, asynchronously on `stream` (so it is ordered
 * behind the kernels that produced the columns on that stream): rank r sends
 * counts[r] * strides[c] bytes of d_send[c]; the root receives rank r's range at
 * d_recv[c] + offsets[r] * strides[c] (its own range is copied device-to-device,
 * or left alone when d_send[c] already points there).  All sends and receives
 * are issued inside one ncclGroupStart/End, so the peers' transfers -- each over
 * its own xGMI link to the root -- run concurrently.  d_recv is read on the root
 * only.  counts has nranks entries on every rank. */
int aesw_gather_columns_device(aesw_comm *comm, int root, int n_cols, const uint8_t *const *d_send,
                               uint8_t *const *d_recv, const uint64_t *counts,
                               const uint32_t *strides, void *stream);

/* ---- tuning / introspection (bench.py, tests) ----------------------------- */
/* name: "waves_shared" / "waves_pbk" (waves per group for launches with one key / with per-block keys; 0 = auto, 1..4,
 * silently limited to what keeps a group's LDS staging below 64 KiB: 3 for the packed layout, 2 dense, 4 values-only),
 * "store_mode" (0 plain, 1 nontemporal: the default, 2 write-through sc1), "nt_stores" (0/1), "grid_cap" (max workgroups per launch,
 * 0 = one per block group), "xcd_remap" (which block groups the workgroups of one XCD take: 0 = dispatch order, 1 = one contiguous eighth of
 * the block groups per XCD (default), C >= 2 = the XCDs take turns in chunks of C groups), "force_table_path" (1), "chunk_blocks" (blocks per stage of the host-pointer
 * pipeline, default 2^15), "batch_streams" (internal streams of aesw_encrypt_witness_batches_device, 1 ... 8, default 3), "split_small" (0 = off: the default; 2 ... 8: a LONE shared- or
 * scheduled-key batch of 2^15 ... 2^17 blocks is dealt as that many sub-ranges of whole 48-block groups onto the internal streams -- an
 * experiment of round 4 that measured 4 - 8 us SLOWER at every size, profiles/r04_study/split_small.md), "stream_check" (0 / 1: check every chunk of aesw_encrypt_witness_stream on the
 * device, result through aesw_last_stream_check; default 0), "stream_poison" (diagnostic for the tests: block index + 1 whose y / z cells the
 * stream overwrites between kernel and check; 0 = off), "key_slots" (round-key slots
 * aesw_schedule_key* cycles through, 1 ... 64, default 4; with 1 every schedule waits for all launches reading the previous key), "copy_threads" (host threads that move a stage from the page-locked bounce buffer into a PAGEABLE destination;
 * -1 = auto: a quarter of the CPUs the process may run on, 1 ... 4; page-locked destinations receive the DMA directly and use none), "lds_pad" (diagnostic: extra LDS bytes per workgroup, lowers residency), "fr_store_mode" / "key_store_mode" (store
 * flavour of the Fr-expanding kernels and of the key-schedule kernel, default 1), "fr_geometry" (0 striding workgroups, 1 one-shot 4 KiB
 * workgroups: the default, 2 one-shot 16 KiB), "assemble_geometry" (Fr form of aesw_assemble_advice_*: 0 striding workgroups,
 * 1 division-free one-shot workgroups on a (chunk, segment, column) grid, 2 / 3 / 4 one-shot workgroups on aligned chunks of a
 * column: 256 threads x 1 piece (4 KiB), 256 x 2 (8 KiB), 128 x 2 (4 KiB): 4 is the default; K < 8 or K > 30 always take the
 * striding kernel), "arena_align_log2" (column alignment of aesw_columns_alloc, 0 = auto), "arena_probe" (candidate backings
 * aesw_columns_alloc measures per unit, -1 = auto: 8 for batches of at least 2^16 blocks, 0 = none: one hipMalloc), "arena_unit" (what a candidate
 * backs: 0 = the whole set of columns in one range; 1 = one column, placed greedily, largest first; 2 = whole sets first, then
 * columns if no whole-set candidate ran the pattern as fast as its fill: the default), "arena_cache" (1 = a freed probed arena keeps its
 * backing for the next aesw_columns_alloc of the same shape: the default; 0 = off, and releases what is cached), "arena_cache_max_mb"
 * (bytes the cache may hold, default 65536), "arena_probe_budget_ms" (wall-time bound of one placement search, default 3000, 0 = none).
 * aesw_get_option reads back every option aesw_set_option accepts, plus "effective_waves_shared" / "effective_waves_pbk" /
 * "effective_waves_key": the group size a packed-layout launch really uses (auto resolved, limits applied; for the key kernel: its witness-only form, without round-key output), and "effective_copy_threads", and the
 * read-only statistics "key_reader_waits" (reader events schedules have waited on), "key_slots_allocated", "key_slots_pinned",
 * "arena_cache_hits", "arena_cached_bytes".
 * Unknown -> INVALID_ARG */
int aesw_set_option(aesw_ctx *ctx, const char *name, int64_t value);
int aesw_get_option(const aesw_ctx *ctx, const char *name, int64_t *value);
/* 1 when mul2/mul3 passed to aesw_create() equal GF(2^8) xtime tables, so the
 * arithmetic MixColumns path is used instead of LDS table lookups. */
int aesw_uses_xtime_path(const aesw_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
