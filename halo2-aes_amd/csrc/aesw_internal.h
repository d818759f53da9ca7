// aesw_internal.h -- kernel parameter blocks and launchers shared by
// aesw_kernels.hip and aesw_api.cpp (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aesw {

struct KeyOut {
    uint8_t *w, *kx, *ky, *kz;  // any may be null
};

struct EncParams {
    const uint8_t *pt;      // n*16
    const uint8_t *keys;    // 16 or n*16; null when rk is used
    const uint32_t *rk;     // 44 words of a key scheduled earlier (aesw_schedule_key_device), or null
    const uint8_t *tables;  // 768: sbox | mul2 | mul3
    const uint32_t *ftab;   // flush descriptors of the layout (build_flush_tables), flush_table_words(layout) words
    uint8_t *x, *y, *z;     // column buffers (16-byte aligned)
    uint8_t *ct;            // n*16 or null
    KeyOut key;             // per-block-key mode only
    uint64_t n;
    uint32_t ngroups;       // block groups (16*waves blocks each); a workgroup strides over them
    uint32_t xcd_remap;     // xcd_group() mode: 0 dispatch order, 1 one contiguous eighth per XCD, C >= 2 turns of C groups
#ifdef AESW_TRACE
    uint64_t *trace;        // tools/trace.py only: 8 x u64 per wave
#endif
};

struct KeyParams {
    const uint8_t *keys;  // n*16
    const uint8_t *tables;
    KeyOut key;
    uint8_t *rk;  // n*176 or null
    uint64_t n;
    uint32_t ngroups;    // filled by the launcher
    uint32_t xcd_remap;  // xcd_group() mode, as in EncParams
};

// key mode: 0 = per-block keys, 1 = shared key expanded in the kernel, 2 = shared key scheduled earlier (p.rk)
hipError_t launch_encrypt(const EncParams &p, int layout, bool xt, int keymode, bool kemit, int waves, int store_mode,
                          uint32_t max_groups_in_flight, uint32_t xcd_remap, uint32_t lds_pad, hipStream_t s);
// flush-descriptor table of a layout (aesw_layout.h "scheduled flush"): size in 32-bit words, and the host-side builder
int flush_table_words(int layout);
void build_flush_tables(int layout, uint32_t *out);
// hipFuncSetAttribute(max dynamic LDS) for every instantiation, once per device: called by aesw_create()
hipError_t warm_launch_attributes();
hipError_t launch_key(const KeyParams &p, int layout, bool xt, int waves, int store_mode, uint32_t xcd_remap, hipStream_t s);
hipError_t launch_table(const uint8_t *tables, uint8_t *t0, uint8_t *t1, uint8_t *t2, uint8_t *t3, hipStream_t s);
struct AssembleParams {
    const uint8_t *x, *y, *z;        // n_blocks slabs
    const uint8_t *kw, *kx, *ky, *kz;  // one key slab (any may be null)
    const void *fr_lut;              // 256 x 32 B
    uint8_t *out;
    uint64_t n_blocks;
    uint32_t k, n_sets;
    uint32_t col_first, col_count;       // the advice columns to write (out holds col_count columns)
    uint32_t sx, sy, sz, kxs, kys, kzs;  // bytes per block / per key
    int packed;
    int geometry;         // Fr cells: 1 = one-shot 4 KiB workgroups on a (chunk, segment, column) grid, 0 = striding workgroups
    uint64_t cap0, capn;  // blocks per column set (set 0 / the others): filled by the launcher
};
hipError_t launch_assemble(const AssembleParams &p, bool as_fr, int store_mode, hipStream_t s);
// placement probe of aesw_columns_alloc: columns x, y, z, w, kx, ky, kz (null = absent) of n blocks
struct ProbeParams {
    uint8_t *col[7];
    uint32_t stride[7];
    uint32_t chunks[7];  // 4 KiB chunks per column (filled by the launcher for the linear fill)
    uint64_t n;
    uint32_t xcd_mode;  // xcd_group() mode of the context
};
hipError_t launch_probe(const ProbeParams &p, bool fill, hipStream_t s);
// aesw_check_witness_device (aesw_check.h): constraint satisfaction of n block slabs and their key slab(s)
struct CheckParams {
    const uint8_t *pt;          // n*16
    const uint8_t *keys;        // null, 16 (one key) or n*16
    const uint8_t *x, *y, *z;   // block columns in `layout` (DENSE or PACKED)
    const uint8_t *ct;          // n*16 or null
    const uint8_t *kw, *kx, *ky, *kz;  // one key slab, or n with per_block_keys
    const uint32_t *table;      // build_check_table(layout): CHK_WORDS words
    const uint8_t *tab768;      // sbox | mul2 | mul3
    uint64_t *report;           // aesw_check_report as 7 x u64
    uint64_t n;
    uint32_t per_block_keys;
    uint32_t skip_shared_key;   // one key slab for the batch: do not check it in this launch (a later chunk of a host-pointer call)
    uint32_t sx, sy, sz, kxs, kys, kzs, bi, img;  // strides, block image bytes, bytes of one wave's image region
};
hipError_t launch_check(const CheckParams &p, hipStream_t s);
hipError_t launch_expand_fr(const uint8_t *cells, uint64_t n_cells, const void *fr_lut, void *out, int store_mode, int geometry, hipStream_t s);

}  // namespace aesw
