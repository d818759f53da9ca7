//! Raw FFI bindings to `libaesw.so` -- the C ABI of `include/aesw.h`.
//!
//! The reference (tkmct/halo2-aes) has no FFI: this is the `extern "C"` seam a maintainer adds so that the value
//! closures of `src/chips/*.rs`, `src/aes128.rs` and `src/key_schedule.rs` read bytes a MI355X produced instead of
//! recomputing them per row (see `../halo2-aes-patch`).  Hand-written, one item per item of the header, in the
//! header's order; `tests/test_rust_shim.py` parses this file and `include/aesw.h` and fails when a function, an
//! argument, a pointer's constness, an integer width, a struct field or a constant differs.
//!
//! Not compiled in the repository's build image (no cargo / rustc there); it needs nothing beyond `core`/`std`.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const AESW_VERSION: c_int = 102;

pub const AESW_AES_ROWS: u32 = 1360; // src/constant.rs:114
pub const AESW_KEY_SCHEDULE_ROWS: u32 = 1760; // src/constant.rs:113
pub const AESW_KEY_ROWS: u32 = 400;
pub const AESW_WORDS_ROWS: u32 = 96;
pub const AESW_TABLE_ROWS: u32 = 66561; // src/table.rs:27-187
pub const AESW_FR_BYTES: u32 = 32;
pub const AESW_BLOCK_COPIES: u32 = 1952;
pub const AESW_KEY_COPIES: u32 = 640;
pub const AESW_COMM_ID_BYTES: u32 = 128;

// enum aesw_status
pub const AESW_OK: c_int = 0;
pub const AESW_ERR_INVALID_ARG: c_int = 1;
pub const AESW_ERR_NO_DEVICE: c_int = 2;
pub const AESW_ERR_HIP: c_int = 3;
pub const AESW_ERR_NOMEM: c_int = 4;
pub const AESW_ERR_CAPACITY: c_int = 5; // the reference panics: "AES calls too many", src/aes128.rs:160-162
pub const AESW_ERR_NO_KEY: c_int = 6; // the reference panics: "Keys should be scheduled", src/aes128.rs:170
pub const AESW_ERR_MISMATCH: c_int = 7;
pub const AESW_ERR_UNSATISFIED: c_int = 8;
pub const AESW_ERR_COMM: c_int = 9;

// enum aesw_layout
pub const AESW_LAYOUT_DENSE: c_int = 0;
pub const AESW_LAYOUT_PACKED: c_int = 1;
pub const AESW_LAYOUT_VALUES: c_int = 2;

// enum aesw_column
pub const AESW_COL_X: c_int = 0;
pub const AESW_COL_Y: c_int = 1;
pub const AESW_COL_Z: c_int = 2;

/// Opaque `aesw_ctx`.
#[repr(C)]
pub struct aesw_ctx {
    _private: [u8; 0],
}

/// Opaque `aesw_comm`.
#[repr(C)]
pub struct aesw_comm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aesw_key_slab {
    pub w: *mut u8,
    pub kx: *mut u8,
    pub ky: *mut u8,
    pub kz: *mut u8,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct aesw_copy_edge {
    pub dst_space: u8,
    pub dst_col: u8,
    pub dst_row: u16,
    pub src_space: u8,
    pub src_col: u8,
    pub src_row: u16,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct aesw_stream_stats {
    pub chunks: u64,
    pub bytes_to_host: u64,
    pub kernel_ns: u64,
    pub d2h_ns: u64,
    pub consumer_ns: u64,
    pub wait_ns: u64,
    pub wall_ns: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aesw_columns {
    pub base: *mut u8,
    pub bytes: u64,
    pub x: *mut u8,
    pub y: *mut u8,
    pub z: *mut u8,
    pub ct: *mut u8,
    pub key: aesw_key_slab,
    pub candidates: u32,
    pub chosen: u32,
    pub probe_us: f32,
    pub fill_us: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct aesw_batch {
    pub d_pt: *const u8,
    pub d_keys: *const u8,
    pub n: u64,
    pub d_x: *mut u8,
    pub d_y: *mut u8,
    pub d_z: *mut u8,
    pub d_ct: *mut u8,
    pub d_key_slab: *const aesw_key_slab,
}

/// Report of `aesw_check_witness_device` (device memory; copy it back after synchronising the stream).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct aesw_check_report {
    pub blocks: u64,
    pub keys: u64,
    pub lookup_failures: u64,
    pub copy_failures: u64,
    pub gate_failures: u64,
    pub input_failures: u64,
    pub first: u64,
}
pub const AESW_CHECK_NONE: u64 = u64::MAX;
impl aesw_check_report {
    /// MockProver::assert_satisfied: no lookup, copy, gate or literal row failed.
    pub fn satisfied(&self) -> bool {
        self.lookup_failures == 0 && self.copy_failures == 0 && self.gate_failures == 0 && self.input_failures == 0
    }
    /// (unit, is_key_slab, kind: 1 lookup / 2 copy / 3 gate / 4 input, row or edge number) of the smallest failing check.
    pub fn first_failure(&self) -> Option<(u64, bool, u32, u32)> {
        if self.first == AESW_CHECK_NONE {
            return None;
        }
        Some((self.first >> 20, (self.first >> 19) & 1 == 1, ((self.first >> 16) & 7) as u32, (self.first & 0xffff) as u32))
    }
}

pub type aesw_chunk_fn = unsafe extern "C" fn(
    user: *mut c_void,
    first_block: u64,
    n_blocks: u64,
    x: *const u8,
    y: *const u8,
    z: *const u8,
) -> c_int;

pub type aesw_column_fn =
    unsafe extern "C" fn(user: *mut c_void, column: u32, cells: *const u8, n_cells: u64) -> c_int;

extern "C" {
    pub fn aesw_version() -> c_int;
    pub fn aesw_strerror(status: c_int) -> *const c_char;
    pub fn aesw_last_error(ctx: *const aesw_ctx) -> *const c_char;
    pub fn aesw_device_count(count: *mut c_int) -> c_int;

    pub fn aesw_create(
        out: *mut *mut aesw_ctx,
        device: c_int,
        sbox: *const u8,
        mul2: *const u8,
        mul3: *const u8,
    ) -> c_int;
    pub fn aesw_destroy(ctx: *mut aesw_ctx);
    pub fn aesw_device(ctx: *const aesw_ctx) -> c_int;

    // ---- geometry (pure host) ----
    pub fn aesw_column_stride(layout: c_int, col: c_int) -> u32;
    pub fn aesw_key_column_stride(layout: c_int, col: c_int) -> u32;
    pub fn aesw_packed_index(col: c_int, idx: *mut i32) -> c_int;
    pub fn aesw_layout_index(layout: c_int, col: c_int, idx: *mut i32) -> c_int;
    pub fn aesw_key_packed_index(col: c_int, idx: *mut i32) -> c_int;
    pub fn aesw_block_placement(k: u32, n_sets: u32, b: u64, set: *mut u32, row: *mut u64) -> c_int;
    pub fn aesw_block_capacity(k: u32, n_sets: u32) -> u64;
    pub fn aesw_selector_tags(
        enc_tag: *mut u8,
        key_tag: *mut u8,
        q_eq_rcon: *mut u8,
        rcon_fixed: *mut u8,
    ) -> c_int;
    pub fn aesw_assemble_selectors(
        k: u32,
        n_sets: u32,
        n_blocks: u64,
        selectors: *mut u8,
        fixed: *mut u8,
    ) -> c_int;
    pub fn aesw_block_copy_graph(edges: *mut aesw_copy_edge) -> c_int;
    pub fn aesw_key_copy_graph(edges: *mut aesw_copy_edge) -> c_int;

    // ---- device-pointer entry points (asynchronous on `stream`, a hipStream_t) ----
    pub fn aesw_schedule_key_device(
        ctx: *mut aesw_ctx,
        d_key: *const u8,
        layout: c_int,
        d_key_slab: *const aesw_key_slab,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_encrypt_witness_device(
        ctx: *mut aesw_ctx,
        d_pt: *const u8,
        d_keys: *const u8,
        per_block_keys: c_int,
        n: u64,
        layout: c_int,
        d_x: *mut u8,
        d_y: *mut u8,
        d_z: *mut u8,
        d_ct: *mut u8,
        d_key_slab: *const aesw_key_slab,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_key_schedule_witness_device(
        ctx: *mut aesw_ctx,
        d_keys: *const u8,
        n: u64,
        layout: c_int,
        d_w: *mut u8,
        d_kx: *mut u8,
        d_ky: *mut u8,
        d_kz: *mut u8,
        d_rk: *mut u8,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_lookup_table_device(
        ctx: *mut aesw_ctx,
        d_t0: *mut u8,
        d_t1: *mut u8,
        d_t2: *mut u8,
        d_t3: *mut u8,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_expand_fr_device(
        ctx: *mut aesw_ctx,
        d_cells: *const u8,
        n_cells: u64,
        d_fr: *mut u8,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_assemble_advice_device(
        ctx: *mut aesw_ctx,
        k: u32,
        n_sets: u32,
        n_blocks: u64,
        layout: c_int,
        d_x: *const u8,
        d_y: *const u8,
        d_z: *const u8,
        d_key_slab: *const aesw_key_slab,
        as_fr: c_int,
        d_out: *mut u8,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_encrypt_witness_batches_device(
        ctx: *mut aesw_ctx,
        batches: *const aesw_batch,
        count: u32,
        per_block_keys: c_int,
        layout: c_int,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_check_witness_device(
        ctx: *mut aesw_ctx,
        d_pt: *const u8,
        d_keys: *const u8,
        per_block_keys: c_int,
        n: u64,
        layout: c_int,
        d_x: *const u8,
        d_y: *const u8,
        d_z: *const u8,
        d_ct: *const u8,
        d_key_slab: *const aesw_key_slab,
        d_report: *mut aesw_check_report,
        stream: *mut c_void,
    ) -> c_int;
    pub fn aesw_last_stream_check(ctx: *const aesw_ctx, out: *mut aesw_check_report) -> c_int;
    pub fn aesw_check_witness(
        ctx: *mut aesw_ctx,
        pt: *const u8,
        keys: *const u8,
        per_block_keys: c_int,
        n: u64,
        layout: c_int,
        x: *const u8,
        y: *const u8,
        z: *const u8,
        ct: *const u8,
        key_slab: *const aesw_key_slab,
        report: *mut aesw_check_report,
    ) -> c_int;
    pub fn aesw_columns_alloc(
        ctx: *mut aesw_ctx,
        n: u64,
        layout: c_int,
        with_key_slab: c_int,
        with_ct: c_int,
        out: *mut aesw_columns,
    ) -> c_int;
    pub fn aesw_columns_free(ctx: *mut aesw_ctx, cols: *mut aesw_columns) -> c_int;

    // ---- host-pointer entry points (synchronous) ----
    pub fn aesw_host_alloc(bytes: usize) -> *mut c_void;
    pub fn aesw_host_free(p: *mut c_void);
    pub fn aesw_encrypt_witness(
        ctx: *mut aesw_ctx,
        pt: *const u8,
        keys: *const u8,
        per_block_keys: c_int,
        n: u64,
        layout: c_int,
        x: *mut u8,
        y: *mut u8,
        z: *mut u8,
        ct: *mut u8,
        key_slab: *const aesw_key_slab,
    ) -> c_int;
    pub fn aesw_encrypt_witness_stream(
        ctx: *mut aesw_ctx,
        pt: *const u8,
        keys: *const u8,
        per_block_keys: c_int,
        n: u64,
        layout: c_int,
        consume: aesw_chunk_fn,
        user: *mut c_void,
    ) -> c_int;
    pub fn aesw_last_stream_stats(ctx: *const aesw_ctx, out: *mut aesw_stream_stats) -> c_int;
    pub fn aesw_assemble_advice_stream(
        ctx: *mut aesw_ctx,
        k: u32,
        n_sets: u32,
        n_blocks: u64,
        layout: c_int,
        d_x: *const u8,
        d_y: *const u8,
        d_z: *const u8,
        d_key_slab: *const aesw_key_slab,
        as_fr: c_int,
        consume: aesw_column_fn,
        user: *mut c_void,
    ) -> c_int;
    pub fn aesw_assemble_advice_host(
        ctx: *mut aesw_ctx,
        k: u32,
        n_sets: u32,
        n_blocks: u64,
        layout: c_int,
        d_x: *const u8,
        d_y: *const u8,
        d_z: *const u8,
        d_key_slab: *const aesw_key_slab,
        as_fr: c_int,
        out: *mut u8,
    ) -> c_int;
    pub fn aesw_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn aesw_host_unregister(p: *mut c_void) -> c_int;
    pub fn aesw_key_schedule_witness(
        ctx: *mut aesw_ctx,
        keys: *const u8,
        n: u64,
        layout: c_int,
        w: *mut u8,
        kx: *mut u8,
        ky: *mut u8,
        kz: *mut u8,
        rk: *mut u8,
    ) -> c_int;
    pub fn aesw_schedule_key(
        ctx: *mut aesw_ctx,
        key: *const u8,
        layout: c_int,
        key_slab: *const aesw_key_slab,
    ) -> c_int;
    pub fn aesw_lookup_table(
        ctx: *mut aesw_ctx,
        t0: *mut u8,
        t1: *mut u8,
        t2: *mut u8,
        t3: *mut u8,
    ) -> c_int;

    // ---- multi-GPU exchange (one process per GPU, RCCL over xGMI) ----
    pub fn aesw_comm_unique_id(id: *mut u8) -> c_int;
    pub fn aesw_comm_create(
        ctx: *mut aesw_ctx,
        nranks: c_int,
        rank: c_int,
        id: *const u8,
        out: *mut *mut aesw_comm,
    ) -> c_int;
    pub fn aesw_comm_destroy(comm: *mut aesw_comm);
    pub fn aesw_comm_set_max_message(comm: *mut aesw_comm, bytes: u64) -> c_int;
    pub fn aesw_comm_last_error() -> *const c_char;
    pub fn aesw_gather_offsets(
        nranks: c_int,
        counts: *const u64,
        offsets: *mut u64,
        total: *mut u64,
    ) -> c_int;
    pub fn aesw_gather_columns_device(
        comm: *mut aesw_comm,
        root: c_int,
        n_cols: c_int,
        d_send: *const *const u8,
        d_recv: *const *mut u8,
        counts: *const u64,
        strides: *const u32,
        stream: *mut c_void,
    ) -> c_int;

    // ---- tuning / introspection ----
    pub fn aesw_set_option(ctx: *mut aesw_ctx, name: *const c_char, value: i64) -> c_int;
    pub fn aesw_get_option(ctx: *const aesw_ctx, name: *const c_char, value: *mut i64) -> c_int;
    pub fn aesw_uses_xtime_path(ctx: *const aesw_ctx) -> c_int;
}
