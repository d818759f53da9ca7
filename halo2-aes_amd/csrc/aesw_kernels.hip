// aesw_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched AES-128
// witness generator.  Hand-written HIP for wave64; no MFMA (pure GF(2^8)/byte
// work, HBM-write bound).  DESIGN.md "kernels" has the rooflines.
//
// Decomposition
//   lane  = one state column (4 packed bytes) of one AES block
//   quad  = one block: ShiftRows is three DPP quad_perm moves, no LDS
//   wave  = 16 blocks; it owns a private LDS staging slab and needs no barrier
//   group = WAVES waves = 16*WAVES consecutive blocks (1..3 waves; every wave's
//           16-block output range is 128-byte-line aligned in every column)
// Data flow per wave: registers -> whole dwords (v_perm_b32) -> per-block LDS
// staging windows (aesw_layout.h Win<>: permanent head + round slots,
// bank-conflict-free strides) -> after every round, the 128-byte lines of the
// output columns that just became complete leave as whole lines, 8 lanes x 16 B
// per line, 8 lines per store instruction, following a fixed schedule whose
// per-lane descriptors (LDS address | global offset) sit in registers
// (aesw_layout.h "scheduled flush").  Partial-line stores are what caps a
// naive per-round flush at ~3 TB/s; whole lines reach ~5.5 TB/s (tools/storebench).
// The S-box / mul2 / mul3 tables live in LDS (768 B); when the host's mul
// tables equal GF(2^8) xtime the packed arithmetic path replaces 8 of the 12
// lookups per round.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "aesw_internal.h"
#include "aesw_lane.h"
#include "aesw_check.h"

namespace aesw {

template <int V> struct IntC { static constexpr int value = V; };
template <bool V> struct BoolC { static constexpr bool value = V; };

constexpr int LANES = 64;
constexpr int BPW = 16;  // blocks per wave
constexpr int TAB_BYTES = 768;
constexpr int RKS_BYTES = 176;  // shared round keys in LDS

// ---------------------------------------------------------------------------
// cross-lane helpers
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// lane w of a quad reads lane (w+k)%4
__device__ __forceinline__ uint32_t quad_rot1(uint32_t v) { return quad_perm<0x39>(v); }  // [1,2,3,0]
__device__ __forceinline__ uint32_t quad_rot2(uint32_t v) { return quad_perm<0x4E>(v); }  // [2,3,0,1]
__device__ __forceinline__ uint32_t quad_rot3(uint32_t v) { return quad_perm<0x93>(v); }  // [3,0,1,2]
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) { return quad_perm<J * 0x55>(v); }

// Lanes of one wave exchange data through LDS without s_barrier: the LDS unit
// executes a wave's DS instructions in issue order; this only stops the
// compiler from moving the reads above the writes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup id -> block group.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 names the XCD class), and each
// XCD has its own L2: mode 0 keeps the dispatch order (all XCDs write inside one moving window), mode 1 gives every XCD one
// contiguous eighth of the groups (eight far-apart write fronts per column), mode C >= 2 lets the XCDs take turns in
// chunks of C groups (each XCD writes C * 16 * waves consecutive blocks, the eight fronts stay C groups apart).
// Bijective on [0, ngroups) for every mode (speed only, never correctness).
__device__ __forceinline__ uint32_t xcd_group(uint32_t id, uint32_t ngroups, uint32_t mode) {
    if (mode == 0) return id;
    const uint32_t xcd = id % 8, seq = id / 8;
    if (mode == 1) {
        const uint32_t q = ngroups / 8, r = ngroups % 8;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
    }
    const uint32_t per = 8 * mode, full = ngroups / per * per;
    if (id >= full) return id;  // the tail that does not fill a whole turn keeps the dispatch order
    return (seq / mode * 8 + xcd) * mode + seq % mode;
}

// ---------------------------------------------------------------------------
// sinks (see aesw_lane.h)
// ---------------------------------------------------------------------------
template <int L>
struct DevSink {
    uint8_t *lds;
    uint32_t pb[3];  // stage + blk*STRIDE + 4w
    uint32_t mb[3];  // stage + blk*STRIDE + MIXW*w
    template <int C>
    __device__ __forceinline__ void plain(int off, uint32_t v) {
        *reinterpret_cast<uint32_t *>(lds + pb[C] + off) = v;
    }
    template <int C>
    __device__ __forceinline__ void mix(int off, int k, uint32_t v) {
        *reinterpret_cast<uint32_t *>(lds + mb[C] + off + 4 * k) = v;
    }
};

struct DevKSink {
    uint8_t *lds;
    uint32_t bx, by, bz, bw;  // stage + blk*stride
    __device__ __forceinline__ void kx(int off, uint32_t v) { *reinterpret_cast<uint32_t *>(lds + bx + off) = v; }
    __device__ __forceinline__ void ky(int off, uint32_t v) { *reinterpret_cast<uint32_t *>(lds + by + off) = v; }
    __device__ __forceinline__ void kz(int off, uint32_t v) { *reinterpret_cast<uint32_t *>(lds + bz + off) = v; }
    __device__ __forceinline__ void words(int off, uint32_t v) { *reinterpret_cast<uint32_t *>(lds + bw + off) = v; }
};

struct NullKSink {
    __device__ __forceinline__ void kx(int, uint32_t) {}
    __device__ __forceinline__ void ky(int, uint32_t) {}
    __device__ __forceinline__ void kz(int, uint32_t) {}
    __device__ __forceinline__ void words(int, uint32_t) {}
};

// ---------------------------------------------------------------------------
// LDS -> HBM
// ---------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int PIECE> struct PieceT;
template <> struct PieceT<16> { using type = u32x4; };
template <> struct PieceT<8> { using type = u32x2; };
template <> struct PieceT<4> { using type = uint32_t; };

// Store flavours: 0 plain, 1 nontemporal (nt), 2 write-through at agent scope
// (sc1).  Plain / nt lines stay dirty in the XCD's L2 and are written back at
// the kernel boundary (that costs dirty bytes / ~6 TB/s after the last wave has
// finished); sc1 stores leave L2 as they are issued, so a launch ends with
// nothing left to flush (MI355X_MICROARCH.md, "stores of each flavour").
// Modes 3..5 (tools/ only) exist in -DAESW_DIAGNOSTIC builds alone: 3 = leave the flush out (compute + staging only),
// 4 = flush only (no AES work, no staging writes: the store schedule fed from whatever LDS holds), 5 = as 4 without the
// LDS reads.  Their output is garbage; they price the parts of a launch.
template <int MODE>
__device__ __forceinline__ void gstore(u32x4 *p, const u32x4 &v) {
    if (MODE == 3) return;
    // hipcc neither counts nor pads an asm store: the trailing s_nop 1 covers the ">64-bit store data
    // overwritten by the next instruction" hazard (cdna guide 5.7 item 1); nothing ever waits on these stores.
    if (MODE == 2 || MODE >= 4) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int MODE>
__device__ __forceinline__ void gstore(u32x2 *p, const u32x2 &v) {
    if (MODE == 3) return;
    if (MODE == 2 || MODE >= 4) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// wave-uniform base (SGPR pair) + 32-bit byte offset per lane: no 64-bit address arithmetic per piece
// All three flavours go through the same asm form (SGPR base + VGPR offset), so that they differ in the cache-policy
// bits and in nothing else: written as C++ stores the plain and nontemporal variants needed 60 more VGPRs for 64-bit
// per-piece addresses (258-346 registers: one wave per SIMD), which made every flavour A/B a residency A/B too.
template <int MODE>
__device__ __forceinline__ void gstore_at(uint8_t *base, uint32_t off, const u32x4 &v) {
    if (MODE == 3) return;
    if (MODE == 2 || MODE >= 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base) : "memory");
    else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ uint8_t *uniform_ptr(uint8_t *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<uint8_t *>(((uint64_t)hi << 32) | lo);
}

// Scheduled whole-line flush (aesw_layout.h "scheduled flush").  A lane holds one descriptor word per store
// instruction of its column (loaded once per workgroup from the host-built table, stage base added):
//   bits 0..15  LDS address of the lane's 16-byte piece      bits 16..30  byte offset in the wave's global range
// Round R issues all its ds_read_b128 first (their LDS round trips overlap), then the stores: 8 whole lines per
// instruction.  FULL (16 valid blocks, the common case): only the last instruction of a round can have unused
// slots (known at compile time whether it has); it and every piece of a partial wave are predicated with
// "offset < nvalid * stride", which unused slots (offset 0x7ff0) fail too.
template <class W, bool ON>
struct ColSched {
    static constexpr int N = ON ? sched_first<W>(10) : 0;
    static constexpr int N4 = (N + 3) / 4;  // the device table holds four instructions' descriptors per lane and 16-byte load
    uint32_t d[N > 0 ? N : 1];
    __device__ __forceinline__ void load(const u32x4 *tab4, int lane, uint32_t stage) {
#pragma unroll
        for (int g = 0; g < N4; ++g) {
            const u32x4 v = tab4[g * LANES + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * g + j < N) d[4 * g + j] = v[j] + stage;
        }
    }
};

template <class W, int R>
struct RoundBatch {
    static constexpr int NI = sched_ninstr<W>(R), FIRST = sched_first<W>(R), NLINES = sched_nlines<W>(R);
    u32x4 v[NI > 0 ? NI : 1];
    template <int NT, class CS>
    __device__ __forceinline__ void load(const uint8_t *lds, const CS &cs) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (NT == 5) v[i] = u32x4{cs.d[FIRST + i], 1u, 2u, 3u};
            else v[i] = *reinterpret_cast<const u32x4 *>(lds + (cs.d[FIRST + i] & 0xffffu));
        }
    }
    template <int NT, bool FULL, class CS>
    __device__ __forceinline__ void store(uint8_t *g, const CS &cs, uint32_t limit) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t off = cs.d[FIRST + i] >> 16;
            // unused line slots (offset 0x7ff0) exist only in a round's last instruction, and only when the round's
            // line count is not a multiple of 8: every other instruction of a full wave stores unconditionally
            constexpr bool last_partial = NLINES % 8 != 0;
            const bool ok = (!FULL || (last_partial && i == NI - 1)) ? off < limit : true;
            if (ok) gstore_at<NT>(g, off, v[i]);
        }
    }
};

template <int L>
struct Scheds {
    ColSched<WinX<L>, Geo<L>::HAS_X> x;
    ColSched<WinY<L>, true> y;
    ColSched<WinZ<L>, true> z;
    static constexpr int OFF_Y = ColSched<WinX<L>, Geo<L>::HAS_X>::N, OFF_Z = OFF_Y + ColSched<WinY<L>, true>::N,
                         TOTAL = OFF_Z + ColSched<WinZ<L>, true>::N;
    // the same in the device layout (groups of four instructions, u32x4 per lane)
    static constexpr int OFF4_Y = ColSched<WinX<L>, Geo<L>::HAS_X>::N4, OFF4_Z = OFF4_Y + ColSched<WinY<L>, true>::N4,
                         TOTAL4 = OFF4_Z + ColSched<WinZ<L>, true>::N4;
};

template <int L, int NT, int R, bool FULL>
__device__ __forceinline__ void flush_round(const uint8_t *lds, uint8_t *gx, uint8_t *gy, uint8_t *gz, const Scheds<L> &sc,
                                            int nvalid) {
    if (NT == 3) return;  // -DAESW_DIAGNOSTIC builds only (store_mode 3): price the flush by leaving it out
    RoundBatch<WinX<L>, R> bx;
    RoundBatch<WinY<L>, R> by;
    RoundBatch<WinZ<L>, R> bz;
    if (Geo<L>::HAS_X) bx.template load<NT>(lds, sc.x);
    by.template load<NT>(lds, sc.y);
    bz.template load<NT>(lds, sc.z);
    if (Geo<L>::HAS_X) bx.template store<NT, FULL>(gx, sc.x, (uint32_t)nvalid * Geo<L>::XS);
    by.template store<NT, FULL>(gy, sc.y, (uint32_t)nvalid * Geo<L>::YS);
    bz.template store<NT, FULL>(gz, sc.z, (uint32_t)nvalid * Geo<L>::ZS);
}

// Fully contiguous: nvalid*STRIDE bytes from LDS stage to g.
template <int PIECE, int STRIDE, int NT>
__device__ __forceinline__ void flush_contig(const uint8_t *lds, uint32_t stage, uint8_t *g, int nvalid, int lane) {
    using V = typename PieceT<PIECE>::type;
    static_assert(STRIDE % PIECE == 0, "piece alignment");
    constexpr int TOTAL = BPW * STRIDE / PIECE;
    const int valid = nvalid * (STRIDE / PIECE);
#pragma unroll
    for (int p0 = 0; p0 < TOTAL; p0 += LANES) {
        const int p = p0 + lane;
        if (p < valid) {
            const V v = *reinterpret_cast<const V *>(lds + stage + p * PIECE);
            gstore<NT>(reinterpret_cast<V *>(g + (size_t)p * PIECE), v);
        }
    }
}

// A whole contiguous range of BYTES bytes (a full wave's sixteen units of one column) as PIECE-byte pieces: exactly
// ceil(BYTES / PIECE / 64) store instructions.
template <int PIECE, int BYTES, int NT>
__device__ __forceinline__ void flush_range(const uint8_t *lds, uint32_t stage, uint8_t *g, int lane) {
    using V = typename PieceT<PIECE>::type;
    static_assert(BYTES % PIECE == 0, "whole pieces");
    constexpr int TOTAL = BYTES / PIECE;
#pragma unroll
    for (int p0 = 0; p0 < TOTAL; p0 += LANES) {
        const int p = p0 + lane;
        if (p0 + LANES <= TOTAL || p < TOTAL) {
            const V v = *reinterpret_cast<const V *>(lds + stage + p * PIECE);
            gstore<NT>(reinterpret_cast<V *>(g + (size_t)p * PIECE), v);
        }
    }
}

// ---------------------------------------------------------------------------
// per-wave LDS footprints
// ---------------------------------------------------------------------------
template <int L>
struct Stage {
    using WX = WinX<L>;
    using WY = WinY<L>;
    using WZ = WinZ<L>;
    static constexpr int SX = Geo<L>::HAS_X ? WX::BYTES : 0, SY = WY::BYTES, SZ = WZ::BYTES;  // window bytes per block
    static constexpr int OX = 0, OY = BPW * SX, OZ = OY + BPW * SY;
    static constexpr int ENC_BYTES = OZ + BPW * SZ;
    using G = Geo<L>;
    static constexpr int KW = 0, KX = BPW * WORDS_ROWS, KY = KX + BPW * G::KXS, KZ = KY + BPW * G::KYS;
    static constexpr int KEY_BYTES = KZ + BPW * G::KZS;
    static constexpr int RK_BYTES_W = BPW * RK_BYTES;  // per-block round keys of a wave
};

// The key staging (per-block keys with key witness) aliases the encrypt windows: the key slab has
// left LDS before the first round writes.  Round keys live in registers in every key mode.
template <int L> __host__ __device__ constexpr int enc_wave_lds(bool kemit) {
    return (kemit && Stage<L>::KEY_BYTES > Stage<L>::ENC_BYTES) ? Stage<L>::KEY_BYTES : Stage<L>::ENC_BYTES;
}

// Residency / addressing guards: a packed group of 3 waves must stay below 64 KiB (16-bit LDS addresses in the
// flush descriptors) and two of them must fit one CU's 160 KiB; dense: 2 waves.
static_assert(TAB_BYTES + RKS_BYTES + 3 * enc_wave_lds<PACKED>(true) <= 65536, "packed windows grew: 3-wave groups no longer addressable");
static_assert(TAB_BYTES + RKS_BYTES + 2 * enc_wave_lds<DENSE>(true) <= 65536, "dense windows grew: 2-wave groups no longer addressable");

// words per layout of the flush-descriptor table the host uploads: per column, groups of four instructions, one u32x4 per
// lane and group (so a lane fetches its descriptors with a quarter of the load instructions); unused tail words are 0
int flush_table_words(int layout) {
    return (layout == DENSE ? Scheds<DENSE>::TOTAL4 : layout == VALUES ? Scheds<VALUES>::TOTAL4 : Scheds<PACKED>::TOTAL4) * LANES * 4;
}
template <class W>
static void build_column_device_layout(uint32_t *out4) {
    constexpr int N = sched_first<W>(10), N4 = (N + 3) / 4;
    uint32_t plain[(N > 0 ? N : 1) * LANES];
    build_flush_table<W>(plain);  // instruction-major: word i*64 + lane (what the tests and the lane model read)
    for (int g = 0; g < N4; ++g)
        for (int lane = 0; lane < LANES; ++lane)
            for (int j = 0; j < 4; ++j) out4[(g * LANES + lane) * 4 + j] = 4 * g + j < N ? plain[(4 * g + j) * LANES + lane] : 0u;
}
template <int L>
static void build_tables_for(uint32_t *out) {
    if (Geo<L>::HAS_X) build_column_device_layout<WinX<L>>(out);
    build_column_device_layout<WinY<L>>(out + Scheds<L>::OFF4_Y * LANES * 4);
    build_column_device_layout<WinZ<L>>(out + Scheds<L>::OFF4_Z * LANES * 4);
}
void build_flush_tables(int layout, uint32_t *out) {
    if (layout == DENSE) build_tables_for<DENSE>(out);
    else if (layout == VALUES) build_tables_for<VALUES>(out);
    else build_tables_for<PACKED>(out);
}

// One key-schedule round for every quad of the wave.
template <int L, class KS, class T>
__device__ __forceinline__ uint32_t key_round_dev(KS &ks, int rho, int w, uint32_t kw, const T &tab) {
    const uint32_t k0 = quad_bcast<0>(kw), k1 = quad_bcast<1>(kw), k2 = quad_bcast<2>(kw), k3 = quad_bcast<3>(kw);
    return emit_key_round<L>(ks, rho, w, k0, k1, k2, k3, rcon(rho - 1), tab);
}

// Key phase of a wave: W[0..15] = key, ten rounds; this lane's word of every round key goes to
// rk[0..10] (registers) and, when rkl is given, also to LDS rkl[blk][44] (key_kernel's rk output).
template <int L, bool EMIT, bool TO_LDS, class T>
__device__ __forceinline__ void key_phase(uint8_t *lds, uint32_t stage, uint32_t rk_off, uint32_t key_word, int blk,
                                          int w, const T &tab, uint32_t (&rk)[11]) {
    using G = Geo<L>;
    using St = Stage<L>;
    uint32_t *rkl = reinterpret_cast<uint32_t *>(lds + rk_off) + blk * 44 + w;
    uint32_t kw = key_word;
    rk[0] = kw;
    if (TO_LDS) rkl[0] = kw;
    if (EMIT) {
        DevKSink ks{lds, stage + St::KX + blk * G::KXS, stage + St::KY + blk * G::KYS, stage + St::KZ + blk * G::KZS,
                    stage + St::KW + blk * WORDS_ROWS};
        ks.words(4 * w, kw);  // src/key_schedule.rs:98-118
#pragma unroll
        for (int rho = 1; rho <= 10; ++rho) {
            kw = key_round_dev<L>(ks, rho, w, kw, tab);
            rk[rho] = kw;
            if (TO_LDS) rkl[4 * rho] = kw;
        }
    } else {
        NullKSink ks;
#pragma unroll
        for (int rho = 1; rho <= 10; ++rho) {
            kw = key_round_dev<L>(ks, rho, w, kw, tab);
            rk[rho] = kw;
            if (TO_LDS) rkl[4 * rho] = kw;
        }
    }
}

template <int L, int NT, bool WHOLE_KZ = false>
__device__ __forceinline__ void key_flush(const uint8_t *lds, uint32_t stage, const KeyOut &o, uint64_t blk0,
                                          int nvalid, int lane) {
    using G = Geo<L>;
    using St = Stage<L>;
    if (o.w) flush_contig<16, WORDS_ROWS, NT>(lds, stage + St::KW, o.w + blk0 * WORDS_ROWS, nvalid, lane);
    if (o.kx) flush_contig<16, G::KXS, NT>(lds, stage + St::KX, o.kx + blk0 * G::KXS, nvalid, lane);
    if (o.ky) flush_contig<16, G::KYS, NT>(lds, stage + St::KY, o.ky + blk0 * G::KYS, nvalid, lane);
    if (o.kz) {
        // packed kz is 200 B per key -- not a multiple of 16 -- but a FULL wave's sixteen keys are one contiguous, 128-byte aligned
        // range of 3 200 B on both sides (blk0 is a multiple of 16; St::KZ and the stage base are multiples of 16): whole-range
        // 16-byte pieces halve the store instructions of this column (round 4: key_kernel +5 ... 7 % in one-process A/Bs on three
        // leases, tools/key_ab.py).  A ragged last wave keeps per-key 8-byte pieces, and so does the key phase fused into
        // encrypt_kernel (WHOLE_KZ false), where the same change measured -0.5 % +- 0.6: left as it was.
#ifdef AESW_KZ_PER_KEY  /* private A/B build (tools/key_ab.py): round 3's per-key 8-byte pieces */
        if (false) {}
#else
        if (WHOLE_KZ && G::KZS % 16 != 0 && nvalid == BPW) flush_range<16, BPW * G::KZS, NT>(lds, stage + St::KZ, o.kz + blk0 * G::KZS, lane);
#endif
        else flush_contig<(G::KZS % 16 == 0 ? 16 : 8), G::KZS, NT>(lds, stage + St::KZ, o.kz + blk0 * G::KZS, nvalid, lane);
    }
}
static_assert((BPW * Geo<PACKED>::KZS) % 16 == 0 && Stage<PACKED>::KZ % 16 == 0 && Stage<PACKED>::KEY_BYTES % 16 == 0, "whole-range kz pieces");

// ---------------------------------------------------------------------------
// encrypt witness kernel
// ---------------------------------------------------------------------------
// KM_PBK: per-block keys (key schedule per quad; KEMIT: also emit its witness).
// KM_SHARED: one key for the batch, expanded per group by wave 0's first quad.
// KM_PRE: one key, scheduled earlier (schedule_key() once, encrypt() many
//         times -- the reference's own call shape, benches/aes128.rs:50-53):
//         the 44 round-key words come from global memory, no barrier for them.
enum : int { KM_PBK = 0, KM_SHARED = 1, KM_PRE = 2 };

// tools/trace.py builds a private copy with -DAESW_TRACE: every wave records the 100 MHz wall clock at
// seven points of its first group (launch timeline study).  Not compiled into the product library.
#ifdef AESW_TRACE
#define AESW_TRACE_POINT(i)                                                                              \
    do {                                                                                                 \
        if (a.trace && lane0 == 0) a.trace[((uint64_t)(blockIdx.x * waves + wave)) * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define AESW_TRACE_POINT(i) do { } while (0)
#endif

template <int L, bool XT, int KM, bool KEMIT, int NT>
__global__ void __launch_bounds__(256, (L == DENSE ? 1 : 2)) encrypt_kernel(const EncParams a) {
    constexpr bool PBK = KM == KM_PBK;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using G = Geo<L>;
    using St = Stage<L>;
    constexpr int WAVE_LDS = enc_wave_lds<L>(KEMIT);

    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: everything derived from it stays scalar
    const int waves = blockDim.x >> 6;
    const int w = lane0 & 3;

    const uint32_t shared_rk = TAB_BYTES;                           // 176 B, KM_SHARED only
    const uint32_t stage0 = TAB_BYTES + RKS_BYTES + wave * WAVE_LDS;  // this wave's slab (below 64 KiB: launch_enc checks)
    // flush descriptors: one dword per lane and store instruction, kept in registers for every group this workgroup handles
    Scheds<L> sc;
    if (NT != 3) {
        const u32x4 *ft4 = reinterpret_cast<const u32x4 *>(a.ftab);
        if (G::HAS_X) sc.x.load(ft4, lane0, stage0 + St::OX);
        sc.y.load(ft4 + Scheds<L>::OFF4_Y * LANES, lane0, stage0 + St::OY);
        sc.z.load(ft4 + Scheds<L>::OFF4_Z * LANES, lane0, stage0 + St::OZ);
    }

    // the first group's inputs first, so their latency hides behind the table load
    // Optional XCD-aware order (workgroups are dealt round-robin over the 8 XCDs): ids that share an
    // XCD walk one contiguous eighth of the block groups.  Bijective for any ngroups (speed only).
    auto remap = [&](uint32_t id) -> uint32_t { return xcd_group(id, a.ngroups, a.xcd_remap); };
    auto group_blk0 = [&](uint32_t grp) { return ((uint64_t)remap(grp) * waves + wave) * BPW; };
    auto group_nvalid = [&](uint64_t b0) {
        const int64_t left = (int64_t)a.n - (int64_t)b0;
        return left >= BPW ? BPW : (left > 0 ? (int)left : 0);
    };
    AESW_TRACE_POINT(0);
    // this lane's word of a group's 16-byte records (plaintexts / per-block keys); 0 past the batch
    auto load_word = [&](const uint8_t *base, uint32_t grp) -> uint32_t {
        if (grp >= a.ngroups) return 0u;
        const uint64_t b0 = group_blk0(grp);
        return (lane0 >> 2) < group_nvalid(b0) ? reinterpret_cast<const uint32_t *>(base)[(b0 + (lane0 >> 2)) * 4 + w] : 0u;
    };
    uint32_t ptw_next = load_word(a.pt, blockIdx.x);
    uint32_t kw_next = PBK ? load_word(a.keys, blockIdx.x) : 0u;
    uint32_t rkr[11];  // this lane's word of every round key (KM_PRE: loaded here; KM_PBK: key phase)
    if (KM == KM_PRE) {
#pragma unroll
        for (int r = 0; r < 11; ++r) rkr[r] = a.rk[4 * r + w];
    }

    // tables -> LDS
    for (int i = tid; i < TAB_BYTES / 4; i += blockDim.x)  // groups may be as small as one wave
        reinterpret_cast<uint32_t *>(lds)[i] = reinterpret_cast<const uint32_t *>(a.tables)[i];
    __syncthreads();
    AESW_TRACE_POINT(1);
    const Tables<XT> tab{lds};

    if (KM == KM_SHARED) {
        if (wave == 0) {
            const uint32_t kw = reinterpret_cast<const uint32_t *>(a.keys)[w];
            NullKSink ks;
            uint32_t k = kw;
            uint32_t *rks = reinterpret_cast<uint32_t *>(lds + shared_rk);
            if (lane0 < 4) rks[w] = k;
#pragma unroll
            for (int rho = 1; rho <= 10; ++rho) {
                k = key_round_dev<L>(ks, rho, w, k, tab);
                if (lane0 < 4) rks[4 * rho + w] = k;
            }
        }
        __syncthreads();
    }
    // A workgroup strides over block groups: with max_groups_in_flight = resident
    // workgroups the table / round-key setup above is paid once per CU slot.
    for (uint32_t grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
        // Everything below is recomputed per group from these two opaque values: otherwise the
        // compiler hoists ~200 loop-invariant LDS/flush addresses out of the loop and spills them.
        int lane = lane0;
        uint32_t stage = stage0;
        asm volatile("" : "+v"(lane), "+v"(stage));
        const int blk = lane >> 2;
        const uint64_t blk0 = group_blk0(grp);
        const int nvalid = group_nvalid(blk0);
        if (nvalid == 0) continue;
        const bool live = blk < nvalid;
        const uint32_t ptw = ptw_next;
        if (grp != blockIdx.x) wave_lds_fence();  // the previous group's flush reads precede this group's writes

        if (PBK) {
            const uint32_t kw = kw_next;
            if (NT < 4) key_phase<L, KEMIT, false>(lds, stage, 0, kw, blk, w, tab, rkr);
            if (KEMIT) {
                wave_lds_fence();
                key_flush<L, NT>(lds, stage, a.key, blk0, nvalid, lane);
            }
            wave_lds_fence();
        }
        const uint32_t *rkp = reinterpret_cast<const uint32_t *>(lds + shared_rk) + w;  // KM_SHARED
        auto rkw = [&](int r) -> uint32_t { return KM == KM_SHARED ? rkp[4 * r] : rkr[r]; };

        DevSink<L> s;
        s.lds = lds;
        s.pb[0] = stage + St::OX + blk * St::SX + 4 * w;
        s.pb[1] = stage + St::OY + blk * St::SY + 4 * w;
        s.pb[2] = stage + St::OZ + blk * St::SZ + 4 * w;
        s.mb[0] = stage + St::OX + blk * St::SX + G::X_MIXW * w;
        s.mb[1] = stage + St::OY + blk * St::SY + G::Y_MIXW * w;
        s.mb[2] = stage + St::OZ + blk * St::SZ + G::Z_MIXW * w;

        uint8_t *gx = uniform_ptr(a.x + blk0 * G::XS);
        uint8_t *gy = uniform_ptr(a.y + blk0 * G::YS);
        uint8_t *gz = uniform_ptr(a.z + blk0 * G::ZS);

        auto round = [&](int relx, int rely, int relz, uint32_t st, uint32_t rkw) -> uint32_t {
            const uint32_t sub = emit_sbox<L>(s, relx, rely, relz, st, tab);
            const uint32_t sh = shift_rows(sub, quad_rot1(sub), quad_rot2(sub), quad_rot3(sub));
            return emit_mix_ark<L>(s, relx, rely, relz, sh, rkw, tab);
        };

        // rows 0..31, then rounds 1..9 (+10); after each round the lines that just
        // became complete are flushed.  Fully unrolled: R is a constant in every copy.
        using WX = typename St::WX;
        using WY = typename St::WY;
        using WZ = typename St::WZ;
        // Two copies of the round chain: a full wave (16 valid blocks) stores without per-piece predicates.
        auto chain = [&](auto full_tag) -> uint32_t {
            constexpr bool FULL = decltype(full_tag)::value;
            uint32_t st = NT < 4 ? emit_head<L>(s, ptw, rkw(0)) : ptw;
            // rounds 1..9 (+10), unrolled through a template parameter so that every window offset and
            // flush bound is an immediate (a pragma-unrolled loop this large falls back to a runtime R)
            auto step = [&](auto rc) {
                constexpr int R = decltype(rc)::value;
                if (NT < 4) st = round(WX::woff(R), WY::woff(R), WZ::woff(R), st, rkw(R));
                if (R == 9 && NT < 4) {
                    const uint32_t sub = emit_sbox<L>(s, WX::woff(10), WY::woff(10), WZ::woff(10), st, tab);
                    const uint32_t sh = shift_rows(sub, quad_rot1(sub), quad_rot2(sub), quad_rot3(sub));
                    st = emit_final_ark<L>(s, WX::woff(10), WY::woff(10), WZ::woff(10), sh, rkw(10));
                }
                wave_lds_fence();
                if (R == 1) {
                    AESW_TRACE_POINT(2);
                    // a striding workgroup's next inputs travel while this group is flushed
                    ptw_next = load_word(a.pt, grp + gridDim.x);
                    if (PBK) kw_next = load_word(a.keys, grp + gridDim.x);
                }
                flush_round<L, NT, R, FULL>(lds, gx, gy, gz, sc, nvalid);
                wave_lds_fence();
                if (R == 1) AESW_TRACE_POINT(3);
                if (R == 5) AESW_TRACE_POINT(4);
                if (R == 9) AESW_TRACE_POINT(5);
            };
            step(IntC<1>{}); step(IntC<2>{}); step(IntC<3>{}); step(IntC<4>{}); step(IntC<5>{});
            step(IntC<6>{}); step(IntC<7>{}); step(IntC<8>{}); step(IntC<9>{});
            return st;
        };
        const uint32_t st = nvalid == BPW ? chain(BoolC<true>{}) : chain(BoolC<false>{});

        if (a.ct && live) reinterpret_cast<uint32_t *>(a.ct)[(blk0 + blk) * 4 + w] = st;
    }
#ifdef AESW_TRACE
    if (a.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of this wave acknowledged
        AESW_TRACE_POINT(6);
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane0 == 0) a.trace[((uint64_t)(blockIdx.x * waves + wave)) * 8 + 7] = ((uint64_t)xcc << 32) | hwid;
    }
#endif
}

// ---------------------------------------------------------------------------
// key-schedule witness kernel (src/key_schedule.rs:80-224 for n keys)
// ---------------------------------------------------------------------------
template <int L, bool XT, int NT>
__global__ void __launch_bounds__(256) key_kernel(const KeyParams a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    using St = Stage<L>;
    // the round-key staging (176 B per key) exists only when the caller asked for round keys (round 4: the witness-only form -- what
    // bench.py and aesw_columns_alloc's key-only arenas run -- neither reserves nor writes it)
    const int WAVE_LDS = St::KEY_BYTES + (a.rk ? St::RK_BYTES_W : 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
    const int blk = lane >> 2, w = lane & 3;
    for (int i = tid; i < TAB_BYTES / 4; i += blockDim.x)  // groups may be as small as one wave
        reinterpret_cast<uint32_t *>(lds)[i] = reinterpret_cast<const uint32_t *>(a.tables)[i];
    __syncthreads();
    const Tables<XT> tab{lds};
    const uint32_t stage = TAB_BYTES + wave * WAVE_LDS;
    const uint32_t rk_w = stage + St::KEY_BYTES;
    const uint32_t grp = xcd_group(blockIdx.x, a.ngroups, a.xcd_remap);  // as in encrypt_kernel
    const uint64_t blk0 = ((uint64_t)grp * waves + wave) * BPW;
    const int64_t left = (int64_t)a.n - (int64_t)blk0;
    const int nvalid = left >= BPW ? BPW : (left > 0 ? (int)left : 0);
    if (nvalid == 0) return;
    const uint32_t kw = blk < nvalid ? reinterpret_cast<const uint32_t *>(a.keys)[(blk0 + blk) * 4 + w] : 0u;
    uint32_t rk_unused[11];
    if (a.rk) key_phase<L, true, true>(lds, stage, rk_w, kw, blk, w, tab, rk_unused);
    else key_phase<L, true, false>(lds, stage, rk_w, kw, blk, w, tab, rk_unused);
    wave_lds_fence();
    key_flush<L, NT, true>(lds, stage, a.key, blk0, nvalid, lane);
    if (a.rk) flush_contig<16, RK_BYTES, NT>(lds, rk_w, a.rk + blk0 * RK_BYTES, nvalid, lane);
}

// ---------------------------------------------------------------------------
// lookup table (src/table.rs:18-192): 66561 rows x 4 byte columns
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) table_kernel(const uint8_t *__restrict__ tables, uint8_t *t0, uint8_t *t1,
                                                   uint8_t *t2, uint8_t *t3) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 66561u) return;
    uint8_t tag, a, b, c;
    if (r < 256u) { tag = 1; a = (uint8_t)r; b = 0; c = 0; }                                         // Tag::U8   :27-53
    else if (r < 512u) { const uint32_t i = r - 256u; tag = 3; a = (uint8_t)i; b = tables[i]; c = 0; }  // Tag::Sbox :57-83
    else if (r < 66048u) { const uint32_t l = r - 512u; tag = 2; a = (uint8_t)(l >> 8); b = (uint8_t)l; c = a ^ b; }  // Tag::Xor :88-116
    else if (r < 66304u) { const uint32_t i = r - 66048u; tag = 4; a = (uint8_t)i; b = tables[256 + i]; c = 0; }      // GfMul2 :120-145
    else if (r < 66560u) { const uint32_t i = r - 66304u; tag = 5; a = (uint8_t)i; b = tables[512 + i]; c = 0; }      // GfMul3 :149-174
    else { tag = 0; a = 0; b = 0; c = 0; }                                                           // empty row :178-187
    t0[r] = tag; t1[r] = a; t2[r] = b; t3[r] = c;
}

// ---------------------------------------------------------------------------
// byte cells -> bn256::Fr Montgomery cells
// ---------------------------------------------------------------------------
// fr_lut: 256 x 32 B (value v -> v*R mod p, little-endian), built on the host.
// One lane writes one 16-byte half cell, so a wave stores 1 KiB contiguously;
// each wave owns UNROLL KiB-pieces per trip and issues all its byte loads, then
// all LUT reads, then all stores, to keep >= 8 KiB in flight per wave.
template <int NT>
__global__ void __launch_bounds__(256) expand_fr_kernel(const uint8_t *__restrict__ cells, uint64_t n_cells,
                                                       const u32x4 *__restrict__ fr_lut, u32x4 *__restrict__ out) {
    __shared__ u32x4 lut[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lut[i] = fr_lut[i];
    __syncthreads();
    constexpr int UNROLL = 8;
    const uint64_t total = n_cells * 2;  // 16-byte pieces
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave * (UNROLL * 64); base < total; base += nwaves * (UNROLL * 64)) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint64_t i = base + j * 64 + lane;
            v[j] = i < total ? cells[i >> 1] : 0u;
        }
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint64_t i = base + j * 64 + lane;
            if (i < total) gstore<NT>(&out[i], lut[v[j] * 2 + (uint32_t)(i & 1)]);
        }
    }
}

// One-shot geometries of the same expansion (round 2, profiles/r02_study: a stream of short-lived workgroups that each
// write a small contiguous chunk in address order is what the memory system writes fastest).
// GEO 1: a workgroup writes 4 KiB (one 16-byte piece per thread, one store per wave) and exits; the LUT is gathered
//        from global memory (8 KiB, L1/L2 resident).  GEO 2: 16 KiB per workgroup, LUT staged in LDS.
template <int NT, int GEO>
__global__ void __launch_bounds__(256) expand_fr_oneshot_kernel(const uint8_t *__restrict__ cells, uint64_t n_cells,
                                                               const u32x4 *__restrict__ fr_lut, u32x4 *__restrict__ out) {
    const uint64_t total = n_cells * 2;
    if (GEO == 1) {
        const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= total) return;
        const uint32_t v = cells[i >> 1];
        gstore<NT>(&out[i], fr_lut[v * 2 + (uint32_t)(i & 1)]);
    } else {
        __shared__ u32x4 lut[512];
        for (int i = threadIdx.x; i < 512; i += 256) lut[i] = fr_lut[i];
        const uint64_t base = (uint64_t)blockIdx.x * 1024;
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t i = base + j * 256 + threadIdx.x;
            v[j] = i < total ? cells[i >> 1] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t i = base + j * 256 + threadIdx.x;
            if (i < total) gstore<NT>(&out[i], lut[v[j] * 2 + (uint32_t)(i & 1)]);
        }
    }
}

// ---------------------------------------------------------------------------
// full advice columns of a K/N circuit (bytes or Fr cells)
// ---------------------------------------------------------------------------
// Cell (column j, row r) -> the slab byte synthesize() puts there (aes_callable
// placement, src/aes128.rs:303-325), 0 when nothing is ever assigned.
__device__ __forceinline__ uint32_t advice_cell(const AssembleParams &a, uint32_t col, uint64_t row) {
    const uint32_t n_adv = 3 * a.n_sets;
    if (col == n_adv) return (row < WORDS_ROWS && a.kw) ? a.kw[row] : 0u;  // words_column
    const uint32_t set = col / 3, c = col - 3 * set;
    const uint64_t rows = (uint64_t)1 << a.k;
    const uint64_t cap0 = rows >= 1760 ? (rows - 1760) / AES_ROWS : 0, capn = rows / AES_ROWS;
    if (set == 0 && row < KEY_ROWS) {
        const uint8_t *kc = c == 0 ? a.kx : c == 1 ? a.ky : a.kz;
        if (!kc) return 0u;
        const int idx = a.packed ? packed_index_key((int)c, (int)row) : (int)row;
        return idx >= 0 ? kc[idx] : 0u;
    }
    const uint64_t base = set == 0 ? KEY_ROWS : 0;
    if (row < base) return 0u;
    const uint64_t rr = row - base;
    const uint64_t bi = rr / AES_ROWS;
    const uint32_t r = (uint32_t)(rr - bi * AES_ROWS);
    if (bi >= (set == 0 ? cap0 : capn)) return 0u;
    const uint64_t b = (set == 0 ? 0 : cap0 + (uint64_t)(set - 1) * capn) + bi;
    if (b >= a.n_blocks) return 0u;
    const uint8_t *sc = c == 0 ? a.x : c == 1 ? a.y : a.z;
    const uint32_t stride = c == 0 ? a.sx : c == 1 ? a.sy : a.sz;
    const int idx = a.packed ? packed_index_enc((int)c, (int)r) : (int)r;
    return idx >= 0 ? sc[b * stride + idx] : 0u;
}

// Round 3: the Fr form of the columns as ONE-SHOT workgroups without a division.  A column of 2^k rows
// is cut into a head (rows [0, base): the 400 key rows of set 0's columns, empty elsewhere) and segments of AES_ROWS rows
// behind it -- exactly the blocks of aes_callable's placement, then never-assigned rows.  Grid: x = 4 KiB chunk of a
// segment's AES_ROWS x 32 B (11 of them, the last one partial), y = segment, z = column; a thread writes one 16-byte half
// cell with one nontemporal store and exits (the geometry that takes expand_fr to 7.3 TB/s).  Row, block and set follow
// from the grid indices by shifts, compares and one multiplication by a constant; the capacities come from the host.
// Measured (tools/asm_geo.py, K = 20, N = 5, 512 MiB): 102.7 us = 5.2 TB/s against 96.6 us = 5.6 TB/s for the striding kernel
// below, byte for byte the same output.  So the divisions were not what held the one-shot form back in round 2: a piece is a
// chain of three dependent loads (packed-index table -> slab byte -> LUT), a one-shot workgroup has nothing else in flight,
// and 2 048 resident workgroups x 4 KiB / (that chain's latency) is the rate; expand_fr, whose chain is two loads long, reaches
// 7.3 TB/s in the same geometry.  Kept as option "assemble_geometry" 1, tested; superseded by the aligned form below.
template <int NT>
__global__ void __launch_bounds__(256) assemble_fr_oneshot_kernel(const AssembleParams a) {
    const uint32_t piece = blockIdx.x * 256 + threadIdx.x;  // 16-byte piece of the segment: 2 per row
    const uint32_t r = piece >> 1;                           // row inside the segment
    if (r >= AES_ROWS) return;
    const uint32_t col = a.col_first + blockIdx.z, seg = blockIdx.y;
    const uint32_t n_adv = 3 * a.n_sets;
    const bool words = col == n_adv;
    const uint32_t set = col / 3, c = col - 3 * set;
    const uint64_t rows = (uint64_t)1 << a.k;
    const uint32_t base = (!words && set == 0) ? KEY_ROWS : 0;
    uint64_t row;
    uint32_t v = 0;
    if (seg == 0) {  // head: key rows of set 0
        if (r >= base || r >= rows) return;  // (a circuit of fewer than 400 rows clips the key slab: K <= 8, found by round 4's K = 7 test)
        row = r;
        const uint8_t *kc = c == 0 ? a.kx : c == 1 ? a.ky : a.kz;
        if (kc) {
            const int idx = a.packed ? packed_index_key((int)c, (int)r) : (int)r;
            if (idx >= 0) v = kc[idx];
        }
    } else {
        const uint64_t bi = seg - 1;
        row = base + bi * AES_ROWS + r;
        if (row >= rows) return;
        if (words) {
            if (row < WORDS_ROWS && a.kw) v = a.kw[row];
        } else if (bi < (set == 0 ? a.cap0 : a.capn)) {
            const uint64_t b = (set == 0 ? 0 : a.cap0 + (uint64_t)(set - 1) * a.capn) + bi;
            if (b < a.n_blocks) {
                const uint8_t *sc = c == 0 ? a.x : c == 1 ? a.y : a.z;
                const uint32_t stride = c == 0 ? a.sx : c == 1 ? a.sy : a.sz;
                const int idx = a.packed ? packed_index_enc((int)c, (int)r) : (int)r;
                if (idx >= 0) v = sc[b * stride + idx];
            }
        }
    }
    const u32x4 *lut = reinterpret_cast<const u32x4 *>(a.fr_lut);  // 8 KiB, L1 / L2 resident
    u32x4 *out = reinterpret_cast<u32x4 *>(a.out) + (((uint64_t)blockIdx.z << a.k) + row) * 2 + (piece & 1);
    gstore<NT>(out, lut[v * 2 + (piece & 1)]);
}

// Geometry 2 / 3 / 4 (4 = the default): the same one-shot workgroups cut on the OUTPUT instead -- a workgroup of THREADS
// threads writes PIECES x THREADS x 16 B of one column, aligned, as expand_fr's GEO 1 does -- at the price of one 32-bit
// division by AES_ROWS per piece (a multiply-high; K <= 30 is checked on the host).  The slab index comes from
// packed_index_enc/_key instead of the table, everything that depends on the column only is scalar, and with PIECES = 2 a
// thread has two independent byte -> LUT -> store chains in flight.  Two things decide the rate: a workgroup should write
// 4 KiB (8 KiB costs ~8 %, as in expand_fr), and the ~100 VALU instructions of index arithmetic in front of the byte load
// should overlap another chain (another ~8 %).  Geometry 2 = 256 threads x 1 piece (4 KiB, one chain), 3 = 256 x 2 (8 KiB,
// two chains), 4 = 128 x 2 (4 KiB, two chains).  Measured (tools/asm_geo.py, N = 5, packed slabs; TB/s written, striding /
// geometry 1 / 2 / 3 / 4 / expand_fr over as many bytes): K = 20: 5.3 / 5.0 / 6.2 / 6.1 - 6.3 / 6.9 / 6.9;
// K = 22: 4.1 / 5.5 / 6.3 / 6.6 / 7.1 / 7.2.  (64 threads x 4 pieces: 6.2; 256 x 4: 5.8.)
template <int NT, int PIECES, int THREADS>
__global__ void __launch_bounds__(THREADS) assemble_fr_aligned_kernel(const AssembleParams a) {
    // every kernel argument in SGPRs before anything else: left alone the compiler loads them where they are first used,
    // behind branches, and a one-shot workgroup then waits for four scalar loads one after the other (worth 3 % in geometry 2)
    asm volatile("" ::"s"(a.x), "s"(a.y), "s"(a.z), "s"(a.kw), "s"(a.kx), "s"(a.ky), "s"(a.kz), "s"(a.fr_lut), "s"(a.out), "s"(a.n_blocks),
                 "s"(a.k), "s"(a.n_sets), "s"(a.col_first), "s"(a.sx), "s"(a.sy), "s"(a.sz), "s"(a.packed), "s"(a.cap0), "s"(a.capn));
    const uint32_t col = a.col_first + blockIdx.y;
    const uint32_t n_adv = 3 * a.n_sets;
    const bool words = col == n_adv;
    const uint32_t set = col / 3, c = col - 3 * set;
    const uint32_t base = (!words && set == 0) ? KEY_ROWS : 0;
    const uint8_t *kc = c == 0 ? a.kx : c == 1 ? a.ky : a.kz;
    const uint8_t *sc = c == 0 ? a.x : c == 1 ? a.y : a.z;
    const uint32_t stride = c == 0 ? a.sx : c == 1 ? a.sy : a.sz;
    const uint64_t cap = set == 0 ? a.cap0 : a.capn;
    const uint64_t b0 = set == 0 ? 0 : a.cap0 + (uint64_t)(set - 1) * a.capn;
    const uint32_t half = threadIdx.x & 1;
    uint32_t row[PIECES], v[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        row[j] = ((blockIdx.x * PIECES + j) * THREADS + threadIdx.x) >> 1;
        v[j] = 0;
        if (words) {
            if (row[j] < WORDS_ROWS && a.kw) v[j] = a.kw[row[j]];
        } else if (row[j] < base) {
            const int idx = a.packed ? packed_index_key((int)c, (int)row[j]) : (int)row[j];
            if (kc && idx >= 0) v[j] = kc[idx];
        } else {
            const uint32_t rr = row[j] - base, bi = rr / AES_ROWS, r = rr - bi * AES_ROWS;
            const int idx = a.packed ? packed_index_enc((int)c, (int)r) : (int)r;
            if (bi < cap && b0 + bi < a.n_blocks && idx >= 0) v[j] = sc[(b0 + bi) * stride + idx];
        }
    }
    const u32x4 *lut = reinterpret_cast<const u32x4 *>(a.fr_lut);
    u32x4 *out = reinterpret_cast<u32x4 *>(a.out) + ((uint64_t)blockIdx.y << (a.k + 1)) + half;
    u32x4 f[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) f[j] = lut[v[j] * 2 + half];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) gstore<NT>(out + (uint64_t)row[j] * 2, f[j]);
}

// (A one-shot geometry like expand_fr's was tried for the cell-indexed form below in round 2 and is 9x SLOWER: a piece here is a chain of
// three dependent loads -- packed-index table, slab byte, LUT -- behind 64-bit divisions, and a thread with a single piece
// has nothing to overlap them with; the striding loop below lets the compiler keep several pieces in flight per lane.)
template <bool AS_FR, int NT>
__global__ void __launch_bounds__(256) assemble_kernel(const AssembleParams a) {
    __shared__ u32x4 lut[AS_FR ? 512 : 1];
    if (AS_FR) {
        for (int i = threadIdx.x; i < 512; i += blockDim.x) lut[i] = reinterpret_cast<const u32x4 *>(a.fr_lut)[i];
        __syncthreads();
    }
    const uint64_t rows = (uint64_t)1 << a.k;
    const uint64_t cells = (uint64_t)a.col_count * rows;  // columns [col_first, col_first + col_count), out starts at col_first
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    if (AS_FR) {
        // one lane = one 16-byte half cell: a wave writes 1 KiB contiguously
        u32x4 *out = reinterpret_cast<u32x4 *>(a.out);
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells * 2; i += stride) {
            const uint64_t cell = i >> 1;
            const uint32_t v = advice_cell(a, a.col_first + (uint32_t)(cell >> a.k), cell & (rows - 1));
            gstore<NT>(&out[i], lut[v * 2 + (uint32_t)(i & 1)]);
        }
    } else {
        // one lane = four consecutive cells of one column (rows is a multiple of 4)
        uint32_t *out = reinterpret_cast<uint32_t *>(a.out);
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells / 4; i += stride) {
            const uint64_t cell = i * 4;
            const uint32_t col = a.col_first + (uint32_t)(cell >> a.k);
            const uint64_t row = cell & (rows - 1);
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v |= advice_cell(a, col, row + j) << (8 * j);
            out[i] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// placement probe (aesw_columns_alloc): a store-only emulation of encrypt_kernel's pattern and a linear fill
// ---------------------------------------------------------------------------
// probe_fronts: one wave per 16 blocks, in xcd_group() order; the key columns leave as one contiguous range per
// column, the encrypt columns in ten slices cut on whole KiB, exactly the traffic shape of a launch (tools/allocbench.hip
// shows the product's time follows this emulation's within a few per cent on every backing).  The values are garbage:
// the arena is uninitialised memory when it is probed.
__global__ void __launch_bounds__(64) probe_fronts_kernel(const ProbeParams a) {
    const int lane = threadIdx.x;
    const uint32_t grp = xcd_group(blockIdx.x, gridDim.x, a.xcd_mode);
    const uint64_t blk0 = (uint64_t)grp * BPW;
    if (blk0 >= a.n) return;
    const int nb = a.n - blk0 < (uint64_t)BPW ? (int)(a.n - blk0) : BPW;
    const u32x4 v = {0x5a5a5a5au, (uint32_t)grp, (uint32_t)lane, 0xa5a5a5a5u};
    for (int c = 3; c < 7; ++c) {
        if (!a.col[c]) continue;
        uint8_t *g = a.col[c] + blk0 * a.stride[c];
        const int len = nb * (int)a.stride[c] / 16 * 16;
        for (int p = lane * 16; p < len; p += LANES * 16) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(g + p));
    }
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 3; ++c) {
            if (!a.col[c]) continue;
            uint8_t *g = a.col[c] + blk0 * a.stride[c];
            const int len = nb * (int)a.stride[c] / 16 * 16;
            const int lo = len / 10 * r / 1024 * 1024, hi = r == 9 ? len : len / 10 * (r + 1) / 1024 * 1024;
            for (int p = lo + lane * 16; p < hi; p += LANES * 16) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(g + p));
        }
}
// probe_fill: workgroup i writes the 4 KiB chunk i of the same columns, one after the other (the fastest writer the part has)
__global__ void __launch_bounds__(256) probe_fill_kernel(const ProbeParams a) {
    uint32_t chunk = blockIdx.x;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)  // chunks[] is 0 for an absent column
        if (c == i && chunk >= a.chunks[i]) { chunk -= a.chunks[i]; c = i + 1; }
    const uint64_t p = (uint64_t)chunk * 4096 + threadIdx.x * 16;
    const u32x4 v = {1u, 2u, 3u, threadIdx.x};
    if (p + 16 <= a.n * a.stride[c]) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(a.col[c] + p));
}

hipError_t launch_probe(const ProbeParams &p0, bool fill, hipStream_t s) {
    if (p0.n == 0) return hipSuccess;
    ProbeParams p = p0;
    if (fill) {
        uint64_t chunks = 0;
        for (int c = 0; c < 7; ++c) {
            const uint64_t k = p.col[c] ? (p.n * p.stride[c] + 4095) / 4096 : 0;
            if (k > 0x7fffffffull) return hipErrorInvalidValue;
            p.chunks[c] = (uint32_t)k;
            chunks += k;
        }
        if (chunks == 0) return hipSuccess;
        if (chunks > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(probe_fill_kernel, dim3((unsigned)chunks), dim3(256), 0, s, p);
    } else {
        const uint64_t groups = (p.n + BPW - 1) / BPW;
        if (groups > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(probe_fronts_kernel, dim3((unsigned)groups), dim3(LANES), 0, s, p);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// check_kernel: MockProver::assert_satisfied over slabs (aesw_check.h).  One wave = one block at a time: its three column
// ranges (and, with per-block keys, its key slab) are copied into the wave's LDS image with coalesced dword loads, then the
// 64 lanes walk the row, edge and gate entries of the layout's check table (LDS, loaded once per workgroup).  Read-bound:
// 3 992 B per block with per-block keys.  Waves stride over the blocks; nothing is written but the report.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One column range of a unit on its way from global memory into the wave's LDS image: BYTES bytes as units of VEC bytes (16 where the
// layout's strides keep both sides 16-byte aligned, 8 for the packed kz and words_column), unit lane + 64 j in lane's j-th register.
// load() only issues the loads; store() is called a block later, so the next unit's bytes travel while the current one is checked.
template <int BYTES, int VEC>
struct Staged {
    static_assert(BYTES % VEC == 0 && (VEC == 16 || VEC == 8), "whole units");
    static constexpr int UNITS = BYTES / VEC, N = (UNITS + LANES - 1) / LANES;
    using V = typename std::conditional<VEC == 16, u32x4, u32x2>::type;
    V v[N];
    __device__ __forceinline__ void load(const uint8_t *src, uint32_t lane) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint32_t i = lane + LANES * j;
            if (i < (uint32_t)UNITS) v[j] = reinterpret_cast<const V *>(src)[i];
        }
    }
    __device__ __forceinline__ void store(uint8_t *dst, uint32_t lane) const {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint32_t i = lane + LANES * j;
            if (i < (uint32_t)UNITS) reinterpret_cast<V *>(dst)[i] = v[j];
        }
    }
};

// The fast path's form of the check table, in LDS (one copy per workgroup): offsets of cells a layout leaves out (CHECK_NONE; only
// on rows whose tag does not read them) point at byte 0, tags that cannot fail (no lookup, U8 range) become 0, and a row entry
// carries the offset of its value table inside t768:  w0 = ox | oy << 16,  w1 = oz | tag << 16 | table offset << 20.
__device__ __forceinline__ void fast_row_entry(uint32_t a, uint32_t b, uint32_t &w0, uint32_t &w1) {
    const uint32_t tag = b >> 16;
    const uint32_t ox = (a & 0xffffu) == CHECK_NONE ? 0u : (a & 0xffffu), oy = (a >> 16) == CHECK_NONE ? 0u : (a >> 16),
                   oz = (b & 0xffffu) == CHECK_NONE ? 0u : (b & 0xffffu);
    const uint32_t tg = tag < 2 ? 0u : tag;
    w0 = ox | oy << 16;
    w1 = oz | tg << 16 | (tg >= 3 ? (tg - 3) * 256u : 0u) << 20;
}
__device__ __forceinline__ void load_fast_table(uint32_t *tab, const uint32_t *t) {
    for (uint32_t r = threadIdx.x; r < (uint32_t)(AES_ROWS + KEY_ROWS); r += blockDim.x) {
        const uint32_t base = r < (uint32_t)AES_ROWS ? CHK_ROWS + 2 * r : CHK_KROWS + 2 * (r - AES_ROWS);
        fast_row_entry(t[base], t[base + 1], tab[base], tab[base + 1]);
    }
    for (uint32_t i = threadIdx.x; i < (uint32_t)BLOCK_COPIES; i += blockDim.x) tab[CHK_EDGES + i] = t[CHK_EDGES + i];
    for (uint32_t i = threadIdx.x; i < (uint32_t)(KEY_COPIES + WORDS_ROWS); i += blockDim.x) tab[CHK_KEDGES + i] = t[CHK_KEDGES + i];
}
// one lookup row, branch-free: 1 if the enabled lookup has no table row
__device__ __forceinline__ uint32_t fast_row_bad(const uint8_t *img, const uint8_t *t768, uint32_t w0, uint32_t w1) {
    const uint32_t x = img[w0 & 0xffffu], y = img[w0 >> 16], z = img[w1 & 0xffffu], tag = (w1 >> 16) & 7u;
    const uint32_t lk = t768[(w1 >> 20) + x];
    const uint32_t want = tag == 2 ? (x ^ y) : lk, got = tag == 2 ? z : y;
    return (tag != 0) & (want != got);
}
template <int ROWS_AT, int NROWS, int EDGES_AT, int NEDGES>
__device__ __forceinline__ uint32_t fast_unit_bad(const uint8_t *img, const uint8_t *t768, const uint32_t *tab, uint32_t lane) {
    uint32_t bad = 0;
    const uint2 *rows = reinterpret_cast<const uint2 *>(tab + ROWS_AT);
#pragma unroll 4
    for (uint32_t r = lane; r < (uint32_t)NROWS; r += LANES) {
        const uint2 w = rows[r];
        bad |= fast_row_bad(img, t768, w.x, w.y);
    }
#pragma unroll 4
    for (uint32_t e = lane; e < (uint32_t)NEDGES; e += LANES) {
        const uint32_t d = tab[EDGES_AT + e];
        bad |= img[d & 0xffffu] != img[d >> 16];
    }
    return bad;
}

// Fast path per block: every lane evaluates its slice branch-free out of registers ("is anything wrong with this unit?"); only a
// unit where some lane says yes is walked again by the exact code of aesw_check.h (the code the CPU model runs), which counts
// and names the failures.  A satisfied witness -- the normal case -- never takes the second walk.
// Pipeline per wave: the loads of block b + stride are issued (into registers) before block b is checked out of LDS.
template <int LAYOUT>
struct ChkLayout {
    static constexpr int SX = AES_ROWS, SY = LAYOUT == DENSE ? AES_ROWS : Geo<PACKED>::YS, SZ = LAYOUT == DENSE ? AES_ROWS : Geo<PACKED>::ZS;
    static constexpr int KXS = KEY_ROWS, KYS = LAYOUT == DENSE ? KEY_ROWS : Geo<PACKED>::KYS, KZS = LAYOUT == DENSE ? KEY_ROWS : Geo<PACKED>::KZS;
    static constexpr int BI = SX + SY + SZ, O_KY = KXS, O_KZ = KXS + KYS, O_W = KXS + KYS + KZS, KI = O_W + WORDS_ROWS;
    static constexpr int IMG = (BI + KI + 15) / 16 * 16;
    static constexpr int KZV = KZS % 16 == 0 && (BI + O_KZ) % 16 == 0 ? 16 : 8, WV = (BI + O_W) % 16 == 0 ? 16 : 8;
    static_assert(SX % 16 == 0 && SY % 16 == 0 && SZ % 16 == 0 && KXS % 16 == 0 && KYS % 16 == 0 && BI % 16 == 0, "16-byte units");
};

template <int LAYOUT>
struct StagedKey {
    using G = ChkLayout<LAYOUT>;
    Staged<G::KXS, 16> kx; Staged<G::KYS, 16> ky; Staged<G::KZS, G::KZV> kz; Staged<WORDS_ROWS, G::WV> w;
    __device__ __forceinline__ void load(const CheckParams &a, uint64_t k, uint32_t lane) {
        kx.load(a.kx + k * G::KXS, lane); ky.load(a.ky + k * G::KYS, lane); kz.load(a.kz + k * G::KZS, lane); w.load(a.kw + k * WORDS_ROWS, lane);
    }
    __device__ __forceinline__ void store(uint8_t *kimg, uint32_t lane) const {
        kx.store(kimg, lane); ky.store(kimg + G::O_KY, lane); kz.store(kimg + G::O_KZ, lane); w.store(kimg + G::O_W, lane);
    }
};

template <int LAYOUT, bool PBK>
__global__ void __launch_bounds__(256) check_kernel(const CheckParams a) {
    using G = ChkLayout<LAYOUT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t check_lds[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(check_lds);
    uint8_t *t768 = check_lds + CHK_WORDS * 4;
    const uint32_t wave = threadIdx.x / LANES, lane = threadIdx.x % LANES;
    uint8_t *img = t768 + 768 + wave * G::IMG;
    uint8_t *kimg = img + G::BI;
    for (uint32_t i = threadIdx.x; i < 768 / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(t768)[i] = reinterpret_cast<const uint32_t *>(a.tab768)[i];
    load_fast_table(tab, a.table);
    __syncthreads();
    const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x / LANES), gwave = (uint64_t)blockIdx.x * (blockDim.x / LANES) + wave;
    CheckAcc acc;
    if (gwave == 0 && lane == 0) { a.report[0] = a.n; a.report[1] = PBK ? a.n : (a.skip_shared_key ? 0 : 1); }
    const uint32_t ct_off = tab[CHK_ROWS + 2 * (AES_ROWS - 16 + (lane & 15)) + 1] & 0xffffu;  // lanes 0..15: z of rows 1344 + lane
    const uint32_t w_off = tab[CHK_GATES + (lane & 15)] & 0xffffu;                              // lanes 0..15: words_column row `lane`
    if (!PBK) {  // one key slab for the whole batch: every wave keeps a copy; the first wave of the grid checks it
        StagedKey<LAYOUT> sk;
        sk.load(a, 0, lane);
        sk.store(kimg, lane);
        wave_lds_sync();
        if (gwave == 0 && !a.skip_shared_key) check_key(img, a.table, t768, a.keys, 0, lane, LANES, acc);
    }
    Staged<G::SX, 16> sx; Staged<G::SY, 16> sy; Staged<G::SZ, 16> sz;
    StagedKey<LAYOUT> skey;
    uint32_t lit = 0, klit = 0;  // the literal rows: plaintext, ciphertext, key bytes (lanes 0..15, one byte each)
    auto fetch = [&](uint64_t b) {
        sx.load(a.x + b * G::SX, lane); sy.load(a.y + b * G::SY, lane); sz.load(a.z + b * G::SZ, lane);
        if (PBK) skey.load(a, b, lane);
        if (lane < 16) {
            lit = a.pt[b * 16 + lane];
            if (a.ct) lit |= (uint32_t)a.ct[b * 16 + lane] << 8;
            if (PBK && a.keys) klit = a.keys[b * 16 + lane];
        }
    };
    if (gwave < a.n) fetch(gwave);
    for (uint64_t b = gwave; b < a.n; b += nwaves) {
        sx.store(img, lane); sy.store(img + G::SX, lane); sz.store(img + G::SX + G::SY, lane);
        if (PBK) skey.store(kimg, lane);
        const uint32_t lit_b = lit, klit_b = klit;
        wave_lds_sync();
        if (b + nwaves < a.n) fetch(b + nwaves);  // in flight while this block is checked
        uint32_t bad = fast_unit_bad<CHK_ROWS, AES_ROWS, CHK_EDGES, BLOCK_COPIES>(img, t768, tab, lane);
        if (lane < 16) {
            bad |= img[lane] != (lit_b & 0xffu);
            if (a.ct) bad |= img[ct_off] != (lit_b >> 8);
        }
        if (__ballot(bad != 0) != 0) check_block(img, a.table, t768, a.pt + b * 16, a.ct ? a.ct + b * 16 : nullptr, b, lane, LANES, acc);
        if (PBK) {
            uint32_t kbad = fast_unit_bad<CHK_KROWS, KEY_ROWS, CHK_KEDGES, KEY_COPIES>(img, t768, tab, lane);
            for (uint32_t r = lane; r < (uint32_t)WORDS_ROWS; r += LANES) {
                const uint32_t gte = tab[CHK_GATES + r];
                kbad |= ((gte >> 24) != 0) & (img[gte & 0xffffu] != ((gte >> 16) & 0xffu));
            }
            if (lane < 16 && a.keys) kbad |= img[w_off] != klit_b;
            if (__ballot(kbad != 0) != 0) check_key(img, a.table, t768, a.keys ? a.keys + b * 16 : nullptr, b, lane, LANES, acc);
        }
        wave_lds_sync();  // the next block overwrites the image
    }
    // failures are the rare case: a lane that found any adds them itself
    if (acc.lookup) atomicAdd(reinterpret_cast<unsigned long long *>(a.report + 2), (unsigned long long)acc.lookup);
    if (acc.copy) atomicAdd(reinterpret_cast<unsigned long long *>(a.report + 3), (unsigned long long)acc.copy);
    if (acc.gate) atomicAdd(reinterpret_cast<unsigned long long *>(a.report + 4), (unsigned long long)acc.gate);
    if (acc.input) atomicAdd(reinterpret_cast<unsigned long long *>(a.report + 5), (unsigned long long)acc.input);
    if (acc.first != ~0ull) atomicMin(reinterpret_cast<unsigned long long *>(a.report + 6), (unsigned long long)acc.first);
}

hipError_t launch_check(const CheckParams &p, hipStream_t s) {
    // the report starts as (0 blocks, 0 keys, no failures, first = none); the kernel's first wave fills in the unit counts
    hipError_t e = hipMemsetAsync(p.report, 0, 6 * sizeof(uint64_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(p.report + 6, 0xff, sizeof(uint64_t), s);
    if (e != hipSuccess || p.n == 0) return e;
    const uint32_t waves = 4;
    const bool dense = p.sy == (uint32_t)AES_ROWS;
    const size_t lds = (size_t)CHK_WORDS * 4 + 768 + (size_t)waves * (dense ? ChkLayout<DENSE>::IMG : ChkLayout<PACKED>::IMG);  // 41 / 46 KiB
    uint64_t groups = (p.n + waves - 1) / waves;
    if (groups > 256 * 3) groups = 256 * 3;  // three workgroups (twelve waves) per CU, every wave strides over its share of the blocks
    const dim3 grid((unsigned)groups), block(waves * LANES);
    if (dense) {
        if (p.per_block_keys) hipLaunchKernelGGL((check_kernel<DENSE, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((check_kernel<DENSE, false>), grid, block, lds, s, p);
    } else {
        if (p.per_block_keys) hipLaunchKernelGGL((check_kernel<PACKED, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((check_kernel<PACKED, false>), grid, block, lds, s, p);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// Dynamic LDS above 48 KiB needs the attribute once per (kernel, device); remember
// it so that the launch path stays free of extra runtime calls.
static hipError_t allow_large_lds(const void *fn, size_t lds) {
    if (lds <= 48 * 1024) return hipSuccess;
    struct Seen { const void *fn; int dev; size_t lds; };
    static Seen seen[2048];
    static int n_seen = 0;
    static std::mutex mu;  // contexts on different host threads launch concurrently
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].fn == fn && seen[i].dev == dev && seen[i].lds >= lds) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && n_seen < 2048) seen[n_seen++] = Seen{fn, dev, lds};
    return e;
}

template <int L, bool XT, int KM, bool KEMIT, int NT>
static hipError_t launch_enc(const EncParams &p0, int waves, uint32_t cap, uint32_t xr, uint32_t lds_pad, hipStream_t stream) {
    const int bpg = waves * BPW;
    const uint64_t groups = (p0.n + bpg - 1) / bpg;
    if (groups == 0) return hipSuccess;
    if (groups > 0x7fffffffull) return hipErrorInvalidValue;
    EncParams p = p0;
    p.ngroups = (uint32_t)groups;
    p.xcd_remap = xr;
    const unsigned grid = cap && cap < groups ? cap : (unsigned)groups;
    const size_t lds = TAB_BYTES + RKS_BYTES + (size_t)waves * enc_wave_lds<L>(KEMIT) + lds_pad;
    // flush descriptors carry 16-bit LDS addresses: a group's staging must end below 64 KiB
    if (TAB_BYTES + RKS_BYTES + (size_t)waves * enc_wave_lds<L>(KEMIT) > 65536 || !p0.ftab) return hipErrorInvalidValue;
    {
        hipError_t e = allow_large_lds(reinterpret_cast<const void *>(&encrypt_kernel<L, XT, KM, KEMIT, NT>), lds);
        if (e != hipSuccess) return e;
    }
    // launched by name, not through a function-pointer variable: a host build with sanitizers silently drops the latter
    hipLaunchKernelGGL((encrypt_kernel<L, XT, KM, KEMIT, NT>), dim3(grid), dim3(waves * LANES), lds, stream, p);
    return hipGetLastError();
}

template <int L, bool XT, int KM, bool KEMIT>
static hipError_t launch_enc_nt(const EncParams &p, int waves, int nt, uint32_t cap, uint32_t xr, uint32_t pad, hipStream_t s) {
#ifdef AESW_DIAGNOSTIC
    if (nt == 3) return launch_enc<L, XT, KM, KEMIT, 3>(p, waves, cap, xr, pad, s);
    if (nt == 4) return launch_enc<L, XT, KM, KEMIT, 4>(p, waves, cap, xr, pad, s);
    if (nt == 5) return launch_enc<L, XT, KM, KEMIT, 5>(p, waves, cap, xr, pad, s);
#endif
    return nt == 2 ? launch_enc<L, XT, KM, KEMIT, 2>(p, waves, cap, xr, pad, s)
         : nt == 1 ? launch_enc<L, XT, KM, KEMIT, 1>(p, waves, cap, xr, pad, s)
                   : launch_enc<L, XT, KM, KEMIT, 0>(p, waves, cap, xr, pad, s);
}

template <int L, bool XT>
static hipError_t launch_enc_mode(const EncParams &p, int km, bool kemit, int waves, int nt, uint32_t cap, uint32_t xr, uint32_t pad, hipStream_t s) {
    if (km == KM_PRE) return launch_enc_nt<L, XT, KM_PRE, false>(p, waves, nt, cap, xr, pad, s);
    if (km == KM_SHARED) return launch_enc_nt<L, XT, KM_SHARED, false>(p, waves, nt, cap, xr, pad, s);
    return kemit ? launch_enc_nt<L, XT, KM_PBK, true>(p, waves, nt, cap, xr, pad, s) : launch_enc_nt<L, XT, KM_PBK, false>(p, waves, nt, cap, xr, pad, s);
}

hipError_t launch_encrypt(const EncParams &p, int layout, bool xt, int keymode, bool kemit, int waves, int nt,
                          uint32_t max_groups_in_flight, uint32_t xcd_remap, uint32_t pad, hipStream_t s) {
    if (waves < 1 || waves > 4 || keymode < 0 || keymode > 2) return hipErrorInvalidValue;
    const uint32_t cap = max_groups_in_flight;
    // a striding workgroup keeps its XCD class (id % 8) only when the stride is a multiple of 8
    const uint32_t xr = (cap == 0 || cap % 8 == 0) ? xcd_remap : 0u;
    if (layout == DENSE)
        return xt ? launch_enc_mode<DENSE, true>(p, keymode, kemit, waves, nt, cap, xr, pad, s)
                  : launch_enc_mode<DENSE, false>(p, keymode, kemit, waves, nt, cap, xr, pad, s);
    if (layout == VALUES)
        return xt ? launch_enc_mode<VALUES, true>(p, keymode, kemit, waves, nt, cap, xr, pad, s)
                  : launch_enc_mode<VALUES, false>(p, keymode, kemit, waves, nt, cap, xr, pad, s);
    return xt ? launch_enc_mode<PACKED, true>(p, keymode, kemit, waves, nt, cap, xr, pad, s)
              : launch_enc_mode<PACKED, false>(p, keymode, kemit, waves, nt, cap, xr, pad, s);
}

template <int L, bool XT, int NT>
static hipError_t launch_key_t(const KeyParams &p0, int waves, uint32_t xr, hipStream_t stream) {
    const int bpg = waves * BPW;
    const uint64_t groups = (p0.n + bpg - 1) / bpg;
    if (groups == 0) return hipSuccess;
    if (groups > 0x7fffffffull) return hipErrorInvalidValue;
    KeyParams p = p0;
    p.ngroups = (uint32_t)groups;
    p.xcd_remap = xr;
    const size_t lds = TAB_BYTES + (size_t)waves * (Stage<L>::KEY_BYTES + (p.rk ? Stage<L>::RK_BYTES_W : 0));
    {
        hipError_t e = allow_large_lds(reinterpret_cast<const void *>(&key_kernel<L, XT, NT>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((key_kernel<L, XT, NT>), dim3((unsigned)groups), dim3(waves * LANES), lds, stream, p);
    return hipGetLastError();
}

template <int L, bool XT>
static hipError_t launch_key_nt(const KeyParams &p, int waves, int nt, uint32_t xr, hipStream_t s) {
    return nt == 2 ? launch_key_t<L, XT, 2>(p, waves, xr, s) : nt == 1 ? launch_key_t<L, XT, 1>(p, waves, xr, s) : launch_key_t<L, XT, 0>(p, waves, xr, s);
}

hipError_t launch_key(const KeyParams &p, int layout, bool xt, int waves, int nt, uint32_t xcd_remap, hipStream_t s) {
    if (waves < 1 || waves > 4) return hipErrorInvalidValue;
    if (layout == DENSE) return xt ? launch_key_nt<DENSE, true>(p, waves, nt, xcd_remap, s) : launch_key_nt<DENSE, false>(p, waves, nt, xcd_remap, s);
    return xt ? launch_key_nt<PACKED, true>(p, waves, nt, xcd_remap, s) : launch_key_nt<PACKED, false>(p, waves, nt, xcd_remap, s);
}

// Dynamic LDS above 48 KiB needs hipFuncSetAttribute once per (kernel, device).  aesw_create() does it here for every
// instantiation a context can launch, so that no launch ever changes a function attribute later -- in particular not
// while the caller's stream is being captured into a hipGraph (tests/test_gpu_graph_capture.py).
template <int L, bool XT, int KM, bool KEMIT>
static hipError_t warm_enc_nt() {
    const size_t lds = TAB_BYTES + RKS_BYTES + 4 * (size_t)enc_wave_lds<L>(KEMIT);  // the largest group any option can ask for
    const size_t cap = lds > 65536 ? 65536 : lds;
    hipError_t e = allow_large_lds(reinterpret_cast<const void *>(&encrypt_kernel<L, XT, KM, KEMIT, 0>), cap);
    if (e == hipSuccess) e = allow_large_lds(reinterpret_cast<const void *>(&encrypt_kernel<L, XT, KM, KEMIT, 1>), cap);
    if (e == hipSuccess) e = allow_large_lds(reinterpret_cast<const void *>(&encrypt_kernel<L, XT, KM, KEMIT, 2>), cap);
    return e;
}
template <int L, bool XT>
static hipError_t warm_layout() {
    hipError_t e = warm_enc_nt<L, XT, KM_PRE, false>();
    if (e == hipSuccess) e = warm_enc_nt<L, XT, KM_SHARED, false>();
    if (e == hipSuccess) e = warm_enc_nt<L, XT, KM_PBK, false>();
    if (e == hipSuccess) e = warm_enc_nt<L, XT, KM_PBK, true>();
    return e;
}
template <int L, bool XT>
static hipError_t warm_key() {
    const size_t lds = TAB_BYTES + 4 * (size_t)(Stage<L>::KEY_BYTES + Stage<L>::RK_BYTES_W);
    hipError_t e = allow_large_lds(reinterpret_cast<const void *>(&key_kernel<L, XT, 0>), lds);
    if (e == hipSuccess) e = allow_large_lds(reinterpret_cast<const void *>(&key_kernel<L, XT, 1>), lds);
    if (e == hipSuccess) e = allow_large_lds(reinterpret_cast<const void *>(&key_kernel<L, XT, 2>), lds);
    return e;
}
template <bool XT>
static hipError_t warm_all() {
    hipError_t e = warm_layout<DENSE, XT>();
    if (e == hipSuccess) e = warm_layout<PACKED, XT>();
    if (e == hipSuccess) e = warm_layout<VALUES, XT>();
    if (e == hipSuccess) e = warm_key<DENSE, XT>();
    if (e == hipSuccess) e = warm_key<PACKED, XT>();
    return e;
}
hipError_t warm_launch_attributes() {
    hipError_t e = warm_all<true>();  // both table paths: "force_table_path" can flip a context later
    if (e == hipSuccess) e = warm_all<false>();
    return e;
}

hipError_t launch_table(const uint8_t *tables, uint8_t *t0, uint8_t *t1, uint8_t *t2, uint8_t *t3, hipStream_t s) {
    hipLaunchKernelGGL(table_kernel, dim3((66561 + 255) / 256), dim3(256), 0, s, tables, t0, t1, t2, t3);
    return hipGetLastError();
}

template <int NT>
static void launch_assemble_aligned(int geo, uint32_t k, uint32_t cols, hipStream_t s, const AssembleParams &p) {
    // 2^k rows x 2 pieces / (PIECES x THREADS) workgroups per column
    if (geo == 2) hipLaunchKernelGGL((assemble_fr_aligned_kernel<NT, 1, 256>), dim3(1u << (k - 7), cols), dim3(256), 0, s, p);
    else if (geo == 3) hipLaunchKernelGGL((assemble_fr_aligned_kernel<NT, 2, 256>), dim3(1u << (k - 8), cols), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((assemble_fr_aligned_kernel<NT, 2, 128>), dim3(1u << (k - 7), cols), dim3(128), 0, s, p);
}

hipError_t launch_assemble(const AssembleParams &p0, bool as_fr, int nt, hipStream_t s) {
    AssembleParams p = p0;
    {   // capacities of aes_callable (src/aes128.rs:303-325): computed here so that no kernel divides
        const uint64_t rows = (uint64_t)1 << p.k;
        p.cap0 = rows >= 1760 ? (rows - 1760) / AES_ROWS : 0;
        p.capn = rows / AES_ROWS;
    }
    const int choice = assemble_kernel_choice(as_fr, p.geometry, p.k, p.col_count);
    if (choice == 1) {
        const uint64_t rows = (uint64_t)1 << p.k;
        const uint64_t segs = 1 + (rows + AES_ROWS - 1) / AES_ROWS;  // head + blocks (+ the tail, clipped in the kernel)
        const dim3 grid((AES_ROWS * 2 + 255) / 256, (unsigned)segs, p.col_count);
        if (nt == 2) hipLaunchKernelGGL((assemble_fr_oneshot_kernel<2>), grid, dim3(256), 0, s, p);
        else if (nt == 1) hipLaunchKernelGGL((assemble_fr_oneshot_kernel<1>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((assemble_fr_oneshot_kernel<0>), grid, dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (choice == 2) {
        if (nt == 2) launch_assemble_aligned<2>(p.geometry, p.k, p.col_count, s, p);
        else if (nt == 1) launch_assemble_aligned<1>(p.geometry, p.k, p.col_count, s, p);
        else launch_assemble_aligned<0>(p.geometry, p.k, p.col_count, s, p);
        return hipGetLastError();
    }
    // the striding kernel: any K, any column count (also where geometry 1's segment grid or the aligned forms do not apply)
    const uint64_t cells = (uint64_t)p.col_count << p.k;
    uint64_t blocks = ((as_fr ? cells * 2 : cells / 4) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) blocks = 1;
    if (!as_fr) hipLaunchKernelGGL((assemble_kernel<false, 0>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (nt == 2) hipLaunchKernelGGL((assemble_kernel<true, 2>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (nt == 1) hipLaunchKernelGGL((assemble_kernel<true, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((assemble_kernel<true, 0>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

template <int GEO>
static hipError_t launch_expand_fr_oneshot(const uint8_t *cells, uint64_t n_cells, const u32x4 *lut, u32x4 *o, int nt, hipStream_t s) {
    const uint64_t per = GEO == 1 ? 256 : 1024;
    const uint64_t blocks = (n_cells * 2 + per - 1) / per;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (nt == 2) hipLaunchKernelGGL((expand_fr_oneshot_kernel<2, GEO>), dim3((unsigned)blocks), dim3(256), 0, s, cells, n_cells, lut, o);
    else if (nt == 1) hipLaunchKernelGGL((expand_fr_oneshot_kernel<1, GEO>), dim3((unsigned)blocks), dim3(256), 0, s, cells, n_cells, lut, o);
    else hipLaunchKernelGGL((expand_fr_oneshot_kernel<0, GEO>), dim3((unsigned)blocks), dim3(256), 0, s, cells, n_cells, lut, o);
    return hipGetLastError();
}

hipError_t launch_expand_fr(const uint8_t *cells, uint64_t n_cells, const void *fr_lut, void *out, int nt, int geometry, hipStream_t s) {
    if (n_cells == 0) return hipSuccess;
    if (geometry == 1) return launch_expand_fr_oneshot<1>(cells, n_cells, reinterpret_cast<const u32x4 *>(fr_lut), reinterpret_cast<u32x4 *>(out), nt, s);
    if (geometry == 2) return launch_expand_fr_oneshot<2>(cells, n_cells, reinterpret_cast<const u32x4 *>(fr_lut), reinterpret_cast<u32x4 *>(out), nt, s);
    uint64_t blocks = (n_cells * 2 + 256 * 8 - 1) / (256 * 8);
    if (blocks > 256 * 8) blocks = 256 * 8;
    const u32x4 *lut = reinterpret_cast<const u32x4 *>(fr_lut);
    u32x4 *o = reinterpret_cast<u32x4 *>(out);
    if (nt == 2) hipLaunchKernelGGL(expand_fr_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, cells, n_cells, lut, o);
    else if (nt == 1) hipLaunchKernelGGL(expand_fr_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, cells, n_cells, lut, o);
    else hipLaunchKernelGGL(expand_fr_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, cells, n_cells, lut, o);
    return hipGetLastError();
}

}  // namespace aesw
