// aesw_lane.h -- what ONE lane computes and emits.
//
// Work decomposition (DESIGN.md "kernel"): a lane owns one AES state column
// (word w = bytes 4w..4w+3, FIPS column-major, byte j in bits 8j) of one block;
// 4 lanes = one block, a 64-lane wavefront = 16 blocks.  All GF(2^8) work is
// done on four packed bytes per 32-bit register, and every value the
// reference's regions assign is assembled into whole dwords with v_perm_b32
// before it is written to the staging slab, so no byte-granular stores exist.
//
// The functions are __host__ __device__ so the same source is exercised on the
// CPU by tests/lane_model (perm() is emulated there) against the oracle before
// any GPU time is spent.  They never touch memory directly: a Sink supplies
//   plain<COL>(off, v): dword v at staging-window byte offset off + 4*w (Win<>::woff(R) + row offset)
//   mix<COL>(off, k, v): dword k of this lane's lcon() record group
// which is LDS on the device and the output arrays in the host model.
#pragma once
#include "aesw_layout.h"

namespace aesw {

// v_perm_b32: result byte i = bytes{hi:lo}[sel.byte[i]]; 0..3 -> lo, 4..7 -> hi,
// 12 -> 0x00.
AESW_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 0xff;
        uint32_t b;
        if (s < 4) b = (lo >> (8 * s)) & 0xff;
        else if (s < 8) b = (hi >> (8 * (s - 4))) & 0xff;
        else if (s == 12) b = 0;
        else b = 0xff;  // not used by this file
        r |= b << (8 * i);
    }
    return r;
#endif
}

// selector: result byte 0 <- source b0, ..., byte 3 <- source b3
AESW_HD constexpr uint32_t SEL(int b0, int b1, int b2, int b3) {
    return (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24);
}
constexpr int Z_ = 12;  // constant zero byte
constexpr int X_ = 0;   // don't care

// GF(2^8) doubling of four packed bytes (== MUL_BY_2 of src/constant.rs:17-31
// when the host's table is the xtime table; aesw_create() checks that).
AESW_HD uint32_t xtime4(uint32_t x) {
    return ((x & 0x7f7f7f7fu) << 1) ^ (((x >> 7) & 0x01010101u) * 0x1bu);
}

// round constants, src/utils.rs:28
AESW_HD constexpr uint32_t rcon(int i) {
    return i == 0 ? 1 : i == 1 ? 2 : i == 2 ? 4 : i == 3 ? 8 : i == 4 ? 16 : i == 5 ? 32 : i == 6 ? 64
         : i == 7 ? 128 : i == 8 ? 27 : 54;
}

// Byte tables: sbox at [0,256), mul2 at [256,512), mul3 at [512,768).
template <bool XT_ARITH>
struct Tables {
    const uint8_t *t;
    AESW_HD uint32_t look4(uint32_t x, int base) const {
        const uint32_t b0 = t[base + (x & 0xff)];
        const uint32_t b1 = t[base + ((x >> 8) & 0xff)];
        const uint32_t b2 = t[base + ((x >> 16) & 0xff)];
        const uint32_t b3 = t[base + (x >> 24)];
        return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    AESW_HD uint32_t sbox4(uint32_t x) const { return look4(x, 0); }  // src/utils.rs:22-24
    AESW_HD uint32_t mul2_4(uint32_t x) const { return XT_ARITH ? xtime4(x) : look4(x, 256); }
    AESW_HD uint32_t mul3_4(uint32_t x, uint32_t m2) const { return XT_ARITH ? (m2 ^ x) : look4(x, 512); }
};

// ShiftRows for the lane owning word w: s0..s3 are the SubBytes words of lanes
// w, w+1, w+2, w+3 (mod 4): shifted[w][j] = subbed[(w+j)%4][j], src/aes128.rs:216-223.
AESW_HD uint32_t shift_rows(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3) {
    return (s0 & 0x000000ffu) | (s1 & 0x0000ff00u) | (s2 & 0x00ff0000u) | (s3 & 0xff000000u);
}

// rows 0..31 of a block: plaintext rows and the initial AddRoundKey
// (src/aes128.rs:176-198).  Returns the state word.
template <int L, class S>
AESW_HD uint32_t emit_head(S &s, uint32_t ptw, uint32_t rk0w) {
    using G = Geo<L>;
    if (G::HAS_X) {
        s.template plain<0>(G::X_H_PT, ptw);
        s.template plain<0>(G::X_H_ARK, ptw);
    }
    if (L == DENSE) {
        s.template plain<1>(0, 0u);
        s.template plain<2>(0, 0u);
    }
    if (G::Y_COPIES) s.template plain<1>(G::Y_H_ARK, rk0w);
    const uint32_t st = ptw ^ rk0w;
    s.template plain<2>(G::Z_H_ARK, st);
    return st;
}

// SubBytes rows of a round (src/aes128.rs:203-209 -> sbox_chip.rs:57-83):
// x = state byte, y = S_BOX[x].  rel* = where this round starts in the block's staging window.
template <int L, class S, class T>
AESW_HD uint32_t emit_sbox(S &s, int relx, int rely, int relz, uint32_t st, const T &tab) {
    using G = Geo<L>;
    if (G::HAS_X) s.template plain<0>(relx + G::X_SBOX, st);
    const uint32_t sub = tab.sbox4(st);
    s.template plain<1>(rely + G::Y_SBOX, sub);
    if (L == DENSE) s.template plain<2>(relz + G::Z_SBOX, 0u);
    return sub;
}

// MixColumns + AddRoundKey rows of rounds 1..9 for the lane's word:
// four lcon() calls (src/aes128.rs:236-248, :268-301), 7 rows each:
//   rows t=0..3: x = sh[t], y = c*sh[t] when c in {2,3}      (gf_mul_chip.rs:59-89 / copy :279-288)
//   row 4: (tmp0, tmp1, i1)  row 5: (tmp2, tmp3, i2)  row 6: (i1, i2, mixed)   (u8_xor_chip.rs:63-100)
// then x = mixed, y = rk, z = mixed ^ rk (src/aes128.rs:250-261).
// Returns the next state word.
template <int L, class S, class T>
AESW_HD uint32_t emit_mix_ark(S &s, int relx, int rely, int relz, uint32_t sh, uint32_t rkw, const T &tab) {
    using G = Geo<L>;
    const uint32_t m2 = tab.mul2_4(sh);
    const uint32_t m3 = tab.mul3_4(sh, m2);
    // T_t = (tmp[t] of matrix rows m=0..3) packed by m; column t of the matrix:
    // t=0: 2,1,1,3   t=1: 3,2,1,1   t=2: 1,3,2,1   t=3: 1,1,3,2
    const uint32_t q0 = perm(m3, m2, SEL(0, X_, X_, 4));
    const uint32_t T0 = perm(sh, q0, SEL(0, 4, 4, 3));
    const uint32_t q1 = perm(m3, m2, SEL(5, 1, X_, X_));
    const uint32_t T1 = perm(sh, q1, SEL(0, 1, 5, 5));
    const uint32_t q2 = perm(m3, m2, SEL(X_, 6, 2, X_));
    const uint32_t T2 = perm(sh, q2, SEL(6, 1, 2, 6));
    const uint32_t q3 = perm(m3, m2, SEL(X_, X_, 7, 3));
    const uint32_t T3 = perm(sh, q3, SEL(7, 7, 2, 3));
    const uint32_t I1 = T0 ^ T1;
    const uint32_t I2 = T2 ^ T3;
    const uint32_t MX = I1 ^ I2;

    // x: per m the 7 bytes [sh0 sh1 sh2 sh3 T0.m T2.m I1.m]
    if (G::HAS_X) {
        const int o = relx + G::X_MIX;
        s.template mix<0>(o, 0, sh);
        uint32_t u = perm(T2, T0, SEL(0, 4, X_, X_));
        u = perm(I1, u, SEL(0, 1, 4, X_));
        s.template mix<0>(o, 1, perm(sh, u, SEL(0, 1, 2, 4)));
        s.template mix<0>(o, 2, perm(T0, sh, SEL(1, 2, 3, 5)));
        u = perm(I1, T2, SEL(1, 5, X_, X_));
        s.template mix<0>(o, 3, perm(sh, u, SEL(0, 1, 4, 5)));
        const uint32_t pb = perm(T2, T0, SEL(2, 6, 3, 7));  // [T0.2 T2.2 T0.3 T2.3]
        s.template mix<0>(o, 4, perm(pb, sh, SEL(2, 3, 4, 5)));
        s.template mix<0>(o, 5, perm(I1, sh, SEL(6, 0, 1, 2)));
        u = perm(I1, pb, SEL(X_, 2, 3, 7));
        s.template mix<0>(o, 6, perm(sh, u, SEL(7, 1, 2, 3)));
    }
    // y
    {
        const int o = rely + G::Y_MIX;
        if (L == DENSE) {
            // per m: [c0*a0|0, c1*a1|0, c2*a2|0, c3*a3|0, T1.m, T3.m, I2.m]
            s.template mix<1>(o, 0, perm(T1, T0, SEL(0, 4, Z_, Z_)));
            uint32_t u = perm(T3, T1, SEL(0, 4, X_, X_));
            s.template mix<1>(o, 1, perm(I2, u, SEL(0, 1, 4, Z_)));
            s.template mix<1>(o, 2, perm(T2, T1, SEL(1, 5, Z_, 1)));
            s.template mix<1>(o, 3, perm(I2, T3, SEL(1, 5, Z_, Z_)));
            u = perm(T3, T2, SEL(2, 6, X_, 6));
            s.template mix<1>(o, 4, perm(T1, u, SEL(0, 1, 6, 3)));
            s.template mix<1>(o, 5, perm(T0, I2, SEL(2, 7, Z_, Z_)));
            u = perm(T3, T1, SEL(7, 3, 7, X_));
            s.template mix<1>(o, 6, perm(I2, u, SEL(0, 1, 2, 7)));
        } else if (L == VALUES) {
            // per m: the two products in t order: m0 [2a0 3a1]  m1 [2a1 3a2]  m2 [2a2 3a3]  m3 [3a0 2a3]
            s.template mix<1>(o, 0, perm(m3, m2, SEL(0, 5, 1, 6)));
            s.template mix<1>(o, 1, perm(m3, m2, SEL(2, 7, 4, 3)));
        } else {
            // per m: the two products in t order, then T1.m, T3.m, I2.m
            uint32_t u = perm(T1, T0, SEL(0, 4, 4, X_));
            s.template mix<1>(o, 0, perm(T3, u, SEL(0, 1, 2, 4)));
            u = perm(T2, T1, SEL(X_, 1, 5, 1));
            s.template mix<1>(o, 1, perm(I2, u, SEL(4, 1, 2, 3)));
            u = perm(I2, T3, SEL(1, 5, X_, 2));
            s.template mix<1>(o, 2, perm(T2, u, SEL(0, 1, 6, 3)));
            u = perm(T3, T1, SEL(2, 6, X_, X_));
            const uint32_t v = perm(T0, I2, SEL(X_, X_, 2, 7));
            s.template mix<1>(o, 3, perm(v, u, SEL(0, 1, 6, 7)));
            u = perm(T3, T1, SEL(7, 3, 7, X_));
            s.template mix<1>(o, 4, perm(I2, u, SEL(0, 1, 2, 7)));
        }
    }
    // z
    {
        const int o = relz + G::Z_MIX;
        if (L == DENSE) {
            // per m: [0 0 0 0 I1.m I2.m MX.m]
            s.template mix<2>(o, 0, 0u);
            uint32_t u = perm(I2, I1, SEL(0, 4, X_, X_));
            s.template mix<2>(o, 1, perm(MX, u, SEL(0, 1, 4, Z_)));
            s.template mix<2>(o, 2, perm(I1, I1, SEL(Z_, Z_, Z_, 1)));
            s.template mix<2>(o, 3, perm(MX, I2, SEL(1, 5, Z_, Z_)));
            s.template mix<2>(o, 4, perm(I2, I1, SEL(Z_, Z_, 2, 6)));
            s.template mix<2>(o, 5, perm(MX, MX, SEL(2, Z_, Z_, Z_)));
            u = perm(I2, I1, SEL(X_, 3, 7, X_));
            s.template mix<2>(o, 6, perm(MX, u, SEL(Z_, 1, 2, 7)));
        } else {
            // per m: [I1.m I2.m MX.m]
            uint32_t u = perm(I2, I1, SEL(0, 4, X_, 1));
            s.template mix<2>(o, 0, perm(MX, u, SEL(0, 1, 4, 3)));
            u = perm(I2, I1, SEL(5, X_, 2, 6));
            s.template mix<2>(o, 1, perm(MX, u, SEL(0, 5, 2, 3)));
            u = perm(I2, I1, SEL(X_, 3, 7, X_));
            s.template mix<2>(o, 2, perm(MX, u, SEL(6, 1, 2, 7)));
        }
    }
    // AddRoundKey rows
    if (G::HAS_X) s.template plain<0>(relx + G::X_ARK, MX);
    if (G::Y_COPIES) s.template plain<1>(rely + G::Y_ARK, rkw);
    const uint32_t nx = MX ^ rkw;
    s.template plain<2>(relz + G::Z_ARK, nx);
    return nx;
}

// Round 10 after its SubBytes rows: mixed = shifted (src/aes128.rs:236-237),
// x = sh, y = rk10, z = ciphertext (:250-261).  rel* = start of the tail.
template <int L, class S>
AESW_HD uint32_t emit_final_ark(S &s, int relx, int rely, int relz, uint32_t sh, uint32_t rkw) {
    using G = Geo<L>;
    if (G::HAS_X) s.template plain<0>(relx + G::X_T_ARK, sh);
    if (G::Y_COPIES) s.template plain<1>(rely + G::Y_T_ARK, rkw);
    const uint32_t ct = sh ^ rkw;
    s.template plain<2>(relz + G::Z_T_ARK, ct);
    return ct;
}

// One key-schedule round rho (1..10) for the lane owning key word w
// (src/key_schedule.rs:122-224).  k0..k3 = the previous round key's four words
// (every lane of the quad sees all four), rc = rcon(rho-1).  KSink:
//   kx/ky/kz(off, v): dword at key-slab byte offset off (already includes 4*w)
//   words(off, v): dword at words_column byte offset off
// Lane roles for the rows that are not per-word: lane 0 -> sbox rows 0..3,
// lane 1 -> rcon xor rows 4..7, lane 2 -> words_column "shift" rows, lane 3 ->
// words_column rc rows.  Returns the lane's new key word.
template <int L, class KS, class T>
AESW_HD uint32_t emit_key_round(KS &ks, int rho, int w, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3,
                                uint32_t rc, const T &tab) {
    using G = Geo<L>;
    const uint32_t rot = (k3 >> 8) | (k3 << 24);  // bytes 13,14,15,12 (:141-154)
    const uint32_t sub = tab.sbox4(rot);          // :156-159
    const uint32_t rconned = sub ^ rc;            // :189-194 (rc word = [rcon,0,0,0], :161-187)
    const uint32_t n0 = k0 ^ rconned;             // :197-204
    const uint32_t n1 = k1 ^ n0;                  // :207-216
    const uint32_t n2 = k2 ^ n1;
    const uint32_t n3 = k3 ^ n2;
    const uint32_t own_old = w == 0 ? k0 : w == 1 ? k1 : w == 2 ? k2 : k3;
    const uint32_t own_new = w == 0 ? n0 : w == 1 ? n1 : w == 2 ? n2 : n3;
    const uint32_t prev_new = w == 0 ? rconned : w == 1 ? n0 : w == 2 ? n1 : n2;
    const int bx = G::KX_ROUND * (rho - 1) + 4 * w;
    const int by = G::KY_ROUND * (rho - 1) + 4 * w;
    const int bz = G::KZ_ROUND * (rho - 1) + 4 * w - G::KZ_SHIFT;
    // xor rows 8+4w..: (old word, previous new word / rconned, new word)
    ks.kx(bx + 8, own_old);
    ks.ky(by + 8, prev_new);
    ks.kz(bz + 8, own_new);
    // range-check rows 24+4w.. (:218-221): x only
    ks.kx(bx + 24, own_new);
    if (L == DENSE) {
        ks.ky(by + 24, 0u);
        ks.kz(bz + 24, 0u);
    }
    if (w < 2) {
        // w==0: rows 0..3 (x = shifted, y = S_BOX[shifted]); w==1: rows 4..7 (sub, rc, rconned)
        ks.kx(bx, w == 0 ? rot : sub);
        ks.ky(by, w == 0 ? sub : rc);
        if (w == 1) ks.kz(bz, rconned);
        else if (L == DENSE) ks.kz(bz, 0u);
    } else {
        // words_column: 16 key bytes, then per round [shifted(4), rc,0,0,0]
        ks.words(16 + 8 * (rho - 1) + 4 * (w - 2), w == 2 ? rot : rc);
    }
    return own_new;
}

}  // namespace aesw
