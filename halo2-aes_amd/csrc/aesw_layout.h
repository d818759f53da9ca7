// aesw_layout.h -- geometry of the witness slabs, shared by the HIP kernels, the
// C ABI's pure-host helpers and the host-side lane model used by the tests.
//
// The slab map is derived from the reference's region call order
// (src/aes128.rs:154-301, src/key_schedule.rs:80-224); DESIGN.md "slab map"
// spells it out.  Nothing here is copied from the reference: the numbers are
// byte offsets of OUR column-major buffers.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AESW_HD __host__ __device__ __forceinline__
#else
#define AESW_HD inline
#endif

namespace aesw {

// DENSE: exact image of the advice rows.  PACKED: assigned cells only, row order.  VALUES: only the cells
// whose value a chip's closure computes -- y of the S-box and mul rows, z of the xor rows; column x and the
// y cells of xor rows are copy_advice() of earlier cells in the reference (src/chips/*.rs) and are omitted.
enum : int { DENSE = 0, PACKED = 1, VALUES = 2 };

constexpr int AES_ROWS = 1360;  // src/constant.rs:114
constexpr int KEY_ROWS = 400;   // 10 rounds x 40 one-row chip regions (src/key_schedule.rs:122-224)
constexpr int WORDS_ROWS = 96;  // 16 + 10 x (4 + 4) rows of words_column
constexpr int RK_BYTES = 176;   // 11 round keys

// ---- encrypt slab -----------------------------------------------------------
// Per column: HEAD bytes (rows 0..31: plaintext rows + initial AddRoundKey,
// src/aes128.rs:176-198), nine ROUND-sized pieces (rounds 1..9, :201-262) and
// TAIL bytes (round 10).  Inside a round: SBOX rows, 16 lcon() records of
// MIXW/4... bytes per lane-word, ARK rows.
template <int L>
struct Geo {
    static constexpr bool HAS_X = L != VALUES;      // column x is emitted at all
    static constexpr bool Y_COPIES = L != VALUES;   // y cells of xor rows (round keys, tmp1/tmp3/i2) are emitted
    // bytes per block
    static constexpr int XS = HAS_X ? 1360 : 0;
    static constexpr int YS = L == DENSE ? 1360 : L == PACKED ? 1056 : 448;
    static constexpr int ZS = L == DENSE ? 1360 : 608;
    // head / round / tail bytes
    static constexpr int X_HEAD = 32, X_ROUND = 144, X_TAIL = 32;
    static constexpr int Y_HEAD = L == DENSE ? 32 : L == PACKED ? 16 : 0, Y_ROUND = L == DENSE ? 144 : L == PACKED ? 112 : 48,
                         Y_TAIL = L == VALUES ? 16 : 32;
    static constexpr int Z_HEAD = L == DENSE ? 32 : 16, Z_ROUND = L == DENSE ? 144 : 64,
                         Z_TAIL = L == DENSE ? 32 : 16;
    // offsets inside the head (relative to the head start)
    static constexpr int X_H_PT = 0, X_H_ARK = 16;        // rows 0..15 pt, rows 16..31 pt again
    static constexpr int Y_H_ARK = L == DENSE ? 16 : 0;   // rk0
    static constexpr int Z_H_ARK = L == DENSE ? 16 : 0;   // pt ^ rk0
    // offsets inside a round (relative to the round start)
    static constexpr int X_SBOX = 0, X_MIX = 16, X_MIXW = 28, X_ARK = 128;
    static constexpr int Y_SBOX = 0, Y_MIX = 16, Y_MIXW = L == DENSE ? 28 : L == PACKED ? 20 : 8,
                         Y_ARK = L == DENSE ? 128 : 96 /* not in VALUES */;
    static constexpr int Z_SBOX = 0 /* dense only: zeros */, Z_MIX = L == DENSE ? 16 : 0,
                         Z_MIXW = L == DENSE ? 28 : 12, Z_ARK = L == DENSE ? 128 : 48;
    // offsets inside the tail (round 10: sbox rows, then ShiftRows ^ rk10)
    static constexpr int X_T_SBOX = 0, X_T_ARK = 16;
    static constexpr int Y_T_SBOX = 0, Y_T_ARK = 16 /* not in VALUES */;
    static constexpr int Z_T_ARK = L == DENSE ? 16 : 0;
    // dwords per lcon record group of one lane-word (4 output bytes x 7 rows)
    static constexpr int X_MIXD = 7, Y_MIXD = L == DENSE ? 7 : L == PACKED ? 5 : 2, Z_MIXD = L == DENSE ? 7 : 3;

    // ---- key slab (per key): rounds of KX_ROUND bytes, no head.  VALUES keeps the PACKED key slab (one key
    // slab per circuit in the reference's call shape; nothing to save there)
    static constexpr int KXS = 400;
    static constexpr int KYS = L == DENSE ? 400 : 240;
    static constexpr int KZS = L == DENSE ? 400 : 200;
    static constexpr int KX_ROUND = 40, KY_ROUND = L == DENSE ? 40 : 24, KZ_ROUND = L == DENSE ? 40 : 20;
    // kz packed drops rows 0..3 of every round (sbox rows have no z)
    static constexpr int KZ_SHIFT = L == DENSE ? 0 : 4;
};

// LDS staging window of one block, one column (DESIGN.md "staging window").
// HBM wants whole 128-byte lines: per-round runs (144/112/64 B at 16-byte
// alignment) write every line two or three times partially and cap at ~3 TB/s,
// whole lines reach ~5.5 TB/s (tools/storebench.hip).  So after every round a
// wave flushes only the lines that are complete; the <=112 not yet flushed
// bytes of a block stay in LDS.  Window layout (bytes, block-relative offset o):
//   [0, PERM_END)            head + the first PERM_R rounds, kept until the end:
//                            the first (<128) bytes of a block share a line with
//                            the previous block's tail, which completes last
//   NSLOT slots of ROUND     rounds PERM_R+1..9, slot = (R-PERM_R-1) % NSLOT
//   TAIL                     round 10
// padded to BYTES = 16 (mod 32) so that the same offset of the 8 blocks of a
// half-wave falls on 8 different 4-bank groups (conflict-free ds_write_b32).
template <int HEAD_, int ROUND_, int TAIL_, int GSTRIDE_>
struct Win {
    static constexpr int HEAD = HEAD_, ROUND = ROUND_, TAIL = TAIL_, GSTRIDE = GSTRIDE_;
    static constexpr int PERM_R = HEAD + ROUND >= 128 ? 1 : (HEAD + 2 * ROUND >= 128 ? 2 : 3);
    static constexpr int NSLOT = 1 + (112 + ROUND - 1) / ROUND;  // rounds that can hold unflushed bytes, +1 being written
    static constexpr int PERM_END = HEAD + PERM_R * ROUND;
    static constexpr int SLOT0 = PERM_END;
    static constexpr int TAIL0 = SLOT0 + NSLOT * ROUND;
    static constexpr int RAW = TAIL0 + TAIL;
    static constexpr int BYTES = RAW + ((16 - RAW % 32) + 32) % 32;
    static_assert(HEAD + 9 * ROUND + TAIL == GSTRIDE, "column geometry");
    static_assert(HEAD % 16 == 0 && ROUND % 16 == 0 && TAIL % 16 == 0 && GSTRIDE % 16 == 0, "16-byte pieces");
    static_assert((16 * GSTRIDE) % 128 == 0, "a wave's 16-block range is line aligned");
    // block-relative offset where round R starts (R = 10: the tail) / ends
    AESW_HD static constexpr int start(int R) { return HEAD + ROUND * (R - 1); }
    AESW_HD static constexpr int end(int R) { return R >= 10 ? GSTRIDE : HEAD + ROUND * R; }
    // window offset where round R's bytes are staged
    AESW_HD static constexpr int woff(int R) {
        return R <= PERM_R ? start(R) : (R >= 10 ? TAIL0 : SLOT0 + ((R - PERM_R - 1) % NSLOT) * ROUND);
    }
};
template <int L> using WinX = Win<Geo<L>::X_HEAD, Geo<L>::X_ROUND, Geo<L>::X_TAIL, 1360>;  // VALUES: a valid type, never staged
template <int L> using WinY = Win<Geo<L>::Y_HEAD, Geo<L>::Y_ROUND, Geo<L>::Y_TAIL, Geo<L>::YS>;
template <int L> using WinZ = Win<Geo<L>::Z_HEAD, Geo<L>::Z_ROUND, Geo<L>::Z_TAIL, Geo<L>::ZS>;

// Whole-line flush after round R (1..9; flush 9 also carries round 10).
// For block b of a wave (column bytes [b*GSTRIDE, (b+1)*GSTRIDE) of the wave's
// line-aligned 16-block range) the lines that just became complete are
// [lo, hi); piece (t, sub) is 16 bytes of line lo+t.  Returns where the piece
// sits in the wave's LDS column stage (block windows of W::BYTES) and where it
// goes in the wave's global range.  This closed form is the SPECIFICATION (when a line leaves, from which LDS
// bytes); the kernel runs the table-driven "scheduled flush" below, which the tests check against it.
struct FlushPiece {
    bool ok;
    int lds_off;  // relative to the wave's stage of this column
    int P;        // byte offset in the wave's 16-block column range
};

template <class W>
AESW_HD constexpr int flush_maxc(int R) {
    return ((R == 1 ? W::end(1) : W::end(R == 9 ? 10 : R) - W::end(R - 1)) + 127) / 128 + (R == 9 ? 1 : 0);
}

template <class W>
AESW_HD FlushPiece flush_piece(int R, int b, int sub, int t, int nvalid) {
    const int rmin = R - (W::NSLOT - 1) < 1 ? 1 : R - (W::NSLOT - 1);
    const int base = b * W::GSTRIDE;
    // Block b's own rounds never flush the line that holds the previous block's tail: it leaves with that
    // block's last flush.  While head + rounds stay short of the first line boundary (z, values-only y) the
    // lower bound is therefore the first line that STARTS inside the block, not the line base falls into.
    const int first = (base + 127) >> 7;
    const int lo_raw = R == 1 ? first : (base + W::end(R - 1)) >> 7;
    const int lo = lo_raw < first ? first : lo_raw;
    const int hi = R == 9 ? (base + W::GSTRIDE + 127) >> 7 : (base + W::end(R)) >> 7;
    const int k = lo + t;
    const int P = 128 * k + 16 * sub;
    int o = P - base;  // block-relative offset of this piece
    bool ok = k < hi && b < nvalid;
    int a = b * W::BYTES;
    int adj = W::woff(rmin) - W::start(rmin);  // window offset = o + adj
    for (int r = rmin + 1; r <= R; ++r) adj = o >= W::start(r) ? W::woff(r) - W::start(r) : adj;
    if (R == 9) {
        adj = o >= W::start(10) ? W::woff(10) - W::start(10) : adj;
        if (o >= W::GSTRIDE) {
            // the tail of this line is the next block's head, which its window kept
            ok = ok && b + 1 < nvalid;
            a += W::BYTES;
            o -= W::GSTRIDE;
            adj = 0;
        }
    }
    return FlushPiece{ok, a + o + adj, P};
}

// ---- scheduled flush (round 2) ---------------------------------------------------------------------------
// The set of 128-byte lines of a wave's 16-block column range that complete in round R is the same for every full
// wave: it only depends on the column geometry.  So the flush is a fixed schedule instead of per-round address
// arithmetic: the lines of round R in address order, eight per store instruction (lane = (line slot lane>>3, 16-byte
// piece lane&7)), and for every (instruction, lane) ONE descriptor word that says where the piece sits in the wave's
// LDS stage and where it goes in the wave's global range.  The kernel loads its descriptors once (one dword per
// lane and instruction, from a table the host builds with build_flush_table()) and keeps them in registers; a piece
// then costs two VALU instructions instead of ~12, and ~55 instead of ~90 store instructions leave per wave.
// flush_piece() above remains the specification of WHEN a line leaves and where its bytes are staged; the tests
// check that the schedule stores every piece exactly once from the same LDS bytes.
//   descriptor = lds_off | P << 16      lds_off: byte offset in the column's wave stage (the kernel adds the stage base;
//                                        the sum must stay below 64 KiB), P: byte offset in the wave's global range;
//   an unused slot has P = SCHED_INVALID_P (above any range), so "P < nvalid*GSTRIDE" is the store predicate of a
//   partial wave and of a partial instruction alike.
constexpr int SCHED_BPW = 16;
constexpr uint32_t SCHED_INVALID_P = 0x7ff0u;

// the round (1..9) whose flush carries block-relative byte o: the head leaves with round 1, round 10 with round 9
template <class W>
AESW_HD constexpr int sched_round_of(int o) {
    for (int R = 1; R <= 8; ++R)
        if (o < W::end(R)) return R;
    return 9;
}
// the round in which line k of the wave's range is complete
template <class W>
AESW_HD constexpr int sched_line_round(int k) {
    int r = 1;
    for (int s = 0; s < 8; ++s) {
        const int P = 128 * k + 16 * s, b = P / W::GSTRIDE, o = P - b * W::GSTRIDE;
        const int q = sched_round_of<W>(o);
        r = q > r ? q : r;
    }
    return r;
}
template <class W>
AESW_HD constexpr int sched_nlines(int R) {
    int n = 0;
    for (int k = 0; k < SCHED_BPW * W::GSTRIDE / 128; ++k) n += sched_line_round<W>(k) == R ? 1 : 0;
    return n;
}
template <class W> AESW_HD constexpr int sched_ninstr(int R) { return (sched_nlines<W>(R) + 7) / 8; }
// index of round R's first instruction in the column's descriptor list; sched_first(10) = their total number
template <class W>
AESW_HD constexpr int sched_first(int R) {
    int n = 0;
    for (int r = 1; r < R; ++r) n += sched_ninstr<W>(r);
    return n;
}
// where block-relative byte o is staged inside the block's window
template <class W>
AESW_HD constexpr int sched_window_offset(int o) {
    if (o < W::HEAD) return o;
    const int r = o >= W::start(10) ? 10 : (o - W::HEAD) / W::ROUND + 1;
    return W::woff(r) + (o - W::start(r));
}
// LDS bank cost of one ds_read_b128 whose lane l reads 16 bytes at addr[l] (addr < 0: lane idle).  gfx950 serves the
// instruction in four passes of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32 -- over 64 banks
// of 4 bytes (MI355X_MICROARCH.md, LDS); a pass needs as many cycles as its busiest bank has distinct dwords.
inline int b128_read_conflict_cost(const int addr[64]) {
    static const int pass_lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                          {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int cost = 0;
    for (int half = 0; half < 2; ++half)
        for (int p = 0; p < 2; ++p) {
            int load[64] = {0};
            int worst = 1;
            for (int i = 0; i < 16; ++i) {
                const int a = addr[pass_lanes[p][i] + 32 * half];
                if (a < 0) continue;
                for (int d = 0; d < 4; ++d) {
                    const int bank = ((a >> 2) + d) & 63;
                    if (++load[bank] > worst) worst = load[bank];
                }
            }
            cost += worst - 1;
        }
    return cost;
}

// Pure host: the column's whole table, sched_first<W>(10) * 64 words, instruction-major (word i*64 + lane).
// Which line takes which slot of its round's instructions does not matter to the global stores (every slot is one
// whole line); it matters to the LDS: the eight lines of an instruction are gathered by ONE ds_read_b128, and line
// starts that collide in the banks cost cycles (PMC, round 2: 58 % of the flush's LDS-active cycles were bank
// conflicts with the lines in address order; this model reproduces the measured 308 conflict cycles per wave as 304).
// The four passes of a ds_read_b128 pair up slots 0-3 and slots 4-7 independently, so a round's lines are dealt into
// QUADS: greedily the cheapest quad that contains the first line still free (its three partners and the split into the
// slot pairs {0,3} / {1,2} by exhaustive search), then pairwise swaps between slots while the summed cost drops.
// Deterministic; ~20 ms per layout.  Unused slots all fall into the round's last instruction.
template <class W>
inline void build_flush_table(uint32_t *out) {
    // the descriptor packs a 16-bit LDS offset below P: every valid P must lie below the "unused slot" sentinel, and a
    // wave's stage of this column must be addressable with 16 bits (the kernel adds the stage base: launch_enc checks the sum)
    static_assert((uint32_t)(SCHED_BPW * W::GSTRIDE) <= SCHED_INVALID_P, "a wave's global range of this column reaches the unused-slot sentinel");
    static_assert(SCHED_BPW * W::BYTES <= 65536, "a wave's LDS stage of this column needs more than 16 address bits");
    static_assert(SCHED_INVALID_P < 0x8000u, "P << 16 must fit the descriptor word's upper half");
    constexpr int NL = SCHED_BPW * W::GSTRIDE / 128;
    int line_round[NL];
    for (int k = 0; k < NL; ++k) line_round[k] = sched_line_round<W>(k);
    auto piece = [](int line, int sub, int *lds, int *P) {
        *P = 128 * line + 16 * sub;
        const int b = *P / W::GSTRIDE, o = *P - b * W::GSTRIDE;
        *lds = b * W::BYTES + sched_window_offset<W>(o);
    };
    auto quad_cost = [&](const int q[4]) {  // four line slots = lanes 0..31 of one instruction; -1 = unused slot
        int addr[64];
        for (int l = 0; l < 64; ++l) addr[l] = -1;
        for (int sl = 0; sl < 4; ++sl)
            if (q[sl] >= 0)
                for (int sub = 0; sub < 8; ++sub) {
                    int P;
                    piece(q[sl], sub, &addr[8 * sl + sub], &P);
                }
        return b128_read_conflict_cost(addr);
    };
    auto best_split = [&](const int m[4], int q[4]) {  // the three ways to pair four members onto slots {0,3} and {1,2}
        static const int pairings[3][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {0, 3, 1, 2}};
        int best = 1 << 30;
        for (const auto &p : pairings) {
            const int t[4] = {m[p[0]], m[p[2]], m[p[3]], m[p[1]]};
            const int c = quad_cost(t);
            if (c < best) {
                best = c;
                for (int i = 0; i < 4; ++i) q[i] = t[i];
            }
        }
        return best;
    };
    int idx = 0;
    for (int R = 1; R <= 9; ++R) {
        int lines[NL], n = 0;
        for (int k = 0; k < NL; ++k)
            if (line_round[k] == R) lines[n++] = k;
        const int ni = (n + 7) / 8;
        int slots[NL + 8];  // the round's line slots, instruction-major; -1 = unused
        for (int i = 0; i < 8 * ni; ++i) slots[i] = -1;
        bool used[NL] = {false};
        int left = n, nq = 0;
        while (left > 0) {
            int f = 0;
            while (used[f]) ++f;
            int best = 1 << 30, pick[4] = {f, -1, -1, -1}, q[4], bq[4] = {lines[f], -1, -1, -1};
            if (left >= 4) {
                for (int a = f + 1; a < n; ++a) {
                    if (used[a]) continue;
                    for (int b = a + 1; b < n; ++b) {
                        if (used[b]) continue;
                        for (int c = b + 1; c < n; ++c) {
                            if (used[c]) continue;
                            const int m[4] = {lines[f], lines[a], lines[b], lines[c]};
                            const int cost = best_split(m, q);
                            if (cost < best) {
                                best = cost;
                                pick[1] = a; pick[2] = b; pick[3] = c;
                                for (int i = 0; i < 4; ++i) bq[i] = q[i];
                            }
                        }
                    }
                }
            } else {  // the last, partial quad: everything that is left
                int m[4] = {-1, -1, -1, -1}, j = 0;
                for (int i = 0; i < n; ++i)
                    if (!used[i]) { m[j] = lines[i]; pick[j] = i; ++j; }
                best_split(m, bq);
            }
            for (int i = 0; i < 4; ++i) {
                if (pick[i] >= 0) { used[pick[i]] = true; --left; }
                slots[4 * nq + i] = bq[i];
            }
            ++nq;
        }
        // refinement: swap two slots (of different quads, or re-pair inside one) while the summed cost drops
        auto cost_of = [&](int quad) { return quad_cost(slots + 4 * quad); };
        const int first_free_instr = (n / 8);  // instructions before this one are full: keep unused slots out of them
        for (bool improved = true; improved;) {
            improved = false;
            for (int s0 = 0; s0 < 8 * ni; ++s0)
                for (int s1 = s0 + 1; s1 < 8 * ni; ++s1) {
                    if (slots[s0] < 0 && slots[s1] < 0) continue;
                    if ((slots[s0] < 0 || slots[s1] < 0) && (s0 / 8 < first_free_instr || s1 / 8 < first_free_instr)) continue;
                    const int q0 = s0 / 4, q1 = s1 / 4;
                    const int before = cost_of(q0) + (q1 != q0 ? cost_of(q1) : 0);
                    if (before == 0) continue;
                    const int t = slots[s0]; slots[s0] = slots[s1]; slots[s1] = t;
                    const int after = cost_of(q0) + (q1 != q0 ? cost_of(q1) : 0);
                    if (after < before) improved = true;
                    else { slots[s1] = slots[s0]; slots[s0] = t; }
                }
        }
        for (int i = 0; i < ni; ++i, ++idx)
            for (int lane = 0; lane < 64; ++lane) {
                const int line = slots[8 * i + (lane >> 3)];
                uint32_t d = SCHED_INVALID_P << 16;  // unused slots: only in the round's last instruction
                if (line >= 0) {
                    int lds, P;
                    piece(line, lane & 7, &lds, &P);
                    d = (uint32_t)lds | ((uint32_t)P << 16);
                }
                out[idx * 64 + lane] = d;
            }
    }
}

// Summed b128_read_conflict_cost of a column's table (tests, tools): extra LDS cycles per wave.
template <class W>
inline int flush_table_conflict_cost(const uint32_t *tab) {
    int cost = 0;
    for (int i = 0; i < sched_first<W>(10); ++i) {
        int addr[64];
        for (int lane = 0; lane < 64; ++lane) {
            const uint32_t d = tab[i * 64 + lane];
            addr[lane] = (d >> 16) == SCHED_INVALID_P ? -1 : (int)(d & 0xffffu);
        }
        cost += b128_read_conflict_cost(addr);
    }
    return cost;
}

// MixColumns matrix rows as the reference writes them (src/aes128.rs:228-233).
constexpr int MIX[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};

// Pure-host: which dense rows of the encrypt slab are assigned, per column.
// Built from the same description the kernels use; the tests compare it with
// the mask the oracle derives by running the reference's call order.
inline void encrypt_assigned_mask(int col, uint8_t mask[AES_ROWS]) {
    for (int r = 0; r < AES_ROWS; ++r) mask[r] = 0;
    auto set = [&](int r, bool x, bool y, bool z) {
        if ((col == 0 && x) || (col == 1 && y) || (col == 2 && z)) mask[r] = 1;
    };
    for (int i = 0; i < 16; ++i) set(i, true, false, false);
    for (int i = 0; i < 16; ++i) set(16 + i, true, true, true);
    for (int R = 1; R <= 9; ++R) {
        const int B = 32 + 144 * (R - 1);
        for (int i = 0; i < 16; ++i) set(B + i, true, true, false);
        for (int k = 0; k < 16; ++k) {
            const int m = k & 3;
            for (int t = 0; t < 4; ++t) set(B + 16 + 7 * k + t, true, MIX[m][t] != 1, false);
            for (int t = 4; t < 7; ++t) set(B + 16 + 7 * k + t, true, true, true);
        }
        for (int i = 0; i < 16; ++i) set(B + 128 + i, true, true, true);
    }
    for (int i = 0; i < 16; ++i) set(1328 + i, true, true, false);
    for (int i = 0; i < 16; ++i) set(1344 + i, true, true, true);
}

// Pure-host: dense rows of the encrypt slab that the VALUES layout keeps (col 1: y of S-box / mul rows,
// col 2: z of xor rows; col 0: none), in row order.
inline void encrypt_values_mask(int col, uint8_t mask[AES_ROWS]) {
    for (int r = 0; r < AES_ROWS; ++r) mask[r] = 0;
    if (col == 0) return;
    if (col == 2) { encrypt_assigned_mask(2, mask); return; }
    // col 1: rows whose lookup is Sbox / GfMul2 / GfMul3
    for (int R = 1; R <= 9; ++R) {
        const int B = 32 + 144 * (R - 1);
        for (int i = 0; i < 16; ++i) mask[B + i] = 1;
        for (int k = 0; k < 16; ++k)
            for (int t = 0; t < 4; ++t) mask[B + 16 + 7 * k + t] = MIX[k & 3][t] != 1;
    }
    for (int i = 0; i < 16; ++i) mask[1328 + i] = 1;
}

inline void key_assigned_mask(int col, uint8_t mask[KEY_ROWS]) {
    for (int r = 0; r < KEY_ROWS; ++r) mask[r] = 0;
    for (int rho = 0; rho < 10; ++rho) {
        const int B = 40 * rho;
        for (int r = 0; r < 40; ++r) {
            const bool x = true, y = r < 24, z = r >= 4 && r < 24;
            if ((col == 0 && x) || (col == 1 && y) || (col == 2 && z)) mask[B + r] = 1;
        }
    }
}

// Which kernel writes the Fr form of aesw_assemble_advice_* for a K / column count under option "assemble_geometry": 0 = the
// striding kernel (any shape), 1 = one-shot workgroups on a (chunk, segment, column) grid -- needs 1 + ceil(2^K / 1360) <= 65535
// segments in grid.y --, 2 = one-shot workgroups on aligned chunks (geometries 2 / 3 / 4: 8 <= K <= 30).  A geometry that cannot
// cover the shape falls back to the striding kernel: an option never turns a valid call into an error.
inline int assemble_kernel_choice(bool as_fr, int geometry, uint32_t k, uint32_t col_count) {
    if (!as_fr || col_count == 0) return 0;
    if (geometry == 1) {
        const uint64_t rows = (uint64_t)1 << k;
        const uint64_t segs = 1 + (rows + AES_ROWS - 1) / AES_ROWS;
        return segs <= 65535 && col_count <= 65535 ? 1 : 0;
    }
    if (geometry >= 2) return k >= 8 && k <= 30 && col_count <= 65535 ? 2 : 0;
    return 0;
}

// Closed forms of the dense-row -> packed-index maps (the prefix counts of encrypt_assigned_mask / key_assigned_mask
// above; -1 = the row is never assigned in that column).  The assemble kernels use these instead of a table so that a
// cell is a chain of two loads (slab byte -> Fr LUT), not three; tests/test_lane_model.py checks them row by row against
// the masks.
AESW_HD int packed_index_enc(int c, int r) {
    if (c == 0) return r;
    if (r < 16) return -1;
    if (r < 32) return r - 16;
    if (r >= 1328) {  // round 10: S-box rows (x, y), then the last AddRoundKey (x, y, z)
        if (r < 1344) return c == 1 ? 1024 + (r - 1328) : -1;
        return (c == 1 ? 1040 : 592) + (r - 1344);
    }
    const int R1 = (r - 32) / 144, q = (r - 32) - 144 * R1;  // rounds 1..9: 112 y and 64 z per round
    const int base = 16 + (c == 1 ? 112 : 64) * R1;
    if (q < 16) return c == 1 ? base + q : -1;                       // SubBytes
    if (q >= 128) return base + (c == 1 ? 96 : 48) + (q - 128);      // AddRoundKey
    const int k = (q - 16) / 7, t = (q - 16) - 7 * k;                 // MixColumns record k, row t of 7
    if (t >= 4) return c == 1 ? base + 18 + 5 * k + (t - 4) : base + 3 * k + (t - 4);
    if (c == 2) return -1;
    const int m = k & 3;  // the two products of MIX row m sit at t = m, m + 1 (m = 3: t = 0, 3)
    const int first = m == 3 ? 0 : m, second = m == 3 ? 3 : m + 1;
    return t == first ? base + 16 + 5 * k : t == second ? base + 17 + 5 * k : -1;
}

AESW_HD int packed_index_key(int c, int r) {
    if (c == 0) return r;
    const int rho = r / 40, j = r - 40 * rho;
    if (c == 1) return j < 24 ? 24 * rho + j : -1;
    return (j >= 4 && j < 24) ? 20 * rho + (j - 4) : -1;
}

// Fixed data for keygen: which chip's selector is enabled on
// each slab row, as the Tag of its lookup (src/table.rs:10-16): 0 none (plain
// copy / assign regions), 1 U8 range, 2 Xor, 3 Sbox, 4 GfMul2, 5 GfMul3.
inline void encrypt_selector_tags(uint8_t tag[AES_ROWS]) {
    for (int r = 0; r < AES_ROWS; ++r) tag[r] = 0;
    for (int i = 0; i < 16; ++i) tag[16 + i] = 2;
    for (int R = 1; R <= 9; ++R) {
        const int B = 32 + 144 * (R - 1);
        for (int i = 0; i < 16; ++i) tag[B + i] = 3;
        for (int k = 0; k < 16; ++k) {
            const int m = k & 3;
            for (int t = 0; t < 4; ++t) tag[B + 16 + 7 * k + t] = MIX[m][t] == 1 ? 0 : (MIX[m][t] == 2 ? 4 : 5);
            for (int t = 4; t < 7; ++t) tag[B + 16 + 7 * k + t] = 2;
        }
        for (int i = 0; i < 16; ++i) tag[B + 128 + i] = 2;
    }
    for (int i = 0; i < 16; ++i) tag[1328 + i] = 3;
    for (int i = 0; i < 16; ++i) tag[1344 + i] = 2;
}

// Key slab rows, plus words_column: rcon[r] = round constant where q_eq_rcon is
// enabled (row 20 + 8*(rho-1), src/key_schedule.rs:161-175), 0 elsewhere.
inline void key_selector_tags(uint8_t tag[KEY_ROWS], uint8_t q_eq_rcon[WORDS_ROWS], uint8_t rcon_fixed[WORDS_ROWS]) {
    for (int r = 0; r < WORDS_ROWS; ++r) q_eq_rcon[r] = rcon_fixed[r] = 0;
    constexpr uint8_t RC[10] = {1, 2, 4, 8, 16, 32, 64, 128, 27, 54};
    for (int rho = 0; rho < 10; ++rho) {
        const int B = 40 * rho;
        for (int r = 0; r < 4; ++r) tag[B + r] = 3;
        for (int r = 4; r < 24; ++r) tag[B + r] = 2;
        for (int r = 24; r < 40; ++r) tag[B + r] = 1;
        q_eq_rcon[20 + 8 * rho] = 1;
        rcon_fixed[20 + 8 * rho] = RC[rho];
    }
}

// ---- equality constraints (the permutation argument), input independent --------------------------------
// Every copy_advice() of one encrypt() call / of schedule_keys(), in the reference's call order.  Cells live in one of
// three spaces: 0 = the block's slab (columns x/y/z of its column set, block-relative row), 1 = the key slab (columns
// x/y/z of set 0, rows 0..399), 2 = words_column (rows 0..95).
struct CopyEdge {
    uint8_t dst_space, dst_col;
    uint16_t dst_row;
    uint8_t src_space, src_col;
    uint16_t src_row;
};
struct CellRef { uint8_t space, col; uint16_t row; };
constexpr int BLOCK_COPIES = 1952;  // 160 sbox + 576 tmp + 608 xor rows x 2
constexpr int KEY_COPIES = 640;     // per round: 4 shift + 4 sbox + 20 xor rows x 2 + 16 range

// round-key byte idx of round `round` as schedule_keys() returns it: the key bytes in words_column, later rounds the z
// cells of the word xor rows (src/key_schedule.rs:197-216)
inline CellRef round_key_cell(int round, int idx) {
    if (round == 0) return CellRef{2, 0, (uint16_t)idx};
    return CellRef{1, 2, (uint16_t)(40 * (round - 1) + 8 + idx)};
}

inline int block_copy_graph(CopyEdge *e) {
    int n = 0;
    auto copy = [&](CellRef src, uint8_t col, int row) {
        e[n++] = CopyEdge{0, col, (uint16_t)row, src.space, src.col, src.row};
        return CellRef{0, col, (uint16_t)row};
    };
    CellRef s[16];
    for (int i = 0; i < 16; ++i) {  // src/aes128.rs:194-198
        copy(CellRef{0, 0, (uint16_t)i}, 0, 16 + i);
        copy(round_key_cell(0, i), 1, 16 + i);
        s[i] = CellRef{0, 2, (uint16_t)(16 + i)};
    }
    for (int R = 1; R <= 10; ++R) {
        const int B = R <= 9 ? 32 + 144 * (R - 1) : 1328;
        CellRef sub[16], mixed[16];
        for (int i = 0; i < 16; ++i) {  // :203-209
            copy(s[i], 0, B + i);
            sub[i] = CellRef{0, 1, (uint16_t)(B + i)};
        }
        if (R <= 9) {
            for (int w = 0; w < 4; ++w)
                for (int m = 0; m < 4; ++m) {  // lcon(), :268-301
                    const int base = B + 16 + 7 * (4 * w + m);
                    CellRef tmp[4];
                    for (int t = 0; t < 4; ++t) {
                        const CellRef c = copy(sub[4 * ((w + t) % 4) + t], 0, base + t);
                        tmp[t] = MIX[m][t] == 1 ? c : CellRef{0, 1, (uint16_t)(base + t)};
                    }
                    copy(tmp[0], 0, base + 4); copy(tmp[1], 1, base + 4);
                    copy(tmp[2], 0, base + 5); copy(tmp[3], 1, base + 5);
                    copy(CellRef{0, 2, (uint16_t)(base + 4)}, 0, base + 6);
                    copy(CellRef{0, 2, (uint16_t)(base + 5)}, 1, base + 6);
                    mixed[4 * w + m] = CellRef{0, 2, (uint16_t)(base + 6)};
                }
        } else {
            for (int w = 0; w < 4; ++w)
                for (int j = 0; j < 4; ++j) mixed[4 * w + j] = sub[4 * ((w + j) % 4) + j];  // :236-237
        }
        const int A = R <= 9 ? B + 128 : 1344;
        for (int i = 0; i < 16; ++i) {  // :250-261
            copy(mixed[i], 0, A + i);
            copy(round_key_cell(R, i), 1, A + i);
            s[i] = CellRef{0, 2, (uint16_t)(A + i)};
        }
    }
    return n;
}

inline int key_copy_graph(CopyEdge *e) {
    int n = 0;
    auto copy = [&](CellRef src, uint8_t space, uint8_t col, int row) {
        e[n++] = CopyEdge{space, col, (uint16_t)row, src.space, src.col, src.row};
        return CellRef{space, col, (uint16_t)row};
    };
    for (int rho = 1; rho <= 10; ++rho) {  // assign_round, src/key_schedule.rs:122-224
        const int B = 40 * (rho - 1), W = 16 + 8 * (rho - 1);
        static const int rot[4] = {13, 14, 15, 12};
        CellRef shifted[4], rconned[4], next[4];
        for (int i = 0; i < 4; ++i) shifted[i] = copy(round_key_cell(rho - 1, rot[i]), 2, 0, W + i);   // :141-154
        for (int i = 0; i < 4; ++i) copy(shifted[i], 1, 0, B + i);                                      // sbox rows
        for (int i = 0; i < 4; ++i) {                                                                   // :189-194
            copy(CellRef{1, 1, (uint16_t)(B + i)}, 1, 0, B + 4 + i);
            copy(CellRef{2, 0, (uint16_t)(W + 4 + i)}, 1, 1, B + 4 + i);
            rconned[i] = CellRef{1, 2, (uint16_t)(B + 4 + i)};
        }
        for (int i = 0; i < 4; ++i) {                                                                   // :197-204
            copy(round_key_cell(rho - 1, i), 1, 0, B + 8 + i);
            copy(rconned[i], 1, 1, B + 8 + i);
            next[i] = CellRef{1, 2, (uint16_t)(B + 8 + i)};
        }
        for (int wd = 1; wd < 4; ++wd)                                                                  // :207-216
            for (int i = 0; i < 4; ++i) {
                copy(round_key_cell(rho - 1, 4 * wd + i), 1, 0, B + 8 + 4 * wd + i);
                copy(next[i], 1, 1, B + 8 + 4 * wd + i);
                next[i] = CellRef{1, 2, (uint16_t)(B + 8 + 4 * wd + i)};
            }
        for (int i = 0; i < 16; ++i) copy(CellRef{1, 2, (uint16_t)(B + 8 + i)}, 1, 0, B + 24 + i);      // :218-221
    }
    return n;
}

}  // namespace aesw
