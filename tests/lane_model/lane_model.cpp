// lane_model.cpp -- CPU execution of the per-lane code in
// halo2-aes_amd/csrc/aesw_lane.h (the same source the HIP kernels compile),
// with the cross-lane steps (DPP quad permutes, LDS staging) replaced by plain
// arrays.  TEST INFRASTRUCTURE: lets `-m "not gpu"` tests compare the lane
// program (perm selectors, slab offsets) with the oracle before any GPU run.
// It is not reachable from the product library.
#include <cstdint>
#include <cstring>

#include "../../halo2-aes_amd/csrc/aesw_lane.h"

using namespace aesw;

namespace {

template <int L>
struct HostSink {
    uint8_t *col[3];  // block base + segment start, per column
    int w;
    static void put(uint8_t *p, uint32_t v) { std::memcpy(p, &v, 4); }
    template <int C> void plain(int off, uint32_t v) { put(col[C] + off + 4 * w, v); }
    template <int C> void mix(int off, int k, uint32_t v) {
        constexpr int mw = C == 0 ? Geo<L>::X_MIXW : C == 1 ? Geo<L>::Y_MIXW : Geo<L>::Z_MIXW;
        put(col[C] + off + mw * w + 4 * k, v);
    }
};

struct HostKSink {
    uint8_t *x, *y, *z, *wd;
    static void put(uint8_t *p, uint32_t v) { if (p) std::memcpy(p, &v, 4); }
    void kx(int off, uint32_t v) { put(x ? x + off : nullptr, v); }
    void ky(int off, uint32_t v) { put(y ? y + off : nullptr, v); }
    void kz(int off, uint32_t v) { put(z ? z + off : nullptr, v); }
    void words(int off, uint32_t v) { put(wd ? wd + off : nullptr, v); }
};

uint32_t ld32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

template <int L, bool XT>
void run(const uint8_t *tab, const uint8_t *pt, const uint8_t *keys, int per_block_keys, int key_only, uint64_t n,
         uint8_t *x, uint8_t *y, uint8_t *z, uint8_t *ct, uint8_t *wd, uint8_t *kx, uint8_t *ky, uint8_t *kz,
         uint8_t *rk_out) {
    using G = Geo<L>;
    Tables<XT> T{tab};
    uint32_t rk[11][4];
    bool have = false;
    for (uint64_t b = 0; b < n; ++b) {
        if (per_block_keys || key_only || !have) {
            const uint8_t *key = keys + ((per_block_keys || key_only) ? 16 * b : 0);
            const bool emit = per_block_keys || key_only;
            for (int w = 0; w < 4; ++w) rk[0][w] = ld32(key + 4 * w);
            HostKSink ks{emit && kx ? kx + (size_t)G::KXS * b : nullptr, emit && ky ? ky + (size_t)G::KYS * b : nullptr,
                         emit && kz ? kz + (size_t)G::KZS * b : nullptr, emit && wd ? wd + (size_t)WORDS_ROWS * b : nullptr};
            for (int w = 0; w < 4; ++w) ks.words(4 * w, rk[0][w]);
            for (int rho = 1; rho <= 10; ++rho)
                for (int w = 0; w < 4; ++w)
                    rk[rho][w] = emit_key_round<L>(ks, rho, w, rk[rho - 1][0], rk[rho - 1][1], rk[rho - 1][2],
                                                   rk[rho - 1][3], rcon(rho - 1), T);
            if (emit && rk_out) std::memcpy(rk_out + (size_t)RK_BYTES * b, rk, RK_BYTES);
            have = true;
        }
        if (key_only) continue;
        uint8_t *bx = x + (size_t)G::XS * b, *by = y + (size_t)G::YS * b, *bz = z + (size_t)G::ZS * b;
        uint32_t st[4], sub[4], sh[4];
        auto sink = [&](int g, int w) {
            return HostSink<L>{{bx + SegX<L>::start(g), by + SegY<L>::start(g), bz + SegZ<L>::start(g)}, w};
        };
        for (int w = 0; w < 4; ++w) {
            auto s = sink(0, w);
            st[w] = emit_head<L>(s, ld32(pt + 16 * b + 4 * w), rk[0][w]);
        }
        for (int R = 1; R <= 9; ++R) {
            const int g = R - 1;  // segment of this round (round 9 lives in segment 8)
            const int rx = g == 0 ? G::X_HEAD : 0, ry = g == 0 ? G::Y_HEAD : 0, rz = g == 0 ? G::Z_HEAD : 0;
            for (int w = 0; w < 4; ++w) { auto s = sink(g, w); sub[w] = emit_sbox<L>(s, rx, ry, rz, st[w], T); }
            for (int w = 0; w < 4; ++w) sh[w] = shift_rows(sub[w], sub[(w + 1) & 3], sub[(w + 2) & 3], sub[(w + 3) & 3]);
            for (int w = 0; w < 4; ++w) { auto s = sink(g, w); st[w] = emit_mix_ark<L>(s, rx, ry, rz, sh[w], rk[R][w], T); }
        }
        {
            const int g = 8, rx = G::X_ROUND, ry = G::Y_ROUND, rz = G::Z_ROUND;
            for (int w = 0; w < 4; ++w) { auto s = sink(g, w); sub[w] = emit_sbox<L>(s, rx, ry, rz, st[w], T); }
            for (int w = 0; w < 4; ++w) sh[w] = shift_rows(sub[w], sub[(w + 1) & 3], sub[(w + 2) & 3], sub[(w + 3) & 3]);
            for (int w = 0; w < 4; ++w) {
                auto s = sink(g, w);
                const uint32_t c = emit_final_ark<L>(s, rx, ry, rz, sh[w], rk[10][w]);
                if (ct) std::memcpy(ct + 16 * b + 4 * w, &c, 4);
            }
        }
    }
}

}  // namespace

extern "C" int lane_model_run(const uint8_t *tab768, const uint8_t *pt, const uint8_t *keys, int per_block_keys,
                              int key_only, uint64_t n, int layout, int xt_arith, uint8_t *x, uint8_t *y, uint8_t *z,
                              uint8_t *ct, uint8_t *wd, uint8_t *kx, uint8_t *ky, uint8_t *kz, uint8_t *rk) {
#define GO(L, XT) run<L, XT>(tab768, pt, keys, per_block_keys, key_only, n, x, y, z, ct, wd, kx, ky, kz, rk)
    if (layout == DENSE) { if (xt_arith) GO(DENSE, true); else GO(DENSE, false); }
    else if (layout == PACKED) { if (xt_arith) GO(PACKED, true); else GO(PACKED, false); }
    else return 1;
    return 0;
#undef GO
}

extern "C" void lane_model_masks(int col, uint8_t *enc_mask, uint8_t *key_mask) {
    encrypt_assigned_mask(col, enc_mask);
    key_assigned_mask(col, key_mask);
}
