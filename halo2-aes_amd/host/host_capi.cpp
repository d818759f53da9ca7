// host_capi.cpp -- MockProver::verify and the C entry points of include/aesw_host.h.
#include <cstdio>
#include <cstring>
#include <set>

#include "../../include/aesw_host.h"
#include "aes_gadget.hpp"

using namespace aesw::host;

namespace aesw {
namespace host {

std::string MockProver::verify() const {
    const Assembly &a = assembly;
    char buf[256];
    for (const Lookup &l : cs.lookups) {
        // project the table onto this lookup's table columns
        std::set<uint64_t> rows;
        uint64_t trows = 0;
        for (const TableColumn &t : l.tables) trows = std::max<uint64_t>(trows, a.table.at(t.index).size());
        for (uint64_t r = 0; r < trows; ++r) {
            uint64_t key = 0;
            for (const TableColumn &t : l.tables) {
                const auto &col = a.table.at(t.index);
                key = key << 8 | (r < col.size() ? col[r] : 0);
            }
            rows.insert(key);
        }
        if (!rows.count(0)) return "lookup '" + l.name + "': the all-zero row (selector off) is not in the table";
        const auto &q = a.selectors.at(l.q.index);
        for (uint64_t r = 0; r < a.n_rows; ++r) {
            if (!q[r]) continue;
            uint64_t key = l.tag;
            for (const Column &c : l.inputs) {
                if (!a.advice_assigned.at(c.index)[r]) {
                    std::snprintf(buf, sizeof buf, "lookup '%s' row %llu: advice column %u not assigned", l.name.c_str(), (unsigned long long)r, c.index);
                    return buf;
                }
                key = key << 8 | a.advice.at(c.index)[r];
            }
            if (!rows.count(key)) {
                std::snprintf(buf, sizeof buf, "lookup '%s' row %llu: inputs not in table", l.name.c_str(), (unsigned long long)r);
                return buf;
            }
        }
    }
    for (const Gate &g : cs.gates) {
        const auto &q = a.selectors.at(g.q.index);
        for (uint64_t r = 0; r < a.n_rows; ++r) {
            if (!q[r]) continue;
            if (!a.advice_assigned.at(g.advice.index)[r] || !a.fixed_assigned.at(g.fixed.index)[r] ||
                a.advice.at(g.advice.index)[r] != a.fixed.at(g.fixed.index)[r]) {
                std::snprintf(buf, sizeof buf, "gate '%s' row %llu not satisfied", g.name.c_str(), (unsigned long long)r);
                return buf;
            }
        }
    }
    for (size_t i = 0; i < a.copies.size(); ++i) {
        const Cell &x = a.copies[i].first, &y = a.copies[i].second;
        if (!a.advice_assigned.at(x.column.index)[x.row] || !a.advice_assigned.at(y.column.index)[y.row] ||
            a.advice.at(x.column.index)[x.row] != a.advice.at(y.column.index)[y.row]) {
            std::snprintf(buf, sizeof buf, "equality constraint %zu not satisfied", i);
            return buf;
        }
    }
    return "";
}

// benches/aes128.rs:30-61 Aes128BenchCircuit / src/aes128.rs:376-407 TestAesCircuit,
// generalised to one plaintext per encrypt() call.
struct AesCircuit {
    aesw_ctx *ctx;
    uint32_t K, N;
    uint8_t key[16];
    std::vector<uint8_t> plaintexts;
    bool skip_schedule_key = false;
    bool bulk_assign = false;
    int layout = AESW_LAYOUT_PACKED;  // PACKED: assigned cells only (default); VALUES: only what the chips' value closures read; DENSE: option
    bool streaming = false;    // ... chunk by chunk, each chunk assigned while the next is produced and copied (configs[4])
    std::shared_ptr<const AesWitness> witness;  // computed lazily, once
    std::vector<std::vector<AssignedCell>> outputs;

    FixedAes128Config configure(ConstraintSystem &meta) { return FixedAes128Config::configure(meta, K, N); }
    void synthesize(FixedAes128Config config, Layouter &layouter) {
        const uint64_t n = plaintexts.size() / 16;
        if (!witness)
            witness = streaming ? AesWitness::prepare_stream(ctx, key, plaintexts.data(), n)
                                : AesWitness::generate(ctx, key, plaintexts.data(), n, layout);
        config.attach_witness(witness);
        config.bulk_assign = bulk_assign;
        load_enc_full_table(layouter, config.tables, ctx);
        if (!skip_schedule_key) config.schedule_key(layouter, key);
        if (streaming)
            witness->stream(ctx, [&](uint64_t first, uint64_t count) {
                for (uint64_t b = first; b < first + count; ++b) outputs.push_back(config.encrypt(layouter, plaintexts.data() + 16 * b));
            });
        else
            for (uint64_t b = 0; b < n; ++b) outputs.push_back(config.encrypt(layouter, plaintexts.data() + 16 * b));
    }
};

// src/key_schedule.rs:245-320 TestCircuit
struct KeyCircuit {
    aesw_ctx *ctx;
    uint8_t key[16];
    struct Config {
        Aes128KeyScheduleConfig ks;
        TableColumn tables[4];
    };
    Config configure(ConstraintSystem &meta) {
        Config c;
        Column advices[3] = {meta.advice_column(), meta.advice_column(), meta.advice_column()};
        for (auto &t : c.tables) t = meta.lookup_table_column();
        const Selector q_u8_range_check = meta.complex_selector(), q_u8_xor = meta.complex_selector(), q_sbox = meta.complex_selector();
        const auto range = U8RangeCheckChip::configure(meta, advices[0], q_u8_range_check, c.tables[0], c.tables[1]);
        const auto xr = U8XorChip::configure(meta, advices[0], advices[1], advices[2], q_u8_xor, c.tables[0], c.tables[1], c.tables[2], c.tables[3]);
        const auto sb = SboxChip::configure(meta, advices[0], advices[1], q_sbox, c.tables[0], c.tables[1], c.tables[2]);
        c.ks = Aes128KeyScheduleConfig::configure(meta, advices, xr, sb, range);
        return c;
    }
    void synthesize(Config config, Layouter &layouter) {
        config.ks.attach_witness(AesWitness::generate(ctx, key, nullptr, 0));
        load_enc_full_table(layouter, config.tables, ctx);
        config.ks.schedule_keys(layouter, key);
    }
};

}  // namespace host
}  // namespace aesw

struct aesw_host_circuit {
    MockProver prover;
    std::vector<std::vector<AssignedCell>> outputs;
};

namespace {
thread_local std::string g_last_error;

template <class F>
int guarded(F &&f) {
    g_last_error.clear();
    try {
        f();
        return AESW_OK;
    } catch (const Panic &p) {
        g_last_error = p.what();
        return p.kind == Panic::Capacity ? AESW_ERR_CAPACITY : p.kind == Panic::NoKey ? AESW_ERR_NO_KEY : AESW_ERR_INVALID_ARG;
    } catch (const Error &e) {
        g_last_error = e.what();
        return e.kind == Error::Mismatch ? AESW_ERR_MISMATCH : e.kind == Error::NotEnoughRowsAvailable ? AESW_ERR_CAPACITY : AESW_ERR_HIP;
    } catch (const std::bad_alloc &) {
        g_last_error = "out of memory";
        return AESW_ERR_NOMEM;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return AESW_ERR_INVALID_ARG;
    }
}
}  // namespace

extern "C" {

const char *aesw_host_last_error(void) { return g_last_error.c_str(); }

int aesw_host_aes_circuit_run(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, const uint8_t key[16], const uint8_t *pts, uint64_t n,
                              int with_witnesses, int skip_schedule_key, int assign_mode, aesw_host_circuit **out) {
    if (!ctx || !key || (n && !pts) || !out || k < 11 || k > 26 || n_sets == 0 || n_sets > 64 || assign_mode < 0 || assign_mode > 4)
        return AESW_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded([&] {
        AesCircuit circuit{ctx, k, n_sets, {0}, std::vector<uint8_t>(pts, pts + 16 * n), skip_schedule_key != 0, assign_mode == 1,
                           assign_mode == 2 || assign_mode == 3 ? AESW_LAYOUT_VALUES : assign_mode == 4 ? AESW_LAYOUT_DENSE : AESW_LAYOUT_PACKED,
                           assign_mode == 3, nullptr, {}};
        std::memcpy(circuit.key, key, 16);
        auto *c = new aesw_host_circuit{MockProver::run(k, circuit, with_witnesses != 0), {}};
        c->outputs = std::move(circuit.outputs);
        *out = c;
    });
}

// The whole circuit without running a single region: what a host with bulk column access does.  Advice columns are the
// device witness placed by aesw_block_placement, selectors / fixed column / equality constraints / table are the
// library's input-independent keygen data.  MockProver::verify() then checks every lookup, the rcon gate and all copies.
int aesw_host_aes_circuit_columns(aesw_ctx *ctx, uint32_t k, uint32_t n_sets, const uint8_t key[16], const uint8_t *pts, uint64_t n,
                                  aesw_host_circuit **out) {
    if (!ctx || !key || (n && !pts) || !out || k < 11 || k > 26 || n_sets == 0 || n_sets > 64) return AESW_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded([&] {
        if (n > aesw_block_capacity(k, n_sets)) throw Panic(Panic::Capacity, "AES calls too many. doesn't fit in the rows");
        std::unique_ptr<aesw_host_circuit> c(new aesw_host_circuit{});
        MockProver &p = c->prover;
        (void)FixedAes128Config::configure(p.cs, k, n_sets);  // columns, lookups, the rcon gate: as configure() registers them
        Assembly &a = p.assembly;
        a.k = k;
        a.n_rows = (uint64_t)1 << k;
        a.advice.assign(p.cs.n_advice, std::vector<uint8_t>(a.n_rows, 0));
        a.advice_assigned.assign(p.cs.n_advice, std::vector<uint8_t>(a.n_rows, 0));
        a.fixed.assign(p.cs.n_fixed, std::vector<uint8_t>(a.n_rows, 0));
        a.fixed_assigned.assign(p.cs.n_fixed, std::vector<uint8_t>(a.n_rows, 0));
        a.selectors.assign(p.cs.n_selectors, std::vector<uint8_t>(a.n_rows, 0));
        a.table.assign(p.cs.n_table, std::vector<uint8_t>(AESW_TABLE_ROWS, 0));
        auto chk = [](int rc, const char *what) { if (rc != AESW_OK) throw Error(Error::Synthesis, std::string(what) + ": " + aesw_strerror(rc)); };
        // device witness, packed: only assigned cells cross PCIe; aesw_packed_index places them
        const int L = AESW_LAYOUT_PACKED;
        const size_t cs[3] = {aesw_column_stride(L, 0), aesw_column_stride(L, 1), aesw_column_stride(L, 2)};
        std::vector<uint8_t> x(n * cs[0]), y(n * cs[1]), z(n * cs[2]);
        std::vector<uint8_t> kw(AESW_WORDS_ROWS), kx(aesw_key_column_stride(L, 0)), ky(aesw_key_column_stride(L, 1)), kz(aesw_key_column_stride(L, 2));
        aesw_key_slab ks{kw.data(), kx.data(), ky.data(), kz.data()};
        chk(aesw_schedule_key(ctx, key, L, &ks), "aesw_schedule_key");
        if (n) chk(aesw_encrypt_witness(ctx, pts, nullptr, 0, n, L, x.data(), y.data(), z.data(), nullptr, nullptr), "aesw_encrypt_witness");
        chk(aesw_lookup_table(ctx, a.table[0].data(), a.table[1].data(), a.table[2].data(), a.table[3].data()), "aesw_lookup_table");
        // advice columns
        const uint32_t words_col = 3 * n_sets;
        int32_t ei[3][AESW_AES_ROWS], ki[3][AESW_KEY_ROWS];  // slab row -> packed index; -1 = a cell the reference never assigns
        for (int j = 0; j < 3; ++j) {
            chk(aesw_packed_index(j, ei[j]), "aesw_packed_index");
            chk(aesw_key_packed_index(j, ki[j]), "aesw_key_packed_index");
        }
        const uint8_t *kcols[3] = {kx.data(), ky.data(), kz.data()};
        for (int j = 0; j < 3; ++j)
            for (uint32_t r = 0; r < AESW_KEY_ROWS; ++r)
                if (ki[j][r] >= 0) { a.advice[j][r] = kcols[j][ki[j][r]]; a.advice_assigned[j][r] = 1; }
        for (uint32_t r = 0; r < AESW_WORDS_ROWS; ++r) { a.advice[words_col][r] = kw[r]; a.advice_assigned[words_col][r] = 1; }
        std::vector<uint32_t> bset(n);
        std::vector<uint64_t> brow(n);
        const uint8_t *cols[3] = {x.data(), y.data(), z.data()};
        for (uint64_t b = 0; b < n; ++b) {
            chk(aesw_block_placement(k, n_sets, b, &bset[b], &brow[b]), "aesw_block_placement");
            for (int j = 0; j < 3; ++j) {
                uint8_t *dst = a.advice[3 * bset[b] + j].data() + brow[b], *asg = a.advice_assigned[3 * bset[b] + j].data() + brow[b];
                const uint8_t *src = cols[j] + b * cs[j];
                for (uint32_t r = 0; r < AESW_AES_ROWS; ++r)
                    if (ei[j][r] >= 0) { dst[r] = src[ei[j][r]]; asg[r] = 1; }
            }
        }
        // selectors + fixed column
        std::vector<uint8_t> sel((size_t)(5 * n_sets + 1) * a.n_rows), fixed(a.n_rows);
        chk(aesw_assemble_selectors(k, n_sets, n, sel.data(), fixed.data()), "aesw_assemble_selectors");
        for (uint32_t s = 0; s < p.cs.n_selectors; ++s) std::memcpy(a.selectors[s].data(), sel.data() + (size_t)s * a.n_rows, a.n_rows);
        a.fixed[0] = fixed;
        a.fixed_assigned[0] = a.selectors[5 * n_sets];  // the round constant is assigned exactly where q_eq_rcon is enabled
        // equality constraints
        std::vector<aesw_copy_edge> be(AESW_BLOCK_COPIES), ke(AESW_KEY_COPIES);
        chk(aesw_block_copy_graph(be.data()), "aesw_block_copy_graph");
        chk(aesw_key_copy_graph(ke.data()), "aesw_key_copy_graph");
        auto cell = [&](uint8_t space, uint8_t col, uint16_t row, uint32_t set, uint64_t row0) {
            if (space == 0) return Cell{Column{Any::Advice, 3 * set + col}, row0 + row};
            if (space == 1) return Cell{Column{Any::Advice, (uint32_t)col}, (uint64_t)row};
            return Cell{Column{Any::Advice, words_col}, (uint64_t)row};
        };
        a.copies.reserve(ke.size() + n * be.size());
        for (const aesw_copy_edge &e : ke) a.copies.emplace_back(cell(e.dst_space, e.dst_col, e.dst_row, 0, 0), cell(e.src_space, e.src_col, e.src_row, 0, 0));
        for (uint64_t b = 0; b < n; ++b)
            for (const aesw_copy_edge &e : be)
                a.copies.emplace_back(cell(e.dst_space, e.dst_col, e.dst_row, bset[b], brow[b]), cell(e.src_space, e.src_col, e.src_row, bset[b], brow[b]));
        // ciphertext cells (what encrypt() returns)
        c->outputs.resize(n);
        for (uint64_t b = 0; b < n; ++b)
            for (uint32_t i = 0; i < 16; ++i) {
                const uint64_t r = brow[b] + AESW_AES_ROWS - 16 + i;
                c->outputs[b].push_back(AssignedCell{Cell{Column{Any::Advice, 3 * bset[b] + 2}, r}, Value::of(a.advice[3 * bset[b] + 2][r])});
            }
        *out = c.release();
    });
}

int aesw_host_key_circuit_run(aesw_ctx *ctx, uint32_t k, const uint8_t key[16], aesw_host_circuit **out) {
    if (!ctx || !key || !out || k < 9 || k > 26) return AESW_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded([&] {
        KeyCircuit circuit{ctx, {0}};
        std::memcpy(circuit.key, key, 16);
        *out = new aesw_host_circuit{MockProver::run(k, circuit, true), {}};
    });
}

void aesw_host_circuit_free(aesw_host_circuit *c) { delete c; }

int aesw_host_circuit_verify(const aesw_host_circuit *c, char *msg, size_t msg_len) {
    if (!c) return AESW_ERR_INVALID_ARG;
    const std::string r = c->prover.verify();
    if (msg && msg_len) std::snprintf(msg, msg_len, "%s", r.c_str());
    return r.empty() ? AESW_OK : AESW_ERR_UNSATISFIED;
}

uint32_t aesw_host_circuit_num_advice(const aesw_host_circuit *c) { return c->prover.cs.n_advice; }
uint32_t aesw_host_circuit_num_selectors(const aesw_host_circuit *c) { return c->prover.cs.n_selectors; }
uint64_t aesw_host_circuit_num_rows(const aesw_host_circuit *c) { return c->prover.assembly.n_rows; }
uint64_t aesw_host_circuit_num_regions(const aesw_host_circuit *c) { return c->prover.assembly.n_regions; }
uint64_t aesw_host_circuit_num_copies(const aesw_host_circuit *c) { return c->prover.assembly.copies.size(); }
uint64_t aesw_host_circuit_closure_calls(const aesw_host_circuit *c) { return c->prover.assembly.closure_calls; }
int aesw_host_circuit_copies(const aesw_host_circuit *c, uint64_t *out) {
    if (!c || !out) return AESW_ERR_INVALID_ARG;
    size_t i = 0;
    for (const auto &p : c->prover.assembly.copies) {
        out[i++] = p.first.column.index; out[i++] = p.first.row;
        out[i++] = p.second.column.index; out[i++] = p.second.row;
    }
    return AESW_OK;
}
const uint8_t *aesw_host_circuit_advice(const aesw_host_circuit *c, uint32_t col) {
    return col < c->prover.assembly.advice.size() ? c->prover.assembly.advice[col].data() : nullptr;
}
const uint8_t *aesw_host_circuit_advice_assigned(const aesw_host_circuit *c, uint32_t col) {
    return col < c->prover.assembly.advice_assigned.size() ? c->prover.assembly.advice_assigned[col].data() : nullptr;
}
const uint8_t *aesw_host_circuit_selector(const aesw_host_circuit *c, uint32_t s) {
    return s < c->prover.assembly.selectors.size() ? c->prover.assembly.selectors[s].data() : nullptr;
}
const uint8_t *aesw_host_circuit_fixed(const aesw_host_circuit *c) {
    return c->prover.assembly.fixed.empty() ? nullptr : c->prover.assembly.fixed[0].data();
}
const uint8_t *aesw_host_circuit_table(const aesw_host_circuit *c, uint32_t col, uint64_t *rows) {
    if (col >= c->prover.assembly.table.size()) return nullptr;
    if (rows) *rows = c->prover.assembly.table[col].size();
    return c->prover.assembly.table[col].data();
}
int aesw_host_circuit_ciphertext(const aesw_host_circuit *c, uint64_t b, uint8_t ct[16]) {
    if (!c || b >= c->outputs.size() || !ct) return AESW_ERR_INVALID_ARG;
    for (int i = 0; i < 16; ++i) {
        const Cell &cell = c->outputs[b][i].cell;
        ct[i] = c->prover.assembly.advice.at(cell.column.index).at(cell.row);
    }
    return AESW_OK;
}
int aesw_host_circuit_poke(aesw_host_circuit *c, uint32_t col, uint64_t row, uint8_t value) {
    if (!c || col >= c->prover.assembly.advice.size() || row >= c->prover.assembly.n_rows) return AESW_ERR_INVALID_ARG;
    c->prover.assembly.advice[col][row] = value;
    return AESW_OK;
}

}  // extern "C"
