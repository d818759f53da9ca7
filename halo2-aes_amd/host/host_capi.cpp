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
    bool values_only = false;  // the device hands over only what the chips' value closures read (AESW_LAYOUT_VALUES)
    bool streaming = false;    // ... chunk by chunk, each chunk assigned while the next is produced and copied (configs[4])
    std::shared_ptr<const AesWitness> witness;  // computed lazily, once
    std::vector<std::vector<AssignedCell>> outputs;

    FixedAes128Config configure(ConstraintSystem &meta) { return FixedAes128Config::configure(meta, K, N); }
    void synthesize(FixedAes128Config config, Layouter &layouter) {
        const uint64_t n = plaintexts.size() / 16;
        if (!witness)
            witness = streaming ? AesWitness::prepare_stream(ctx, key, plaintexts.data(), n)
                                : AesWitness::generate(ctx, key, plaintexts.data(), n, values_only);
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
    if (!ctx || !key || (n && !pts) || !out || k < 11 || k > 26 || n_sets == 0 || n_sets > 64 || assign_mode < 0 || assign_mode > 3)
        return AESW_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded([&] {
        AesCircuit circuit{ctx, k, n_sets, {0}, std::vector<uint8_t>(pts, pts + 16 * n), skip_schedule_key != 0, assign_mode == 1, assign_mode == 2, assign_mode == 3, nullptr, {}};
        std::memcpy(circuit.key, key, 16);
        auto *c = new aesw_host_circuit{MockProver::run(k, circuit, with_witnesses != 0), {}};
        c->outputs = std::move(circuit.outputs);
        *out = c;
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
