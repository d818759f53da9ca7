// halo2_lite.hpp -- the sliver of the halo2 front end the AES gadget touches,
// so that the host side of the boundary can keep the reference's
// configure()/synthesize() surface in C++ (the image has no Rust toolchain).
//
// Mirrors, by name and meaning, what src/aes128.rs:9-13 and
// src/key_schedule.rs:16-21 import from halo2_proofs v0.3.0: ConstraintSystem,
// Column<Advice|Fixed>, Selector, TableColumn, Layouter (SimpleFloorPlanner
// semantics: every assign_region closure runs twice, a shape pass that does
// not evaluate value closures and the real pass), Region, AssignedCell,
// Value, Error, and a MockProver that checks lookups, gates and equality.
// Field elements are uint64_t: every value on this path is a byte.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace aesw {
namespace host {

using Fp = uint64_t;

// plonk::Error
struct Error : std::runtime_error {
    enum Kind { Synthesis, NotEnoughRowsAvailable, Mismatch } kind;
    Error(Kind k, const std::string &what) : std::runtime_error(what), kind(k) {}
};
// a Rust panic!/expect in the reference
struct Panic : std::runtime_error {
    enum Kind { Capacity, NoKey, Other } kind;
    Panic(Kind k, const std::string &what) : std::runtime_error(what), kind(k) {}
};

struct Value {
    bool known = false;
    Fp v = 0;
    static Value of(Fp x) { return Value{true, x}; }
    static Value unknown() { return Value{}; }
};

enum class Any : uint8_t { Advice, Fixed };
struct Column {
    Any type = Any::Advice;
    uint32_t index = 0;
};
struct Selector { uint32_t index = 0; };
struct TableColumn { uint32_t index = 0; };
struct Cell {
    Column column;
    uint64_t row = 0;
};

class Region;
struct AssignedCell {
    Cell cell;
    Value val;
    Value value() const { return val; }
    // AssignedCell::copy_advice: assign the same value at (column, offset) and constrain equal
    AssignedCell copy_advice(Region &region, Column column, uint64_t offset) const;
};

// A lookup argument: inputs (q*tag, q*advice...) against table columns; a gate q*(advice - fixed).
struct Lookup {
    std::string name;
    Selector q;
    Fp tag;
    std::vector<Column> inputs;       // advice columns, in order
    std::vector<TableColumn> tables;  // tag column first
};
struct Gate {
    std::string name;
    Selector q;
    Column advice, fixed;
};

class ConstraintSystem {
public:
    Column advice_column() { return Column{Any::Advice, n_advice++}; }
    Column fixed_column() { return Column{Any::Fixed, n_fixed++}; }
    Selector selector() { return Selector{n_selectors++}; }
    Selector complex_selector() { return Selector{n_selectors++}; }
    TableColumn lookup_table_column() { return TableColumn{n_table++}; }
    void enable_equality(Column) {}
    void enable_constant(Column) {}
    void lookup(Lookup l) { lookups.push_back(std::move(l)); }
    void create_gate(Gate g) { gates.push_back(std::move(g)); }
    uint32_t n_advice = 0, n_fixed = 0, n_selectors = 0, n_table = 0;
    std::vector<Lookup> lookups;
    std::vector<Gate> gates;
};

// Everything synthesize() writes: the advice matrix is what the prover commits to.
struct Assembly {
    uint32_t k = 0;
    uint64_t n_rows = 0;
    std::vector<std::vector<uint8_t>> advice, advice_assigned, fixed, fixed_assigned, selectors, table;
    std::vector<std::pair<Cell, Cell>> copies;
    uint64_t n_regions = 0;
    uint64_t closure_calls = 0;  // value closures evaluated (tests: shape pass must not evaluate any)
};

class Region {
public:
    virtual ~Region() = default;
    virtual AssignedCell assign_advice(Column column, uint64_t offset, const std::function<Value()> &to) = 0;
    virtual void assign_fixed(Column column, uint64_t offset, const std::function<Value()> &to) = 0;
    virtual void enable_selector(Selector s, uint64_t offset) = 0;
    virtual void constrain_equal(Cell a, Cell b) = 0;
};

inline AssignedCell AssignedCell::copy_advice(Region &region, Column column, uint64_t offset) const {
    const Value v = val;
    AssignedCell out = region.assign_advice(column, offset, [v] { return v; });
    region.constrain_equal(out.cell, cell);
    return out;
}

class Table {  // layouter.assign_table
public:
    explicit Table(Assembly &a) : a_(a) {}
    void assign_cell(TableColumn col, uint64_t row, Fp v) {
        auto &c = a_.table.at(col.index);
        if (row >= c.size()) c.resize(row + 1, 0);
        if (v > 255) throw Error(Error::Synthesis, "table value out of byte range");
        c[row] = (uint8_t)v;
    }
private:
    Assembly &a_;
};

// SimpleFloorPlanner / SingleChipLayouter: "position the region starting at the
// earliest row for which none of the columns are in use" [upstream v0.3.0].
class Layouter {
public:
    Layouter(const ConstraintSystem &cs, Assembly &a, bool with_witnesses = true)
        : a_(a), with_witnesses_(with_witnesses), h_adv_(cs.n_advice, 0), h_fix_(cs.n_fixed, 0), h_sel_(cs.n_selectors, 0) {}

    template <class T>
    T assign_region(const std::string &, const std::function<T(Region &)> &assignment) {
        ShapeRegion shape;
        (void)assignment(shape);  // pass 1: shape only, value closures are not evaluated
        uint64_t start = 0;
        for (const auto &c : shape.columns) start = std::max(start, height(c));
        for (const auto &c : shape.columns) height(c) = start + shape.row_count;
        a_.n_regions++;
        if (start + shape.row_count > a_.n_rows) throw Error(Error::NotEnoughRowsAvailable, "not enough rows available");
        AssignRegion real(a_, start, with_witnesses_);
        return assignment(real);  // pass 2
    }
    void assign_table(const std::string &, const std::function<void(Table &)> &assignment) {
        Table t(a_);
        assignment(t);
    }
    uint64_t column_height(Column c) const { return c.type == Any::Advice ? h_adv_.at(c.index) : h_fix_.at(c.index); }
    const std::vector<std::pair<Cell, Cell>> &copies() const { return a_.copies; }  // (copy, original) pairs so far

private:
    struct RegionColumn {
        uint8_t kind;  // 0 advice, 1 fixed, 2 selector
        uint32_t index;
    };
    uint64_t &height(const RegionColumn &c) { return c.kind == 0 ? h_adv_.at(c.index) : c.kind == 1 ? h_fix_.at(c.index) : h_sel_.at(c.index); }

    class ShapeRegion : public Region {
    public:
        std::vector<RegionColumn> columns;
        uint64_t row_count = 0;
        void use(uint8_t kind, uint32_t index, uint64_t offset) {
            bool seen = false;
            for (const auto &c : columns) seen = seen || (c.kind == kind && c.index == index);
            if (!seen) columns.push_back(RegionColumn{kind, index});
            row_count = std::max(row_count, offset + 1);
        }
        AssignedCell assign_advice(Column column, uint64_t offset, const std::function<Value()> &) override {
            use(0, column.index, offset);
            return AssignedCell{Cell{column, offset}, Value::unknown()};
        }
        void assign_fixed(Column column, uint64_t offset, const std::function<Value()> &) override { use(1, column.index, offset); }
        void enable_selector(Selector s, uint64_t offset) override { use(2, s.index, offset); }
        void constrain_equal(Cell, Cell) override {}
    };

    class AssignRegion : public Region {
    public:
        AssignRegion(Assembly &a, uint64_t start, bool with_witnesses) : a_(a), start_(start), with_witnesses_(with_witnesses) {}
        AssignedCell assign_advice(Column column, uint64_t offset, const std::function<Value()> &to) override {
            const uint64_t row = start_ + offset;
            Value v = Value::unknown();
            if (with_witnesses_) {  // keygen ignores the closure entirely
                v = to();
                a_.closure_calls++;
                if (!v.known) throw Error(Error::Synthesis, "witness value unknown");
                if (v.v > 255) throw Error(Error::Synthesis, "advice value out of byte range");
                a_.advice.at(column.index).at(row) = (uint8_t)v.v;
                a_.advice_assigned.at(column.index).at(row) = 1;
            }
            return AssignedCell{Cell{column, row}, v};
        }
        void assign_fixed(Column column, uint64_t offset, const std::function<Value()> &to) override {
            const Value v = to();
            a_.fixed.at(column.index).at(start_ + offset) = (uint8_t)v.v;
            a_.fixed_assigned.at(column.index).at(start_ + offset) = 1;
        }
        void enable_selector(Selector s, uint64_t offset) override { a_.selectors.at(s.index).at(start_ + offset) = 1; }
        void constrain_equal(Cell a, Cell b) override { a_.copies.emplace_back(a, b); }

    private:
        Assembly &a_;
        uint64_t start_;
        bool with_witnesses_;
    };

    Assembly &a_;
    bool with_witnesses_;
    std::vector<uint64_t> h_adv_, h_fix_, h_sel_;
};

// dev::MockProver: run = configure + synthesize into an Assembly; verify =
// assert_satisfied() for the argument kinds this gadget uses.
class MockProver {
public:
    ConstraintSystem cs;
    Assembly assembly;

    // circuit must provide: Config configure(ConstraintSystem&); void synthesize(Config, Layouter&)
    template <class Circuit>
    static MockProver run(uint32_t k, Circuit &circuit, bool with_witnesses = true) {
        MockProver p;
        auto config = circuit.configure(p.cs);
        Assembly &a = p.assembly;
        a.k = k;
        a.n_rows = (uint64_t)1 << k;
        a.advice.assign(p.cs.n_advice, std::vector<uint8_t>(a.n_rows, 0));
        a.advice_assigned.assign(p.cs.n_advice, std::vector<uint8_t>(a.n_rows, 0));
        a.fixed.assign(p.cs.n_fixed, std::vector<uint8_t>(a.n_rows, 0));
        a.fixed_assigned.assign(p.cs.n_fixed, std::vector<uint8_t>(a.n_rows, 0));
        a.selectors.assign(p.cs.n_selectors, std::vector<uint8_t>(a.n_rows, 0));
        a.table.assign(p.cs.n_table, std::vector<uint8_t>());
        Layouter layouter(p.cs, a, with_witnesses);
        circuit.synthesize(config, layouter);
        return p;
    }

    // Empty string when satisfied, else the first failure.
    std::string verify() const;
};

}  // namespace host
}  // namespace aesw
