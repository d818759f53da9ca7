"""The C-ABI shared library: loads, exports every symbol include/aesw.h
declares, and its pure-host entry points behave (no compute calls: no GPU)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol

ROOT = Path(__file__).resolve().parent.parent


def _declared(header="aesw.h"):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aesw_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pkg):
    lib = pkg.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libaesw.so does not export %s" % n
    assert sorted(pkg.api.SYMBOLS) == names, "api.py binds a different set than include/aesw.h declares"
    host_names = _declared("aesw_host.h")
    assert sorted(pkg.api.HOST_SYMBOLS) == host_names, "api.py binds a different set than include/aesw_host.h declares"
    out = subprocess.run(["nm", "-D", "--defined-only", str(pkg.api.LIB_PATH)], stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r" T (aesw_\w+)", out))
    assert set(names) <= exported
    # the device library carries no host-mirror code: that lives in libaesw_host.so, above the C ABI
    assert not any(n.startswith("aesw_host_") and n not in ("aesw_host_alloc", "aesw_host_free", "aesw_host_register", "aesw_host_unregister") for n in exported)
    hlib = pkg.api.load_host_library()
    for n in host_names:
        assert hasattr(hlib, n), "libaesw_host.so does not export %s" % n
    out = subprocess.run(["nm", "-D", "--defined-only", str(pkg.api.HOST_LIB_PATH)], stdout=subprocess.PIPE, text=True).stdout
    assert set(host_names) <= set(re.findall(r" T (aesw_\w+)", out))
    assert b"gfx950" not in pkg.api.HOST_LIB_PATH.read_bytes()   # host code only


def test_no_diagnostic_kernels_in_the_product(pkg):
    """The "leave the flush out" (store_mode 3) instantiations exist only in -DAESW_DIAGNOSTIC builds."""
    out = subprocess.run(["nm", "-C", str(pkg.api.LIB_PATH)], stdout=subprocess.PIPE, text=True).stdout
    enc = [l for l in out.splitlines() if "encrypt_kernel<" in l]
    assert enc, "no encrypt_kernel symbols found"
    assert not any(re.search(r"encrypt_kernel<[^>]*, 3>", l) for l in enc)


def test_library_is_gfx950_code(pkg):
    """The product .so carries a gfx950 code object (hipcc --offload-arch=gfx950)."""
    data = pkg.api.LIB_PATH.read_bytes()
    assert b"gfx950" in data
    assert b"encrypt_kernel" in data and b"key_kernel" in data


def test_version_and_strerror(pkg):
    lib = pkg.load_library()
    assert lib.aesw_version() == 102
    assert lib.aesw_strerror(0) == b"ok"
    assert b"AES calls too many" in lib.aesw_strerror(5)   # the reference's panic text, src/aes128.rs:161
    assert b"Keys should be scheduled" in lib.aesw_strerror(6)  # src/aes128.rs:170
    assert lib.aesw_strerror(12345) == b"unknown status"


def test_geometry(pkg, oracle):
    assert [pkg.column_stride(pkg.LAYOUT_DENSE, c) for c in range(3)] == [1360, 1360, 1360]
    assert [pkg.column_stride(pkg.LAYOUT_PACKED, c) for c in range(3)] == [1360, 1056, 608]
    assert [pkg.key_column_stride(pkg.LAYOUT_DENSE, c) for c in range(3)] == [400, 400, 400]
    assert [pkg.key_column_stride(pkg.LAYOUT_PACKED, c) for c in range(3)] == [400, 240, 200]
    assert pkg.column_stride(7, 0) == 0 and pkg.column_stride(0, 3) == 0
    for c in range(3):
        assert np.array_equal(pkg.packed_index(c), oracle.packed_index(c))
        assert np.array_equal(pkg.key_packed_index(c), oracle.key_packed_index(c))
    # algorithmic bytes per block: live cells + inputs
    live = sum(pkg.column_stride(pkg.LAYOUT_PACKED, c) for c in range(3))
    klive = 96 + sum(pkg.key_column_stride(pkg.LAYOUT_PACKED, c) for c in range(3))
    assert live + 16 == 3040 and live + klive + 32 == 3992


def test_values_layout_geometry(pkg, oracle):
    """AESW_LAYOUT_VALUES = the cells whose selector says a chip closure computes them (oracle circuit), nothing else."""
    V = pkg.LAYOUT_VALUES
    assert [pkg.column_stride(V, c) for c in range(3)] == [0, 448, 608]
    assert [pkg.key_column_stride(V, c) for c in range(3)] == [pkg.key_column_stride(pkg.LAYOUT_PACKED, c) for c in range(3)]
    for c in range(3):
        idx = pkg.layout_index(V, c)
        kept = idx >= 0
        assert np.array_equal(kept, oracle.values_mask(c))
        assert np.array_equal(idx[kept], np.arange(kept.sum()))          # row order, no gaps
        assert np.array_equal(pkg.layout_index(pkg.LAYOUT_PACKED, c), pkg.packed_index(c))
        assert np.array_equal(pkg.layout_index(pkg.LAYOUT_DENSE, c), np.arange(1360))
    # every kept z cell is an assigned z cell; every kept y cell is an assigned y cell that is not an xor row
    enc, _, _, _ = pkg.selector_tags()
    assert np.array_equal(pkg.layout_index(V, 1) >= 0, np.isin(enc, (3, 4, 5)))
    assert np.array_equal(pkg.layout_index(V, 2) >= 0, enc == 2)
    with pytest.raises(pkg.AeswError):
        pkg.layout_index(7, 1)


def test_copy_graphs_equal_the_copies_of_synthesize(pkg, oracle):
    """aesw_block_copy_graph / aesw_key_copy_graph, placed with aesw_block_placement, == every copy_advice() the restated
    synthesize() records, in the same order (K=16, N=3: blocks in three column sets)."""
    k, n_sets, n = 16, 3, 120
    rng = np.random.default_rng(21)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    be, ke = pkg.block_copy_graph(), pkg.key_copy_graph()
    assert len(be) == 1952 and len(ke) == 640
    words_col = 3 * n_sets

    def place(space, col, row, set_, row0):
        col, row = col.astype(np.uint64), row.astype(np.uint64)
        c = np.where(space == 0, 3 * set_ + col, np.where(space == 1, col, words_col))
        r = np.where(space == 0, row0 + row, row)
        return c.astype(np.uint64), r.astype(np.uint64)

    expect = []
    sc, sr = place(ke["src_space"], ke["src_col"], ke["src_row"], 0, 0)
    dc, dr = place(ke["dst_space"], ke["dst_col"], ke["dst_row"], 0, 0)
    expect.append(np.stack([sc, sr, dc, dr], axis=1))
    for b in range(n):
        s, r0 = pkg.block_placement(k, n_sets, b)
        sc, sr = place(be["src_space"], be["src_col"], be["src_row"], s, r0)
        dc, dr = place(be["dst_space"], be["dst_col"], be["dst_row"], s, r0)
        expect.append(np.stack([sc, sr, dc, dr], axis=1))
    expect = np.concatenate(expect)
    with oracle.circuit(k, n_sets, key, pts) as c:
        got = c.copies()
    assert got.shape == expect.shape
    assert np.array_equal(got, expect)


def test_block_placement_mirrors_aes_callable(pkg, oracle):
    """aesw_block_placement == where the oracle's restated aes_callable() puts blocks."""
    assert pkg.block_capacity(20, 5) == 769 + 4 * 771 == 3853
    assert pkg.block_capacity(20, 4) == 3082 and pkg.block_capacity(20, 3) == 2311
    assert pkg.block_placement(20, 3, 0) == (0, 400)
    assert pkg.block_placement(20, 3, 768) == (0, 400 + 768 * 1360)
    assert pkg.block_placement(20, 3, 769) == (1, 0)
    with pytest.raises(pkg.AeswError) as e:
        pkg.block_placement(20, 5, 3853)           # the reference panics here (benches/aes128.rs asks for 6000)
    assert e.value.status == 5
    pts = np.zeros((120, 16), np.uint8)
    with oracle.circuit(16, 3, np.zeros(16, np.uint8), pts, record_copies=False) as c:
        for b in range(120):
            assert c.block_placement(b) == pkg.block_placement(16, 3, b)
    assert pkg.block_capacity(10, 2) == 0          # 2^10 < 1760: set 0 holds nothing, set 1 holds 0 (1024 < 1360)


def test_selector_tags_match_oracle_circuit(pkg, oracle):
    """Fixed selector data == the selectors the restated synthesize() enables (K=16, N=2)."""
    enc, key, q, rc = pkg.selector_tags()
    rng = np.random.default_rng(9)
    with oracle.circuit(16, 2, rng.integers(0, 256, 16, dtype=np.uint8), rng.integers(0, 256, (50, 16), dtype=np.uint8),
                        record_copies=False) as c:
        n_sets = 2
        for s in range(n_sets):
            sel = {1: c.selector(5 * s + 0), 2: c.selector(5 * s + 1), 3: c.selector(5 * s + 2),
                   4: c.selector(5 * s + 3), 5: c.selector(5 * s + 4)}
            expect = np.zeros(c.num_rows, np.uint8)
            if s == 0:
                expect[:400] = key
            for b in range(50):
                bs, row = c.block_placement(b)
                if bs == s:
                    expect[row:row + 1360] = enc
            for tag, col in sel.items():
                assert np.array_equal(col, (expect == tag).astype(np.uint8)), "set %d tag %d" % (s, tag)
        q_sel = c.selector(5 * n_sets)
        assert np.array_equal(q_sel[:96], q) and not q_sel[96:].any()
        assert np.array_equal(c.fixed()[:96], rc)


def test_block_placement_sweep(pkg):
    """aes_callable() as a pure function, against a direct restatement of src/aes128.rs:303-325."""
    def reference_walk(k, n_sets, count_blocks):
        current, count, out = 0, 0, []
        for _ in range(count_blocks):
            max_row = 2 ** k - (1760 if current == 0 else 0)
            if max_row >= count * 1360 + 1360:
                pass
            elif current < n_sets - 1:
                current, count = current + 1, 0
            else:
                return out, True   # panic
            out.append((current, (400 if current == 0 else 0) + count * 1360))
            count += 1
        return out, False
    for k, n_sets in ((11, 1), (11, 3), (12, 2), (13, 5), (16, 4), (20, 5)):
        cap = pkg.block_capacity(k, n_sets)
        walk, panicked = reference_walk(k, n_sets, cap + 1)
        assert panicked and len(walk) == cap
        step = max(1, cap // 200)
        for b in list(range(0, cap, step)) + [cap - 1] if cap else []:
            assert pkg.block_placement(k, n_sets, b) == walk[b]
        with pytest.raises(pkg.AeswError):
            pkg.block_placement(k, n_sets, cap)


def test_assemble_selectors_equals_synthesize(pkg, oracle):
    """Whole-circuit selector and fixed columns (keygen data) == what the restated synthesize() enables."""
    k, n_sets, n = 14, 3, 30            # 2^14 rows: 10 + 12 + 12 blocks fit; 30 fills sets 0,1 and part of 2
    assert pkg.block_capacity(k, n_sets) == 34
    sel, fixed = pkg.assemble_selectors(k, n_sets, n)
    with oracle.circuit(k, n_sets, np.zeros(16, np.uint8), np.zeros((n, 16), np.uint8), record_copies=False) as c:
        assert c.status == 0 and c.num_selectors == sel.shape[0]
        for s in range(c.num_selectors):
            assert np.array_equal(sel[s], c.selector(s)), "selector %d" % s
        assert np.array_equal(fixed, c.fixed())
    with pytest.raises(pkg.AeswError) as e:
        pkg.assemble_selectors(k, n_sets, 35)
    assert e.value.status == 5


def test_no_cpu_path(pkg):
    """Without a gfx950 device the library refuses to create a context."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.AeswError) as e:
        pkg.Context(0)
    assert e.value.status == 2
    assert pkg.device_count() == 0


def test_missing_library_fails_loudly(pkg, tmp_path):
    with pytest.raises(FileNotFoundError):
        pkg.api.load_library(tmp_path / "libaesw.so")


def test_product_does_not_touch_the_oracle():
    """Nothing under halo2-aes_amd/ or include/ may reference oracle/ or the lane model (test infrastructure)."""
    for p in list((ROOT / "halo2-aes_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.is_file() and p.suffix in (".py", ".h", ".hip", ".cpp", ".hpp"):
            text = p.read_text()
            for word in ("aesw_oracle", "oracle_lib", "libaesw_oracle", "liblane_model", "oracle/"):
                assert word not in text, "%s mentions %s" % (p, word)
