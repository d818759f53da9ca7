"""The Rust side of the boundary (rust/) pinned to include/aesw.h without a Rust toolchain.

There is no cargo / rustc in the build image, so nothing under rust/ can be compiled here.  What CAN be checked is the
one thing that is silent undefined behaviour when it drifts: that every prototype, #[repr(C)] struct, callback type and
constant of rust/aesw-sys/src/lib.rs says exactly what include/aesw.h says.  Both files are parsed into the same
canonical form (name, [(parameter, type)], return type) and compared item by item; the test then MUTATES the header text
in memory (one more argument, a dropped const, a widened integer, swapped struct fields, a changed constant) and requires
the comparison to notice.  Further: the unified diff of rust/halo2-aes-patch applies cleanly to the reference's files
(only where /root/reference exists: never on the GPU box) and the cursor arithmetic of src/aesw.rs -- one row per chip
call, sixteen for the plaintext region -- replayed in Python lands on the oracle's region order (400 key rows, 1 360 rows
per block) with every value-closure row reading a cell the PACKED layout really holds.
"""
import re
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "aesw.h"
RUST = ROOT / "rust" / "aesw-sys" / "src" / "lib.rs"
PATCH = ROOT / "rust" / "halo2-aes-patch" / "aesw-witness.patch"
AESW_RS = ROOT / "rust" / "halo2-aes-patch" / "src" / "aesw.rs"
REFERENCE = Path("/root/reference")

# ---------------------------------------------------------------------------------------------------------------------
# canonical types: ("int", bits, signed) | ("ptr", const, pointee) | ("void",) | ("struct", name) | ("fn", name) | ...
# ---------------------------------------------------------------------------------------------------------------------
C_INT = {"int": ("int", "c_int"), "uint8_t": ("int", "u8"), "uint16_t": ("int", "u16"), "uint32_t": ("int", "u32"),
         "uint64_t": ("int", "u64"), "int32_t": ("int", "i32"), "int64_t": ("int", "i64"), "size_t": ("int", "usize"),
         "char": ("int", "c_char"), "void": ("void",), "float": ("float", "f32")}
RUST_INT = {"c_int": ("int", "c_int"), "u8": ("int", "u8"), "u16": ("int", "u16"), "u32": ("int", "u32"), "u64": ("int", "u64"),
            "i32": ("int", "i32"), "i64": ("int", "i64"), "usize": ("int", "usize"), "c_char": ("int", "c_char"), "c_void": ("void",), "f32": ("float", "f32")}
NAMED = ("aesw_ctx", "aesw_comm", "aesw_key_slab", "aesw_copy_edge", "aesw_stream_stats", "aesw_columns", "aesw_batch", "aesw_check_report")
CALLBACKS = ("aesw_chunk_fn", "aesw_column_fn")


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", " ", text)


def c_type(decl):
    """'const uint8_t *const *d_send' -> (canonical type, 'd_send'); arrays decay to pointers."""
    decl = decl.strip()
    array = False
    m = re.search(r"\[[^\]]*\]\s*$", decl)
    if m:
        array = True
        decl = decl[:m.start()].strip()
    m = re.match(r"^(.*?)([A-Za-z_]\w*)$", decl)
    assert m, decl
    tpart, name = m.group(1).strip(), m.group(2)
    if not tpart:  # unnamed parameter such as "void"
        tpart, name = name, ""
    toks = re.findall(r"\*|[A-Za-z_]\w*", tpart)
    # base type with its own constness, then pointer levels each with the constness of the POINTER
    base_const = False
    base = None
    i = 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        elif toks[i] in ("struct", "enum"):
            pass
        else:
            base = toks[i]
        i += 1
    assert base, decl
    if base in C_INT:
        t = C_INT[base]
    elif base in NAMED:
        t = ("struct", base)
    elif base in CALLBACKS:
        t = ("fn", base)
    else:
        raise AssertionError("unknown C type %r in %r" % (base, decl))
    pointee_const = base_const
    while i < len(toks):
        assert toks[i] == "*", decl
        t = ("ptr", pointee_const, t)
        pointee_const = False
        i += 1
        while i < len(toks) and toks[i] == "const":
            pointee_const = True  # constness of the pointer just built = constness seen by the next level
            i += 1
    if array:  # "const uint8_t sbox[256]" is a pointer to const uint8_t; aesw.h has no arrays of pointers
        assert t[0] != "ptr", decl
        t = ("ptr", base_const, t)
    return t, name


def rust_type(text):
    text = text.strip()
    m = re.match(r"^\*(const|mut)\s+(.*)$", text, flags=re.S)
    if m:
        return ("ptr", m.group(1) == "const", rust_type(m.group(2)))
    if text in RUST_INT:
        return RUST_INT[text]
    if text in NAMED:
        return ("struct", text)
    if text in CALLBACKS:
        return ("fn", text)
    raise AssertionError("unknown Rust type %r" % text)


def split_top(text, sep=","):
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [o.strip() for o in out if o.strip()]


def parse_header(text):
    text = strip_c_comments(text)
    funcs, structs, callbacks, consts = {}, {}, {}, {}
    for m in re.finditer(r"#define\s+(AESW_\w+)\s+(\d+)u?\b", text):
        consts[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"enum\s+aesw_\w+\s*\{(.*?)\}", text, flags=re.S):
        for item in split_top(m.group(1)):
            k, v = item.split("=")
            consts[k.strip()] = int(v.strip())
    for m in re.finditer(r"typedef\s+struct\s+(aesw_\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for line in m.group(2).split(";"):
            line = line.strip()
            if not line:
                continue
            # "uint8_t dst_space, dst_col" declares two fields of one type
            first = split_top(line)
            t0, n0 = c_type(first[0])
            fields.append((n0, t0))
            for extra in first[1:]:
                star = extra.count("*")
                assert star == 0, "pointer declarator lists are not used in aesw.h"
                fields.append((extra.strip(), t0))
        structs[m.group(1)] = fields
    for m in re.finditer(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(aesw_\w+)\s*\)\s*\((.*?)\)\s*;", text, flags=re.S):
        ret, _ = c_type(m.group(1).strip() + " _")
        callbacks[m.group(2)] = ([c_type(a)[::-1] for a in split_top(m.group(3))], ret)
    body = re.sub(r"typedef\s+struct\s+aesw_\w+\s*\{.*?\}\s*aesw_\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"typedef[^;]*;", " ", body)
    body = re.sub(r"enum\s+aesw_\w+\s*\{.*?\}\s*;", " ", body, flags=re.S)
    body = re.sub(r"#[^\n]*", " ", body)
    body = body.replace('extern "C" {', " ").replace("}", " ")
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^(.*?)\b(aesw_\w+)\s*\((.*)\)$", stmt)
        if not m:
            assert "(" not in stmt, "unparsed declaration: %r" % stmt
            continue
        ret, _ = c_type(m.group(1).strip() + " _")
        args = split_top(m.group(3))
        params = [] if args == ["void"] else [c_type(a)[::-1] for a in args]
        funcs[m.group(2)] = (params, ret)
    return funcs, structs, callbacks, consts


def parse_rust(text):
    text = strip_rust_comments(text)
    funcs, structs, callbacks, consts = {}, {}, {}, {}
    for m in re.finditer(r"pub\s+const\s+(AESW_\w+)\s*:\s*\w+\s*=\s*(\d+)\s*;", text):
        consts[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub\s+struct\s+(aesw_\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in split_top(m.group(2)):
            fm = re.match(r"^(?:pub\s+)?(\w+)\s*:\s*(.*)$", f, flags=re.S)
            assert fm, f
            if fm.group(1) == "_private":
                continue
            fields.append((fm.group(1), rust_type(fm.group(2))))
        if m.group(1) not in ("aesw_ctx", "aesw_comm"):  # opaque handles: forward declarations in aesw.h
            structs[m.group(1)] = fields
    for m in re.finditer(r"pub\s+type\s+(aesw_\w+)\s*=\s*unsafe\s+extern\s+\"C\"\s+fn\s*\((.*?)\)\s*->\s*([^;]+);", text, flags=re.S):
        params = []
        for a in split_top(m.group(2)):
            n, t = a.split(":", 1)
            params.append((n.strip(), rust_type(t)))
        callbacks[m.group(1)] = (params, rust_type(m.group(3)))
    ext = re.search(r"extern\s+\"C\"\s*\{(.*)\}", text, flags=re.S)
    assert ext, "no extern \"C\" block"
    for m in re.finditer(r"pub\s+fn\s+(aesw_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext.group(1), flags=re.S):
        params = []
        for a in split_top(m.group(2)):
            n, t = a.split(":", 1)
            params.append((n.strip(), rust_type(t)))
        funcs[m.group(1)] = (params, rust_type(m.group(3)) if m.group(3) else ("void",))
    return funcs, structs, callbacks, consts


def differences(c, r):
    """Human-readable list of every disagreement between the two parses."""
    out = []
    cf, cs, cc, ck = c
    rf, rs, rc, rk = r
    for what, a, b in (("function", cf, rf), ("struct", cs, rs), ("callback", cc, rc)):
        for name in sorted(set(a) - set(b)):
            out.append("%s %s: in aesw.h, not in lib.rs" % (what, name))
        for name in sorted(set(b) - set(a)):
            out.append("%s %s: in lib.rs, not in aesw.h" % (what, name))
    for name in sorted(set(cf) & set(rf)):
        (cp, cr), (rp, rr) = cf[name], rf[name]
        if cr != rr:
            out.append("%s: return type %r vs %r" % (name, cr, rr))
        if len(cp) != len(rp):
            out.append("%s: %d parameters in aesw.h, %d in lib.rs" % (name, len(cp), len(rp)))
            continue
        for i, ((cn, ct), (rn, rt)) in enumerate(zip(cp, rp)):
            if ct != rt:
                out.append("%s: parameter %d (%s / %s): %r vs %r" % (name, i, cn, rn, ct, rt))
            if cn and cn != rn:
                out.append("%s: parameter %d is named %s in aesw.h and %s in lib.rs" % (name, i, cn, rn))
    for name in sorted(set(cc) & set(rc)):
        (cp, cr), (rp, rr) = cc[name], rc[name]
        if cr != rr or [t for _, t in cp] != [t for _, t in rp]:
            out.append("callback %s differs: %r vs %r" % (name, (cp, cr), (rp, rr)))
    for name in sorted(set(cs) & set(rs)):
        if name in ("aesw_ctx", "aesw_comm"):
            continue
        if cs[name] != rs[name]:
            out.append("struct %s: fields %r vs %r" % (name, cs[name], rs[name]))
    for name in sorted(set(ck) | set(rk)):
        if ck.get(name) != rk.get(name):
            out.append("constant %s: %r in aesw.h, %r in lib.rs" % (name, ck.get(name), rk.get(name)))
    return out


@pytest.fixture(scope="module")
def parsed():
    return parse_header(HEADER.read_text()), parse_rust(RUST.read_text())


def test_the_parsers_see_everything(parsed):
    (cf, cs, cc, ck), (rf, rs, rc, rk) = parsed
    # every aesw_ symbol the library exports is a parsed prototype (tests/test_abi.py pins header <-> library)
    assert len(cf) >= 46 and "aesw_encrypt_witness_device" in cf and "aesw_gather_columns_device" in cf
    assert len(cf["aesw_encrypt_witness_device"][0]) == 12
    assert cf["aesw_gather_columns_device"][0][3] == ("d_send", ("ptr", True, ("ptr", True, ("int", "u8"))))
    assert cf["aesw_gather_columns_device"][0][4] == ("d_recv", ("ptr", True, ("ptr", False, ("int", "u8"))))
    assert cf["aesw_create"][0][2] == ("sbox", ("ptr", True, ("int", "u8")))  # const uint8_t sbox[256] decays to a pointer
    assert cf["aesw_packed_index"][0][1] == ("idx", ("ptr", False, ("int", "i32")))
    assert set(cs) == {"aesw_key_slab", "aesw_copy_edge", "aesw_stream_stats", "aesw_columns", "aesw_batch", "aesw_check_report"}
    assert [n for n, _ in cs["aesw_check_report"]] == ["blocks", "keys", "lookup_failures", "copy_failures", "gate_failures", "input_failures", "first"]
    assert [n for n, _ in cs["aesw_copy_edge"]] == ["dst_space", "dst_col", "dst_row", "src_space", "src_col", "src_row"]
    assert set(cc) == {"aesw_chunk_fn", "aesw_column_fn"}
    assert ck["AESW_AES_ROWS"] == 1360 and ck["AESW_ERR_COMM"] == 9 and ck["AESW_LAYOUT_VALUES"] == 2


def test_rust_bindings_equal_the_header(parsed):
    c, r = parsed
    diff = differences(c, r)
    assert not diff, "\n".join(diff)


@pytest.mark.parametrize("name,old,new,expect", [
    ("one more argument", "int aesw_expand_fr_device(aesw_ctx *ctx, const uint8_t *d_cells, uint64_t n_cells, uint8_t *d_fr,\n                          void *stream);",
     "int aesw_expand_fr_device(aesw_ctx *ctx, const uint8_t *d_cells, uint64_t n_cells, uint8_t *d_fr,\n                          int flags, void *stream);",
     "aesw_expand_fr_device: 6 parameters in aesw.h, 5 in lib.rs"),
    ("dropped const", "int aesw_last_stream_stats(const aesw_ctx *ctx, aesw_stream_stats *out);",
     "int aesw_last_stream_stats(aesw_ctx *ctx, aesw_stream_stats *out);", "aesw_last_stream_stats: parameter 0"),
    ("widened integer", "int aesw_block_placement(uint32_t k, uint32_t n_sets, uint64_t b, uint32_t *set, uint64_t *row);",
     "int aesw_block_placement(uint32_t k, uint64_t n_sets, uint64_t b, uint32_t *set, uint64_t *row);", "aesw_block_placement: parameter 1"),
    ("pointer became a value", "int aesw_device_count(int *count);", "int aesw_device_count(int count);", "aesw_device_count: parameter 0"),
    ("return type", "uint64_t aesw_block_capacity(uint32_t k, uint32_t n_sets);", "uint32_t aesw_block_capacity(uint32_t k, uint32_t n_sets);",
     "aesw_block_capacity: return type"),
    ("struct fields swapped", "    uint64_t kernel_ns;\n    uint64_t d2h_ns;", "    uint64_t d2h_ns;\n    uint64_t kernel_ns;", "struct aesw_stream_stats"),
    ("struct field narrowed", "    uint16_t dst_row;", "    uint8_t dst_row;", "struct aesw_copy_edge"),
    ("callback argument", "uint64_t first_block, uint64_t n_blocks, const uint8_t *x,", "uint64_t first_block, uint32_t n_blocks, const uint8_t *x,",
     "callback aesw_chunk_fn"),
    ("constant", "#define AESW_KEY_ROWS 400u", "#define AESW_KEY_ROWS 401u", "constant AESW_KEY_ROWS"),
    ("status code", "AESW_ERR_NO_KEY = 6,", "AESW_ERR_NO_KEY = 16,", "constant AESW_ERR_NO_KEY"),
    ("new function", "int aesw_version(void);", "int aesw_version(void);\nint aesw_new_entry_point(aesw_ctx *ctx);", "function aesw_new_entry_point: in aesw.h, not in lib.rs"),
])
def test_a_drifted_header_is_noticed(parsed, name, old, new, expect):
    text = HEADER.read_text()
    assert text.count(old) == 1, "mutation anchor for %r not found exactly once" % name
    diff = differences(parse_header(text.replace(old, new)), parsed[1])
    assert any(expect in d for d in diff), (name, diff)


def test_every_exported_function_is_bound(parsed):
    """The library's dynamic symbols, the header and lib.rs name the same entry points (when the library is built)."""
    lib = ROOT / "halo2-aes_amd" / "libaesw.so"
    if not lib.exists():
        pytest.skip("libaesw.so not built")
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("aesw_") and " T " in l}
    assert exported == set(parsed[1][0]), (sorted(exported - set(parsed[1][0])), sorted(set(parsed[1][0]) - exported))


def test_integration_md_points_at_the_files_instead_of_carrying_a_copy():
    text = (ROOT / "INTEGRATION.md").read_text()
    assert "rust/aesw-sys/src/lib.rs" in text and "rust/halo2-aes-patch/aesw-witness.patch" in text
    assert 'extern "C" {' not in text, "INTEGRATION.md must not carry its own copy of the extern block"


def test_cargo_manifest_and_build_script():
    toml = (ROOT / "rust" / "aesw-sys" / "Cargo.toml").read_text()
    assert 'links = "aesw"' in toml and 'build = "build.rs"' in toml and 'name = "aesw-sys"' in toml
    build = (ROOT / "rust" / "aesw-sys" / "build.rs").read_text()
    assert "rustc-link-lib=dylib=aesw" in build and "rustc-link-lib=dylib=amdhip64" in build and "AESW_LIB_DIR" in build


# ---------------------------------------------------------------------------------------------------------------------
# the patch
# ---------------------------------------------------------------------------------------------------------------------
PATCHED_FILES = ["Cargo.toml", "benches/aes128.rs", "benches/key_schedule.rs", "src/aes128.rs", "src/chips/gf_mul_chip.rs",
                 "src/chips/sbox_chip.rs", "src/chips/u8_range_check_chip.rs", "src/chips/u8_xor_chip.rs", "src/key_schedule.rs",
                 "src/lib.rs", "src/main.rs"]


def test_patch_touches_only_the_value_sources():
    text = PATCH.read_text()
    files = re.findall(r"^\+\+\+ b/(\S+)", text, flags=re.M)
    assert files == PATCHED_FILES
    removed = [l[1:].strip() for l in text.splitlines() if l.startswith("-") and not l.startswith("---")]
    added = [l[1:].strip() for l in text.splitlines() if l.startswith("+") and not l.startswith("+++")]
    # the six value sources of SURVEY 8(b) are what goes away ...
    for gone in ("xor_bytes(", "|| sub_byte(&x_copied.value_field().evaluate()),", "Fp::from($dict[*v.to_bytes().first().unwrap() as usize] as u64)",
                 "|| Value::known(Fp::from(p as u64)),", "|| Value::known(Fp::from(byte as u64)),", "|| Value::known(Fp::from(0)),"):
        assert any(gone in r for r in removed), gone
    # ... and buffer reads come in; configure(), selectors, lookups and copy_advice() calls are untouched
    for new in ("|| aesw::z_at(at),", "|| aesw::y_at(at),", "|| crate::aesw::y_at(at),", "|| aesw::x_at(pt_at.offset(i)),", "|| crate::aesw::word_at(i),"):
        assert any(new == a for a in added), new
    assert not any("copy_advice" in r and "let " not in r for r in removed)
    assert not any(w in r for r in removed for w in ("meta.lookup", "enable(", "constrain_equal", "assign_fixed"))
    # small: hunks with two lines of context, never a whole reference file
    assert len(text.splitlines()) < 300


@pytest.mark.skipif(not REFERENCE.exists() or shutil.which("patch") is None, reason="needs /root/reference and patch(1): this container only")
def test_patch_applies_to_the_reference(tmp_path):
    for f in PATCHED_FILES:
        (tmp_path / f).parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(REFERENCE / f, tmp_path / f)
    r = subprocess.run(["patch", "-p1", "--fuzz=0", "-d", str(tmp_path), "-i", str(PATCH)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "fuzz" not in r.stdout and "offset" not in r.stdout, r.stdout
    xor = (tmp_path / "src/chips/u8_xor_chip.rs").read_text()
    assert "xor_bytes" not in xor and "aesw::z_at(at)" in xor and xor.count("copy_advice(") == 2
    # every closure the patched files still hand to assign_advice for a VALUE is a buffer read
    for f in ("src/chips/u8_xor_chip.rs", "src/chips/sbox_chip.rs", "src/chips/gf_mul_chip.rs"):
        body = (tmp_path / f).read_text()
        assert len(re.findall(r"\|\| (?:crate::)?aesw::[xyz]_at\(at\)", body)) == 1, f
        assert len(re.findall(r"let at = (?:crate::)?aesw::advance\(1\);", body)) == 1, f


def test_cursor_walk_of_aesw_rs_matches_the_oracle_region_order():
    """Replay the gadget's call sequence with the cursor rules of src/aesw.rs (advance(1) per chip call and per lcon copy
    region, advance(16) for the plaintext region; enter_key / enter_next_block reset the row) and check it against the
    oracle's layouter-derived order: the row each chip lands on carries that chip's selector, and the cell its closure
    reads exists in the PACKED layout."""
    import oracle_lib
    orc = oracle_lib.Oracle()
    text = AESW_RS.read_text()
    assert "pub fn advance(rows: usize) -> At" in text and "pub fn enter_next_block()" in text and "pub fn enter_key()" in text
    XOR, SBOX, MUL2, MUL3, RANGE, COPY, PT = "xor", "sbox", "mul2", "mul3", "range", "copy", "pt"
    # key schedule, src/key_schedule.rs:122-224 (the chips only: words_column regions do not move the cursor)
    key_calls = []
    for _ in range(10):
        key_calls += [SBOX] * 4 + [XOR] * 4 + [XOR] * 4 + [XOR] * 12 + [RANGE] * 16
    # encrypt, src/aes128.rs:154-301
    matrix = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]
    enc_calls = [(PT, 16)] + [(XOR, 1)] * 16
    for rnd in range(1, 11):
        enc_calls += [(SBOX, 1)] * 16
        if rnd != 10:
            for _w in range(4):
                for col in matrix:
                    enc_calls += [({1: COPY, 2: MUL2, 3: MUL3}[c], 1) for c in col] + [(XOR, 1)] * 3
        enc_calls += [(XOR, 1)] * 16
    with orc.circuit(12, 1, np.zeros(16, np.uint8), np.zeros((1, 16), np.uint8), record_copies=False) as c:
        sel = {RANGE: c.selector(0), XOR: c.selector(1), SBOX: c.selector(2), MUL2: c.selector(3), MUL3: c.selector(4)}
    row = 0
    for kind in key_calls:
        assert sel[kind][row] == 1, (kind, row)
        row += 1
    assert row == 400
    idx = [orc.packed_index(cc) for cc in range(3)]
    kidx = [orc.key_packed_index(cc) for cc in range(3)]
    # value closures read y (sbox) or z (xor) of key rows: those cells exist in the packed key slab
    for r, kind in enumerate(key_calls):
        if kind == SBOX:
            assert kidx[1][r] >= 0
        if kind == XOR:
            assert kidx[2][r] >= 0
    row = 0
    for kind, rows in enc_calls:
        base = 400 + row
        if kind == PT:
            assert all(idx[0][row + i] >= 0 for i in range(16))
            assert not any(s[base:base + 16].any() for s in sel.values())
        elif kind == COPY:
            assert not any(s[base] for s in sel.values()) and idx[0][row] >= 0 and idx[1][row] < 0
        else:
            assert sel[kind][base] == 1, (kind, row)
            assert idx[2 if kind == XOR else 1][row] >= 0, (kind, row)
        row += rows
    assert row == 1360
