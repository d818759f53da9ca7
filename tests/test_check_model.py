"""aesw_check.h -- the device checker's own source -- on the CPU (tests/lane_model), against the oracle.

What is checked is what the reference's only executable correctness tests check (MockProver::assert_satisfied,
src/aes128.rs:409-418, src/key_schedule.rs:385-392): lookups, the round-constant gate, copy constraints -- restated per slab.
Here: (1) a witness the oracle wrote satisfies everything, in both layouts and both key modes; (2) EVERY single-cell change of a
block slab or a key slab is caught (every assigned cell takes part in a lookup, a copy or a literal row); (3) for a sample of such
changes the verdict -- and what kind of constraint notices first -- agrees with the oracle's MockProver-style verifier on the
same cell of a real K = 11 circuit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

KINDS = {1: "lookup", 2: "copy", 3: "gate", 4: "input"}


@pytest.fixture(scope="module")
def model(pkg):
    import __graft_entry__ as ge
    L = C.CDLL(str(ge.build_lane_model()))
    L.lane_model_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64] + [C.c_void_p] * 9
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _tab(oracle):
    t = oracle.t
    return np.concatenate([np.frombuffer(bytes(getattr(t, n)), np.uint8) for n in ("sbox", "mul2", "mul3")]).copy()


def _check(model, tab, layout, pt, keys, pbk, w, k, ct=None):
    rep = np.zeros(7, np.uint64)
    rc = model.lane_model_check(_p(tab), layout, _p(pt), _p(keys), 1 if pbk else 0, pt.shape[0], _p(w.x), _p(w.y), _p(w.z), _p(ct),
                                _p(k.w), _p(k.kx), _p(k.ky), _p(k.kz), _p(rep))
    assert rc == 0
    f = int(rep[6])
    first = None if f == 2 ** 64 - 1 else (f >> 20, bool((f >> 19) & 1), (f >> 16) & 7, f & 0xFFFF)
    return {"lookup": int(rep[2]), "copy": int(rep[3]), "gate": int(rep[4]), "input": int(rep[5]), "first": first,
            "blocks": int(rep[0]), "keys": int(rep[1])}


def _witness(oracle, layout, n, pbk, seed):
    rng = np.random.default_rng(seed)
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (n, 16) if pbk else 16, dtype=np.uint8)
    w = oracle.encrypt_witness(pt, keys, layout=layout)
    k = oracle.key_schedule_witness(keys, layout=layout)
    return pt, keys, w, k


@pytest.mark.parametrize("layout", [ol.DENSE, ol.PACKED])
@pytest.mark.parametrize("pbk", [False, True])
def test_an_oracle_witness_satisfies_every_constraint(model, oracle, layout, pbk):
    tab = _tab(oracle)
    pt, keys, w, k = _witness(oracle, layout, 37, pbk, 1)
    r = _check(model, tab, layout, pt, keys, pbk, w, k, ct=w.ct)
    assert r == {"lookup": 0, "copy": 0, "gate": 0, "input": 0, "first": None, "blocks": 37, "keys": 37 if pbk else 1}
    # without the key bytes the literal rows of words_column are not compared, everything else is
    assert _check(model, tab, layout, pt, None if not pbk else keys, pbk, w, k)["first"] is None
    # the zero vector of the reference's own tests (src/aes128.rs:389-402)
    z = np.zeros((3, 16), np.uint8)
    wz, kz = oracle.encrypt_witness(z, z[0], layout=layout), oracle.key_schedule_witness(z[0], layout=layout)
    assert _check(model, tab, layout, z, z[0], False, wz, kz, ct=wz.ct)["first"] is None


@pytest.mark.parametrize("layout", [ol.DENSE, ol.PACKED])
def test_every_single_cell_change_is_caught(model, oracle, layout):
    """Flip one bit of one ASSIGNED cell of block 1 (of 3) / of the key slab: the checker must object, name that unit, and object
    to nothing once the cell is restored.  3 024 + 936 cells."""
    tab = _tab(oracle)
    pt, key, w, k = _witness(oracle, layout, 3, False, 2)
    masks = [oracle.assigned_mask(c) for c in range(3)] if hasattr(oracle, "assigned_mask") else None
    strides = ol.ENC_STRIDE[layout]
    missed = []
    for ci, col in enumerate((w.x, w.y, w.z)):
        s = strides[ci]
        for i in range(s):
            if layout == ol.DENSE and not _dense_assigned(oracle, ci)[i]:
                continue  # a cell the reference never assigns: no constraint reads it
            col[s + i] ^= 0x10
            r = _check(model, tab, layout, pt, key, False, w, k)
            col[s + i] ^= 0x10
            if r["first"] is None or r["first"][0] != 1 or r["first"][1]:
                missed.append(("block", ci, i, r["first"]))
    kstr = ol.KEY_STRIDE[layout]
    for ci, col in enumerate((k.kx, k.ky, k.kz, k.w)):
        s = kstr[ci] if ci < 3 else ol.WORDS_ROWS
        for i in range(s):
            if ci < 3 and layout == ol.DENSE and not _dense_key_assigned(oracle, ci)[i]:
                continue
            col[i] ^= 0x04
            r = _check(model, tab, layout, pt, key, False, w, k)
            col[i] ^= 0x04
            if r["first"] is None:
                missed.append(("key", ci, i, None))
    assert not missed, missed[:10]
    assert _check(model, tab, layout, pt, key, False, w, k)["first"] is None


def _dense_assigned(oracle, col):
    m = np.zeros(ol.AES_ROWS, np.uint8)
    assert oracle.L.aesw_o_encrypt_assigned_mask(col, _p(m)) == 0
    return m


def _dense_key_assigned(oracle, col):
    m = np.zeros(400, np.uint8)
    assert oracle.L.aesw_o_key_assigned_mask(col, _p(m)) == 0
    return m


def test_verdicts_agree_with_the_oracles_mockprover_on_a_real_circuit(model, oracle, pkg):
    """The same cell changed in a slab (checker) and in a K = 11, N = 2 circuit holding that block in set 1 (the oracle's
    restated synthesize() + MockProver-style verify): both accept the untouched witness, both reject every change, and the kind
    the oracle names (it looks at lookups, then the gate, then copies) is a kind the checker counted."""
    tab = _tab(oracle)
    rng = np.random.default_rng(3)
    pt, key = rng.integers(0, 256, (1, 16), dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    w, k = oracle.encrypt_witness(pt, key, layout=ol.DENSE), oracle.key_schedule_witness(key, layout=ol.DENSE)
    n_sets = 2
    cells = [("block", c, r) for c in range(3) for r in (0, 15, 16, 31, 32, 47, 48, 52, 54, 159, 160, 176, 1327, 1328, 1343, 1344, 1359)
             if _dense_assigned(oracle, c)[r]]
    cells += [("key", c, r) for c in range(3) for r in (0, 3, 4, 7, 8, 23, 24, 39, 399) if _dense_key_assigned(oracle, c)[r]]
    cells += [("words", 0, r) for r in (0, 15, 16, 19, 20, 21, 95)]
    with oracle.circuit(11, n_sets, key, pt) as circ:
        assert circ.verify()[0] == 0
        bset, brow = circ.block_placement(0)
        assert (bset, brow) == (1, 0)
        for space, c, r in cells:
            arr = {"block": (w.x, w.y, w.z)[c] if space == "block" else None, "key": (k.kx, k.ky, k.kz)[c] if space == "key" else None,
                   "words": k.w}[space]
            ccol, crow = (3 * bset + c, brow + r) if space == "block" else ((c, r) if space == "key" else (3 * n_sets, r))
            old = int(arr[r])
            arr[r] = old ^ 0x21
            circ.poke(ccol, crow, old ^ 0x21)
            rep = _check(model, tab, ol.DENSE, pt, key, False, w, k)
            rc, msg = circ.verify()
            arr[r] = old
            circ.poke(ccol, crow, old)
            assert rc != 0 and rep["first"] is not None, (space, c, r, msg, rep)
            named = "lookup" if "lookup" in msg else "gate" if "gate" in msg else "copy" if "copy" in msg else None
            assert named is not None, msg
            # a plaintext / key literal has no constraint of its own in the circuit (the oracle sees the copies that read it)
            assert rep[named] > 0 or (named == "copy" and rep["input"] > 0), (space, c, r, msg, rep)
        assert circ.verify()[0] == 0
    assert _check(model, tab, ol.DENSE, pt, key, False, w, k)["first"] is None


def test_first_failure_is_the_smallest_unit(model, oracle):
    tab = _tab(oracle)
    pt, keys, w, k = _witness(oracle, ol.PACKED, 20, True, 4)
    sx, sy, sz = ol.ENC_STRIDE[ol.PACKED]
    w.z[7 * sz + 100] ^= 1      # block 7: an xor row's output
    w.y[12 * sy + 5] ^= 1       # block 12
    k.kz[15 * 200 + 3] ^= 1     # key slab 15
    r = _check(model, tab, ol.PACKED, pt, keys, True, w, k)
    assert r["first"][0] == 7 and not r["first"][1] and r["lookup"] >= 2 and r["copy"] >= 2
    w.z[7 * sz + 100] ^= 1
    w.y[12 * sy + 5] ^= 1
    r = _check(model, tab, ol.PACKED, pt, keys, True, w, k)
    assert r["first"][0] == 15 and r["first"][1]


def test_the_tables_are_inputs_of_the_check_too(model):
    """src/table.rs builds the lookup table from whatever src/constant.rs holds; so does the checker.  A witness made with a random
    S-box and random (non-xtime) mul tables satisfies the check under THOSE tables -- including the oracle's own MockProver on a
    real circuit -- and fails it under the reference's; FIPS tables (S_BOX[255] = 22) differ from the reference's in exactly the
    rows whose S-box input is 0xff."""
    rng = np.random.default_rng(6)
    tables = (rng.permutation(256).astype(np.uint8), rng.integers(0, 256, 256, dtype=np.uint8), rng.integers(0, 256, 256, dtype=np.uint8))
    o2 = ol.Oracle(tables=tables)
    pt, keys = rng.integers(0, 256, (30, 16), dtype=np.uint8), rng.integers(0, 256, (30, 16), dtype=np.uint8)
    for layout in (ol.DENSE, ol.PACKED):
        w, k = o2.encrypt_witness(pt, keys, layout=layout), o2.key_schedule_witness(keys, layout=layout)
        assert _check(model, np.concatenate(tables), layout, pt, keys, True, w, k, ct=w.ct)["first"] is None
        ref = _check(model, _tab(ol.Oracle()), layout, pt, keys, True, w, k, ct=w.ct)
        assert ref["lookup"] > 1000 and ref["copy"] == 0 and ref["gate"] == 0 and ref["input"] == 0  # the copies and literals do not depend on the tables
    with o2.circuit(11, 2, keys[0], pt[:1]) as circ:
        assert circ.verify()[0] == 0
    # the reference's table against FIPS-197's: one entry apart
    o = ol.Oracle()
    pt[3], keys[3] = 0xFF, 0  # pt ^ key = 0xff in every byte: round 1's S-box rows read entry 255
    w, k = o.encrypt_witness(pt, keys, layout=ol.PACKED), o.key_schedule_witness(keys, layout=ol.PACKED)
    fips = _tab(o)
    assert fips[255] == 23
    fips[255] = 22
    r = _check(model, fips, ol.PACKED, pt, keys, True, w, k)
    assert r["lookup"] >= 16 and r["first"][0] <= 3 and r["copy"] == 0
