"""The device program's shared source (aesw_lane.h + the staging windows and
whole-line flush math of aesw_layout.h) run on the CPU, against the oracle.
This is host logic: it catches wrong v_perm selectors, slab offsets, window
slots and flush indices without a GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol


@pytest.fixture(scope="module")
def model(pkg):
    import __graft_entry__ as ge
    L = C.CDLL(str(ge.build_lane_model()))
    L.lane_model_run.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int] + [C.c_void_p] * 9
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


GUARD = 256


def _run(model, tab, pt, keys, pbk, key_only, layout, xt):
    n = pt.shape[0] if not key_only else keys.reshape(-1, 16).shape[0]
    sx, sy, sz = ol.ENC_STRIDE[layout]
    kxs, kys, kzs = ol.KEY_STRIDE[layout]
    bufs = {k: np.full(n * s + GUARD, 0xAB, np.uint8) for k, s in
            (("x", sx), ("y", sy), ("z", sz), ("ct", 16), ("w", 96), ("kx", kxs), ("ky", kys), ("kz", kzs), ("rk", 176))}
    rc = model.lane_model_run(_p(tab), _p(pt), _p(keys), pbk, key_only, n, layout, xt, *[_p(bufs[k]) for k in
                              ("x", "y", "z", "ct", "w", "kx", "ky", "kz", "rk")])
    assert rc == 0
    return bufs


def _check(bufs, name, exp):
    got = bufs[name]
    exp = exp.reshape(-1)
    assert np.array_equal(got[:exp.size], exp), "%s differs at %s" % (name, np.nonzero(got[:exp.size] != exp)[0][:8])
    assert np.all(got[exp.size:] == 0xAB), "%s: wrote past the end of the buffer" % name


@pytest.mark.parametrize("layout", [ol.DENSE, ol.PACKED, ol.VALUES])
@pytest.mark.parametrize("xt", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 7, 15, 16, 17, 31, 33, 100])
def test_encrypt_matches_oracle(model, oracle, layout, xt, n):
    rng = np.random.default_rng(1000 + n)
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    pt[0] = keys[0] ^ 0xFF
    tab = np.concatenate(oracle.tables())
    for pbk in (0, 1):
        k = keys if pbk else keys[0].copy()
        bufs = _run(model, tab, pt, k, pbk, 0, layout, xt)
        exp = oracle.encrypt_witness(pt, k, layout=layout)
        for c in "xyz":
            _check(bufs, c, getattr(exp, c))
        _check(bufs, "ct", exp.ct)
        if pbk:
            kexp = oracle.key_schedule_witness(keys, layout=layout)
            for c in ("w", "kx", "ky", "kz", "rk"):
                _check(bufs, c, getattr(kexp, c))


@pytest.mark.parametrize("layout", [ol.DENSE, ol.PACKED])
def test_key_only_matches_oracle(model, oracle, layout):
    rng = np.random.default_rng(77)
    keys = rng.integers(0, 256, (37, 16), dtype=np.uint8)
    tab = np.concatenate(oracle.tables())
    bufs = _run(model, tab, np.zeros((37, 16), np.uint8), keys, 0, 1, layout, 1)
    kexp = oracle.key_schedule_witness(keys, layout=layout)
    for c in ("w", "kx", "ky", "kz", "rk"):
        _check(bufs, c, getattr(kexp, c))


@pytest.mark.parametrize("layout", [ol.DENSE, ol.PACKED, ol.VALUES])
def test_arbitrary_tables(model, layout):
    """mul2/mul3 need not be xtime tables: the table path follows whatever the host passes."""
    rng = np.random.default_rng(5)
    tables = (rng.permutation(256).astype(np.uint8), rng.integers(0, 256, 256, dtype=np.uint8),
              rng.integers(0, 256, 256, dtype=np.uint8))
    o2 = ol.Oracle(tables=tables)
    pt = rng.integers(0, 256, (40, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (40, 16), dtype=np.uint8)
    bufs = _run(model, np.concatenate(tables), pt, keys, 1, 0, layout, 0)
    exp = o2.encrypt_witness(pt, keys, layout=layout)
    for c in "xyz":
        _check(bufs, c, getattr(exp, c))


def test_masks_match_oracle(model, oracle):
    for col in range(3):
        em, km = np.zeros(1360, np.uint8), np.zeros(400, np.uint8)
        model.lane_model_masks(col, _p(em), _p(km))
        assert np.array_equal(em, oracle.assigned_mask(col))
        assert np.array_equal(km, oracle.key_assigned_mask(col))


def test_closed_form_packed_index_matches_the_masks(model, oracle):
    """assemble's arithmetic dense-row -> packed-index map == the prefix count of the oracle's assigned masks."""
    for col in range(3):
        for key, mask in ((0, oracle.assigned_mask(col)), (1, oracle.key_assigned_mask(col))):
            want = np.where(mask.astype(bool), np.cumsum(mask.astype(np.int64)) - 1, -1)
            got = np.array([model.lane_model_packed_index(key, col, r) for r in range(len(mask))])
            assert np.array_equal(got, want), (col, key, np.flatnonzero(got != want)[:8])


def test_values_mask_matches_oracle(model, oracle):
    """The VALUES layout keeps exactly the cells a chip closure computes (y: S-box / mul rows, z: xor rows)."""
    for col in range(3):
        m = np.zeros(1360, np.uint8)
        model.lane_model_values_mask(col, _p(m))
        assert np.array_equal(m.astype(bool), oracle.values_mask(col))
    assert oracle.values_mask(1).sum() == 448 and oracle.values_mask(2).sum() == 608


def test_window_geometry(model):
    """Staging windows: permanent head >= one line, slots cover the unflushed
    bytes, stride = 16 (mod 32) bytes for conflict-free LDS writes."""
    expect = {(0, 0): 496, (0, 1): 496, (0, 2): 496, (1, 0): 496, (1, 1): 400, (1, 2): 368, (2, 1): 368, (2, 2): 368}
    rounds = {(0, 0): 144, (0, 1): 144, (0, 2): 144, (1, 0): 144, (1, 1): 112, (1, 2): 64, (2, 1): 48, (2, 2): 64}
    for (layout, col), nbytes in expect.items():
        out = (C.c_int * 6)()
        model.lane_model_window(layout, col, out)
        perm_r, nslot, perm_end, tail0, raw, total = list(out)
        assert perm_end >= 128
        assert (nslot - 1) * rounds[(layout, col)] >= 112
        assert total == nbytes and total % 32 == 16 and total >= raw
        banks = {(b * total // 4) % 32 for b in range(8)}
        assert len(banks) == 8 and all(v % 4 == 0 for v in banks)


def test_flush_incremental_equals_closed_form(model):
    """The kernel's scheduled flush (descriptor table) == the closed form (flush_piece) for every
    round, lane, piece and number of valid blocks, in all six column geometries."""
    assert model.lane_model_flush_forms_disagree() == 0


def test_flush_stores_every_piece_exactly_once(model):
    """No 16-byte piece of a wave's column range is stored twice or skipped, for 1..16 valid blocks and all eight
    column geometries (a line flushed twice would be invisible in the bytes and cost 1-9 % of the bandwidth)."""
    assert model.lane_model_flush_not_exactly_once() == 0


def test_golden_vectors(model, oracle):
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "slab_vectors.npz")
    tab = np.concatenate(oracle.tables())
    for layout, name in ((ol.DENSE, "dense"), (ol.PACKED, "packed"), (ol.VALUES, "values")):
        bufs = _run(model, tab, g["pt"], g["keys"], 1, 0, layout, 1)
        for c in "xyz":
            _check(bufs, c, g["%s_%s" % (name, c)])
        for c in ("w", "kx", "ky", "kz", "rk"):
            _check(bufs, c, g["%s_%s" % (name, c)])
        bufs = _run(model, tab, g["pt"], g["keys"][4].copy(), 0, 0, layout, 1)
        for c in "xyz":
            _check(bufs, c, g["%s_shared_%s" % (name, c)])


def test_flush_table_is_dealt_for_the_lds_banks(model):
    """build_flush_table() deals each round's lines into quads that collide little in the LDS banks: the modelled extra
    cycles of the flush's ds_read_b128 (four passes of 16 lanes, 64 banks; PMC agrees with the model to 2 %,
    profiles/r02_study/lds_conflicts.md) are far below those of the same lines in address order."""
    import ctypes as C
    for layout, limit in ((ol.PACKED, 0.40), (ol.DENSE, 0.25), (ol.VALUES, 0.55)):
        built, plain = C.c_int(), C.c_int()
        model.lane_model_flush_conflict_costs(layout, C.byref(built), C.byref(plain))
        assert 0 < built.value <= limit * plain.value, (layout, built.value, plain.value)


def test_assemble_geometry_falls_back_to_the_striding_kernel_where_it_cannot_cover_the_grid(model):
    """"assemble_geometry" 1 puts 1 + ceil(2^K / 1360) segments into grid.y (<= 65535), i.e. K <= 26; the entry
    points accept K up to 30 (device) / 28 (host).  The launcher's dispatch (assemble_kernel_choice, shared source) must fall
    back to the striding kernel (0) there instead of failing, as the aligned geometries 2-4 do outside 8 <= K <= 30."""
    f = model.lane_model_assemble_kernel_choice
    for k in range(2, 33):
        segs = 1 + -(-(1 << k) // 1360)
        assert f(1, 1, k, 16) == (1 if segs <= 65535 else 0), k
        for geo in (2, 3, 4):
            assert f(1, geo, k, 16) == (2 if 8 <= k <= 30 else 0), (geo, k)
        assert f(1, 0, k, 16) == 0 and f(0, 1, k, 16) == 0 and f(0, 4, k, 16) == 0  # byte cells: always the striding kernel
    assert f(1, 1, 26, 16) == 1 and f(1, 1, 27, 16) == 0 and f(1, 1, 28, 16) == 0
    assert f(1, 1, 20, 65535) == 1 and f(1, 1, 20, 65536) == 0 and f(1, 4, 20, 65536) == 0 and f(1, 4, 20, 0) == 0
