"""The reference's own layout renderings (prints/*.png, written by its dev-graph tests) against the slab map.

tests/golden/layout_runs.json holds, per advice column, the pixels halo2's CircuitLayout painted black (= a cell the
circuit assigned).  The oracle's assigned-cell masks, placed as synthesize() places them and rasterised like the
renderer does (row r covers pixel rows floor(r*H/2^K) .. floor((r+1)*H/2^K)), must reproduce them: 400 / 240 / 200
assigned key cells in rows 0..399 of set 0, 96 rows of words_column, blocks of exactly 1360 rows, and -- through the
aliasing of the 16-row and 20-row unassigned stretches of y and z against the 16-rows-per-pixel grid -- the order of the
rows INSIDE a slab.  This is the one pin of the row order the reference itself provides (everything else is derived)."""
import json
from pathlib import Path

import numpy as np

import oracle_lib as ol

G = json.loads((Path(__file__).resolve().parent / "golden" / "layout_runs.json").read_text())


def _image(col, H):
    v = np.zeros(H, bool)
    for s, n in col["black_runs"]:
        v[s:s + n] = True
    return v


def _raster(mask, k, H):
    rows = 1 << k
    r = np.nonzero(mask)[0].astype(np.int64)
    p0, p1 = r * H // rows, (r + 1) * H // rows
    d = np.zeros(H + 2, np.int64)
    np.add.at(d, p0, 1)
    np.add.at(d, p1 + 1, -1)
    return np.cumsum(d)[:H] > 0


def _diff(mask, k, col, H, skip_tail=24):
    # the last pixels of the drawing area carry the frame of the plot
    return int((_raster(mask, k, H)[:H - skip_tail] != _image(col, H)[:H - skip_tail]).sum())


def test_key_schedule_png(oracle):
    g = G["key_schedule"]
    H, k = g["pixel_rows"], g["k"]
    assert (g["n_columns"], g["n_advice"], k) == (13, 4, 17)      # 3 advice + words_column; 4 table, 1 fixed, 4 selectors
    for c in range(3):
        m = np.zeros(1 << k, bool)
        m[:400] = oracle.key_assigned_mask(c)
        assert _diff(m, k, g["columns"][c], H) <= 2, "key slab column %d" % c
    w = np.zeros(1 << k, bool)
    w[:96] = True
    assert _diff(w, k, g["columns"][3], H) <= 3                   # 96 rows of words_column at 4 rows per pixel
    wrong = np.zeros(1 << k, bool)
    wrong[:400] = np.roll(oracle.key_assigned_mask(2), 8)         # a z column with its rows elsewhere in the round
    assert _diff(wrong, k, g["columns"][2], H) > 15


def _circuit_masks(oracle, k, n_blocks_per_set):
    em = [oracle.assigned_mask(c).astype(bool) for c in range(3)]
    km = [oracle.key_assigned_mask(c).astype(bool) for c in range(3)]
    out = {}
    for s, nb in enumerate(n_blocks_per_set):
        for c in range(3):
            m = np.zeros(1 << k, bool)
            base = 0
            if s == 0:
                m[:400] = km[c]
                base = 400                                         # blocks of set 0 start right behind the key rows
            for b in range(nb):
                m[base + 1360 * b:base + 1360 * (b + 1)] = em[c]
            out[(s, c)] = m
    return out


def test_aes128_png(oracle):
    g = G["aes128"]
    H, k = g["pixel_rows"], g["k"]
    assert (g["n_columns"], g["n_advice"], k) == (23, 7, 19)      # N = 2: 6 + words_column; 4 table, 1 fixed, 11 selectors
    masks = _circuit_masks(oracle, k, (385, 385))
    for (s, c), m in masks.items():
        d = _diff(m, k, g["columns"][3 * s + c], H)
        # x and y: pixel-exact up to the frame; z has 1 727 runs per column, of which a handful of pixel boundaries
        # fall on the renderer's floating-point rounding
        assert d <= (16 if c == 2 else 2), "set %d column %d: %d pixels differ" % (s, c, d)
    w = np.zeros(1 << k, bool)
    w[:96] = True
    assert _diff(w, k, g["columns"][6], H) <= 1
    # the picture is sensitive to the row order inside a slab: move z's rows by one lcon() record and the
    # aliasing pattern no longer matches; a block stride other than 1360 loses the match entirely
    z = oracle.assigned_mask(2).astype(bool)
    m = np.zeros(1 << k, bool)
    for b in range(385):
        m[400 + 1360 * b:400 + 1360 * (b + 1)] = np.roll(z, 7)
    m[:400] = oracle.key_assigned_mask(2)
    assert _diff(m, k, g["columns"][2], H) > 400
    m = np.zeros(1 << k, bool)
    for b in range(385):
        m[1352 * b:1352 * b + 1352] = z[:1352]                    # set 1 with blocks 1352 rows apart
    assert _diff(m, k, g["columns"][5], H) > 400


def test_blocks_per_set_of_the_rendered_revision():
    """385 blocks x 1360 rows fill each set of the K = 19 rendering to within one block: AES_ROWS = 1360 (src/constant.rs:114)."""
    g = G["aes128"]
    H = g["pixel_rows"]
    last = max(s + n for s, n in g["columns"][3]["black_runs"] if s + n < H - 24)   # x column of set 1
    rows = last * (1 << g["k"]) / H
    assert abs(rows - 385 * 1360) <= 2 * (1 << g["k"]) / H


def _slab_selectors(oracle):
    """Per slab row, which chip's selector the restated synthesize() enables: [range, xor, sbox, mul2, mul3] for the
    1360 rows of a block and the 400 key rows, q_eq_rcon and the fixed column for the 96 rows of words_column."""
    with oracle.circuit(12, 1, np.zeros(16, np.uint8), np.zeros((1, 16), np.uint8), record_copies=False) as c:
        sel = [c.selector(i) for i in range(c.num_selectors)]
        fixed = c.fixed()
    enc = [s[400:400 + 1360].astype(bool) for s in sel[:5]]
    key = [s[:400].astype(bool) for s in sel[:5]]
    return enc, key, sel[5][:96].astype(bool), fixed[:96] != 0


def test_aes128_png_selector_columns(oracle):
    """The eleven selector columns and the fixed column of the rendering: [range, xor, sbox, mul2, mul3] per column set
    (configure()'s order, src/aes128.rs:63-68), then q_eq_rcon (src/key_schedule.rs:50).  A selector cell is black where
    the selector is enabled, so this pins WHICH chip sits on the rows of a slab at the picture's resolution (16 rows per
    pixel): the S-box, xor and range rows of every round and how many mul-by-2 / mul-by-3 rows each stretch of lcon() records
    holds (src/aes128.rs:228-248, :268-301)."""
    g = G["aes128"]
    H, k = g["pixel_rows"], g["k"]
    cols = g["fixed_and_selectors"]
    assert len(cols) == 12
    enc, key, q, fx = _slab_selectors(oracle)

    def placed(s, t, block=None):
        m = np.zeros(1 << k, bool)
        base = 400 if s == 0 else 0
        if s == 0:
            m[:400] = key[t]
        e = enc[t] if block is None else block
        for b in range(385):
            m[base + 1360 * b:base + 1360 * (b + 1)] = e
        return m

    w = np.zeros(1 << k, bool)
    w[:96] = fx
    assert _diff(w, k, cols[0], H) <= 1                                   # round constants in the fixed column
    for s in range(2):
        for t in range(5):
            d = _diff(placed(s, t), k, cols[1 + 5 * s + t], H)
            assert d <= 12, "set %d selector %d: %d pixels differ" % (s, t, d)
    w[:96] = q
    assert _diff(w, k, cols[11], H) <= 1
    # sensitivity: mul2 and mul3 exchanged breaks ~860 pixels of each column.  (What 16 rows per pixel can NOT tell apart
    # is the order of the lcon() records inside a round: word-major vs matrix-row-major moves the mul rows by < 4 rows.)
    assert _diff(placed(0, 3, enc[4]), k, cols[4], H) > 400 and _diff(placed(0, 4, enc[3]), k, cols[5], H) > 400


def test_key_schedule_png_selector_columns(oracle):
    g = G["key_schedule"]
    H, k = g["pixel_rows"], g["k"]
    cols = g["fixed_and_selectors"]
    assert len(cols) == 5                                                # fixed, range, xor, sbox, q_eq_rcon
    _, key, q, fx = _slab_selectors(oracle)
    for col, mask in zip(cols, (fx, key[0], key[1], key[2], q)):
        m = np.zeros(1 << k, bool)
        m[:len(mask)] = mask
        assert _diff(m, k, col, H) <= 3
