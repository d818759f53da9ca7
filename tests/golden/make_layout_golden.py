#!/usr/bin/env python3
"""Extracts the only layout evidence the reference holds -- the two circuit-layout renderings its own dev-graph
tests wrote (prints/aes128-layout.png, src/aes128.rs:437-456; prints/key-schedule-layout.png,
src/key_schedule.rs:394-412) -- into tests/golden/layout_runs.json: per advice column the run-length encoding of the
pixels halo2's CircuitLayout painted black ("this cell was assigned").  Data only; run here, where /root/reference exists.

What the images turn out to be (measured by this script, checked by tests/test_layout_pngs.py):
  * both are 2048 x 32768 with a 58-pixel title; the 2^K rows map onto the 32710 pixel rows below it;
  * key-schedule-layout.png: 13 columns = 4 advice + 9 fixed/selector, K = 17 (4.007 rows per pixel);
  * aes128-layout.png: 23 columns = 7 advice (N = 2 column sets + words_column) + 16 fixed/selector, K = 19 (16.03 rows
    per pixel), 385 blocks in EACH set -- an older revision of the test than the source now shows (N = 3, K = 20, 1 000
    blocks; and today's aes_callable() reserves 1 760 rows in set 0, which leaves room for 384 blocks there).
"""
import json
from pathlib import Path

import numpy as np
from PIL import Image

Image.MAX_IMAGE_PIXELS = None
REF = Path("/root/reference/prints")
OUT = Path(__file__).resolve().parent / "layout_runs.json"
Y0, H = 58, 32710


def runs(v):
    """[[start, length], ...] of the True runs of a boolean vector."""
    d = np.diff(np.concatenate([[0], v.astype(np.int8), [0]]))
    s, e = np.nonzero(d == 1)[0], np.nonzero(d == -1)[0]
    return [[int(a), int(b - a)] for a, b in zip(s, e)]


def extract(name, n_columns, n_advice, k, fixed_from):
    """advice columns 0..n_advice-1, then the fixed column and the selector columns (the four lookup-table columns in
    between are drawn as one table region, not cell by cell, and carry no information)"""
    a = np.array(Image.open(REF / name).convert("RGB"))
    assert a.shape == (32768, 2048, 3)
    black = (a == 0).all(axis=2)
    cw = 2048 / n_columns
    cols, fixed = [], []
    for c in range(n_columns):
        x = int((c + 0.5) * cw)  # the middle of the column: away from the region borders
        entry = {"x": x, "black_runs": runs(black[Y0:Y0 + H, x])}
        if c < n_advice:
            cols.append(entry)
        elif c >= fixed_from:
            fixed.append(entry)
    return {"image": name, "size": [2048, 32768], "title_rows": Y0, "pixel_rows": H, "k": k, "n_columns": n_columns,
            "n_advice": n_advice, "columns": cols, "fixed_and_selectors": fixed}


def main():
    out = {"note": "black = assigned cell in halo2 dev-graph CircuitLayout; extracted by tests/golden/make_layout_golden.py",
           "key_schedule": extract("key-schedule-layout.png", 13, 4, 17, 8),
           "aes128": extract("aes128-layout.png", 23, 7, 19, 11)}
    OUT.write_text(json.dumps(out, separators=(",", ":")))
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
