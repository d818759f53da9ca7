"""Host-side constants of the AES-128 gadget.

These are the values the host hands to ``aesw_create`` -- the device never
bakes them in.  ``reference_tables()`` reproduces the reference's
``src/constant.rs:1-47`` *including* its ``S_BOX[255] == 23`` (FIPS-197 has 22):
the gadget's lookup table (``src/table.rs:75``) and its witness path
(``src/utils.rs:22-24``) share that constant, so bit-exact parity with the
reference means using it.  ``fips_tables()`` is the corrected table for the day
upstream fixes the typo.  tests/test_oracle_pins.py checks
the generated tables against the text of constant.rs (fixture in tests/golden).
"""
from __future__ import annotations

import functools

import numpy as np

AES_ROWS = 1360            # src/constant.rs:114
KEY_SCHEDULE_ROWS = 1760   # src/constant.rs:113
KEY_ROWS = 400
WORDS_ROWS = 96
TABLE_ROWS = 66561
ROUND_CONSTANT = (1, 2, 4, 8, 16, 32, 64, 128, 27, 54)  # src/utils.rs:28

LAYOUT_DENSE = 0
LAYOUT_PACKED = 1
LAYOUT_VALUES = 2   # only closure-computed cells: y of S-box/mul rows (448 B/block), z of xor rows (608 B/block)


def _xtime(a: int) -> int:
    return ((a << 1) ^ (0x1B if a & 0x80 else 0)) & 0xFF


def _gf_mul(a: int, b: int) -> int:
    r = 0
    while b:
        if b & 1:
            r ^= a
        a = _xtime(a)
        b >>= 1
    return r


@functools.lru_cache(maxsize=None)
def _sbox_fips_bytes() -> bytes:
    """The GF(2^8) inverse search takes ~25 - 45 ms in Python: done once per process."""
    return _sbox_fips_compute().tobytes()


def _sbox_fips() -> np.ndarray:
    return np.frombuffer(_sbox_fips_bytes(), dtype=np.uint8).copy()


def _sbox_fips_compute() -> np.ndarray:
    inv = [0] * 256
    for x in range(1, 256):
        for y in range(1, 256):
            if _gf_mul(x, y) == 1:
                inv[x] = y
                break
    out = np.zeros(256, dtype=np.uint8)
    for x in range(256):
        s = r = inv[x]
        for _ in range(4):
            r = ((r << 1) | (r >> 7)) & 0xFF
            s ^= r
        out[x] = s ^ 0x63
    return out


def fips_tables():
    """(sbox, mul2, mul3) per FIPS-197."""
    sbox = _sbox_fips()
    mul2 = np.array([_xtime(i) for i in range(256)], dtype=np.uint8)
    mul3 = np.array([_xtime(i) ^ i for i in range(256)], dtype=np.uint8)
    return sbox, mul2, mul3


def reference_tables():
    """(sbox, mul2, mul3) exactly as src/constant.rs:1-47 has them."""
    sbox, mul2, mul3 = fips_tables()
    sbox = sbox.copy()
    sbox[255] = 23  # src/constant.rs:14
    return sbox, mul2, mul3
