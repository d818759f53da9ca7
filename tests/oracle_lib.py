"""ctypes wrapper of oracle/libaesw_oracle.so (the CPU oracle).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from collections import namedtuple
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "libaesw_oracle.so"

AES_ROWS, KEY_ROWS, WORDS_ROWS, TABLE_ROWS = 1360, 400, 96, 66561
DENSE, PACKED, VALUES = 0, 1, 2
ENC_STRIDE = {DENSE: (1360, 1360, 1360), PACKED: (1360, 1056, 608), VALUES: (0, 448, 608)}
KEY_STRIDE = {DENSE: (400, 400, 400), PACKED: (400, 240, 200), VALUES: (400, 240, 200)}

OWitness = namedtuple("OWitness", "x y z ct")
OKeyWitness = namedtuple("OKeyWitness", "w kx ky kz rk")


class Tables(C.Structure):
    _fields_ = [("sbox", C.c_uint8 * 256), ("mul2", C.c_uint8 * 256), ("mul3", C.c_uint8 * 256)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def ensure_built():
    src = [ROOT / "oracle" / "aesw_oracle.c", ROOT / "oracle" / "aesw_oracle.h"]
    if not LIB.exists() or any(s.exists() and s.stat().st_mtime > LIB.stat().st_mtime for s in src):
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "-B", "libaesw_oracle.so"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB


class Oracle:
    def __init__(self, tables=None):
        ensure_built()
        L = self.L = C.CDLL(str(LIB))
        V, I, U64, U32 = C.c_void_p, C.c_int, C.c_uint64, C.c_uint32
        L.aesw_o_encrypt_witness.argtypes = [V, V, V, I, U64, I, V, V, V, V, I]
        L.aesw_o_key_schedule_witness.argtypes = [V, V, U64, I, V, V, V, V, V, I]
        L.aesw_o_lookup_table.argtypes = [V, V, V, V, V]
        L.aesw_o_xor_bytes.argtypes = [U64, U64, C.POINTER(U64)]
        L.aesw_o_sub_byte.argtypes = [V, U64]
        L.aesw_o_sub_byte.restype = C.c_uint8
        L.aesw_o_round_constant.argtypes = [U32]
        L.aesw_o_round_constant.restype = U64
        L.aesw_o_circuit_synthesize.argtypes = [U32, U32, V, V, V, U64, I]
        L.aesw_o_circuit_synthesize.restype = V
        L.aesw_o_key_circuit_synthesize.argtypes = [U32, V, V, I]
        L.aesw_o_key_circuit_synthesize.restype = V
        L.aesw_o_circuit_free.argtypes = [V]
        for name in ("status", "num_advice", "num_selectors"):
            getattr(L, "aesw_o_circuit_" + name).argtypes = [V]
        for name in ("num_rows", "num_regions", "num_copies"):
            f = getattr(L, "aesw_o_circuit_" + name)
            f.argtypes = [V]
            f.restype = U64
        L.aesw_o_circuit_copies.argtypes = [V, V]
        L.aesw_o_circuit_column_height.argtypes = [V, U32]
        L.aesw_o_circuit_column_height.restype = U64
        for name in ("advice", "advice_assigned", "selector"):
            f = getattr(L, "aesw_o_circuit_" + name)
            f.argtypes = [V, U32]
            f.restype = C.POINTER(C.c_uint8)
        L.aesw_o_circuit_fixed.argtypes = [V]
        L.aesw_o_circuit_fixed.restype = C.POINTER(C.c_uint8)
        L.aesw_o_circuit_round_keys.argtypes = [V, V]
        L.aesw_o_circuit_round_key_cells.argtypes = [V, V, V]
        L.aesw_o_circuit_ciphertext.argtypes = [V, U64, V]
        L.aesw_o_circuit_block_placement.argtypes = [V, U64, C.POINTER(U32), C.POINTER(U64)]
        L.aesw_o_circuit_verify.argtypes = [V, C.c_char_p, C.c_size_t]
        L.aesw_o_circuit_poke.argtypes = [V, U32, U64, C.c_uint8]
        self.t = Tables()
        if tables is None:
            L.aesw_o_reference_tables(C.byref(self.t))
        else:
            for name, arr in zip(("sbox", "mul2", "mul3"), tables):
                C.memmove(getattr(self.t, name), np.ascontiguousarray(arr, np.uint8).ctypes.data, 256)

    # ---- tables
    def tables(self):
        return tuple(np.frombuffer(bytes(getattr(self.t, n)), np.uint8).copy() for n in ("sbox", "mul2", "mul3"))

    def fips_tables(self):
        t = Tables()
        self.L.aesw_o_fips_tables(C.byref(t))
        return tuple(np.frombuffer(bytes(getattr(t, n)), np.uint8).copy() for n in ("sbox", "mul2", "mul3"))

    def xor_bytes(self, x: int, y: int) -> int:
        z = C.c_uint64()
        rc = self.L.aesw_o_xor_bytes(x, y, C.byref(z))
        assert rc == 0
        return int(z.value)

    # ---- slab level
    def assigned_mask(self, col: int) -> np.ndarray:
        m = np.zeros(AES_ROWS, np.uint8)
        assert self.L.aesw_o_encrypt_assigned_mask(col, _p(m)) == 0
        return m

    def key_assigned_mask(self, col: int) -> np.ndarray:
        m = np.zeros(KEY_ROWS, np.uint8)
        assert self.L.aesw_o_key_assigned_mask(col, _p(m)) == 0
        return m

    def packed_index(self, col: int) -> np.ndarray:
        idx = np.zeros(AES_ROWS, np.int32)
        assert self.L.aesw_o_encrypt_packed_index(col, _p(idx), None) == 0
        return idx

    def key_packed_index(self, col: int) -> np.ndarray:
        idx = np.zeros(KEY_ROWS, np.int32)
        assert self.L.aesw_o_key_packed_index(col, _p(idx), None) == 0
        return idx

    def values_mask(self, col: int) -> np.ndarray:
        """Block-relative rows whose cell value a chip closure computes (not a copy_advice): y where the
        S-box / mul2 / mul3 selector is enabled, z where the xor selector is; derived from the selectors the
        restated synthesize() enables (src/chips/*.rs), not from the product's index tables."""
        if not hasattr(self, "_vmask"):
            with self.circuit(12, 1, np.zeros(16, np.uint8), np.zeros((1, 16), np.uint8), record_copies=False) as c:
                sel = [c.selector(i)[KEY_ROWS:KEY_ROWS + AES_ROWS] for i in range(5)]  # range, xor, sbox, mul2, mul3
            self._vmask = {0: np.zeros(AES_ROWS, bool), 1: (sel[2] | sel[3] | sel[4]).astype(bool), 2: sel[1].astype(bool)}
        return self._vmask[col]

    def encrypt_witness(self, pt, keys, layout=PACKED, threads=8) -> OWitness:
        pt = np.ascontiguousarray(pt, np.uint8).reshape(-1, 16)
        keys = np.ascontiguousarray(keys, np.uint8)
        n = pt.shape[0]
        if layout == VALUES:
            d = self.encrypt_witness(pt, keys, layout=DENSE, threads=threads)
            cols = [np.ascontiguousarray(getattr(d, c).reshape(n, AES_ROWS)[:, self.values_mask(i)]).reshape(-1)
                    for i, c in enumerate("xyz")]
            return OWitness(cols[0], cols[1], cols[2], d.ct)
        pbk = 0 if keys.size == 16 else 1
        sx, sy, sz = ENC_STRIDE[layout]
        x, y, z = np.zeros(n * sx, np.uint8), np.zeros(n * sy, np.uint8), np.zeros(n * sz, np.uint8)
        ct = np.zeros((n, 16), np.uint8)
        rc = self.L.aesw_o_encrypt_witness(C.byref(self.t), _p(pt), _p(keys), pbk, n, layout, _p(x), _p(y), _p(z), _p(ct),
                                           threads)
        if rc:
            raise RuntimeError("oracle encrypt_witness rc=%d" % rc)
        return OWitness(x, y, z, ct)

    def key_schedule_witness(self, keys, layout=PACKED, threads=8) -> OKeyWitness:
        if layout == VALUES:
            layout = PACKED  # key slabs of the VALUES layout are the packed ones
        keys = np.ascontiguousarray(keys, np.uint8).reshape(-1, 16)
        n = keys.shape[0]
        kxs, kys, kzs = KEY_STRIDE[layout]
        w = np.zeros(n * WORDS_ROWS, np.uint8)
        kx, ky, kz = np.zeros(n * kxs, np.uint8), np.zeros(n * kys, np.uint8), np.zeros(n * kzs, np.uint8)
        rk = np.zeros((n, 176), np.uint8)
        rc = self.L.aesw_o_key_schedule_witness(C.byref(self.t), _p(keys), n, layout, _p(w), _p(kx), _p(ky), _p(kz),
                                                _p(rk), threads)
        if rc:
            raise RuntimeError("oracle key_schedule_witness rc=%d" % rc)
        return OKeyWitness(w, kx, ky, kz, rk)

    def lookup_table(self) -> np.ndarray:
        t = np.zeros((4, TABLE_ROWS), np.uint8)
        assert self.L.aesw_o_lookup_table(C.byref(self.t), *[_p(t[i]) for i in range(4)]) == 0
        return t

    # ---- circuit level
    def circuit(self, k: int, n_sets: int, key, pts, record_copies=True) -> "Circuit":
        key = np.ascontiguousarray(key, np.uint8).reshape(16)
        pts = np.ascontiguousarray(pts, np.uint8).reshape(-1, 16)
        h = self.L.aesw_o_circuit_synthesize(k, n_sets, C.byref(self.t), _p(key), _p(pts), pts.shape[0],
                                             1 if record_copies else 0)
        if not h:
            raise MemoryError("oracle circuit allocation failed")
        return Circuit(self.L, h)

    def key_circuit(self, k: int, key, record_copies=True) -> "Circuit":
        key = np.ascontiguousarray(key, np.uint8).reshape(16)
        h = self.L.aesw_o_key_circuit_synthesize(k, C.byref(self.t), _p(key), 1 if record_copies else 0)
        if not h:
            raise MemoryError("oracle circuit allocation failed")
        return Circuit(self.L, h)


class Circuit:
    def __init__(self, L, h):
        self.L, self.h = L, C.c_void_p(h)

    def close(self):
        if self.h:
            self.L.aesw_o_circuit_free(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def status(self):
        return self.L.aesw_o_circuit_status(self.h)

    @property
    def num_advice(self):
        return self.L.aesw_o_circuit_num_advice(self.h)

    @property
    def num_selectors(self):
        return self.L.aesw_o_circuit_num_selectors(self.h)

    @property
    def num_rows(self):
        return self.L.aesw_o_circuit_num_rows(self.h)

    @property
    def num_regions(self):
        return self.L.aesw_o_circuit_num_regions(self.h)

    @property
    def num_copies(self):
        return self.L.aesw_o_circuit_num_copies(self.h)

    def copies(self) -> np.ndarray:
        """[num_copies, 4] = (source column, source row, copy column, copy row), in copy_advice() call order."""
        out = np.zeros((self.num_copies, 4), np.uint64)
        assert self.L.aesw_o_circuit_copies(self.h, _p(out)) == 0
        return out

    def column_height(self, col):
        return self.L.aesw_o_circuit_column_height(self.h, col)

    def _col(self, fn, idx):
        p = fn(self.h, idx)
        return np.ctypeslib.as_array(p, shape=(self.num_rows,)).copy()

    def advice(self, col):
        return self._col(self.L.aesw_o_circuit_advice, col)

    def advice_assigned(self, col):
        return self._col(self.L.aesw_o_circuit_advice_assigned, col)

    def selector(self, s):
        return self._col(self.L.aesw_o_circuit_selector, s)

    def fixed(self):
        return np.ctypeslib.as_array(self.L.aesw_o_circuit_fixed(self.h), shape=(self.num_rows,)).copy()

    def round_keys(self):
        rk = np.zeros(176, np.uint8)
        assert self.L.aesw_o_circuit_round_keys(self.h, _p(rk)) == 0
        return rk

    def round_key_cells(self):
        col = np.zeros(176, np.uint32)
        row = np.zeros(176, np.uint64)
        assert self.L.aesw_o_circuit_round_key_cells(self.h, _p(col), _p(row)) == 0
        return col, row

    def ciphertext(self, b):
        ct = np.zeros(16, np.uint8)
        assert self.L.aesw_o_circuit_ciphertext(self.h, b, _p(ct)) == 0
        return ct

    def block_placement(self, b):
        s, r = C.c_uint32(), C.c_uint64()
        assert self.L.aesw_o_circuit_block_placement(self.h, b, C.byref(s), C.byref(r)) == 0
        return int(s.value), int(r.value)

    def poke(self, col, row, value):
        assert self.L.aesw_o_circuit_poke(self.h, col, row, value) == 0

    def verify(self):
        buf = C.create_string_buffer(256)
        rc = self.L.aesw_o_circuit_verify(self.h, buf, 256)
        return rc, buf.value.decode()
