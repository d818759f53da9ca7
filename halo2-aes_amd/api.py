"""ctypes binding of include/aesw.h plus a thin tensor-level wrapper.

PyTorch is plumbing only: device memory (``torch.empty(..., device="cuda")``),
the current HIP stream and ``torch.distributed``.  All compute happens in
``libaesw.so`` (hand-written gfx950 kernels) through the C ABI; if the library
is missing or no gfx950 device is usable this module raises -- there is no
fallback path.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple
from pathlib import Path

import numpy as np

from . import constants as K

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libaesw.so"            # HIP kernels + the C ABI of include/aesw.h
HOST_LIB_PATH = _PKG / "libaesw_host.so"  # C++ mirror of the reference's host interface (include/aesw_host.h), above the C ABI

STATUS = {
    0: "AESW_OK", 1: "AESW_ERR_INVALID_ARG", 2: "AESW_ERR_NO_DEVICE", 3: "AESW_ERR_HIP", 4: "AESW_ERR_NOMEM",
    5: "AESW_ERR_CAPACITY", 6: "AESW_ERR_NO_KEY", 7: "AESW_ERR_MISMATCH", 8: "AESW_ERR_UNSATISFIED", 9: "AESW_ERR_COMM",
}
OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_CAPACITY, ERR_NO_KEY, ERR_MISMATCH = range(8)


class AeswError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        msg = "%s (%d): %s" % (STATUS.get(status, "?"), status, _strerror(status))
        if detail:
            msg += " -- " + detail
        super().__init__(msg)


class StreamStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("chunks", "bytes_to_host", "kernel_ns", "d2h_ns", "consumer_ns", "wait_ns", "wall_ns")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class KeySlab(C.Structure):
    _fields_ = [("w", C.c_void_p), ("kx", C.c_void_p), ("ky", C.c_void_p), ("kz", C.c_void_p)]


class Columns(C.Structure):
    """aesw_columns: one device allocation holding every output column of a batch (aesw_columns_alloc)."""
    _fields_ = [("base", C.c_void_p), ("bytes", C.c_uint64), ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p),
                ("ct", C.c_void_p), ("key", KeySlab), ("candidates", C.c_uint32), ("chosen", C.c_uint32),
                ("probe_us", C.c_float), ("fill_us", C.c_float)]


class Batch(C.Structure):
    """aesw_batch: one batch of aesw_encrypt_witness_batches_device."""
    _fields_ = [("d_pt", C.c_void_p), ("d_keys", C.c_void_p), ("n", C.c_uint64), ("d_x", C.c_void_p), ("d_y", C.c_void_p),
                ("d_z", C.c_void_p), ("d_ct", C.c_void_p), ("d_key_slab", C.POINTER(KeySlab))]


class CheckReport(C.Structure):
    """aesw_check_report: what aesw_check_witness_device found."""
    _fields_ = [("blocks", C.c_uint64), ("keys", C.c_uint64), ("lookup_failures", C.c_uint64), ("copy_failures", C.c_uint64),
                ("gate_failures", C.c_uint64), ("input_failures", C.c_uint64), ("first", C.c_uint64)]


class _DevView:
    """A raw device range as a __cuda_array_interface__ object, so torch can wrap it without owning it."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3, "strides": None}


# every symbol include/aesw.h declares: (restype, argtypes)
_P, _I, _U64, _U32, _I64 = C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_int64
SYMBOLS = {
    "aesw_version": (_I, []),
    "aesw_strerror": (C.c_char_p, [_I]),
    "aesw_last_error": (C.c_char_p, [_P]),
    "aesw_device_count": (_I, [C.POINTER(_I)]),
    "aesw_create": (_I, [C.POINTER(_P), _I, _P, _P, _P]),
    "aesw_destroy": (None, [_P]),
    "aesw_device": (_I, [_P]),
    "aesw_column_stride": (_U32, [_I, _I]),
    "aesw_key_column_stride": (_U32, [_I, _I]),
    "aesw_packed_index": (_I, [_I, _P]),
    "aesw_layout_index": (_I, [_I, _I, _P]),
    "aesw_key_packed_index": (_I, [_I, _P]),
    "aesw_block_placement": (_I, [_U32, _U32, _U64, C.POINTER(_U32), C.POINTER(_U64)]),
    "aesw_block_capacity": (_U64, [_U32, _U32]),
    "aesw_selector_tags": (_I, [_P, _P, _P, _P]),
    "aesw_block_copy_graph": (_I, [_P]),
    "aesw_key_copy_graph": (_I, [_P]),
    "aesw_assemble_selectors": (_I, [_U32, _U32, _U64, _P, _P]),
    "aesw_schedule_key_device": (_I, [_P, _P, _I, C.POINTER(KeySlab), _P]),
    "aesw_schedule_key": (_I, [_P, _P, _I, C.POINTER(KeySlab)]),
    "aesw_encrypt_witness_device": (_I, [_P, _P, _P, _I, _U64, _I, _P, _P, _P, _P, C.POINTER(KeySlab), _P]),
    "aesw_encrypt_witness_batches_device": (_I, [_P, C.POINTER(Batch), _U32, _I, _I, _P]),
    "aesw_key_schedule_witness_device": (_I, [_P, _P, _U64, _I, _P, _P, _P, _P, _P, _P]),
    "aesw_lookup_table_device": (_I, [_P, _P, _P, _P, _P, _P]),
    "aesw_expand_fr_device": (_I, [_P, _P, _U64, _P, _P]),
    "aesw_check_witness_device": (_I, [_P, _P, _P, _I, _U64, _I, _P, _P, _P, _P, C.POINTER(KeySlab), _P, _P]),
    "aesw_check_witness": (_I, [_P, _P, _P, _I, _U64, _I, _P, _P, _P, _P, C.POINTER(KeySlab), C.POINTER(CheckReport)]),
    "aesw_last_stream_check": (_I, [_P, C.POINTER(CheckReport)]),
    "aesw_assemble_advice_device": (_I, [_P, _U32, _U32, _U64, _I, _P, _P, _P, C.POINTER(KeySlab), _I, _P, _P]),
    "aesw_columns_alloc": (_I, [_P, _U64, _I, _I, _I, C.POINTER(Columns)]),
    "aesw_columns_free": (_I, [_P, C.POINTER(Columns)]),
    "aesw_encrypt_witness": (_I, [_P, _P, _P, _I, _U64, _I, _P, _P, _P, _P, C.POINTER(KeySlab)]),
    "aesw_key_schedule_witness": (_I, [_P, _P, _U64, _I, _P, _P, _P, _P, _P]),
    "aesw_encrypt_witness_stream": (_I, [_P, _P, _P, _I, _U64, _I, _P, _P]),
    "aesw_lookup_table": (_I, [_P, _P, _P, _P, _P]),
    "aesw_last_stream_stats": (_I, [_P, _P]),
    "aesw_assemble_advice_host": (_I, [_P, _U32, _U32, _U64, _I, _P, _P, _P, C.POINTER(KeySlab), _I, _P]),
    "aesw_host_register": (_I, [_P, C.c_size_t]),
    "aesw_host_unregister": (_I, [_P]),
    "aesw_assemble_advice_stream": (_I, [_P, _U32, _U32, _U64, _I, _P, _P, _P, C.POINTER(KeySlab), _I, _P, _P]),
    "aesw_host_alloc": (_P, [C.c_size_t]),
    "aesw_host_free": (None, [_P]),
    "aesw_comm_unique_id": (_I, [_P]),
    "aesw_comm_create": (_I, [_P, _I, _I, _P, C.POINTER(_P)]),
    "aesw_comm_destroy": (None, [_P]),
    "aesw_comm_set_max_message": (_I, [_P, _U64]),
    "aesw_comm_last_error": (C.c_char_p, []),
    "aesw_gather_offsets": (_I, [_I, _P, _P, C.POINTER(_U64)]),
    "aesw_gather_columns_device": (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    "aesw_set_option": (_I, [_P, C.c_char_p, _I64]),
    "aesw_get_option": (_I, [_P, C.c_char_p, C.POINTER(_I64)]),
    "aesw_uses_xtime_path": (_I, [_P]),
}

# include/aesw_host.h: the host-side mirror of the reference's circuits
HOST_SYMBOLS = {
    "aesw_host_aes_circuit_run": (_I, [_P, _U32, _U32, _P, _P, _U64, _I, _I, _I, C.POINTER(_P)]),
    "aesw_host_circuit_copies": (_I, [_P, _P]),
    "aesw_host_aes_circuit_columns": (_I, [_P, _U32, _U32, _P, _P, _U64, C.POINTER(_P)]),
    "aesw_host_key_circuit_run": (_I, [_P, _U32, _P, C.POINTER(_P)]),
    "aesw_host_circuit_free": (None, [_P]),
    "aesw_host_circuit_verify": (_I, [_P, C.c_char_p, C.c_size_t]),
    "aesw_host_circuit_num_advice": (_U32, [_P]),
    "aesw_host_circuit_num_selectors": (_U32, [_P]),
    "aesw_host_circuit_num_rows": (_U64, [_P]),
    "aesw_host_circuit_num_regions": (_U64, [_P]),
    "aesw_host_circuit_num_copies": (_U64, [_P]),
    "aesw_host_circuit_closure_calls": (_U64, [_P]),
    "aesw_host_circuit_advice": (C.POINTER(C.c_uint8), [_P, _U32]),
    "aesw_host_circuit_advice_assigned": (C.POINTER(C.c_uint8), [_P, _U32]),
    "aesw_host_circuit_selector": (C.POINTER(C.c_uint8), [_P, _U32]),
    "aesw_host_circuit_fixed": (C.POINTER(C.c_uint8), [_P]),
    "aesw_host_circuit_table": (C.POINTER(C.c_uint8), [_P, _U32, C.POINTER(_U64)]),
    "aesw_host_circuit_ciphertext": (_I, [_P, _U64, _P]),
    "aesw_host_circuit_poke": (_I, [_P, _U32, _U64, C.c_uint8]),
    "aesw_host_last_error": (C.c_char_p, []),
}

_lib = None
_host_lib = None


def load_host_library(path: Path | None = None) -> C.CDLL:
    """Load libaesw_host.so (in-tree): the host-side mirror.  It links against libaesw.so and holds no device code."""
    global _host_lib
    if _host_lib is not None and path is None:
        return _host_lib
    load_library()  # libaesw.so first: the mirror's NEEDED entry resolves to the copy already mapped
    p = Path(path) if path else HOST_LIB_PATH
    if not p.exists():
        raise FileNotFoundError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    lib = C.CDLL(str(p))
    for name, (res, args) in HOST_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _host_lib = lib
    return lib


def load_library(path: Path | None = None) -> C.CDLL:
    """Load libaesw.so (in-tree).  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    # torch ships its own libamdhip64.so.7 and dlopen()s it by path.  Import it
    # FIRST so libaesw.so's NEEDED libamdhip64.so.7 binds to that copy: two HIP
    # runtimes in one process do not share the device (hipGetDeviceCount fails
    # in the second one).  A host without torch (the Rust caller) uses /opt/rocm's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not p.exists():
        raise FileNotFoundError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback implementation." % p)
    lib = C.CDLL(str(p))  # RTLD_LOCAL: two builds of the library can sit in one process (tools/ab_lib.py)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _strerror(status: int) -> str:
    try:
        return load_library().aesw_strerror(status).decode()
    except Exception:  # library not built yet
        return ""


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ---- pure-host geometry -----------------------------------------------------------

def column_stride(layout: int, col: int) -> int:
    return int(load_library().aesw_column_stride(layout, col))


def key_column_stride(layout: int, col: int) -> int:
    return int(load_library().aesw_key_column_stride(layout, col))


def packed_index(col: int) -> np.ndarray:
    idx = np.zeros(K.AES_ROWS, dtype=np.int32)
    rc = load_library().aesw_packed_index(col, _np_ptr(idx))
    if rc:
        raise AeswError(rc)
    return idx


def layout_index(layout: int, col: int) -> np.ndarray:
    """dense row -> index in column `col` of `layout` (-1: the layout leaves the cell out)."""
    idx = np.zeros(K.AES_ROWS, dtype=np.int32)
    rc = load_library().aesw_layout_index(layout, col, _np_ptr(idx))
    if rc:
        raise AeswError(rc)
    return idx


def key_packed_index(col: int) -> np.ndarray:
    idx = np.zeros(K.KEY_ROWS, dtype=np.int32)
    rc = load_library().aesw_key_packed_index(col, _np_ptr(idx))
    if rc:
        raise AeswError(rc)
    return idx


def block_placement(k: int, n_sets: int, b: int):
    """(set, first row) of the b-th encrypt() call; raises AeswError(CAPACITY) where the reference panics."""
    s, r = C.c_uint32(), C.c_uint64()
    rc = load_library().aesw_block_placement(k, n_sets, b, C.byref(s), C.byref(r))
    if rc:
        raise AeswError(rc)
    return int(s.value), int(r.value)


def block_capacity(k: int, n_sets: int) -> int:
    return int(load_library().aesw_block_capacity(k, n_sets))


def selector_tags():
    """(enc_tag[1360], key_tag[400], q_eq_rcon[96], rcon_fixed[96]): fixed selector data for keygen."""
    e, k = np.zeros(K.AES_ROWS, np.uint8), np.zeros(K.KEY_ROWS, np.uint8)
    q, c = np.zeros(K.WORDS_ROWS, np.uint8), np.zeros(K.WORDS_ROWS, np.uint8)
    rc = load_library().aesw_selector_tags(_np_ptr(e), _np_ptr(k), _np_ptr(q), _np_ptr(c))
    if rc:
        raise AeswError(rc)
    return e, k, q, c


COPY_EDGE = np.dtype([("dst_space", np.uint8), ("dst_col", np.uint8), ("dst_row", np.uint16),
                      ("src_space", np.uint8), ("src_col", np.uint8), ("src_row", np.uint16)])


def block_copy_graph() -> np.ndarray:
    """The 1 952 copy_advice() edges of one encrypt() call (structured array, see aesw_copy_edge)."""
    e = np.zeros(1952, dtype=COPY_EDGE)
    rc = load_library().aesw_block_copy_graph(_np_ptr(e))
    if rc:
        raise AeswError(rc)
    return e


def key_copy_graph() -> np.ndarray:
    """The 640 copy_advice() edges of schedule_keys()."""
    e = np.zeros(640, dtype=COPY_EDGE)
    rc = load_library().aesw_key_copy_graph(_np_ptr(e))
    if rc:
        raise AeswError(rc)
    return e


def assemble_selectors(k: int, n_sets: int, n_blocks: int):
    """(selectors[(5*n_sets+1), 2^k], fixed[2^k]) of a whole circuit, as keygen lays them out."""
    sel = np.zeros((5 * n_sets + 1, 1 << k), np.uint8)
    fixed = np.zeros(1 << k, np.uint8)
    rc = load_library().aesw_assemble_selectors(k, n_sets, n_blocks, _np_ptr(sel), _np_ptr(fixed))
    if rc:
        raise AeswError(rc)
    return sel, fixed


def device_count() -> int:
    n = C.c_int(0)
    load_library().aesw_device_count(C.byref(n))
    return int(n.value)


def host_alloc(nbytes: int) -> np.ndarray:
    """Page-locked uint8 host buffer (aesw_host_alloc): D2H lands in it without a bounce copy."""
    lib = load_library()
    p = lib.aesw_host_alloc(nbytes)
    if not p:
        raise MemoryError("aesw_host_alloc(%d) failed" % nbytes)
    buf = (C.c_uint8 * nbytes).from_address(p)
    arr = np.frombuffer(buf, dtype=np.uint8)
    _pinned[arr.ctypes.data] = p
    return arr


def host_register(arr: np.ndarray):
    """Page-lock memory the caller already owns (hipHostRegister); pair with host_unregister."""
    rc = load_library().aesw_host_register(_np_ptr(arr), arr.nbytes)
    if rc:
        raise AeswError(rc, "aesw_host_register")


def host_unregister(arr: np.ndarray):
    rc = load_library().aesw_host_unregister(_np_ptr(arr))
    if rc:
        raise AeswError(rc, "aesw_host_unregister")


def host_free(arr: np.ndarray):
    p = _pinned.pop(arr.ctypes.data, None)
    if p:
        load_library().aesw_host_free(p)


_pinned = {}

Witness = namedtuple("Witness", "x y z ct key")
KeyWitness = namedtuple("KeyWitness", "w kx ky kz rk")


class ArenaWitness(Witness):
    """A Witness whose tensors are views of an arena of aesw_columns_alloc; `.columns` is the aesw_columns handle it is
    freed by (Context.free_columns), so slicing or replacing a member tensor cannot orphan the arena."""


class Context:
    """One aesw_ctx: a device plus the host's three byte tables."""

    def __init__(self, device: int = 0, tables=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        sbox, mul2, mul3 = tables if tables is not None else K.reference_tables()
        self._tables = tuple(np.ascontiguousarray(t, dtype=np.uint8) for t in (sbox, mul2, mul3))
        for t in self._tables:
            if t.shape != (256,):
                raise ValueError("tables must be three uint8[256] arrays")
        rc = self._lib.aesw_create(C.byref(self._h), device, *[_np_ptr(t) for t in self._tables])
        if rc:
            self._h = C.c_void_p()
            raise AeswError(rc, "aesw_create(device=%d)" % device)
        self.device = device
        self._arenas = {}  # y pointer -> Columns of alloc_columns()

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            for _group, comm in list(getattr(self, "_sharding_comms", {}).values()):  # communicators sharding.gather_columns made for this context
                comm.close()
            self.__dict__.pop("_sharding_comms", None)
            for cols in list(getattr(self, "_arenas", {}).values()):
                self._lib.aesw_columns_free(self._h, C.byref(cols))
            self._arenas = {}
            self._lib.aesw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, what: str = ""):
        if rc:
            raise AeswError(rc, (what + " " + self._lib.aesw_last_error(self._h).decode()).strip())

    # -- options
    def set_option(self, name: str, value: int):
        self._check(self._lib.aesw_set_option(self._h, name.encode(), int(value)), "set_option(%s)" % name)

    def get_option(self, name: str) -> int:
        v = C.c_int64()
        self._check(self._lib.aesw_get_option(self._h, name.encode(), C.byref(v)), "get_option(%s)" % name)
        return int(v.value)

    @property
    def uses_xtime_path(self) -> bool:
        return bool(self._lib.aesw_uses_xtime_path(self._h))

    # -- tensor plumbing
    def _torch(self):
        import torch
        return torch

    def _dev(self):
        return self._torch().device("cuda", self.device)

    def _stream(self):
        return C.c_void_p(self._torch().cuda.current_stream(self.device).cuda_stream)

    def _u8(self, t, what):
        torch = self._torch()
        if not isinstance(t, torch.Tensor) or t.dtype != torch.uint8 or not t.is_cuda or t.device.index != self.device:
            raise TypeError("%s must be a uint8 tensor on cuda:%d" % (what, self.device))
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % what)
        return t

    def alloc_witness(self, n: int, layout: int = K.LAYOUT_PACKED, want_ct: bool = False, key_slab: bool = False,
                      n_keys: int | None = None):
        torch = self._torch()
        dev = self._dev()
        cols = [torch.empty(n * column_stride(layout, c), dtype=torch.uint8, device=dev) for c in range(3)]
        ct = torch.empty((n, 16), dtype=torch.uint8, device=dev) if want_ct else None
        key = None
        if key_slab:
            m = n if n_keys is None else n_keys
            key = KeyWitness(torch.empty(m * K.WORDS_ROWS, dtype=torch.uint8, device=dev),
                             *[torch.empty(m * key_column_stride(layout, c), dtype=torch.uint8, device=dev)
                               for c in range(3)], None)
        return Witness(cols[0], cols[1], cols[2], ct, key)

    def alloc_columns(self, n: int, layout: int = K.LAYOUT_PACKED, want_ct: bool = False, key_slab: bool = False, key_only: bool = False):
        """alloc_witness through the C ABI's arena (aesw_columns_alloc): every column of the batch placed together, each on
        a 2 MiB boundary.  From 2^16 blocks on the physical backing is chosen by measurement (option "arena_probe"; a shape
        this context has placed and freed before comes out of its placement cache without a search, last_arena["candidates"]
        == 0); with "arena_probe" 0 it is one hipMalloc with columns on 2^"arena_align_log2"-byte boundaries (that option is
        ignored on the probed path).  The returned ArenaWitness's tensors are views of the arena, which lives until
        free_columns(witness) or the Context is closed."""
        torch = self._torch()
        cols = Columns()
        if key_only:
            key_slab = True
        self._check(self._lib.aesw_columns_alloc(self._h, n, layout, 2 if key_only else (1 if key_slab else 0), 1 if want_ct else 0, C.byref(cols)),
                    "aesw_columns_alloc")
        dev = self._dev()

        def view(ptr, nbytes):
            if not ptr or not nbytes:
                return torch.empty(0, dtype=torch.uint8, device=dev)
            return torch.as_tensor(_DevView(ptr, nbytes), device=dev)

        x, y, z = (view(getattr(cols, c), n * column_stride(layout, i) if not key_only else 0) for i, c in enumerate("xyz"))
        ct = view(cols.ct, n * 16).view(n, 16) if want_ct else None
        key = None
        if key_slab:
            key = KeyWitness(view(cols.key.w, n * K.WORDS_ROWS), *[view(getattr(cols.key, c), n * key_column_stride(layout, i))
                                                                   for i, c in enumerate(("kx", "ky", "kz"))], None)
        wit = ArenaWitness(x, y, z, ct, key)
        wit.columns = cols
        self._arenas[id(cols)] = cols
        self.last_arena = {"candidates": int(cols.candidates), "chosen": int(cols.chosen), "probe_us": float(cols.probe_us),
                           "fill_us": float(cols.fill_us), "bytes": int(cols.bytes)}
        return wit

    def free_columns(self, wit) -> None:
        """Release the arena behind a Witness from alloc_columns (its tensors must not be used afterwards)."""
        cols = getattr(wit, "columns", None)
        if cols is None or self._arenas.pop(id(cols), None) is None:
            raise ValueError("not a witness of alloc_columns (or already freed)")
        wit.columns = None
        self._check(self._lib.aesw_columns_free(self._h, C.byref(cols)), "aesw_columns_free")

    # -- device entry points
    def schedule_key(self, key, layout: int = K.LAYOUT_PACKED, key_slab: bool = True):
        """FixedAes128Config::schedule_key (src/aes128.rs:143-152): expand `key`
        (uint8[16] on the device) once; later encrypt_witness(pt, None) calls use it.
        Returns its key-schedule witness (KeyWitness, rk=None) when key_slab."""
        torch = self._torch()
        key = self._u8(key, "key")
        if key.numel() != 16:
            raise ValueError("key must be uint8[16]")
        out = ks = None
        if key_slab:
            dev = self._dev()
            out = KeyWitness(torch.empty(K.WORDS_ROWS, dtype=torch.uint8, device=dev),
                             *[torch.empty(key_column_stride(layout, c), dtype=torch.uint8, device=dev) for c in range(3)],
                             None)
            ks = KeySlab(*[t.data_ptr() for t in out[:4]])
        rc = self._lib.aesw_schedule_key_device(self._h, key.data_ptr(), layout, C.byref(ks) if ks is not None else None,
                                                self._stream())
        self._check(rc, "aesw_schedule_key_device")
        return out

    def encrypt_witness(self, pt, keys, layout: int = K.LAYOUT_PACKED, out: Witness | None = None,
                        want_ct: bool = False, key_slab: bool = False) -> Witness:
        """Batched FixedAes128Config::encrypt witness (src/aes128.rs:154-265).

        pt: uint8[n,16] on the device.  keys: None (the key given to
        schedule_key(): the reference's schedule_key once + encrypt n times),
        uint8[16] (one shared key expanded inside the call) or uint8[n,16]
        (per-block keys).  Asynchronous on torch's current stream.
        """
        pt = self._u8(pt, "pt")
        if pt.dim() != 2 or pt.shape[1] != 16:
            raise ValueError("pt must be [n,16]")
        n = pt.shape[0]
        if keys is None:
            pbk = 0
            if key_slab:
                raise ValueError("the key slab of a scheduled key is returned by schedule_key()")
        else:
            keys = self._u8(keys, "keys")
        if keys is None:
            pass
        elif keys.numel() == 16 and keys.dim() == 1:
            pbk = 0
        elif keys.dim() == 2 and tuple(keys.shape) == (n, 16):
            pbk = 1
        else:
            raise ValueError("keys must be [16] (shared) or [n,16] (per block)")
        if out is None:
            out = self.alloc_witness(n, layout, want_ct, key_slab, n_keys=n if pbk else 1)
        for c, name in enumerate("xyz"):
            self._u8(out[c], name)
            if out[c].numel() < n * column_stride(layout, c):
                raise ValueError("column %s too small" % name)
        ks = None
        if out.key is not None:
            ks = KeySlab(*[t.data_ptr() if t is not None else None for t in out.key[:4]])
        rc = self._lib.aesw_encrypt_witness_device(
            self._h, pt.data_ptr(), keys.data_ptr() if keys is not None else None, pbk, n, layout,
            out.x.data_ptr() if out.x.numel() else None, out.y.data_ptr(),
            out.z.data_ptr(), out.ct.data_ptr() if out.ct is not None else None,
            C.byref(ks) if ks is not None else None, self._stream())
        self._check(rc, "aesw_encrypt_witness_device")
        return out

    def encrypt_witness_batches(self, batches, per_block_keys: bool, layout: int = K.LAYOUT_PACKED):
        """aesw_encrypt_witness_batches_device: `batches` is a list of (pt, keys, out) with pt uint8[n,16], keys None /
        uint8[16] / uint8[n,16] (all batches in the same key mode) and out a Witness whose columns hold n blocks (its .ct and
        .key are written when present).  The batches are dealt round-robin onto the context's internal streams behind torch's
        current stream and joined back into it: ramp and tail of one launch overlap its neighbours."""
        arr = (Batch * len(batches))()
        keep = []
        for i, (pt, keys, out) in enumerate(batches):
            pt = self._u8(pt, "pt")
            n = pt.shape[0]
            if pt.dim() != 2 or pt.shape[1] != 16:
                raise ValueError("pt must be [n,16]")
            if keys is not None:
                keys = self._u8(keys, "keys")
                if keys.numel() != (n * 16 if per_block_keys else 16):
                    raise ValueError("keys must hold 16 bytes, or n*16 with per_block_keys")
            elif per_block_keys:
                raise ValueError("per_block_keys needs keys")
            for c, name in enumerate("xyz"):
                if out[c].numel() < n * column_stride(layout, c):
                    raise ValueError("column %s of batch %d too small" % (name, i))
            ks = None
            if out.key is not None:
                ks = KeySlab(*[t.data_ptr() if t is not None else None for t in out.key[:4]])
                keep.append(ks)
            arr[i] = Batch(pt.data_ptr(), keys.data_ptr() if keys is not None else None, n,
                           out.x.data_ptr() if out.x.numel() else None, out.y.data_ptr(), out.z.data_ptr(),
                           out.ct.data_ptr() if out.ct is not None else None, C.pointer(ks) if ks is not None else None)
            keep.append((pt, keys))
        rc = self._lib.aesw_encrypt_witness_batches_device(self._h, arr, len(batches), 1 if per_block_keys else 0, layout, self._stream())
        self._check(rc, "aesw_encrypt_witness_batches_device")

    def key_schedule_witness(self, keys, layout: int = K.LAYOUT_PACKED, want_rk: bool = True, out: KeyWitness | None = None) -> KeyWitness:
        """Aes128KeyScheduleConfig::schedule_keys witness for n keys (src/key_schedule.rs:80-224); into `out`
        (e.g. alloc_columns(n, layout, key_only=True).key) when given."""
        torch = self._torch()
        keys = self._u8(keys, "keys")
        if keys.dim() == 1:
            keys = keys.reshape(1, 16)
        if keys.dim() != 2 or keys.shape[1] != 16:
            raise ValueError("keys must be [n,16]")
        n = keys.shape[0]
        dev = self._dev()
        if out is not None:
            w, kx, ky, kz = out[:4]
            for t, need in ((w, n * K.WORDS_ROWS), (kx, n * key_column_stride(layout, 0)), (ky, n * key_column_stride(layout, 1)),
                            (kz, n * key_column_stride(layout, 2))):
                if self._u8(t, "out").numel() < need:
                    raise ValueError("out column too small")
        else:
            w = torch.empty(n * K.WORDS_ROWS, dtype=torch.uint8, device=dev)
            kx, ky, kz = [torch.empty(n * key_column_stride(layout, c), dtype=torch.uint8, device=dev) for c in range(3)]
        rk = torch.empty((n, 176), dtype=torch.uint8, device=dev) if want_rk else None
        rc = self._lib.aesw_key_schedule_witness_device(
            self._h, keys.data_ptr(), n, layout, w.data_ptr(), kx.data_ptr(), ky.data_ptr(), kz.data_ptr(),
            rk.data_ptr() if rk is not None else None, self._stream())
        self._check(rc, "aesw_key_schedule_witness_device")
        return KeyWitness(w, kx, ky, kz, rk)

    def lookup_table(self):
        """load_enc_full_table (src/table.rs:18-192): four uint8[66561] columns on the device."""
        torch = self._torch()
        t = torch.empty((4, K.TABLE_ROWS), dtype=torch.uint8, device=self._dev())
        rc = self._lib.aesw_lookup_table_device(self._h, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                                t[3].data_ptr(), self._stream())
        self._check(rc, "aesw_lookup_table_device")
        return t

    def expand_fr(self, cells, out=None):
        """uint8 cells -> [n,32] uint8 bn256::Fr Montgomery cells (Fp::from(u64), src/utils.rs:23)."""
        torch = self._torch()
        cells = self._u8(cells, "cells")
        n = cells.numel()
        if out is None:
            out = torch.empty((n, 32), dtype=torch.uint8, device=self._dev())
        rc = self._lib.aesw_expand_fr_device(self._h, cells.data_ptr(), n, out.data_ptr(), self._stream())
        self._check(rc, "aesw_expand_fr_device")
        return out

    def check_witness(self, pt, keys, witness: Witness, key_witness: KeyWitness, layout: int = K.LAYOUT_PACKED, ct=None, sync: bool = True):
        """MockProver::assert_satisfied over a batch on the device (aesw_check_witness_device; src/aes128.rs:409-418): every
        enabled lookup, every copy_advice() pair, the round-constant gate and the literal rows of n blocks and their key slab(s).
        keys: None, uint8[16] or uint8[n,16] (per-block keys: key_witness then holds n key slabs).  Returns a dict (counts,
        `satisfied`, `first` = None or (unit, is_key_slab, kind, index)) after synchronising the stream; with sync=False the
        uint64[7] device tensor the report was written to."""
        torch = self._torch()
        pt = self._u8(pt, "pt")
        n = pt.shape[0]
        pbk = keys is not None and self._u8(keys, "keys").numel() != 16
        ks = KeySlab(*[t.data_ptr() for t in key_witness[:4]])
        rep = torch.empty(7, dtype=torch.int64, device=self._dev())
        rc = self._lib.aesw_check_witness_device(
            self._h, pt.data_ptr(), keys.data_ptr() if keys is not None else None, 1 if pbk else 0, n, layout, witness.x.data_ptr(),
            witness.y.data_ptr(), witness.z.data_ptr(), ct.data_ptr() if ct is not None else None, C.byref(ks), rep.data_ptr(), self._stream())
        self._check(rc, "aesw_check_witness_device")
        if not sync:
            return rep
        torch.cuda.current_stream().synchronize()
        v = [int(x) & 0xFFFFFFFFFFFFFFFF for x in rep.cpu().tolist()]
        first = None if v[6] == 0xFFFFFFFFFFFFFFFF else (v[6] >> 20, bool((v[6] >> 19) & 1), (v[6] >> 16) & 7, v[6] & 0xFFFF)
        return {"blocks": v[0], "keys": v[1], "lookup_failures": v[2], "copy_failures": v[3], "gate_failures": v[4], "input_failures": v[5],
                "first": first, "satisfied": not any(v[2:6])}

    def last_stream_check(self):
        """aesw_last_stream_check: what option "stream_check" found over the chunks of the last encrypt_witness_stream call."""
        rep = CheckReport()
        self._check(self._lib.aesw_last_stream_check(self._h, C.byref(rep)), "aesw_last_stream_check")
        f = int(rep.first)
        out = {k_: int(getattr(rep, k_)) for k_ in ("blocks", "keys", "lookup_failures", "copy_failures", "gate_failures", "input_failures")}
        out.update(first=None if f == 0xFFFFFFFFFFFFFFFF else (f >> 20, bool((f >> 19) & 1), (f >> 16) & 7, f & 0xFFFF),
                   satisfied=not any(out[k_] for k_ in ("lookup_failures", "copy_failures", "gate_failures", "input_failures")))
        return out

    def check_witness_host(self, pt, keys, cols, key_cols, layout: int = K.LAYOUT_PACKED, ct=None):
        """aesw_check_witness: the same check for a witness in HOST memory (numpy uint8 arrays: cols = (x, y, z), key_cols =
        (w, kx, ky, kz)); uploaded and checked in stages of "chunk_blocks" blocks."""
        import numpy as np
        pt = np.ascontiguousarray(pt, np.uint8).reshape(-1, 16)
        n = pt.shape[0]
        keys = None if keys is None else np.ascontiguousarray(keys, np.uint8)
        pbk = keys is not None and keys.size != 16
        keep = [np.ascontiguousarray(a, np.uint8) for a in (*cols, *key_cols)] + ([np.ascontiguousarray(ct, np.uint8)] if ct is not None else [])
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        ks = KeySlab(*[a.ctypes.data for a in keep[3:7]])
        rep = CheckReport()
        rc = self._lib.aesw_check_witness(self._h, p(pt), p(keys) if keys is not None else None, 1 if pbk else 0, n, layout, p(keep[0]), p(keep[1]),
                                          p(keep[2]), p(keep[7]) if ct is not None else None, C.byref(ks), C.byref(rep))
        self._check(rc, "aesw_check_witness")
        f = int(rep.first)
        first = None if f == 0xFFFFFFFFFFFFFFFF else (f >> 20, bool((f >> 19) & 1), (f >> 16) & 7, f & 0xFFFF)
        out = {k_: int(getattr(rep, k_)) for k_ in ("blocks", "keys", "lookup_failures", "copy_failures", "gate_failures", "input_failures")}
        out.update(first=first, satisfied=not any(out[k_] for k_ in ("lookup_failures", "copy_failures", "gate_failures", "input_failures")))
        return out

    def assemble_advice(self, k: int, n_sets: int, witness: Witness, key_witness: KeyWitness | None, n_blocks: int,
                        layout: int = K.LAYOUT_PACKED, as_fr: bool = False, out=None):
        """All advice columns of a FixedAes128Config<K, n_sets> circuit as the prover holds them:
        [(3*n_sets+1), 2^k] bytes, or [(3*n_sets+1), 2^k, 32] Fr cells with as_fr (into `out` when given)."""
        torch = self._torch()
        ncol = 3 * n_sets + 1
        shape = (ncol, 1 << k, 32) if as_fr else (ncol, 1 << k)
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=self._dev())
        elif tuple(out.shape) != shape:
            raise ValueError("out must have shape %r" % (shape,))
        self._u8(out, "out")
        ks = KeySlab(*[t.data_ptr() for t in key_witness[:4]]) if key_witness is not None else None
        rc = self._lib.aesw_assemble_advice_device(
            self._h, k, n_sets, n_blocks, layout, witness.x.data_ptr(), witness.y.data_ptr(), witness.z.data_ptr(),
            C.byref(ks) if ks is not None else None, 1 if as_fr else 0, out.data_ptr(), self._stream())
        self._check(rc, "aesw_assemble_advice_device")
        return out

    # -- host entry points (numpy in, numpy out)
    def encrypt_witness_host(self, pt: np.ndarray, keys: np.ndarray, layout: int = K.LAYOUT_PACKED,
                             want_ct: bool = False, key_slab: bool = False, out_cols=None):
        pt = np.ascontiguousarray(pt, dtype=np.uint8).reshape(-1, 16)
        n = pt.shape[0]
        if keys is None:  # the key given to schedule_key()
            pbk = 0
            if key_slab:
                raise ValueError("the key slab of a scheduled key is returned by schedule_key()")
        else:
            keys = np.ascontiguousarray(keys, dtype=np.uint8)
            pbk = 0 if keys.size == 16 else 1
            if pbk and keys.size != n * 16:
                raise ValueError("keys must hold 16 or n*16 bytes")
        cols = list(out_cols) if out_cols is not None else [np.empty(n * column_stride(layout, c), dtype=np.uint8)
                                                            for c in range(3)]
        ct = np.empty((n, 16), dtype=np.uint8) if want_ct else None
        key = ks = None
        if key_slab:
            m = n if pbk else 1
            key = KeyWitness(np.empty(m * K.WORDS_ROWS, np.uint8),
                             *[np.empty(m * key_column_stride(layout, c), np.uint8) for c in range(3)], None)
            ks = KeySlab(*[a.ctypes.data for a in key[:4]])
        rc = self._lib.aesw_encrypt_witness(self._h, _np_ptr(pt), _np_ptr(keys) if keys is not None else None, pbk, n, layout,
                                            *[_np_ptr(c) for c in cols],
                                            _np_ptr(ct) if ct is not None else None, C.byref(ks) if ks is not None else None)
        self._check(rc, "aesw_encrypt_witness")
        return Witness(cols[0], cols[1], cols[2], ct, key)

    def encrypt_witness_stream(self, pt: np.ndarray, keys, consume, layout: int = K.LAYOUT_PACKED):
        """aesw_encrypt_witness_stream: consume(first_block, n_blocks, x, y, z) is called per chunk with
        numpy views of page-locked buffers (valid only during the call) while the next chunk is in flight."""
        pt = np.ascontiguousarray(pt, dtype=np.uint8).reshape(-1, 16)
        n = pt.shape[0]
        pbk = 0
        if keys is not None:
            keys = np.ascontiguousarray(keys, dtype=np.uint8)
            pbk = 0 if keys.size == 16 else 1
        strides = [column_stride(layout, c) for c in range(3)]
        err = []

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8))
        def cb(_user, first, count, x, y, z):
            try:
                cols = [np.ctypeslib.as_array(p, shape=(count * s,)) if s else np.empty(0, np.uint8)  # VALUES: no x
                        for p, s in zip((x, y, z), strides)]
                r = consume(int(first), int(count), *cols)
                return int(r or 0)
            except Exception as e:  # never let an exception cross the C boundary
                err.append(e)
                return 1

        rc = self._lib.aesw_encrypt_witness_stream(self._h, _np_ptr(pt), _np_ptr(keys) if keys is not None else None, pbk, n,
                                                   layout, C.cast(cb, C.c_void_p), None)
        if err:
            raise err[0]
        self._check(rc, "aesw_encrypt_witness_stream")

    def last_stream_stats(self) -> dict:
        """Where the time of the last streaming call went (aesw_last_stream_stats)."""
        st = StreamStats()
        self._check(self._lib.aesw_last_stream_stats(self._h, C.byref(st)), "aesw_last_stream_stats")
        return st.as_dict()

    def assemble_advice_stream(self, k: int, n_sets: int, witness: "Witness", key_witness, n_blocks: int, consume,
                               layout: int = K.LAYOUT_PACKED, as_fr: bool = True):
        """aesw_assemble_advice_stream: consume(column, cells) per advice column with a numpy view of the page-locked
        buffer (2^k bytes, or [2^k, 32] Fr cells), valid only during the call, while the next column is in flight."""
        self._torch().cuda.current_stream(self.device).synchronize()  # the slabs must be complete
        err = []

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint64)
        def cb(_user, col, cells, n_cells):
            try:
                a = np.ctypeslib.as_array(cells, shape=(n_cells * (32 if as_fr else 1),))
                r = consume(int(col), a.reshape(n_cells, 32) if as_fr else a)
                return int(r or 0)
            except Exception as e:  # never let an exception cross the C boundary
                err.append(e)
                return 1

        ks = KeySlab(*[t.data_ptr() for t in key_witness[:4]]) if key_witness is not None else None
        rc = self._lib.aesw_assemble_advice_stream(
            self._h, k, n_sets, n_blocks, layout, witness.x.data_ptr(), witness.y.data_ptr(), witness.z.data_ptr(),
            C.byref(ks) if ks is not None else None, 1 if as_fr else 0, C.cast(cb, C.c_void_p), None)
        if err:
            raise err[0]
        self._check(rc, "aesw_assemble_advice_stream")

    def assemble_advice_host(self, k: int, n_sets: int, witness: "Witness", key_witness, n_blocks: int, out: np.ndarray,
                             layout: int = K.LAYOUT_PACKED, as_fr: bool = True):
        """aesw_assemble_advice_host: all advice columns into the host array `out` ((3*n_sets+1) << k cells of 1 or 32 bytes);
        direct DMA when `out` is page-locked (api.host_alloc / host_register)."""
        self._torch().cuda.current_stream(self.device).synchronize()  # the slabs must be complete
        need = ((3 * n_sets + 1) << k) * (32 if as_fr else 1)
        if out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or out.size < need:
            raise ValueError("out must be a contiguous uint8 array of at least %d bytes" % need)
        ks = KeySlab(*[t.data_ptr() for t in key_witness[:4]]) if key_witness is not None else None
        rc = self._lib.aesw_assemble_advice_host(
            self._h, k, n_sets, n_blocks, layout, witness.x.data_ptr(), witness.y.data_ptr(), witness.z.data_ptr(),
            C.byref(ks) if ks is not None else None, 1 if as_fr else 0, _np_ptr(out))
        self._check(rc, "aesw_assemble_advice_host")
        return out

    def key_schedule_witness_host(self, keys: np.ndarray, layout: int = K.LAYOUT_PACKED) -> KeyWitness:
        keys = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1, 16)
        n = keys.shape[0]
        w = np.empty(n * K.WORDS_ROWS, np.uint8)
        kx, ky, kz = [np.empty(n * key_column_stride(layout, c), np.uint8) for c in range(3)]
        rk = np.empty((n, 176), np.uint8)
        rc = self._lib.aesw_key_schedule_witness(self._h, _np_ptr(keys), n, layout, _np_ptr(w), _np_ptr(kx), _np_ptr(ky),
                                                 _np_ptr(kz), _np_ptr(rk))
        self._check(rc, "aesw_key_schedule_witness")
        return KeyWitness(w, kx, ky, kz, rk)

    def lookup_table_host(self) -> np.ndarray:
        t = np.empty((4, K.TABLE_ROWS), dtype=np.uint8)
        rc = self._lib.aesw_lookup_table(self._h, *[_np_ptr(t[i]) for i in range(4)])
        self._check(rc, "aesw_lookup_table")
        return t


class Comm:
    """aesw_comm: the RCCL communicator of the C ABI (one process per GPU).  `unique_id()` on one rank, carried to
    the others by the host's own means, then Comm(ctx, nranks, rank, id) on every rank (collective)."""

    def __init__(self, ctx: "Context", nranks: int, rank: int, uid: bytes | None = None):
        self._lib = ctx._lib
        self._h = C.c_void_p()
        self.nranks, self.rank, self.ctx = nranks, rank, ctx
        buf = (C.c_uint8 * 128).from_buffer_copy(uid) if uid is not None else None
        rc = self._lib.aesw_comm_create(ctx._h, nranks, rank, buf, C.byref(self._h))
        if rc:
            raise AeswError(rc, self._lib.aesw_comm_last_error().decode())

    @staticmethod
    def unique_id() -> bytes:
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.aesw_comm_unique_id(buf)
        if rc:
            raise AeswError(rc, lib.aesw_comm_last_error().decode())
        return bytes(buf)

    def close(self):
        if self._h:
            self._lib.aesw_comm_destroy(self._h)
            self._h = C.c_void_p()

    def set_max_message(self, nbytes: int):
        rc = self._lib.aesw_comm_set_max_message(self._h, nbytes)
        if rc:
            raise AeswError(rc)

    def gather_columns(self, columns, counts, strides, root: int = 0, out=None):
        """aesw_gather_columns_device on torch's current stream.  columns: this rank's flat uint8 device tensors;
        returns the gathered tensors on `root` (allocated unless `out` is given), None elsewhere."""
        import torch
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        strides_a = np.ascontiguousarray(strides, dtype=np.uint32)
        if counts.size != self.nranks or len(columns) != strides_a.size:
            raise ValueError("counts/strides do not match ranks / columns")
        total = int(counts.sum())
        n = len(columns)
        send = (C.c_void_p * n)(*[c.data_ptr() if c.numel() else None for c in columns])
        recv = None
        if self.rank == root:
            if out is None:
                out = [torch.empty(total * int(s), dtype=torch.uint8, device=c.device) for c, s in zip(columns, strides_a)]
            recv = (C.c_void_p * n)(*[o.data_ptr() if o.numel() else None for o in out])
        stream = C.c_void_p(torch.cuda.current_stream(self.ctx.device).cuda_stream)
        rc = self._lib.aesw_gather_columns_device(self._h, root, n, send, recv, _np_ptr(counts), _np_ptr(strides_a), stream)
        if rc:
            raise AeswError(rc, self._lib.aesw_comm_last_error().decode())
        return out if self.rank == root else None


class HostCircuit:
    """What the C++ host mirror's synthesize() assigned (include/aesw_host.h):
    MockProver::run of the reference's circuits with device-fed value closures."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, C.c_void_p(handle)

    @staticmethod
    def _host():
        return load_host_library()

    @classmethod
    def aes(cls, ctx: "Context", k: int, n_sets: int, key, pts, with_witnesses: bool = True,
            skip_schedule_key: bool = False, bulk_assign: bool = False, values_only: bool = False,
            streaming: bool = False, dense: bool = False) -> "HostCircuit":
        """load_enc_full_table, schedule_key(key), encrypt(pts[b]) for every block: TestAesCircuit /
        Aes128BenchCircuit (src/aes128.rs:376-407, benches/aes128.rs:30-61).  The device witness travels in the
        PACKED layout (assigned cells only) unless values_only / streaming (VALUES) or dense is asked for."""
        key = np.ascontiguousarray(key, np.uint8).reshape(16)
        pts = np.ascontiguousarray(pts, np.uint8).reshape(-1, 16)
        h = C.c_void_p()
        lib = cls._host()
        mode = 3 if streaming else (2 if values_only else (1 if bulk_assign else (4 if dense else 0)))
        rc = lib.aesw_host_aes_circuit_run(ctx._h, k, n_sets, _np_ptr(key), _np_ptr(pts), pts.shape[0],
                                           1 if with_witnesses else 0, 1 if skip_schedule_key else 0, mode, C.byref(h))
        if rc:
            raise AeswError(rc, lib.aesw_host_last_error().decode())
        return cls(lib, h.value)

    @classmethod
    def aes_columns(cls, ctx: "Context", k: int, n_sets: int, key, pts) -> "HostCircuit":
        """The same circuit without running a region: whole advice columns from the device witness, selectors, fixed
        column, table and equality constraints from the library's keygen data (aesw_host_aes_circuit_columns)."""
        key = np.ascontiguousarray(key, np.uint8).reshape(16)
        pts = np.ascontiguousarray(pts, np.uint8).reshape(-1, 16)
        h = C.c_void_p()
        lib = cls._host()
        rc = lib.aesw_host_aes_circuit_columns(ctx._h, k, n_sets, _np_ptr(key), _np_ptr(pts), pts.shape[0], C.byref(h))
        if rc:
            raise AeswError(rc, lib.aesw_host_last_error().decode())
        return cls(lib, h.value)

    @classmethod
    def key_schedule(cls, ctx: "Context", k: int, key) -> "HostCircuit":
        """key_schedule.rs TestCircuit (src/key_schedule.rs:245-320)."""
        key = np.ascontiguousarray(key, np.uint8).reshape(16)
        h = C.c_void_p()
        lib = cls._host()
        rc = lib.aesw_host_key_circuit_run(ctx._h, k, _np_ptr(key), C.byref(h))
        if rc:
            raise AeswError(rc, lib.aesw_host_last_error().decode())
        return cls(lib, h.value)

    def close(self):
        if self._h:
            self._lib.aesw_host_circuit_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def verify(self):
        """mock.assert_satisfied(): (0, "") or (AESW_ERR_UNSATISFIED, first failure)."""
        buf = C.create_string_buffer(256)
        rc = self._lib.aesw_host_circuit_verify(self._h, buf, 256)
        return rc, buf.value.decode()

    num_advice = property(lambda self: self._lib.aesw_host_circuit_num_advice(self._h))
    num_selectors = property(lambda self: self._lib.aesw_host_circuit_num_selectors(self._h))
    num_rows = property(lambda self: self._lib.aesw_host_circuit_num_rows(self._h))
    num_regions = property(lambda self: self._lib.aesw_host_circuit_num_regions(self._h))
    num_copies = property(lambda self: self._lib.aesw_host_circuit_num_copies(self._h))
    closure_calls = property(lambda self: self._lib.aesw_host_circuit_closure_calls(self._h))

    def _arr(self, p, n):
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def advice(self, col):
        return self._arr(self._lib.aesw_host_circuit_advice(self._h, col), self.num_rows)

    def advice_assigned(self, col):
        return self._arr(self._lib.aesw_host_circuit_advice_assigned(self._h, col), self.num_rows)

    def selector(self, s):
        return self._arr(self._lib.aesw_host_circuit_selector(self._h, s), self.num_rows)

    def fixed(self):
        return self._arr(self._lib.aesw_host_circuit_fixed(self._h), self.num_rows)

    def table(self, col):
        n = C.c_uint64()
        p = self._lib.aesw_host_circuit_table(self._h, col, C.byref(n))
        return self._arr(p, n.value)

    def copies(self) -> np.ndarray:
        """[num_copies, 4] = (copy column, copy row, original column, original row)."""
        out = np.zeros((self.num_copies, 4), np.uint64)
        rc = self._lib.aesw_host_circuit_copies(self._h, _np_ptr(out))
        if rc:
            raise AeswError(rc)
        return out

    def ciphertext(self, b):
        ct = np.zeros(16, np.uint8)
        rc = self._lib.aesw_host_circuit_ciphertext(self._h, b, _np_ptr(ct))
        if rc:
            raise AeswError(rc)
        return ct

    def poke(self, col, row, value):
        rc = self._lib.aesw_host_circuit_poke(self._h, col, row, value)
        if rc:
            raise AeswError(rc)
