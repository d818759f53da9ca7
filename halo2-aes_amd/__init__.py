"""halo2_aes_amd -- MI355X-native batched witness generator for the
tkmct/halo2-aes AES-128 gadget.

The directory is called ``halo2-aes_amd`` (not a Python identifier); import it
through ``__graft_entry__.load_package()`` / ``tests/conftest.py``, which
register it as the module ``halo2_aes_amd``.

Layout:
  csrc/        HIP kernels (gfx950) + the C ABI of include/aesw.h
  host/        C++ mirror of the reference's interface (FixedAes128Config, chips,
               Aes128KeyScheduleConfig, load_enc_full_table, MockProver) above the C ABI
  api.py       ctypes binding of the C ABI, tensor plumbing (torch)
  sharding.py  one-process-per-GPU block sharding and the optional RCCL gather
  constants.py the host's byte tables (src/constant.rs) and row constants
  _build.py    hipcc / gcc recipes for the in-tree .so files
"""
from . import constants
from .constants import (AES_ROWS, KEY_ROWS, KEY_SCHEDULE_ROWS, LAYOUT_DENSE, LAYOUT_PACKED, LAYOUT_VALUES, TABLE_ROWS, WORDS_ROWS,
                        fips_tables, reference_tables)
from .api import (AeswError, Comm, Context, HostCircuit, assemble_selectors, KeyWitness, Witness, block_capacity, block_copy_graph, block_placement,
                  column_stride, key_copy_graph,
                  device_count, key_column_stride, key_packed_index, layout_index, load_library, packed_index, selector_tags)
from . import sharding

__all__ = [
    "constants", "AES_ROWS", "KEY_ROWS", "KEY_SCHEDULE_ROWS", "LAYOUT_DENSE", "LAYOUT_PACKED", "LAYOUT_VALUES", "TABLE_ROWS",
    "WORDS_ROWS", "fips_tables", "reference_tables", "AeswError", "Comm", "Context", "HostCircuit", "assemble_selectors", "KeyWitness", "Witness",
    "block_capacity", "block_copy_graph", "block_placement", "column_stride", "key_copy_graph", "device_count", "key_column_stride", "key_packed_index",
    "layout_index", "load_library", "packed_index", "selector_tags", "sharding",
]
