#!/usr/bin/env python3
"""A/B of two builds of the library in ONE process: usage ab_lib.py OTHER.so [c1|c2] [LOG2N] [layout]
(the default build vs OTHER.so, same buffers, interleaved rounds)."""
import ctypes as C
import statistics
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
import bench  # noqa: E402
other = C.CDLL(str(Path(sys.argv[1]).resolve()))  # an older build may lack newer symbols: bind only what this tool calls
for name in ("aesw_create", "aesw_schedule_key_device", "aesw_encrypt_witness_device", "aesw_last_error", "aesw_set_option"):
    res, args = pkg.api.SYMBOLS[name]
    getattr(other, name).restype, getattr(other, name).argtypes = res, args
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
log2n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
layout = {"packed": pkg.LAYOUT_PACKED, "dense": pkg.LAYOUT_DENSE, "values": pkg.LAYOUT_VALUES}[sys.argv[4] if len(sys.argv) > 4 else "packed"]
pbk = workload == "c2"
ctx = pkg.Context(0)
r = bench.Runner(pkg, ctx, torch, 1 << log2n, pbk, layout, pbk, 11)
# a context of the other library, same tables
t = [x.copy() for x in pkg.reference_tables()]
h2 = C.c_void_p()
assert other.aesw_create(C.byref(h2), 0, t[0].ctypes.data_as(C.c_void_p), t[1].ctypes.data_as(C.c_void_p), t[2].ctypes.data_as(C.c_void_p)) == 0
if not pbk:
    assert other.aesw_schedule_key_device(h2, r.keys.data_ptr(), layout, None, None) == 0
    torch.cuda.synchronize()
res = {"default": [], "other": []}
steps = 50 if log2n <= 17 else 10
for _ in range(7):
    for name, lib, h in (("default", pkg.load_library(), ctx._h), ("other", other, h2)):
        r.lib, r.h = lib, h
        w, ms, _ = r.run(steps, 3, True)
        res[name].append(ms * 1e3)
for name in res:
    med = statistics.median(res[name])
    print("%-8s %9.2f us  %6.0f GB/s" % (name, med, r.bytes_per_block * r.n / med / 1e3))
