#!/usr/bin/env python3
"""key_kernel A/B in ONE process: the in-tree library against tools/libaesw_kz8.so (the same source built with -DAESW_KZ_PER_KEY:
packed kz flushed as per-key 8-byte pieces, as up to round 3), 2^20 keys into the same probed key-only arena, interleaved rounds.
Build the variant first:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DAESW_KZ_PER_KEY -o tools/libaesw_kz8.so
halo2-aes_amd/csrc/aesw_kernels.hip halo2-aes_amd/csrc/aesw_api.cpp halo2-aes_amd/csrc/aesw_arena.cpp halo2-aes_amd/csrc/aesw_comm.cpp"""
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
nk = 1 << 20
keys = torch.randint(0, 256, (nk, 16), dtype=torch.uint8, device="cuda")
ctx = pkg.Context(0)
ka = ctx.alloc_columns(nk, pkg.LAYOUT_PACKED, key_slab=True, key_only=True)
print("arena", ctx.last_arena)
other = C.CDLL(str(ROOT / "tools" / "libaesw_kz8.so"))
for name in ("aesw_create", "aesw_key_schedule_witness_device"):
    res, args = pkg.api.SYMBOLS[name]
    getattr(other, name).restype, getattr(other, name).argtypes = res, args
t = [x.copy() for x in pkg.reference_tables()]
h2 = C.c_void_p()
assert other.aesw_create(C.byref(h2), 0, *[x.ctypes.data_as(C.c_void_p) for x in t]) == 0
libs = {"whole-range 16 B kz": (pkg.load_library(), ctx._h), "per-key 8 B kz (r03)": (other, h2)}
k = ka.key
res = {n: [] for n in libs}
for _ in range(7):
    for n, (lib, h) in libs.items():
        def go():
            assert lib.aesw_key_schedule_witness_device(h, keys.data_ptr(), nk, pkg.LAYOUT_PACKED, k.w.data_ptr(), k.kx.data_ptr(), k.ky.data_ptr(),
                                                        k.kz.data_ptr(), None, None) == 0
        go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.default_stream())
        for _ in range(10):
            go()
        e1.record(torch.cuda.default_stream())
        torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) * 100)
for n in libs:
    med = statistics.median(res[n])
    print("%-22s %7.1f us -> %5.0f GB/s algorithmic (%.3f)   min %.1f max %.1f" % (n, med, 952 * nk / med / 1e3, 952 * nk / med / 8e6, min(res[n]), max(res[n])))
