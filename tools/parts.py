#!/usr/bin/env python3
"""Price the parts of a launch in ONE process with the -DAESW_DIAGNOSTIC build (tools/libaesw_diag.so):
store_mode 2 = the product kernel, 3 = compute + staging without the flush, 4 = flush only (LDS reads + stores),
5 = stores only.  usage: parts.py [c1|c2] [LOG2N] [lds_pad ...]"""
import ctypes as C
import statistics
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
DIAG = ROOT / "tools" / "libaesw_diag.so"
if not DIAG.exists():
    csrc = ROOT / "halo2-aes_amd" / "csrc"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DAESW_DIAGNOSTIC", "-o", str(DIAG),
                           str(csrc / "aesw_kernels.hip"), str(csrc / "aesw_api.cpp"), str(csrc / "aesw_arena.cpp"), str(csrc / "aesw_comm.cpp")])
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
import bench  # noqa: E402
lib = C.CDLL(str(DIAG))
for name, (res, args) in pkg.api.SYMBOLS.items():
    getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pads = [int(x) for x in sys.argv[3:]] or [0]
layout = pkg.LAYOUT_PACKED
pbk = workload == "c2"
ctx = pkg.Context(0)
r = bench.Runner(pkg, ctx, torch, 1 << log2n, pbk, layout, pbk, 11)
t = [x.copy() for x in pkg.reference_tables()]
variants = []
for pad in pads:
    for mode in (2, 3, 4, 5):
        h = C.c_void_p()
        assert lib.aesw_create(C.byref(h), 0, *[x.ctypes.data_as(C.c_void_p) for x in t]) == 0
        assert lib.aesw_set_option(h, b"store_mode", mode) == 0
        assert lib.aesw_set_option(h, b"lds_pad", pad) == 0
        if not pbk:
            assert lib.aesw_schedule_key_device(h, r.keys.data_ptr(), layout, None, None) == 0
        variants.append((pad, mode, h))
torch.cuda.synchronize()
res = {(p, m): [] for p, m, _ in variants}
steps = 50 if log2n <= 17 else 10
for _ in range(5):
    for p, m, h in variants:
        r.lib, r.h = lib, h
        w, ms, _ = r.run(steps, 3, True)
        res[(p, m)].append(ms * 1e3)
names = {2: "product kernel", 3: "compute + staging, no flush", 4: "flush only (LDS reads + stores)", 5: "stores only"}
for (p, m), v in res.items():
    med = statistics.median(v)
    print("%s 2^%d  lds_pad %6d  mode %d %-34s %9.2f us  %6.0f GB/s" % (workload, log2n, p, m, names[m], med, r.bytes_per_block * r.n / med / 1e3))
