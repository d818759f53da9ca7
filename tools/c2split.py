#!/usr/bin/env python3
"""configs[2] cost split in one process: shared key / per-block keys without key slab / with key slab, 2^20 blocks."""
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
ctx = pkg.Context(0)
for o in sys.argv[1:]:
    k, v = o.split("=")
    ctx.set_option(k, int(v))
n = 1 << 20
cases = [("shared key", False, False), ("per-block keys, no key slab", True, False), ("per-block keys + key slab", True, True)]
runners = [(name, bench.Runner(pkg, ctx, torch, n, pbk, pkg.LAYOUT_PACKED, ks, 5)) for name, pbk, ks in cases]
res = {name: [] for name, _ in runners}
for _ in range(5):
    for name, r in runners:
        if not r.pbk:
            ctx.schedule_key(r.keys, layout=pkg.LAYOUT_PACKED, key_slab=False)
        w, ms, _ = r.run(10, 2, True)
        res[name].append(ms * 1e3)
for name, r in runners:
    med = statistics.median(res[name])
    print("%-30s %8.1f us  written %6.0f GB/s" % (name, med, r.out_bytes_per_step / med / 1e3))
