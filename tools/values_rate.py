#!/usr/bin/env python3
"""Device rate of the three layouts in one process (shared key, scheduled): 2^16 and 2^20 blocks."""
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
for log2n in (16, 20):
    n = 1 << log2n
    runners = [(name, bench.Runner(pkg, ctx, torch, n, False, lay, False, 5)) for name, lay in
               (("packed", pkg.LAYOUT_PACKED), ("dense", pkg.LAYOUT_DENSE), ("values", pkg.LAYOUT_VALUES))]
    res = {name: [] for name, _ in runners}
    for _ in range(5):
        for name, r in runners:
            ctx.schedule_key(r.keys, layout=r.layout, key_slab=False)
            w, ms, _ = r.run(50 if log2n <= 16 else 10, 3, True)
            res[name].append(ms * 1e3)
    for name, r in runners:
        med = statistics.median(res[name])
        print("2^%d %-7s %8.2f us  %.3e blocks/s  written %6.0f GB/s" % (log2n, name, med, n / med * 1e6, r.out_bytes_per_step / med / 1e3))
    del runners
    torch.cuda.empty_cache()
