#!/usr/bin/env python3
"""Launch time vs batch size in one process: t(n) = a + b*n separates the fixed per-launch cost
from the streaming rate."""
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
rows = []
for log2n in (14, 15, 16, 17, 18, 19, 20):
    n = 1 << log2n
    r = bench.Runner(pkg, ctx, torch, n, False, pkg.LAYOUT_PACKED, False, 99)
    steps = max(10, min(200, (1 << 23) // n))
    ms = []
    for _ in range(5):
        w, m, _ = r.run(steps, 3, True)
        ms.append(m * 1e3)
    med = statistics.median(ms)
    rows.append((n, med))
    print("2^%d blocks: %9.2f us  %6.0f GB/s  (%d sets, %d steps)" % (log2n, med, 3040 * n / med / 1e3, r.nsets, steps))
    del r
    torch.cuda.empty_cache()
# least squares a + b n
import numpy as np
n = np.array([r[0] for r in rows], float)
t = np.array([r[1] for r in rows], float)
b, a = np.polyfit(n, t, 1)
print("fit: fixed %.2f us + %.3f us per 2^16 blocks  -> streaming %.0f GB/s" % (a, b * 65536, 3040 / b / 1e3))
