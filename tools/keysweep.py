#!/usr/bin/env python3
"""key_kernel alone: waves per group, 2^20 keys, both layouts, one process."""
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
for layout, name in ((pkg.LAYOUT_PACKED, "packed"), (pkg.LAYOUT_DENSE, "dense")):
    ctxs = []
    for w in ((4, 1, 1), (4, 1, 0), (4, 2, 1), (4, 2, 0), (2, 1, 1), (2, 1, 0), (1, 1, 1), (1, 2, 1)):  # (waves per group, store mode, xcd_remap)
        c = pkg.Context(0)
        c.set_option("waves_pbk", w[0])
        c.set_option("key_store_mode", w[1])
        c.set_option("xcd_remap", w[2])
        ctxs.append((w, c))
    res = {w: [] for w, _ in ctxs}
    for _ in range(5):
        for w, c in ctxs:
            c.key_schedule_witness(keys, layout=layout, want_rk=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                kw = c.key_schedule_witness(keys, layout=layout, want_rk=False)
            e1.record()
            torch.cuda.synchronize()
            res[w].append(e0.elapsed_time(e1) / 10 * 1e3)
    out = sum(pkg.key_column_stride(layout, c) for c in range(3)) + 96
    for w, _ in ctxs:
        med = statistics.median(res[w])
        print("%-7s waves,store,xcd %s  %8.1f us  written %6.0f GB/s  algorithmic %6.0f GB/s" % (name, w, med, out * nk / med / 1e3, 952 * nk / med / 1e3))
