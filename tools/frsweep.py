#!/usr/bin/env python3
"""Geometry (striding + LDS LUT / one-shot 4 KiB / one-shot 16 KiB) and store flavour (plain / nt / sc1) of expand_fr, one process.
(tools/asm_ab.py: the assemble kernel against another build of the library.)"""
import statistics
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
ctxs = []
for geo in (0, 1, 2):
    for m in (0, 1, 2):
        c = pkg.Context(0)
        c.set_option("fr_store_mode", m)
        c.set_option("fr_geometry", geo)
        ctxs.append(((geo, m), c))
cells = torch.randint(0, 256, (1 << 26,), dtype=torch.uint8, device="cuda")
outs = [torch.empty((1 << 26, 32), dtype=torch.uint8, device="cuda") for _ in range(2)]
res = {m: [] for m, _ in ctxs}
for rnd in range(5):
    for m, c in ctxs:
        c.expand_fr(cells, outs[0]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(6):
            c.expand_fr(cells, outs[i & 1])
        e1.record(); torch.cuda.synchronize()
        res[m].append(e0.elapsed_time(e1) / 6 * 1e3)
for m, _ in ctxs:
    med = statistics.median(res[m])
    print("expand_fr geometry %d store_mode %d: %8.1f us  %6.0f GB/s written" % (m[0], m[1], med, (1 << 31) / med / 1e3))

