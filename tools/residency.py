#!/usr/bin/env python3
"""Residency sensitivity of the witness kernel: the diagnostic option "lds_pad" lowers the number of one-wave workgroups a CU
holds from 7 to 3; 2^20 blocks with per-block keys and 2^16 blocks with the scheduled key, one process, interleaved rounds,
probed arena (profiles/r03_study/README.md 6)."""
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

for log2n, pbk in ((20, True), (16, False)):
    n = 1 << log2n
    pads = [0, 2300, 6200, 12000, 20000]
    ctxs = []
    for p in pads:
        c = pkg.Context(0)
        c.set_option("lds_pad", p)
        c.set_option("waves_shared", 1)
        c.set_option("waves_pbk", 1)
        ctxs.append(c)
    base = bench.Runner(pkg, ctxs[0], torch, n, pbk, pkg.LAYOUT_PACKED, pbk, 1234, arena=True)
    print("arena:", base.arena_info[:1])
    res = {p: [] for p in pads}
    steps = 50 if log2n <= 17 else 10
    for rnd in range(5):
        for p, c in zip(pads, ctxs):
            if not pbk:
                c.schedule_key(base.keys, layout=pkg.LAYOUT_PACKED, key_slab=False)
            base.h = c._h
            base.ctx = c
            w, ms, _ = base.run(steps, 3, True)
            res[p].append(ms * 1e3)
    for p in pads:
        lds = 768 + 176 + 20224 + p
        med = statistics.median(res[p])
        print("2^%d %s lds_pad %5d (%d B per one-wave workgroup -> %d per CU): %8.2f us -> %.3f" % (
            log2n, "pbk" if pbk else "scheduled key", p, lds, 163840 // lds, med, base.bytes_per_block * n / med / 8e6))
    base.ctx = ctxs[0]
    base.close()
