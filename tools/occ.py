#!/usr/bin/env python3
"""Residency study: waves per group x extra LDS (lds_pad lowers the groups resident per CU).
One process, interleaved rounds.  usage: occ.py LOG2N [c1|c2] [packed|dense]"""
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

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
workload = sys.argv[2] if len(sys.argv) > 2 else "c1"
layout = {"packed": pkg.LAYOUT_PACKED, "dense": pkg.LAYOUT_DENSE, "values": pkg.LAYOUT_VALUES}[sys.argv[3] if len(sys.argv) > 3 else "packed"]
variants = []
for w in (4, 2, 1):
    for pad_kb in (0, 8, 16, 24, 40):
        variants.append({"waves_shared": w, "waves_pbk": w, "lds_pad": pad_kb * 1024})
ctxs = []
for v in variants:
    c = pkg.Context(0)
    for k, val in v.items():
        c.set_option(k, val)
    ctxs.append(c)
n = 1 << log2n
pbk = workload == "c2"
base = bench.Runner(pkg, ctxs[0], torch, n, pbk, layout, pbk, 1234)
results = {i: [] for i in range(len(variants))}
steps = 50 if log2n <= 17 else 10
for rnd in range(5):
    for i, c in enumerate(ctxs):
        if not pbk:
            c.schedule_key(base.keys, layout=layout, key_slab=False)
        base.h = c._h
        base.ctx = c
        try:
            w, ms, _ = base.run(steps, 3, True)
            results[i].append(ms * 1e3)
        except Exception as e:
            results[i].append(float("nan"))
bpb = base.bytes_per_block
for i, v in enumerate(variants):
    med = statistics.median(results[i])
    print("%-60s median %8.2f us  min %8.2f us  -> %6.0f GB/s" % (v, med, min(results[i]), bpb * n / med / 1e3), flush=True)
