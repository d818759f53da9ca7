#!/usr/bin/env python3
"""A/B launch options in ONE process, interleaved rounds (perf deltas across
processes/boxes are noise: cdna guide rule 24).  Prints median/min launch time."""
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

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
workload = sys.argv[2] if len(sys.argv) > 2 else "c1"
layout = {"packed": pkg.LAYOUT_PACKED, "dense": pkg.LAYOUT_DENSE, "values": pkg.LAYOUT_VALUES}[sys.argv[3] if len(sys.argv) > 3 else "packed"]
variants = []
if len(sys.argv) > 4 and sys.argv[4] == "cap":
    for w in (4, 3, 2):
        for cap in (0, 256 * (8 // w), 256 * (8 // w) * 3 // 2, 256 * (8 // w) // 2):
            variants.append({"waves_shared": w, "waves_pbk": w, "grid_cap": cap})
elif len(sys.argv) > 4 and sys.argv[4] == "cap3":
    for w in (3, 1):
        for cap in (0, 256, 512, 768, 1024, 2048):
            variants.append({"waves_shared": w, "waves_pbk": w, "grid_cap": cap * (3 // w)})
elif len(sys.argv) > 4 and sys.argv[4] == "waves":
    for w in (4, 3, 2, 1):
        variants.append({"waves_shared": w, "waves_pbk": w})
elif len(sys.argv) > 4 and sys.argv[4] == "cap4":
    for cap in (0, 128, 192, 224, 256, 288, 320, 384, 512):
        variants.append({"waves_shared": 4, "waves_pbk": 4, "grid_cap": cap})
elif len(sys.argv) > 4 and sys.argv[4] == "store":
    for w in (3, 1):
        for m in (0, 1, 2):
            variants.append({"waves_shared": w, "waves_pbk": w, "store_mode": m})
elif len(sys.argv) > 4 and sys.argv[4] == "xcd":
    for w in (4, 2):
        for xr in (0, 1):
            variants.append({"waves_shared": w, "waves_pbk": w, "xcd_remap": xr})
else:
    for nt in (0, 1):
        for w in (4, 3, 2, 1):
            variants.append({"nt_stores": nt, "waves_shared": w, "waves_pbk": w})
ctxs = []
for v in variants:
    c = pkg.Context(0)
    for k, val in v.items():
        c.set_option(k, val)
    ctxs.append(c)
n = 1 << log2n
pbk = workload == "c2"
# the shared output columns live in a probed arena (round 3: placement decides 10 - 20 % of a launch; on memory chosen by
# measurement the options are compared at the rate the product runs at).  AESW_SWEEP_ARENA=0: plain tensors as in rounds 1 - 2
import os  # noqa: E402
runners = [bench.Runner(pkg, c, torch, n, pbk, layout, pbk, 1234, arena=os.environ.get("AESW_SWEEP_ARENA", "1") != "0") for c in ctxs[:1]]
print("arena:", runners[0].arena_info)
# share one runner's buffers across contexts: only the ctx handle differs
base = runners[0]
results = {i: [] for i in range(len(variants))}
steps = 50 if log2n <= 17 else 10
for rnd in range(7):
    for i, c in enumerate(ctxs):
        if not pbk:
            c.schedule_key(base.keys, layout=layout, key_slab=False)
        base.h = c._h
        base.ctx = c
        w, ms, _ = base.run(steps, 3, True)
        results[i].append(ms * 1e3)
bpb = base.bytes_per_block
for i, v in enumerate(variants):
    med = statistics.median(results[i])
    print("%-55s median %8.2f us  min %8.2f us  -> %6.0f GB/s (median)" % (v, med, min(results[i]), bpb * n / med / 1e3))
