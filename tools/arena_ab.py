#!/usr/bin/env python3
"""the headline workload (2^20 blocks, per-block keys + key witness, packed) with its output columns
  tensors   as separate torch tensors (what bench.py did until round 3)
  malloc    in one hipMalloc arena (aesw_columns_alloc, arena_probe 0)
  set       in a probed arena, candidates = whole sets   (arena_unit 0)
  column    in a probed arena, candidates = single columns, greedy (arena_unit 1)
  auto      probed arena, the default: whole sets first, columns when no set candidate is fast (arena_unit 2)
ONE variant per process (who allocates first gets different memory: variants in one process are not comparable); run the
four back to back on one lease: tools/arena_ab.sh.  bench.py's own Runner (hipGraph of `steps` launches, HIP events).
usage: arena_ab.py VARIANT [LOG2N] [c2|c1] [arena_probe]      (c1 = one scheduled key, no key witness: BASELINE configs[1])"""
import statistics
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
import bench  # noqa: E402

variant = sys.argv[1]
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << lg
ctx = pkg.Context(0)
pbk = not (len(sys.argv) > 3 and sys.argv[3] == "c1")
force = int(sys.argv[4]) if len(sys.argv) > 4 else -1
L = pkg.LAYOUT_PACKED
t0 = time.perf_counter()
if variant == "tensors":
    r = bench.Runner(pkg, ctx, torch, n, pbk, L, pbk, 11, arena=False)
else:
    ctx.set_option("arena_probe", 0 if variant == "malloc" else force)
    ctx.set_option("arena_unit", {"column": 1, "set": 0}.get(variant, 2))  # "auto": sets first, columns if no set is fast
    r = bench.Runner(pkg, ctx, torch, n, pbk, L, pbk, 11, arena=True)
torch.cuda.synchronize()
setup = time.perf_counter() - t0
v = []
for rnd in range(5):
    w, ms, _ = r.run(20 if lg >= 19 else 100, 3, True)
    v.append(ms * 1e3)
med = statistics.median(v)
info = " ".join("[%d cand, probe %.0f / fill %.0f us]" % (a["candidates"], a["probe_us"], a["fill_us"]) for a in r.arena_info if a["candidates"])
print("%-8s median %8.2f us  min %8.2f  max %8.2f  -> %6.0f GB/s  (%.3f of 8 TB/s)  set-up %.2f s  %s" % (
    variant, med, min(v), max(v), r.bytes_per_block * n / med / 1e3, r.bytes_per_block * n / med / 8e6, setup, info))
