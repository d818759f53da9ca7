#!/usr/bin/env python3
"""does a LONE configs[1]-sized launch get shorter when it is dealt as 2-4 line-aligned sub-ranges onto the
context's internal streams (option "split_small")?   usage: split_small.py [arena|plain]
Per size (2^15, 2^16, 2^17 blocks, one scheduled key, packed columns) and split (0 = one launch, 2, 3, 4): a hipGraph of 100
launches in sequence on one stream, microseconds per launch = graph time / 100, median of 5 replays, two rounds.
Kill criterion (set before measuring): keep the option on by default only if 2^16 blocks drop from ~35 us to <= 32 us."""
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

arena = (sys.argv[1] if len(sys.argv) > 1 else "arena") == "arena"
ctx = pkg.Context(0)
for rnd in range(2):
    for lg in (15, 16, 17):
        n = 1 << lg
        r = bench.Runner(pkg, ctx, torch, n, False, pkg.LAYOUT_PACKED, False, 77 + lg, arena=arena)
        row = []
        for split in (0, 2, 3, 4):
            ctx.set_option("split_small", split)
            r.prepare(100, 5, True)
            assert r.graph is not None
            us = statistics.median(r.timed()[1] for _ in range(5)) * 1e3
            row.append("split %d: %6.2f us (%.3f)" % (split, us, bench.BYTES_SHARED * n / us / 8e6))
        ctx.set_option("split_small", 0)
        print("round %d  2^%d blocks, %s:  %s" % (rnd, lg, "probed arena" if arena else "plain tensors", "   ".join(row)), flush=True)
        r.close()
ctx.close()
