#!/usr/bin/env python3
"""Store flavour (0 plain, 1 nontemporal, 2 write-through sc1) by workload, layout and batch size, default group sizes, one process."""
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
ctxs = []
for m in (0, 1, 2):
    c = pkg.Context(0)
    c.set_option("store_mode", m)
    ctxs.append(c)
lay = {"packed": pkg.LAYOUT_PACKED, "dense": pkg.LAYOUT_DENSE, "values": pkg.LAYOUT_VALUES}
cases = [(w, l, n) for w in ("c1", "c2") for l in ("packed", "dense", "values") for n in (14, 16, 18, 20)]
for w, l, log2n in cases:
    pbk = w == "c2"
    n = 1 << log2n
    r = bench.Runner(pkg, ctxs[0], torch, n, pbk, lay[l], pbk, 7)
    res = [[], [], []]
    steps = 100 if log2n <= 16 else (30 if log2n <= 18 else 10)
    for _ in range(5):
        for m, c in enumerate(ctxs):
            if not pbk:
                c.schedule_key(r.keys, layout=lay[l], key_slab=False)
            r.h, r.ctx = c._h, c
            _, ms, _ = r.run(steps, 3, True)
            res[m].append(ms * 1e3)
    med = [statistics.median(x) for x in res]
    best = med.index(min(med))
    print("%s %-6s 2^%-2d  plain %8.2f  nt %8.2f  sc1 %8.2f us   best: %s (%+.1f %% vs sc1)" %
          (w, l, log2n, med[0], med[1], med[2], ("plain", "nt", "sc1")[best], (med[2] / med[best] - 1) * 100), flush=True)
    del r
    torch.cuda.empty_cache()
