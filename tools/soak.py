#!/usr/bin/env python3
"""Soak: many byte-exact comparisons against the oracle at sizes where timing-dependent faults show
(store-data hazards, LDS ordering), random layout / key mode / size / launch options per iteration."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import oracle_lib as ol  # noqa: E402

ge.build()
pkg = ge.load_package()
orc = ol.Oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
threads = min(16, os.cpu_count() or 8)
t0 = time.time()
for it in range(iters):
    layout = int(rng.integers(0, 3))
    keymode = int(rng.integers(0, 3))
    n = int(rng.integers(1 << 16, 1 << 18)) + int(rng.integers(0, 64))
    ctx = pkg.Context(0)
    ctx.set_option("waves_shared", int(rng.integers(0, 5)))
    ctx.set_option("waves_pbk", int(rng.integers(0, 5)))
    ctx.set_option("store_mode", int(rng.integers(0, 3)))
    if rng.integers(0, 4) == 0:
        ctx.set_option("grid_cap", int(rng.integers(1, 2048)))
    ctx.set_option("xcd_remap", int(rng.choice([0, 1, 1, 2, 7, 64, 1000, 100000])))
    arena = bool(rng.integers(0, 2))  # round 3: output columns in a probed arena (virtual-memory backed) or in plain tensors
    ctx.set_option("arena_probe", int(rng.integers(1, 4)))
    ctx.set_option("arena_unit", int(rng.integers(0, 3)))
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    dpt = torch.from_numpy(pt).cuda()
    out = ctx.alloc_columns(n, layout, want_ct=True, key_slab=keymode == 2) if arena else None
    if keymode == 0:
        ctx.schedule_key(torch.from_numpy(keys[0]).cuda(), layout=layout, key_slab=False)
        got = ctx.encrypt_witness(dpt, None, layout=layout, want_ct=True, out=out)
        kh = keys[0]
    elif keymode == 1:
        got = ctx.encrypt_witness(dpt, torch.from_numpy(keys[0]).cuda(), layout=layout, want_ct=True, out=out)
        kh = keys[0]
    else:
        got = ctx.encrypt_witness(dpt, torch.from_numpy(keys).cuda(), layout=layout, want_ct=True, key_slab=True, out=out)
        kh = keys
    torch.cuda.synchronize()
    exp = orc.encrypt_witness(pt, kh, layout=layout, threads=threads)
    for c in "xyz":
        a, e = getattr(got, c).cpu().numpy(), getattr(exp, c)
        if not np.array_equal(a, e):
            bad = np.nonzero(a != e)[0]
            print("MISMATCH iter %d layout %d keymode %d n %d column %s: %d bytes, first %d" % (it, layout, keymode, n, c, bad.size, bad[0]))
            sys.exit(1)
    if not np.array_equal(got.ct.cpu().numpy(), exp.ct):
        print("MISMATCH ct iter", it)
        sys.exit(1)
    if keymode == 2:
        kexp = orc.key_schedule_witness(keys, layout=layout, threads=threads)
        for c in ("w", "kx", "ky", "kz"):
            if not np.array_equal(getattr(got.key, c).cpu().numpy(), getattr(kexp, c)):
                print("MISMATCH key slab", c, "iter", it)
                sys.exit(1)
    ctx.close()
    if it % 10 == 9:
        print("iter %d ok (%.0f s)" % (it + 1, time.time() - t0), flush=True)
print("soak ok: %d iterations" % iters)
