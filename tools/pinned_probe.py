#!/usr/bin/env python3
"""Where does the host-pointer path (page-locked outputs) lose its link rate inside bench.py's process?
Times aesw_encrypt_witness into the same page-locked buffers at several points of a bench-like sequence."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa
import __graft_entry__ as ge
ge.build()
pkg = ge.load_package()
import bench
ctx = pkg.Context(0)
nn = 1 << 20
hpt = pkg.api.host_alloc(nn * 16).reshape(nn, 16)
hpt[:] = np.random.default_rng(1).integers(0, 256, (nn, 16), dtype=np.uint8)
outs = [pkg.api.host_alloc(nn * pkg.column_stride(pkg.LAYOUT_PACKED, c)) for c in range(3)]
key = torch.arange(16, dtype=torch.uint8).cuda()


def probe(tag):
    ctx.schedule_key(key, layout=pkg.LAYOUT_PACKED, key_slab=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.encrypt_witness_host(hpt, None, layout=pkg.LAYOUT_PACKED, out_cols=outs)
        ts.append(time.perf_counter() - t0)
    print("%-40s %s GB/s" % (tag, " ".join("%.1f" % (nn * 3024 / t / 1e9) for t in ts)), flush=True)


probe("fresh")
r = bench.Runner(pkg, ctx, torch, 1 << 16, False, pkg.LAYOUT_PACKED, False, 3)
r.run(50, 5, False)
probe("after eager launches")
r.run(50, 5, True)
probe("after a hipGraph")
del r
torch.cuda.empty_cache()
probe("after empty_cache")
r = bench.Runner(pkg, ctx, torch, 1 << 20, True, pkg.LAYOUT_PACKED, True, 3)
r.run(10, 2, True)
del r
torch.cuda.empty_cache()
probe("after the 2^20 per-block-key runner")
big = np.empty(3 << 30, np.uint8); big[::4096] = 1
probe("after touching 3 GiB of pageable memory")
outs2 = [pkg.api.host_alloc(nn * pkg.column_stride(pkg.LAYOUT_PACKED, c)) for c in range(3)]
outs, old = outs2, outs
probe("fresh page-locked outputs, late")
outs = old
probe("the first outputs again")
