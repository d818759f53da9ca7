#!/usr/bin/env python3
"""key_kernel into a probed key-only arena (aesw_columns_alloc with_key_slab = 2): waves per workgroup x store flavour, 2^20
keys, one process, interleaved rounds; the arena's probe / fill times are the yardstick."""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
nk = 1 << 20
keys = torch.randint(0, 256, (nk, 16), dtype=torch.uint8, device="cuda")
base = pkg.Context(0)
ka = base.alloc_columns(nk, pkg.LAYOUT_PACKED, key_slab=True, key_only=True)
print("arena", base.last_arena)
variants = [(w, m) for w in (1, 2, 3, 4) for m in (1, 2)]
ctxs = []
for w, m in variants:
    c = pkg.Context(0)
    c.set_option("waves_pbk", w)
    c.set_option("key_store_mode", m)
    ctxs.append(c)
res = {v: [] for v in variants}
for _ in range(5):
    for v, c in zip(variants, ctxs):
        c.key_schedule_witness(keys, layout=pkg.LAYOUT_PACKED, want_rk=False, out=ka.key)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c.key_schedule_witness(keys, layout=pkg.LAYOUT_PACKED, want_rk=False, out=ka.key)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 10 * 1e3)
for v in variants:
    med = statistics.median(res[v])
    print("waves %d store %d: %7.1f us -> %5.0f GB/s algorithmic (%.3f), eff waves %s" % (v[0], v[1], med, 952 * nk / med / 1e3, 952 * nk / med / 1e3 / 8000, ctxs[variants.index(v)].get_option("effective_waves_key")))
