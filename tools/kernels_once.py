#!/usr/bin/env python3
"""One launch of every secondary kernel, for a rocprofv3 --pmc pass (expected HBM write bytes printed)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
ctx = pkg.Context(0)
g = torch.Generator(device="cpu").manual_seed(3)
nk = 1 << 16
keys = torch.randint(0, 256, (nk, 16), dtype=torch.uint8, generator=g).cuda()
for layout, name in ((pkg.LAYOUT_PACKED, "packed"), (pkg.LAYOUT_DENSE, "dense")):
    ctx.key_schedule_witness(keys, layout=layout, want_rk=False)
    print("key_kernel %s: expected %d bytes" % (name, nk * (96 + sum(pkg.key_column_stride(layout, c) for c in range(3)))))
cells = torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, generator=g).cuda()
ctx.expand_fr(cells)
print("expand_fr: expected %d bytes" % (32 << 20))
k, n_sets = 16, 2
n = pkg.block_capacity(k, n_sets)
pts = torch.randint(0, 256, (n, 16), dtype=torch.uint8, generator=g).cuda()
kw = ctx.schedule_key(keys[0].contiguous())
w = ctx.encrypt_witness(pts, None)
ctx.assemble_advice(k, n_sets, w, kw, n, as_fr=True)
ctx.assemble_advice(k, n_sets, w, kw, n, as_fr=False)
print("assemble fr: expected %d bytes; bytes: expected %d" % (7 * 65536 * 32, 7 * 65536))
ctx.lookup_table()
print("table: expected %d bytes" % (4 * 66561))
for lay, name in ((pkg.LAYOUT_DENSE, "dense"), (pkg.LAYOUT_VALUES, "values")):
    ctx.encrypt_witness(pts, keys[:n].contiguous(), layout=lay, key_slab=True)
torch.cuda.synchronize()
