#!/usr/bin/env python3
"""The Fr form of assemble: striding workgroups (assemble_geometry 0), the division-free one-shot grid (1) and one-shot
workgroups on aligned chunks of the output (2: 256 threads x 1 piece, 3: 256 x 2, 4: 128 x 2 = the default), with expand_fr over as many bytes as the
yardstick.  K from argv (default 20), N = 5 (16 columns x 2^K cells x 32 B), full capacity; all outputs compared byte for byte."""
import statistics
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
k, n_sets = (int(sys.argv[1]) if len(sys.argv) > 1 else 20), 5
NGEO = 5
LAYOUT = pkg.LAYOUT_DENSE if len(sys.argv) > 2 and sys.argv[2] == "dense" else pkg.LAYOUT_PACKED  # dense slabs: no packed-index arithmetic
ctx = pkg.Context(0)
nn = pkg.block_capacity(k, n_sets)
pt = torch.randint(0, 256, (nn, 16), dtype=torch.uint8, device="cuda")
key = torch.randint(0, 256, (16,), dtype=torch.uint8, device="cuda")
kw = ctx.schedule_key(key, layout=LAYOUT, key_slab=True)
wit = ctx.encrypt_witness(pt, None, layout=LAYOUT)
outs = {}
res = {g: [] for g in range(NGEO)}
for rnd in range(5):
    for geo in range(NGEO):
        ctx.set_option("assemble_geometry", geo)
        out = ctx.assemble_advice(k, n_sets, wit, kw, nn, layout=LAYOUT, as_fr=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = ctx.assemble_advice(k, n_sets, wit, kw, nn, layout=LAYOUT, as_fr=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        res[geo].append(e0.elapsed_time(e1) / 10 * 1e3)
        outs[geo] = out
assert all(torch.equal(outs[0], outs[g]) for g in range(1, NGEO)), "the geometries disagree"
nbytes = outs[0].numel()
for geo in range(NGEO):
    med = statistics.median(res[geo])
    print("assemble_geometry %d: %8.2f us per %d MiB  -> %6.0f GB/s written" % (geo, med, nbytes >> 20, nbytes / med / 1e3))

# yardstick: expand_fr (same LUT expansion, two-load chain, aligned 4 KiB workgroups) writing the same number of bytes
cells = torch.randint(0, 256, (nbytes // 32,), dtype=torch.uint8, device="cuda")
fo = ctx.expand_fr(cells)
ts = []
for rnd in range(5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.expand_fr(cells, out=fo)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
med = statistics.median(ts)
print("expand_fr (yardstick):  %8.2f us per %d MiB  -> %6.0f GB/s written" % (med, nbytes >> 20, nbytes / med / 1e3))
