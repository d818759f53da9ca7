#!/usr/bin/env python3
"""Does the placement of the seven output columns change the launch time?  One process, the headline workload
(2^20 blocks, per-block keys + key witness, packed), the same kernel; the columns are carved from ONE device buffer
at different alignments / gaps, or are separate torch tensors (what bench.py does).  Interleaved rounds, medians."""
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
lib = pkg.load_library()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
ctx = pkg.Context(0)
L = pkg.LAYOUT_PACKED
strides = [pkg.column_stride(L, c) for c in range(3)] + [96] + [pkg.key_column_stride(L, c) for c in range(3)]
g = torch.Generator(device="cpu").manual_seed(5)
pt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, generator=g).cuda()
keys = torch.randint(0, 256, (n, 16), dtype=torch.uint8, generator=g).cuda()
BIG = 72 << 30
big = torch.empty(BIG, dtype=torch.uint8, device="cuda")
base = big.data_ptr()
base += (-base) % (2 << 20)
LIMIT = big.data_ptr() + BIG


def carve(align, skew, start):
    """column c starts at the next multiple of `align` plus c * skew"""
    ptrs, off = [], start
    for c, s in enumerate(strides):
        off += (-off) % align
        ptrs.append(base + off + c * skew)
        off += n * s + c * skew
    return ptrs, off


variants = {}
for name, align, skew in (("back to back (128 B aligned)", 128, 0), ("2 MiB aligned", 2 << 20, 0),
                          ("64 MiB aligned", 64 << 20, 0), ("256 MiB aligned", 256 << 20, 0), ("512 MiB aligned", 512 << 20, 0),
                          ("1 GiB aligned", 1 << 30, 0), ("2 GiB aligned", 2 << 30, 0), ("4 GiB aligned", 4 << 30, 0),
                          ("1 GiB aligned + c x 4 KiB", 1 << 30, 4096), ("1 GiB aligned + c x 2 MiB", 1 << 30, 2 << 20),
                          ("1 GiB aligned + c x 146 MiB", 1 << 30, 146 << 20)):
    a, end = carve(align, skew, 0)
    b, end2 = carve(align, skew, end + (64 << 20))
    # every column range must lie inside the buffer BEFORE anything is launched (an out-of-range store faults the GPU)
    for ptrs in (a, b):
        for p_, s_ in zip(ptrs, strides):
            assert big.data_ptr() <= p_ and p_ + n * s_ <= LIMIT and p_ % 16 == 0, (name, hex(p_))
    assert base + end2 <= LIMIT, name
    variants[name] = (a, b)
if len(sys.argv) > 2 and sys.argv[2] == "where":  # the same back-to-back layout at different places of the buffer
    variants = {}
    for gib in (0, 8, 16, 24, 32, 40, 48, 56):
        a, end = carve(128, 0, gib << 30)
        b, end2 = carve(128, 0, end + (64 << 20))
        for ptrs in (a, b):
            for p_, s_ in zip(ptrs, strides):
                assert big.data_ptr() <= p_ and p_ + n * s_ <= LIMIT and p_ % 16 == 0, (gib, hex(p_))
        variants["back to back at %2d GiB" % gib] = (a, b)
    # and the 4 GiB-aligned layout with its two sets swapped in place (same addresses, other roles)
    a, end = carve(4 << 30, 0, 0)
    b, end2 = carve(4 << 30, 0, end + (64 << 20))
    assert base + end2 <= LIMIT
    variants["4 GiB aligned"] = (a, b)
    variants["columns 1 GiB apart, from 30 GiB"] = ([base + (30 << 30) + (c << 30) for c in range(7)], [base + (38 << 30) + (c << 30) for c in range(7)])
    for ptrs in variants["columns 1 GiB apart, from 30 GiB"]:
        for p_, s_ in zip(ptrs, strides):
            assert p_ + n * s_ <= LIMIT
sep = [[torch.empty(n * s, dtype=torch.uint8, device="cuda") for s in strides] for _ in range(2)]
variants["separate torch tensors"] = ([t.data_ptr() for t in sep[0]], [t.data_ptr() for t in sep[1]])


def launch(ptrs, stream):
    ks = pkg.api.KeySlab(ptrs[3], ptrs[4], ptrs[5], ptrs[6])
    rc = lib.aesw_encrypt_witness_device(ctx._h, pt.data_ptr(), keys.data_ptr(), 1, n, L, ptrs[0], ptrs[1], ptrs[2], None, C.byref(ks), stream)
    assert rc == 0, rc


steps = 10 if lg >= 19 else 50
res = {k: [] for k in variants}
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rnd in range(7):
    for name, (a, b) in variants.items():
        for i in range(2):
            launch(a if i & 1 else b, sp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            launch(a if i & 1 else b, sp)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / steps * 1e3)
for name, v in res.items():
    med = statistics.median(v)
    print("%-34s median %8.2f us  min %8.2f  max %8.2f  -> %6.0f GB/s" % (name, med, min(v), max(v), 3992 * n / med / 1e3))
