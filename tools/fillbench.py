#!/usr/bin/env python3
"""How fast can ANY kernel write 199 MB (the headline launch's bytes)?  torch fill_ and zero_ over a ring of
buffers larger than the Infinity Cache, launches back to back in a hipGraph, next to the witness kernel."""
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
for log2n in (16, 20):
    nbytes = 3024 << log2n
    ring = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(max(2, -(-(800 << 20) // nbytes)))]
    views32 = [r.view(torch.int32) for r in ring]
    steps = 100 if log2n == 16 else 10

    def run(fn):
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=s):
            for i in range(steps):
                fn(i)
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / steps * 1e3)
        return statistics.median(ts)

    t_fill8 = run(lambda i: ring[i % len(ring)].fill_(7))
    t_fill32 = run(lambda i: views32[i % len(ring)].fill_(7))
    t_zero = run(lambda i: ring[i % len(ring)].zero_())
    ctx = pkg.Context(0)
    r = bench.Runner(pkg, ctx, torch, 1 << log2n, False, pkg.LAYOUT_PACKED, False, 5)
    _, ms, _ = r.run(steps, 5, True)
    print("2^%d blocks (%.1f MB): fill_ u8 %.2f us (%.0f GB/s)  fill_ i32 %.2f us (%.0f GB/s)  zero_ %.2f us (%.0f GB/s)  witness kernel %.2f us (%.0f GB/s)" %
          (log2n, nbytes / 1e6, t_fill8, nbytes / t_fill8 / 1e3, t_fill32, nbytes / t_fill32 / 1e3, t_zero, nbytes / t_zero / 1e3, ms * 1e3, nbytes / ms / 1e6))
    del ring, views32, r
    torch.cuda.empty_cache()
