#!/usr/bin/env python3
"""Independent batches through aesw_encrypt_witness_batches_device on 1 / 2 / 3 internal streams, captured into one hipGraph
(bench.overlapped_launches): microseconds per batch.
argv: log2(blocks) [pbk].  Round 3: a 2^16-block launch costs 35 us alone and 29 us when its ramp and tail overlap its
neighbours' bodies -- the time of a linear fill of its bytes."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
import bench  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pbk = len(sys.argv) > 2 and sys.argv[2] == "pbk"
n = 1 << log2n
ctx = pkg.Context(0)
steps = 200 if log2n <= 17 else 40
bpb = bench.BYTES_PBK if pbk else bench.BYTES_SHARED
for rnd in range(2):
    res = bench.overlapped_launches(pkg, ctx, torch, n, pbk, (1, 2, 3), steps, True)
    print("output sets: %d, last arena: %s" % (bench.overlapped_launches.last_sets, ctx.last_arena))
    for br, us in res.items():
        print("2^%d blocks%s, %d stream(s): %8.2f us per launch -> %5.0f GB/s (%.3f of 8 TB/s)" % (
            log2n, " per-block keys" if pbk else "", br, us, bpb * n / us / 1e3, bpb * n / us / 8e6))
ctx.close()
