#!/usr/bin/env python3
"""End to end on one MI355X, the reference's own circuit shape (src/main.rs: K = 20, N = 4, 3 000 blocks):

1. the device generates the witness of every encrypt() call (one kernel launch),
2. the host-side mirror of FixedAes128Config runs synthesize() on it -- every value closure is a read of the
   device witness -- and MockProver checks lookups, the rcon gate and all copy constraints,
3. the same circuit again from the values-only witness (1 056 B per block over PCIe instead of 3 024),
4. the whole advice matrix as 32-byte bn256::Fr cells straight from the device (bulk column assignment),
5. selectors and the fixed column for keygen, from pure-host geometry.

Run:  python examples/end_to_end.py [blocks]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def main(n=3000, k=20, n_sets=4):
    ge.build()
    pkg = ge.load_package()
    ctx = pkg.Context(0)
    rng = np.random.default_rng(2024)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    assert n <= pkg.block_capacity(k, n_sets), "the reference would panic: AES calls too many"

    # 1. device witness (packed columns), resident in HBM
    t0 = time.perf_counter()
    key_witness = ctx.schedule_key(torch.from_numpy(key).cuda())
    wit = ctx.encrypt_witness(torch.from_numpy(pts).cuda(), None, want_ct=True)
    torch.cuda.synchronize()
    print("1. witness of %d blocks on the device: %.2f ms (includes the first-launch set-up)" % (n, (time.perf_counter() - t0) * 1e3))

    # 1b. the reference's own correctness test -- MockProver's constraint satisfaction -- on the device, before any circuit is built
    t0 = time.perf_counter()
    rep = ctx.check_witness(torch.from_numpy(pts).cuda(), torch.from_numpy(key).cuda(), wit, key_witness, ct=wit.ct)
    assert rep["satisfied"] and rep["blocks"] == n and rep["keys"] == 1, rep
    print("1b. every lookup, copy constraint, gate and literal row of %d blocks + the key slab checked on the device: %.2f ms" % (
        n, (time.perf_counter() - t0) * 1e3))

    # 2. synthesize() + MockProver on the full witness
    t0 = time.perf_counter()
    with pkg.HostCircuit.aes(ctx, k, n_sets, key, pts) as mock:
        rc, msg = mock.verify()
        assert rc == 0, msg
        regions, advice = mock.num_regions, [mock.advice(c) for c in range(mock.num_advice)]
        assert np.array_equal(mock.ciphertext(n - 1), wit.ct[n - 1].cpu().numpy())
    print("2. synthesize() + assert_satisfied(): %d regions, %.2f s" % (regions, time.perf_counter() - t0))

    # 3. the same from the values-only witness
    t0 = time.perf_counter()
    with pkg.HostCircuit.aes(ctx, k, n_sets, key, pts, values_only=True) as mock:
        assert mock.verify() == (0, "")
        assert all(np.array_equal(mock.advice(c), advice[c]) for c in range(mock.num_advice))
    print("3. the same circuit from the values-only witness: %.2f s" % (time.perf_counter() - t0))

    # 3b. no region at all: whole columns + the library's keygen data (selectors, fixed column, table, equality constraints)
    t0 = time.perf_counter()
    with pkg.HostCircuit.aes_columns(ctx, k, n_sets, key, pts) as mock:
        built = time.perf_counter() - t0
        assert all(np.array_equal(mock.advice(c), advice[c]) for c in range(mock.num_advice))
        t0 = time.perf_counter()
        assert mock.verify() == (0, "")
        print("3b. the same circuit from whole columns and keygen data, no region run: built in %.3f s (MockProver check %.2f s)" %
              (built, time.perf_counter() - t0))

    # 4. bulk columns: (3N+1) x 2^K Fr cells from the device == what synthesize() assigned
    fr = ctx.assemble_advice(k, n_sets, wit, key_witness, n, as_fr=True)
    cols = ctx.assemble_advice(k, n_sets, wit, key_witness, n)
    torch.cuda.synchronize()
    assert all(np.array_equal(cols[c].cpu().numpy(), advice[c]) for c in range(len(advice)))
    print("4. advice matrix as Fr cells: %s, %.1f MB" % (tuple(fr.shape), fr.numel() / 1e6))

    # 5. keygen data
    sel, fixed = pkg.assemble_selectors(k, n_sets, n)
    print("5. selectors %s (enabled rows: %d), fixed column rcon rows: %d" % (sel.shape, int(sel.sum()), int((fixed != 0).sum())))
    print("ok")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
