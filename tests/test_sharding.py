"""Multi-GPU path on CPU: block sharding and the column gather with the gloo
backend, world_size 2 and 3 (the same code runs over RCCL with backend nccl)."""
import os
import socket

import numpy as np
import pytest


def test_shard_range_partitions(pkg):
    sh = pkg.sharding
    for n in (0, 1, 7, 16, 1 << 16, (1 << 24) + 5):
        for world in (1, 2, 3, 8):
            ranges = [sh.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = sh.shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    assert sh.shard_range(1 << 24, 3, 8) == (3 << 21, 4 << 21)   # BASELINE config 3: 2^21 per GPU
    with pytest.raises(ValueError):
        sh.shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, layout, tmp, max_msg):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    sys.path.insert(0, str(root / "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_lib as ol
    pkg = ge.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)
        pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
        key = rng.integers(0, 256, 16, dtype=np.uint8)
        lo, hi = pkg.sharding.shard_range(n, rank, world)
        # the per-rank witness comes from the oracle here (no GPU in this test): the
        # sharding/gather plumbing is what is under test
        w = ol.Oracle().encrypt_witness(pt[lo:hi], key, layout=layout) if hi > lo else None
        strides = [pkg.column_stride(layout, c) for c in range(3)]
        cols = [torch.from_numpy(getattr(w, c)) if w is not None else torch.empty(0, dtype=torch.uint8) for c in "xyz"]
        full = pkg.sharding.gather_columns(cols, pkg.sharding.shard_sizes(n, world), strides, dst=0, max_message_bytes=max_msg)
        if rank == 0:
            exp = ol.Oracle().encrypt_witness(pt, key, layout=layout)
            ok = all(np.array_equal(f.numpy(), getattr(exp, c)) for f, c in zip(full, "xyz"))
            (tmp / "result").write_text("ok" if ok else "mismatch")
        else:
            assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


# world size 8 is the shape of BASELINE configs[3] (2^24 blocks over 8 GPUs): eight ranks, ragged shards (203 = 8 x 25 + 3),
# every range cut into >= 3 messages; and five blocks over eight ranks, so that three ranks have nothing to send
@pytest.mark.parametrize("world,n,max_msg", [(2, 37, 1 << 30), (3, 5, 1 << 30), (2, 1, 1 << 30), (2, 37, 1000), (8, 203, 5000), (8, 5, 200)])
def test_gather_columns_gloo(pkg, tmp_path, world, n, max_msg):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, pkg.LAYOUT_PACKED, tmp_path, max_msg), nprocs=world, join=True)
    assert (tmp_path / "result").read_text() == "ok"


def _subgroup_worker(rank, world, port, tmp):
    """Gather inside a SUBGROUP whose group ranks differ from the global ranks (global 1, 2 -> group 0, 1)."""
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        group = dist.new_group([1, 2])  # collective over all ranks
        if rank in (1, 2):
            g = rank - 1
            counts, strides = [3, 2], [5, 7]
            cols = [torch.full((counts[g] * s,), 10 * g + c, dtype=torch.uint8) for c, s in enumerate(strides)]
            full = pkg.sharding.gather_columns(cols, counts, strides, dst=0, group=group)
            if g == 0:
                ok = all(torch.equal(f, torch.cat([torch.full((counts[r] * s,), 10 * r + c, dtype=torch.uint8) for r in range(2)]))
                         for c, (f, s) in enumerate(zip(full, strides)))
                (tmp / "sub").write_text("ok" if ok else "mismatch")
            else:
                assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_columns_in_a_subgroup(pkg, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_subgroup_worker, args=(3, _free_port(), tmp_path), nprocs=3, join=True)
    assert (tmp_path / "sub").read_text() == "ok"


def test_gather_offsets_pure_host(pkg):
    """aesw_gather_offsets: the index math of the C-ABI gather (exclusive prefix sum), no device needed."""
    import ctypes as C
    lib = pkg.load_library()
    counts = np.array([5, 0, 7, 1], np.uint64)
    offs = np.zeros(4, np.uint64)
    total = C.c_uint64()
    assert lib.aesw_gather_offsets(4, counts.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), C.byref(total)) == 0
    assert offs.tolist() == [0, 5, 5, 12] and total.value == 13
    assert lib.aesw_gather_offsets(0, counts.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), None) == 1


def test_shard_range_and_gather_offsets_agree(pkg):
    """the block range a rank generates (sharding.shard_range, Python) and the place the C ABI's gather
    puts it on the root (aesw_gather_offsets, exclusive prefix sum of the counts) are the same partition -- for BASELINE
    configs[3] (2^24 blocks, 8 GPUs: 2^21 each) and for sizes that do not divide."""
    import ctypes as C
    lib = pkg.load_library()
    sh = pkg.sharding
    for n, world in ((1 << 24, 8), ((1 << 24) + 5, 8), (1000003, 7), (5, 8), (0, 4), (1 << 20, 1)):
        counts = np.array(sh.shard_sizes(n, world), np.uint64)
        offs = np.zeros(world, np.uint64)
        total = C.c_uint64()
        assert lib.aesw_gather_offsets(world, counts.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), C.byref(total)) == 0
        assert total.value == n
        for r in range(world):
            lo, hi = sh.shard_range(n, r, world)
            assert (int(offs[r]), int(offs[r] + counts[r])) == (lo, hi), (n, world, r)
    assert sh.shard_sizes(1 << 24, 8) == [1 << 21] * 8
