"""Two ranks, one per process, each generating its contiguous shard of the batch on the GPU
through the HIP path, then the optional column gather (gloo here; RCCL with backend nccl on a
multi-GPU node).  Rank 0 compares the gathered columns with the oracle's full batch."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, tmp):
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
        ctx = pkg.Context(0)  # a multi-GPU node would use LOCAL_RANK here
        rng = np.random.default_rng(4242)
        pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
        key = rng.integers(0, 256, 16, dtype=np.uint8)
        lo, hi = pkg.sharding.shard_range(n, rank, world)
        ctx.schedule_key(torch.from_numpy(key).cuda(), key_slab=False)
        w = ctx.encrypt_witness(torch.from_numpy(pt[lo:hi]).cuda(), None, layout=pkg.LAYOUT_PACKED)
        torch.cuda.synchronize()
        strides = [pkg.column_stride(pkg.LAYOUT_PACKED, c) for c in range(3)]
        full = pkg.sharding.gather_columns([w.x.cpu(), w.y.cpu(), w.z.cpu()], pkg.sharding.shard_sizes(n, world), strides, dst=0)
        if rank == 0:
            exp = ol.Oracle().encrypt_witness(pt, key, layout=ol.PACKED)
            ok = all(np.array_equal(f.numpy(), getattr(exp, c)) for f, c in zip(full, "xyz"))
            (tmp / "result").write_text("ok" if ok else "mismatch")
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_and_gather(pkg, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), 10007, tmp_path), nprocs=2, join=True)
    assert (tmp_path / "result").read_text() == "ok"
