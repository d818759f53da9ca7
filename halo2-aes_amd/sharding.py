"""Multi-GPU: one process per GPU, blocks sharded by contiguous index range.

Blocks are independent: rank g of G takes blocks
[g*n/G, (g+1)*n/G).  There is NO collective on the data path -- every rank
writes its own slab range.  Because a column's slabs are contiguous per block,
each rank's output is one contiguous byte range per column, so handing the
whole witness to one consumer is a single *gather* per column
(``gather_columns``; RCCL when the backend is "nccl", gloo in the CPU tests).
A gather maps well onto xGMI -- each peer has its own link to the root -- but
the root's ingest (7 x ~153 GB/s) is far below one GPU's generation rate, so
it is optional and timed separately by bench.py, never part of `value`.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block range [lo, hi) of `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard request n=%d rank=%d world=%d" % (n, rank, world))
    return n * rank // world, n * (rank + 1) // world


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_columns(columns: Sequence, counts: Sequence[int], strides: Sequence[int], dst: int = 0, group=None,
                   max_message_bytes: int = 1 << 30):
    """Gather per-rank column slices on rank `dst`.

    columns: this rank's column tensors (uint8, flat, counts[rank]*stride bytes).
    counts:  blocks per rank (len == world).  strides: bytes per block per column.
    Returns the full columns on `dst` (list of tensors), None elsewhere.
    Uses send/recv pairs so ragged shard sizes need no padding; with the nccl
    backend every peer->root transfer rides its own xGMI link.  A rank's range of
    one column travels as messages of at most `max_message_bytes` (BASELINE
    configs[3] moves 2.9 GB per column and rank).
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if len(counts) != world or len(columns) != len(strides):
        raise ValueError("counts/strides do not match world size / columns")
    if max_message_bytes <= 0:
        raise ValueError("max_message_bytes must be positive")
    total = sum(counts)
    offs = [sum(counts[:r]) for r in range(world)]
    # One grouped batch of point-to-point operations: with the nccl backend (RCCL) the group is
    # issued as a single ncclGroupStart/End, so the root's receives from all peers proceed
    # concurrently, each over its own xGMI link, instead of one peer after the other.
    ops = []
    outs = None
    if rank == dst:
        outs = [torch.empty(total * s, dtype=torch.uint8, device=c.device) for c, s in zip(columns, strides)]
        for c, (col, s) in enumerate(zip(columns, strides)):
            for r in range(world):
                view = outs[c][offs[r] * s:(offs[r] + counts[r]) * s]
                if r == dst:
                    view.copy_(col[:counts[r] * s])
                elif counts[r]:
                    for o in range(0, view.numel(), max_message_bytes):
                        ops.append(dist.P2POp(dist.irecv, view[o:o + max_message_bytes], r, group))
    elif counts[rank]:
        for col, s in zip(columns, strides):
            mine = col[:counts[rank] * s].contiguous()
            for o in range(0, mine.numel(), max_message_bytes):
                ops.append(dist.P2POp(dist.isend, mine[o:o + max_message_bytes], dst, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return outs
