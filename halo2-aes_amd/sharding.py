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


last_gather_path = None  # "aesw_gather_columns_device (RCCL)" or "torch.distributed point-to-point": what the last call used


def _aesw_comm(ctx, group):
    """The C-ABI RCCL communicator of (ctx, group), created on first use: group rank 0 makes the unique id and the
    existing process group carries it to the others.  It is kept ON the context, next to a reference to the group (so the
    group's id() cannot be handed to another object while the entry lives), and goes away with the context."""
    import torch
    import torch.distributed as dist
    from . import api

    comms = ctx.__dict__.setdefault("_sharding_comms", {})  # id(group) -> (group, api.Comm)
    key = id(group) if group is not None else None
    if key in comms and comms[key][0] is group:
        return comms[key][1]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = torch.zeros(128, dtype=torch.uint8, device="cuda:%d" % ctx.device)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(api.Comm.unique_id()), dtype=torch.uint8))
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(uid, src=src, group=group)
    comm = api.Comm(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))
    comms[key] = (group, comm)
    return comm


def gather_columns(columns: Sequence, counts: Sequence[int], strides: Sequence[int], dst: int = 0, group=None,
                   max_message_bytes: int = 1 << 30, ctx=None, force_torch: bool = False):
    """Gather per-rank column slices on group rank `dst`.

    columns: this rank's column tensors (uint8, flat, counts[rank]*stride bytes).
    counts:  blocks per rank (len == world).  strides: bytes per block per column.
    Returns the full columns on `dst` (list of tensors), None elsewhere.

    With the nccl backend and `ctx` (the rank's aesw Context) the exchange is the C ABI's
    aesw_gather_columns_device -- RCCL send/recv inside one ncclGroupStart/End, enqueued on torch's current
    stream, so it is ordered behind the kernels that produced the columns without a host synchronisation; a host
    without torch calls the same entry point.  Otherwise (gloo in the CPU tests) it falls back to
    torch.distributed point-to-point operations with the same index math.  A rank's range of one column travels
    as messages of at most `max_message_bytes`; sender and receiver must cut a range into the same pieces, so the
    RCCL path uses the group-wide MINIMUM of the ranks' values (one int64 all-reduce per call).
    `force_torch=True` (or AESW_GATHER_PATH=torch in the environment) keeps the torch point-to-point path even with
    the nccl backend: the RCCL leg of the C ABI has not yet run with more than one real rank.
    """
    import os
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if len(counts) != world or len(columns) != len(strides):
        raise ValueError("counts/strides do not match world size / columns")
    if max_message_bytes <= 0:
        raise ValueError("max_message_bytes must be positive")
    global last_gather_path
    force_torch = force_torch or os.environ.get("AESW_GATHER_PATH", "") == "torch"
    if ctx is not None and not force_torch and dist.get_backend(group) == "nccl" and all(c.is_cuda for c in columns):
        last_gather_path = "aesw_gather_columns_device (RCCL send/recv over xGMI, C ABI)"
        comm = _aesw_comm(ctx, group)
        # sender and receiver must cut a range into the same pieces: agree on the MINIMUM of the ranks' requests.  One int64
        # all-reduce on EVERY call -- whether to communicate must not depend on rank-local state, or the collectives of ranks
        # that repeat a request and ranks that change theirs stop pairing up
        t = torch.tensor([int(max_message_bytes)], dtype=torch.int64, device="cuda:%d" % ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        comm.set_max_message(int(t.item()))
        return comm.gather_columns([c[:counts[rank] * s] for c, s in zip(columns, strides)], counts, strides, root=dst)
    last_gather_path = "torch.distributed point-to-point (%s)" % dist.get_backend(group)
    total = sum(counts)
    offs = [sum(counts[:r]) for r in range(world)]
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)  # P2POp wants global ranks
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
                        ops.append(dist.P2POp(dist.irecv, view[o:o + max_message_bytes], peer(r), group))
    elif counts[rank]:
        for col, s in zip(columns, strides):
            mine = col[:counts[rank] * s].contiguous()
            for o in range(0, mine.numel(), max_message_bytes):
                ops.append(dist.P2POp(dist.isend, mine[o:o + max_message_bytes], peer(dst), group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return outs
