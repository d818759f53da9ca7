#!/usr/bin/env python3
"""bench.py -- AES-128 block witnesses/s and achieved HBM GB/s on N MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by torch.distributed.run, one rank per GPU.
A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: one aesw_encrypt_witness_device launch over
BASELINE.json configs[2] -- 2^20 blocks, per-block keys (the GPU key-schedule
path, src/key_schedule.rs:80-224) with the key-schedule witness, packed advice
columns, 1 MI355X -- per rank (weak scaling: every rank generates its own
2^20-block shard, no collective on the data path; halo2-aes_amd/sharding.py).
Rank 0 prints ONE JSON line.

Timed region: the K steps are captured in ONE hipGraph; after W untimed warm-up
steps the graph is replayed REPLAYS times, each replay bracketed by barrier +
synchronize and timed on its own (wall clock and HIP events on the launch
stream).  `value` and `ms_per_step` come from the MEDIAN replay (exactly K
steps); every replay, min and max are in `timing`.

roofline.achieved = algorithmic bytes per launch (3992 B/block per-block keys,
3040 B/block shared key: SURVEY.md 8(d)) / mean launch duration of
the median replay from HIP events; frac_events and frac_wall are both given.
cpu_baseline = the CPU oracle (a port of the reference's value path, NOT the
reference: Rust + halo2 cannot be built here) timed on this box's host cores on
a bounded sample of the same workload shape, N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
BYTES_SHARED = 3040     # 3024 live cells written + 16 B plaintext read per block
BYTES_PBK = 3992        # + 936 key-schedule cells + 16 B key read per block
BYTES_VALUES = 1072     # AESW_LAYOUT_VALUES: 448 + 608 closure-computed cells written + 16 B read per block
SEED = 0xA35128
REPLAYS = 5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--layout", choices=["packed", "dense", "values"], default="packed")
    ap.add_argument("--workload", choices=["c1", "c2"], default="c2",
                    help="c2 = 2^20 blocks per-block keys + key witness (BASELINE configs[2], the headline); c1 = 2^16 blocks shared key (configs[1])")
    ap.add_argument("--log2-blocks", type=int, default=None, help="override the batch size (per rank)")
    ap.add_argument("--replays", type=int, default=REPLAYS, help="timed replays of the K-step graph (median reported)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--c3-log2-blocks", type=int, default=None, help="N>1 extra: blocks per GPU of the configs[3] run (default 21)")
    ap.add_argument("--c4-log2-blocks", type=int, default=24, help="N=1 extra: blocks of the configs[4] streaming run")
    ap.add_argument("--c4-rank-log2-blocks", type=int, default=21,
                    help="N>1: blocks EVERY rank streams to its own host in the configs[4] leg (2^24 over 8 GPUs = 2^21 per GPU)")
    ap.add_argument("--arena", choices=["auto", "on", "off"], default="auto",
                    help="output columns of a set in ONE device allocation (aesw_columns_alloc) instead of separate tensors")
    ap.add_argument("--gather-path", choices=["torch", "cabi"], default="torch",
                    help="N>1: what the timed `gather` and `c3` phases use: torch.distributed point-to-point send/recv (RCCL through torch's "
                         "communicator: the default, a path that has run on hardware) or the C ABI's own RCCL gather (aesw_gather_columns_device: "
                         "never run with more than one real rank, DESIGN 7); the C ABI's gather is ALSO run as the last phase (`gather_cabi`)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from the host instead of a hipGraph")
    ap.add_argument("--option", action="append", default=[], help="name=value passed to aesw_set_option")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo to rehearse on one GPU)")
    return ap.parse_args()


class Runner:
    """One workload: inputs resident in HBM, a ring of output buffer sets larger
    than the 256 MiB Infinity Cache so consecutive steps do not rewrite cached lines."""

    def __init__(self, pkg, ctx, torch, n, per_block_keys, layout, key_slab, seed, arena=False):
        self.pkg, self.ctx, self.torch, self.n = pkg, ctx, torch, n
        self.arena = arena
        self.pbk, self.layout, self.key_slab = per_block_keys, layout, key_slab
        self.key_witness = None  # the key-schedule witness of the scheduled key (shared-key workloads)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.pt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, generator=g).cuda()
        self.keys = torch.randint(0, 256, (n, 16) if per_block_keys else (16,), dtype=torch.uint8, generator=g).cuda()
        if not per_block_keys:
            # the reference's call shape (benches/aes128.rs:50-53): schedule_key once, then encrypt() per block
            self.key_witness = ctx.schedule_key(self.keys, layout=layout, key_slab=True)
            torch.cuda.synchronize()
        per_set = sum(pkg.column_stride(layout, c) for c in range(3)) * n
        if key_slab:
            per_set += (96 + sum(pkg.key_column_stride(layout, c) for c in range(3))) * n
        self.out_bytes_per_step = per_set
        # the ring exists so that a step never rewrites lines the 256 MiB Infinity Cache still holds from the previous one: sets
        # are rotated until 640 MB lie between two writes of the same byte; a set that is that large by itself needs no partner
        self.nsets = 1 if per_set >= (640 << 20) else max(2, min(8, -(-(640 << 20) // per_set)))
        # arena: one device allocation per set with aligned column bases (aesw_columns_alloc); else one tensor per column
        self.sets, self.arena_info = [], []
        torch.cuda.synchronize()
        t_setup = time.perf_counter()
        for _ in range(self.nsets):
            if arena:
                self.sets.append(ctx.alloc_columns(n, layout, key_slab=key_slab))
                self.arena_info.append(dict(ctx.last_arena))
            else:
                self.sets.append(ctx.alloc_witness(n, layout, want_ct=False, key_slab=key_slab, n_keys=n))
        torch.cuda.synchronize()
        self.setup_s = time.perf_counter() - t_setup  # wall time of placing the output columns (the arena's search), outside every timed region
        self.lib = pkg.load_library()
        self.h = ctx._h
        self._ks = [pkg.api.KeySlab(*[t.data_ptr() for t in s.key[:4]]) if key_slab else None for s in self.sets]
        self.bytes_per_block = BYTES_PBK if (per_block_keys and key_slab) else BYTES_SHARED
        if layout == pkg.LAYOUT_VALUES:  # only closure-computed cells: 448 + 608 B written, 16 B read per block
            self.bytes_per_block = BYTES_VALUES + (936 + 16 if (per_block_keys and key_slab) else 0)
        self.graph = None
        self.graph_steps = 0

    def close(self):
        """Release the arenas (separate tensors go back to torch's allocator by themselves)."""
        if self.arena:
            for w in self.sets:
                self.ctx.free_columns(w)
        self.sets, self._ks, self.graph = [], [], None

    def launch(self, i, stream):
        s = self.sets[i % self.nsets]
        ks = self._ks[i % self.nsets]
        rc = self.lib.aesw_encrypt_witness_device(
            self.h, self.pt.data_ptr(), self.keys.data_ptr() if self.pbk else None, 1 if self.pbk else 0, self.n, self.layout,
            s.x.data_ptr(), s.y.data_ptr(), s.z.data_ptr(), None, C.byref(ks) if ks is not None else None, stream)
        if rc:
            raise RuntimeError("aesw_encrypt_witness_device rc=%d %s" % (rc, self.lib.aesw_last_error(self.h).decode()))

    def prepare(self, steps, warmup, use_graph):
        """W untimed warm-up steps, then (optionally) one hipGraph holding all K launches."""
        torch = self.torch
        stream = torch.cuda.current_stream()
        sp = C.c_void_p(stream.cuda_stream)
        for i in range(warmup):
            self.launch(i, sp)
        torch.cuda.synchronize()
        self.graph, self.graph_steps = None, steps
        if use_graph:
            try:
                cap = torch.cuda.Stream()
                cap.wait_stream(stream)
                if not self.pbk:
                    # a scheduled-key launch can only be captured on the stream its key was scheduled on (aesw.h)
                    with torch.cuda.stream(cap):
                        self.ctx.schedule_key(self.keys, layout=self.layout, key_slab=False)
                    cap.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=cap):
                    csp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                    for i in range(steps):
                        self.launch(i, csp)
                graph.replay()  # untimed: instantiate / upload
                torch.cuda.synchronize()
                self.graph = graph
            except Exception as e:  # pragma: no cover - depends on the runtime
                print("bench: hipGraph capture unavailable (%s); launching from the host" % e, file=sys.stderr)
                self.graph = None

    def timed(self, barrier=None):
        """EXACTLY K steps between barrier + synchronize on both sides.
        Returns (wall seconds, mean launch duration in ms from HIP events on the launch stream)."""
        torch = self.torch
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if barrier:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        if self.graph is not None:
            self.graph.replay()
        else:
            for i in range(self.graph_steps):
                self.launch(i, sp)
        e1.record()
        while not e1.query():  # poll: a blocking wait adds its wake-up latency to a short timed region
            pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()  # this rank's K steps are done; the MAX over ranks is taken by the caller
        if barrier:
            barrier()
        return t1 - t0, e0.elapsed_time(e1) / self.graph_steps

    def run(self, steps, warmup, use_graph, barrier=None):
        """One timed region (tools/ and the extras): (wall seconds, mean launch ms, graphed)."""
        self.prepare(steps, warmup, use_graph)
        wall, ms = self.timed(barrier)
        return wall, ms, self.graph is not None


def overlapped_launches(pkg, ctx, torch, n, per_block_keys, branch_counts, steps, arena):
    """Independent batches through aesw_encrypt_witness_batches_device with "batch_streams" = b for every b in branch_counts:
    the `steps` batches of one call are dealt round-robin onto b internal streams (fork behind the caller's stream, join back
    into it), so the ramp and tail of one launch overlap the body of the next; the call is captured into one hipGraph.
    Shared key passed by pointer (the scheduled-key form can only be captured on the stream its key was scheduled on); every
    stream writes its own output sets.  Returns {b: microseconds per batch = graph time / steps, median of 5 replays}."""
    lib = pkg.load_library()
    g = torch.Generator(device="cpu").manual_seed(SEED + 11)
    pt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, generator=g).cuda()
    keys = torch.randint(0, 256, (n, 16) if per_block_keys else (16,), dtype=torch.uint8, generator=g).cuda()
    per_set = (3024 + (936 if per_block_keys else 0)) * n
    # every stream owns its output sets (two streams never write the same bytes at once); as many per stream as keep a byte
    # from being rewritten while the 256 MiB Infinity Cache may still hold it
    ring = 1 if per_set >= (640 << 20) else min(8, -(-(640 << 20) // per_set))
    per_stream = {b: max(1, -(-ring // b)) for b in branch_counts}
    nsets = max(b * per_stream[b] for b in branch_counts)
    sets = [ctx.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=per_block_keys) if arena
            else ctx.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=False, key_slab=per_block_keys, n_keys=n) for _ in range(nsets)]
    ks = [pkg.api.KeySlab(*[t.data_ptr() for t in w.key[:4]]) if per_block_keys else None for w in sets]
    overlapped_launches.last_sets = nsets
    try:
        out = {}
        for b in branch_counts:
            ctx.set_option("batch_streams", b)
            arr = (pkg.api.Batch * steps)()
            for i in range(steps):  # batch i runs on internal stream i % b: give it one of that stream's sets
                j = (i % b) * per_stream[b] + (i // b) % per_stream[b]
                w = sets[j]
                arr[i] = pkg.api.Batch(pt.data_ptr(), keys.data_ptr(), n, w.x.data_ptr(), w.y.data_ptr(), w.z.data_ptr(), None,
                                       C.pointer(ks[j]) if ks[j] is not None else None)

            def call(sp):
                rc = lib.aesw_encrypt_witness_batches_device(ctx._h, arr, steps, 1 if per_block_keys else 0, pkg.LAYOUT_PACKED, sp)
                if rc:
                    raise RuntimeError("aesw_encrypt_witness_batches_device rc=%d %s" % (rc, lib.aesw_last_error(ctx._h).decode()))

            call(C.c_void_p(torch.cuda.current_stream().cuda_stream))  # untimed: creates the internal streams outside the capture
            torch.cuda.synchronize()
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=cap):
                call(C.c_void_p(torch.cuda.current_stream().cuda_stream))
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                graph.replay()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / steps * 1e6)
            out[b] = sorted(ts)[2]
            del graph
        return out
    finally:
        ctx.set_option("batch_streams", 3)
        if arena:
            for w in sets:
                ctx.free_columns(w)


def parity_gate(runner, blocks=4096, seed=SEED + 99):
    """"Parity gate before any number is reported".  After the timed replays, `blocks` blocks of EVERY output set
    the graph wrote -- the first and last quarter-thousand, the rest random over the batch -- are copied out of the columns the
    timed launches filled (x, y, z and, with per-block keys, words_column / kx / ky / kz) and compared byte for byte with the CPU
    oracle (test infrastructure, used here as the checker only; src/aes128.rs:154-301, src/key_schedule.rs:80-224).
    Outside every timed region.  Returns the record for the line; mismatches != 0 makes bench.py exit non-zero."""
    import numpy as np
    import oracle_lib
    torch, pkg, n = runner.torch, runner.pkg, runner.n
    olay = {pkg.LAYOUT_PACKED: oracle_lib.PACKED, pkg.LAYOUT_DENSE: oracle_lib.DENSE, pkg.LAYOUT_VALUES: oracle_lib.VALUES}[runner.layout]
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(seed)
    edge = min(256, n // 2)
    mid = max(0, min(blocks, n) - 2 * edge)
    idx = np.unique(np.concatenate([np.arange(edge), np.arange(n - edge, n), rng.integers(0, n, mid)])).astype(np.int64)
    didx = torch.from_numpy(idx).cuda()
    pt = runner.pt.index_select(0, didx).cpu().numpy()
    if runner.pbk:
        keys = runner.keys.index_select(0, didx).cpu().numpy()
    else:
        keys = runner.keys.cpu().numpy()
    e = orc.encrypt_witness(pt, keys, layout=olay)
    want = {"x": e.x, "y": e.y, "z": e.z}
    strides = {c: pkg.column_stride(runner.layout, i) for i, c in enumerate("xyz")}
    if runner.key_slab and runner.pbk:
        k = orc.key_schedule_witness(keys, layout=olay)
        want.update({"w": k.w, "kx": k.kx, "ky": k.ky, "kz": k.kz})
        strides.update({"w": 96, **{c: pkg.key_column_stride(runner.layout, i) for i, c in enumerate(("kx", "ky", "kz"))}})
    mismatches, cells, bad_cols = 0, 0, []
    for si, s in enumerate(runner.sets[:min(runner.nsets, max(1, runner.graph_steps))]):
        cols = {"x": s.x, "y": s.y, "z": s.z}
        if "w" in want:
            cols.update({"w": s.key.w, "kx": s.key.kx, "ky": s.key.ky, "kz": s.key.kz})
        for c, exp in want.items():
            st = strides[c]
            if st == 0:
                continue
            got = cols[c].view(n, st).index_select(0, didx).cpu().numpy().reshape(-1)
            bad = int(np.count_nonzero(got != np.asarray(exp).reshape(-1)))
            cells += got.size
            if bad:
                mismatches += bad
                bad_cols.append("set %d column %s: %d bytes" % (si, c, bad))
    device_check = None
    if runner.layout != pkg.LAYOUT_VALUES and (runner.key_slab or runner.key_witness is not None):
        # ... and EVERY block of every set against every constraint of the reference's circuit (aesw_check_witness_device: the
        # MockProver::assert_satisfied of src/aes128.rs:409-418 on the device): lookups, copy constraints, the rcon gate, the
        # plaintext and key literals.  Not a recomputation and not the oracle: the reference's own notion of a valid witness.
        tot = {"blocks": 0, "keys": 0, "lookup_failures": 0, "copy_failures": 0, "gate_failures": 0, "input_failures": 0}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        runner.ctx.check_witness(runner.pt[:64], runner.keys[:64] if runner.pbk else runner.keys, runner.sets[0],
                                 runner.sets[0].key if runner.pbk else runner.key_witness, layout=runner.layout)  # builds the check table, untimed
        e0.record()
        reps = [runner.ctx.check_witness(runner.pt, runner.keys, s, s.key if runner.pbk else runner.key_witness, layout=runner.layout, sync=False)
                for s in runner.sets[:min(runner.nsets, max(1, runner.graph_steps))]]
        e1.record()
        torch.cuda.synchronize()
        for r in reps:
            v = r.cpu().tolist()
            for k_, x_ in zip(tot, v[:6]):
                tot[k_] += int(x_)
        device_check = dict(tot, ms_per_set=e0.elapsed_time(e1) / len(reps),
                            satisfied=not (tot["lookup_failures"] or tot["copy_failures"] or tot["gate_failures"] or tot["input_failures"]),
                            what="aesw_check_witness_device over every block of every set: each enabled lookup, each copy_advice() pair, the "
                                 "rcon gate and the plaintext / key literal rows (the reference's MockProver criterion)")
        if not device_check["satisfied"]:
            mismatches += 1
            bad_cols.append("device check: %r" % tot)
    return {"blocks": int(idx.size), "sets_checked": min(runner.nsets, max(1, runner.graph_steps)), "columns": sorted(want), "device_check": device_check,
            "bytes_compared": cells, "mismatches": mismatches, "where": bad_cols[:8],
            "sample": "blocks 0..%d, %d..%d and %d random ones of the columns the timed graph wrote, every byte against oracle/aesw_oracle.c" % (
                edge - 1, n - edge, n - 1, int(idx.size) - 2 * edge)}


def usable_cpus():
    """CPUs this process can really run on: the affinity mask, cut down by a cgroup CPU quota if there is one
    (a GPU box hands a 1-GPU job a share of its cores; os.cpu_count() still reports all of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(per_block_keys, n_target_seconds=12.0):
    """The CPU oracle on this host's cores: 1 thread (the reference synthesizes
    single-threaded) on a bounded sample of the headline workload's shape.  Output arrays are
    allocated and touched BEFORE the clock starts: the timed region is the oracle's work only."""
    import numpy as np
    import oracle_lib
    from oracle_lib import _p
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(SEED + 2)
    sx, sy, sz = oracle_lib.ENC_STRIDE[oracle_lib.PACKED]
    kxs, kys, kzs = oracle_lib.KEY_STRIDE[oracle_lib.PACKED]

    def buffers(n):
        sizes = [n * sx, n * sy, n * sz, n * 16]
        if per_block_keys:
            sizes += [n * oracle_lib.WORDS_ROWS, n * kxs, n * kys, n * kzs, n * 176]
        bufs = [np.empty(b, np.uint8) for b in sizes]
        for b in bufs:
            b.fill(1)  # first touch: page faults happen here, not inside the timed region
        return bufs

    def once(pt, keys, threads, bufs):
        n = pt.shape[0]
        t0 = time.perf_counter()
        rc = orc.L.aesw_o_encrypt_witness(C.byref(orc.t), _p(pt), _p(keys), 1 if per_block_keys else 0, n, oracle_lib.PACKED,
                                          _p(bufs[0]), _p(bufs[1]), _p(bufs[2]), _p(bufs[3]), threads)
        if per_block_keys and not rc:  # the key-schedule witness of every block's key (src/key_schedule.rs:80-224)
            rc = orc.L.aesw_o_key_schedule_witness(C.byref(orc.t), _p(keys), n, oracle_lib.PACKED, _p(bufs[4]), _p(bufs[5]),
                                                   _p(bufs[6]), _p(bufs[7]), _p(bufs[8]), threads)
        dt = time.perf_counter() - t0
        if rc:
            raise RuntimeError("oracle rc=%d" % rc)
        return dt

    def inputs(n):
        return (rng.integers(0, 256, (n, 16), dtype=np.uint8),
                rng.integers(0, 256, (n, 16) if per_block_keys else 16, dtype=np.uint8))

    probe = 2048
    rate = probe / once(*inputs(probe), 1, buffers(probe))
    n = int(max(4096, min(1 << 20, rate * n_target_seconds)))
    pt, keys = inputs(n)
    bufs = buffers(n)
    dt1 = once(pt, keys, 1, bufs)
    cores = os.cpu_count() or 1
    usable = usable_cpus()
    threads = max(1, min(usable, 256))  # the oracle caps its threads at 256 (oracle/aesw_oracle.c run_jobs)
    # a larger sample for the all-core leg, or thread start-up dominates it
    n_all = int(min(1 << 20, max(n, n * threads // 8)))
    pt_a, keys_a = inputs(n_all)
    bufs_a = buffers(n_all)
    once(pt_a[:4096], keys_a[:4096] if per_block_keys else keys_a, threads, bufs_a)  # one small call first (code and tables warm)
    dtn = once(pt_a, keys_a, threads, bufs_a)
    return {
        "value": n / dt1, "unit": "blocks/s", "cores": 1, "kind": "port",
        "sample": "%d blocks, %s, packed layout, oracle/aesw_oracle.c single thread (%.1f s), outputs pre-allocated and touched" % (
            n, "per-block keys + key-schedule witness" if per_block_keys else "shared key", dt1),
        "all_cores": {"value": n_all / dtn, "cores": threads, "threads": threads, "logical_cpus_of_the_host": cores,
                      "usable_cpus": usable, "sample": "%d blocks in %.2f s" % (n_all, dtn),
                      "note": "threads = the CPUs this job may use (affinity mask and cgroup quota), not the host's logical CPU count"},
    }


def c4_stream(pkg, ctx, pt4, key4, lay, olay, verify=True, samples=6, check_stream=True):
    """BASELINE configs[4] on this rank's GPU: aesw_encrypt_witness_stream over pt4 with a CHEAP consumer -- the
    pt ^ rk0 check on every chunk (what a host's assign loop would at least have to touch) plus a copy of the first
    256 blocks of `samples` chunks.  The oracle comparison of those copies happens AFTER the stream has ended and
    outside every reported time (it is the checker's time, not a host-assign time).
    Consumer to match in the reference: region.assign_advice of src/aes128.rs:176-192, benches/aes128.rs:44-56."""
    import numpy as np
    n4 = pt4.shape[0]
    strides = [pkg.column_stride(lay, c) for c in range(3)]
    chunk = ctx.get_option("chunk_blocks")
    sampled = set(int(v) for v in np.linspace(0, -(-n4 // chunk) - 1, samples).astype(int))
    bad = [0]
    kept = []

    def consume(first, count, x, y, z):
        p = pt4[first:first + count]
        if not np.array_equal(z.reshape(count, strides[2])[:, :16], p ^ key4):
            bad[0] += 1
        if first // chunk in sampled:
            m = min(count, 256)
            kept.append((first, m, [np.array(c[:m * s_]) if s_ else None for c, s_ in ((x, strides[0]), (y, strides[1]), (z, strides[2]))]))
        return 0

    ctx.encrypt_witness_stream(pt4[:min(n4, 1 << 16)], None, lambda *args: 0, layout=lay)  # sizes the context's buffers
    t0 = time.perf_counter()
    ctx.encrypt_witness_stream(pt4, None, consume, layout=lay)
    dt = time.perf_counter() - t0
    st = ctx.last_stream_stats()
    chk = {"blocks": 0, "satisfied": True}
    if check_stream and lay != pkg.LAYOUT_VALUES:
        # the same stream once more with option "stream_check": every chunk checked on the device behind its kernel
        # (aesw_check_witness_device) on its way to the host -- all n4 blocks certified; timed by itself so that the figure above
        # stays comparable with earlier rounds
        ctx.set_option("stream_check", 1)
        try:
            t1 = time.perf_counter()
            ctx.encrypt_witness_stream(pt4, None, lambda *args: 0, layout=lay)
            chk = dict(ctx.last_stream_check(), seconds=time.perf_counter() - t1)
        finally:
            ctx.set_option("stream_check", 0)
        if not chk["satisfied"] or chk["blocks"] != n4:
            bad[0] += 1
    per = sum(strides)
    if verify:
        import oracle_lib
        orc = oracle_lib.Oracle()
        for first, m, cols in kept:
            e = orc.encrypt_witness(pt4[first:first + m], key4, layout=olay)
            for got, exp in zip(cols, (e.x, e.y, e.z)):
                if got is not None and not np.array_equal(got, exp):
                    bad[0] += 1
    return {"blocks": n4, "blocks_per_s": n4 / dt, "seconds": dt, "GBps_to_host": n4 * per / dt / 1e9,
            "kernel_s": st["kernel_ns"] * 1e-9, "kernel_blocks_per_s": n4 / (st["kernel_ns"] * 1e-9),
            "d2h_s": st["d2h_ns"] * 1e-9, "d2h_GBps": st["bytes_to_host"] / (st["d2h_ns"] * 1e-9) / 1e9,
            "consumer_s": st["consumer_ns"] * 1e-9, "wait_s": st["wait_ns"] * 1e-9, "chunks": st["chunks"],
            "sampled_chunks_verified_after_the_stream": len(kept) if verify else 0, "mismatches": bad[0],
            "stream_check": {k_: v_ for k_, v_ in chk.items() if k_ != "first"}}


class Watchdog:
    """A stalled collective must not look like success: each phase of the N>1 tail gets its own budget; on expiry
    rank 0 prints the line with the phase that stalled and EVERY rank exits non-zero."""

    def __init__(self, rank, line):
        self.rank, self.line, self.timer, self.phase, self.optional_key = rank, line, None, None, None

    def arm(self, phase, seconds, optional_key=None):
        """optional_key: the phase is an optional extra measured after everything else (the C ABI's own RCCL gather, never run
        with real ranks before): a stall is recorded under line[optional_key] as a non-fatal error and the job still ends with
        status 0 -- the scaling measurement it rides on is already in the line and must not be voided by it."""
        self.disarm()
        self.phase, self.optional_key = phase, optional_key
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self):
        if self.optional_key is not None:
            if self.rank == 0:
                self.line[self.optional_key] = {"error": "phase '%s' timed out" % self.phase, "fatal": False}
                print(json.dumps(self.line), flush=True)
            os._exit(0)
        if self.rank == 0:
            self.line["error"] = "phase '%s' timed out" % self.phase
            print(json.dumps(self.line), flush=True)
        os._exit(3)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        print("bench: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, a.gpus), file=sys.stderr)
    import torch
    import __graft_entry__ as ge
    ge.build()
    pkg = ge.load_package()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dist = None
    if world > 1 or os.environ.get("AESW_BENCH_FORCE_DIST"):  # the env switch rehearses the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if "AESW_BENCH_DEVICE" in os.environ:  # rehearsal: several ranks on one GPU (gloo only)
            local_rank = int(os.environ["AESW_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)
    dev = local_rank if dist is not None else 0
    torch.cuda.set_device(dev)
    ctx = pkg.Context(dev)
    for opt in a.option:
        k, v = opt.split("=")
        ctx.set_option(k, int(v))
    layout = {"packed": pkg.LAYOUT_PACKED, "dense": pkg.LAYOUT_DENSE, "values": pkg.LAYOUT_VALUES}[a.layout]
    host_bufs = None
    if rank == 0 and dist is None and not a.no_extras:
        # Page-locked buffers of the PCIe-inclusive extras are taken first, as a host would at start-up.
        try:
            nn_host = 1 << 20
            host_bufs = (pkg.api.host_alloc(nn_host * 16).reshape(nn_host, 16),
                         [pkg.api.host_alloc(nn_host * pkg.column_stride(pkg.LAYOUT_PACKED, c)) for c in range(3)])
        except Exception:
            host_bufs = None

    def barrier():
        if dist is not None:
            dist.barrier()

    pbk = a.workload == "c2"
    lg = a.log2_blocks or (20 if pbk else 16)
    n = 1 << lg
    # the per-rank workload at N>1 IS the N=1 headline workload, so the driver's 1/2/4/8 curve compares like with like
    # "auto" = on: the probed arena beat one tensor per column on every lease of profiles/r03_study/arena_ab.md (0.80-0.86 against 0.68-0.76)
    use_arena = a.arena != "off"
    runner = Runner(pkg, ctx, torch, n, pbk, layout, pbk, SEED + (2 if pbk else 1) + rank, arena=use_arena)
    wl = ("2^%d blocks, per-block keys (+ key-schedule witness), %s advice columns (BASELINE configs[2])" if pbk else
          "2^%d blocks, one shared key, %s advice columns (BASELINE configs[1])") % (lg, a.layout)
    runner.prepare(a.steps, a.warmup, not a.no_graph)
    graphed = runner.graph is not None
    replays = []
    for _ in range(max(1, a.replays)):
        wall, ms_launch = runner.timed(barrier)
        if dist is not None:
            t = torch.tensor([wall, ms_launch], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, ms_launch = float(t[0]), float(t[1])
        replays.append((wall, ms_launch))
    order = sorted(range(len(replays)), key=lambda i: replays[i][0])
    wall, ms_launch = replays[order[len(order) // 2]]  # the median replay: exactly K steps
    total_blocks = n * world * a.steps
    value = total_blocks / wall
    bytes_launch = runner.bytes_per_block * n
    achieved = bytes_launch / (ms_launch * 1e-3) / 1e9
    achieved_wall = bytes_launch * a.steps / wall / 1e9

    # parity gate on what the timed replays wrote (outside the timed region; every rank checks its own shard)
    try:
        gate = parity_gate(runner)
    except Exception as e:
        gate = {"blocks": 0, "mismatches": -1, "error": str(e)}
    if dist is not None:
        gt = torch.tensor([float(gate["mismatches"] != 0), float(gate["blocks"])], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        gate["ranks_with_mismatches"], gate["blocks_all_ranks"] = int(gt[0]), int(gt[1])
    gate_failed = gate["mismatches"] != 0 or gate.get("ranks_with_mismatches", 0) != 0

    traffic = None
    tp = ROOT / "profiles" / "traffic.json"
    if tp.exists():
        try:
            traffic = json.loads(tp.read_text()).get("%s_%s" % (a.workload, a.layout))
        except Exception:
            traffic = None

    line = {
        "metric": "AES-128 block witnesses/sec", "value": value, "unit": "blocks/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall * 1e3 / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": wl, "blocks_per_gpu": n, "layout": a.layout, "sharding": "blocks by rank, no collective",
                   "launch": "hipGraph of %d launches" % a.steps if graphed else "host launches",
                   "output_ring_sets": runner.nsets,
                   "columns": "one probed arena per set (aesw_columns_alloc)" if use_arena else "one tensor per column",
                   "arena_probe": runner.arena_info if use_arena else None,
                   "arena_setup_s": runner.setup_s,
                   "arena_setup_note": ("wall time of aesw_columns_alloc for the %d output set(s): candidate backings built, the store pattern and a "
                                        "linear fill timed on each; paid once per allocation, outside the timed region -- "
                                        "extra.headline_plain_tensors is the same graph into columns that were not probed" % runner.nsets)
                                       if use_arena else "torch.empty per column"},
        "parity_gate": gate,
        "timing": {"replays": len(replays), "reported": "median replay (exactly %d steps)" % a.steps,
                   "ms_per_step_wall": [w * 1e3 / a.steps for w, _ in replays],
                   "ms_per_step_events": [m for _, m in replays],
                   "min_ms_per_step": min(w for w, _ in replays) * 1e3 / a.steps,
                   "max_ms_per_step": max(w for w, _ in replays) * 1e3 / a.steps,
                   "timed_seconds_total": sum(w for w, _ in replays)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "frac_events": achieved / HBM_PEAK_GBPS,
                     "frac_wall": achieved_wall / HBM_PEAK_GBPS,
                     "frac_events_min": bytes_launch / (max(m for _, m in replays) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "frac_events_max": bytes_launch / (min(m for _, m in replays) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "traffic": traffic,
                     "kernel": "aesw::encrypt_kernel", "launch_ms": ms_launch,
                     "algorithmic_bytes_per_block": runner.bytes_per_block,
                     "algorithmic_bytes_per_launch": bytes_launch,
                     "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc WRITE_SIZE + 2 x FETCH_SIZE)",
                     "traffic_source": "committed PMC passes of this command (profiles/traffic.json), not collected in this run"},
        "achieved_hbm_GBps_all_gpus": achieved * world,
    }

    if gate_failed:
        line["error"] = "parity gate failed: the timed launches' output differs from the oracle"
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        sys.exit(5)

    extras = {}
    if rank == 0 and dist is None and not a.no_extras:
        import numpy as np
        if use_arena:
            # the SAME K-step graph into columns nobody probed (one torch tensor per column), same process, same inputs: what a
            # caller that brings its own buffers gets on this lease, next to the probed figure above
            try:
                rp = Runner(pkg, ctx, torch, n, pbk, layout, pbk, SEED + (2 if pbk else 1) + rank, arena=False)
                rp.prepare(a.steps, a.warmup, not a.no_graph)
                runs = sorted((rp.timed() for _ in range(max(1, a.replays))), key=lambda t: t[0])
                w_p, ms_p = runs[len(runs) // 2]
                ach = rp.bytes_per_block * n / (ms_p * 1e-3) / 1e9
                extras["headline_plain_tensors"] = {
                    "blocks_per_s": n * a.steps / w_p, "ms_per_step": w_p * 1e3 / a.steps, "launch_ms": ms_p, "achieved_GBps": ach,
                    "frac": ach / HBM_PEAK_GBPS, "setup_s": rp.setup_s, "parity_gate": parity_gate(rp),
                    "note": "the headline workload and graph, output columns = one torch.empty per column (no aesw_columns_alloc): "
                            "median of %d replays of exactly %d steps" % (len(runs), a.steps)}
                if extras["headline_plain_tensors"]["parity_gate"]["mismatches"]:
                    line["error"] = "parity gate failed (plain tensors)"
                rp.close()
                del rp
            except Exception as e:
                extras["headline_plain_tensors"] = {"error": str(e)}
        runner.close()
        del runner
        torch.cuda.empty_cache()
        # every host-path extra schedules its own key (the headline runner has per-block keys: nothing scheduled yet)
        hkey = torch.from_numpy(np.random.default_rng(SEED + 5).integers(0, 256, 16, dtype=np.uint8)).cuda()
        ctx.schedule_key(hkey, layout=pkg.LAYOUT_PACKED, key_slab=False)
        torch.cuda.synchronize()
        try:  # PCIe-inclusive rate of the host-pointer entry point (never `value`).  Runs before the large
            # runners below: for a few hundred ms after gigabytes of device memory are freed, device-to-host
            # copies run at ~36 GB/s instead of ~55 GB/s (tools/pinned_probe.py).
            time.sleep(0.3)
            nn = 1 << 20
            if host_bufs is None:
                raise RuntimeError("page-locked host buffers unavailable")
            hpt, pinned_outs = host_bufs   # page-locked input, like the outputs
            hpt[:] = np.random.default_rng(SEED).integers(0, 256, (nn, 16), dtype=np.uint8)
            res = {"blocks": nn, "note": "aesw_encrypt_witness: H2D + kernels + overlapped D2H, packed layout, 2^15-block chunks"}
            for kind in ("pinned", "pageable", "pageable_1_copy_thread"):
                if kind == "pinned":
                    outs = pinned_outs
                else:
                    outs = [np.empty(nn * pkg.column_stride(pkg.LAYOUT_PACKED, c), np.uint8) for c in range(3)]
                ctx.set_option("copy_threads", 1 if kind == "pageable_1_copy_thread" else -1)
                ctx.encrypt_witness_host(hpt, None, layout=pkg.LAYOUT_PACKED, out_cols=outs)  # sizes the context's buffers
                dts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    ctx.encrypt_witness_host(hpt, None, layout=pkg.LAYOUT_PACKED, out_cols=outs)
                    dts.append(time.perf_counter() - t0)
                dt = sorted(dts)[1]  # median of three calls
                res[kind] = {"blocks_per_s": nn / dt, "GBps_to_host": nn * 3024 / dt / 1e9}
                if kind == "pageable":
                    res[kind]["copy_threads"] = ctx.get_option("effective_copy_threads")
                del outs
            ctx.set_option("copy_threads", -1)
            vouts = [np.empty(0, np.uint8), pinned_outs[1][:nn * 448], pinned_outs[2][:nn * 608]]
            ctx.encrypt_witness_host(hpt, None, layout=pkg.LAYOUT_VALUES, out_cols=vouts)
            dts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.encrypt_witness_host(hpt, None, layout=pkg.LAYOUT_VALUES, out_cols=vouts)
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[1]
            res["pinned_values_layout"] = {"blocks_per_s": nn / dt, "GBps_to_host": nn * 1056 / dt / 1e9}
            pkg.api.host_free(hpt)
            for o in pinned_outs:
                pkg.api.host_free(o)
            del pinned_outs, hpt, vouts
            # the link itself: one page-locked 1 GiB device-to-host copy, for scale
            dsrc = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            hdst = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
            hdst.copy_(dsrc, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                hdst.copy_(dsrc, non_blocking=True)
            torch.cuda.synchronize()
            res["raw_d2h_GBps"] = 3 * (1 << 30) / (time.perf_counter() - t0) / 1e9
            del dsrc, hdst
            extras["pcie_inclusive"] = res
        except Exception as e:
            extras["pcie_inclusive"] = {"error": str(e)}
        try:  # BASELINE configs[4] on one GPU: 2^24 blocks streamed to a checking consumer, the parts timed separately
            import oracle_lib
            n4 = 1 << a.c4_log2_blocks
            pt4 = np.random.default_rng(SEED + 4).integers(0, 256, (n4, 16), dtype=np.uint8)
            key4 = hkey.cpu().numpy()
            c4 = {"blocks": n4, "note": "aesw_encrypt_witness_stream, scheduled key: chunk i+1's kernel and D2H overlap the consumer of chunk i; "
                                        "kernel_s / d2h_s are device time summed over chunks (HIP events), consumer_s / wait_s host time; "
                                        "the consumer checks pt ^ rk0 on every chunk and keeps 256 blocks of 6 sampled chunks, which are "
                                        "compared with the oracle after the stream has ended (outside every reported time)"}
            for name, lay, olay in (("values", pkg.LAYOUT_VALUES, oracle_lib.VALUES), ("packed", pkg.LAYOUT_PACKED, oracle_lib.PACKED)):
                c4[name] = c4_stream(pkg, ctx, pt4, key4, lay, olay)
            del pt4
            extras["c4"] = c4
        except Exception as e:
            extras["c4"] = {"error": str(e)}
        try:  # whole Fr advice columns of a K=20, N=5 circuit delivered to the host, column by column
            k, n_sets = 20, 5
            nn = pkg.block_capacity(k, n_sets)
            apt = torch.randint(0, 256, (nn, 16), dtype=torch.uint8, device="cuda")
            kw = ctx.schedule_key(hkey, layout=pkg.LAYOUT_PACKED, key_slab=True)
            wit = ctx.encrypt_witness(apt, None, layout=pkg.LAYOUT_PACKED)
            torch.cuda.synchronize()
            sink = np.empty((1 << k, 32), np.uint8)  # the host's advice polynomial buffer: one bulk copy per column

            def take(col, cells):
                np.copyto(sink, cells)
                return 0

            ctx.assemble_advice_stream(k, n_sets, wit, kw, nn, lambda col, cells: 0, layout=pkg.LAYOUT_PACKED, as_fr=True)
            t0 = time.perf_counter()
            ctx.assemble_advice_stream(k, n_sets, wit, kw, nn, take, layout=pkg.LAYOUT_PACKED, as_fr=True)
            dt = time.perf_counter() - t0
            st = ctx.last_stream_stats()
            # and without the consumer's copy: DMA straight into the host's (page-locked) advice buffer
            whole = pkg.api.host_alloc(16 * (32 << k))
            ctx.assemble_advice_host(k, n_sets, wit, kw, nn, whole, layout=pkg.LAYOUT_PACKED, as_fr=True)
            t0 = time.perf_counter()
            ctx.assemble_advice_host(k, n_sets, wit, kw, nn, whole, layout=pkg.LAYOUT_PACKED, as_fr=True)
            dt_direct = time.perf_counter() - t0
            pkg.api.host_free(whole)
            del whole
            # ... and into an ordinary (pageable) array: every column crosses the bounce buffer, moved on by copy_threads threads
            plain = np.empty(16 * (32 << k), np.uint8)
            ctx.assemble_advice_host(k, n_sets, wit, kw, nn, plain, layout=pkg.LAYOUT_PACKED, as_fr=True)
            t0 = time.perf_counter()
            ctx.assemble_advice_host(k, n_sets, wit, kw, nn, plain, layout=pkg.LAYOUT_PACKED, as_fr=True)
            dt_plain = time.perf_counter() - t0
            del plain
            extras["fr_columns_to_host"] = {
                "direct_pageable": {"seconds": dt_plain, "blocks_per_s": nn / dt_plain, "GBps_to_host": 16 * (32 << k) / dt_plain / 1e9,
                                    "copy_threads": ctx.get_option("effective_copy_threads"),
                                    "note": "aesw_assemble_advice_host into an ordinary numpy array"},
                "direct_pinned": {"seconds": dt_direct, "blocks_per_s": nn / dt_direct, "GBps_to_host": 16 * (32 << k) / dt_direct / 1e9,
                                  "note": "aesw_assemble_advice_host into one page-locked buffer (aesw_host_alloc / aesw_host_register): no consumer copy"},
                "circuit": "K=20, N=5: 16 advice columns x 2^20 Fr cells (512 MiB), %d blocks" % nn,
                "seconds": dt, "blocks_per_s": nn / dt, "GBps_to_host": st["bytes_to_host"] / dt / 1e9,
                "kernel_s": st["kernel_ns"] * 1e-9, "d2h_s": st["d2h_ns"] * 1e-9, "consumer_s": st["consumer_ns"] * 1e-9,
                "wait_s": st["wait_ns"] * 1e-9,
                "note": "aesw_assemble_advice_stream: assemble column j+1 and copy it while the host bulk-copies column j "
                        "(one 32 MiB memcpy per column instead of 2^20 assign_advice calls); compare host_synthesize"}
            del apt, wit, sink
        except Exception as e:
            extras["fr_columns_to_host"] = {"error": str(e)}
        for name, nn, xpbk, lay in (("c1_packed", 1 << 16, False, pkg.LAYOUT_PACKED),
                                    ("c1_packed_2p20", 1 << 20, False, pkg.LAYOUT_PACKED),
                                    ("c1_values", 1 << 16, False, pkg.LAYOUT_VALUES),
                                    ("c1_values_2p20", 1 << 20, False, pkg.LAYOUT_VALUES),
                                    ("c1_dense", 1 << 16, False, pkg.LAYOUT_DENSE),
                                    ("c2_dense", 1 << 20, True, pkg.LAYOUT_DENSE)):
            try:
                r = Runner(pkg, ctx, torch, nn, xpbk, lay, xpbk, SEED + 7, arena=use_arena)
                steps = 50 if nn <= (1 << 16) else 12
                r.prepare(steps, 3, not a.no_graph)
                runs = sorted((r.timed() for _ in range(3)), key=lambda t: t[1])
                w, ms = runs[1]
                extras[name] = {"blocks": nn, "blocks_per_s": nn * steps / w, "launch_ms": ms,
                                "achieved_GBps": r.bytes_per_block * nn / (ms * 1e-3) / 1e9,
                                "frac": r.bytes_per_block * nn / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                "written_GBps": r.out_bytes_per_step / (ms * 1e-3) / 1e9}
                r.close()
                del r
                torch.cuda.empty_cache()
            except Exception as e:  # keep the headline even if an extra fails
                extras[name] = {"error": str(e)}
        try:  # independent batches on several streams: ramp and tail of a launch overlap its neighbours
            ov = {"note": "microseconds per batch (graph time / batches) of independent batches through aesw_encrypt_witness_batches_device "
                          "with batch_streams = 1, 2 (, 3), captured into one hipGraph; shared key by pointer; the headline above is "
                          "the 1-stream form, as its roofline entry is defined per kernel"}
            for name, nn, xpbk, steps, counts in (("c1_packed", 1 << 16, False, 200, (1, 2, 3)), ("c2_packed", 1 << 20, True, 40, (1, 2))):
                row = {}
                bpb = BYTES_PBK if xpbk else BYTES_SHARED
                for br, us in overlapped_launches(pkg, ctx, torch, nn, xpbk, counts, steps, use_arena).items():
                    row["streams_%d" % br] = {"us_per_launch": us, "blocks_per_s": nn / us * 1e6,
                                              "achieved_GBps": bpb * nn / us / 1e3, "frac": bpb * nn / us / 1e3 / HBM_PEAK_GBPS}
                torch.cuda.empty_cache()
                ov[name] = row
            extras["overlapped_batches"] = ov
        except Exception as e:
            extras["overlapped_batches"] = {"error": str(e)}
        try:  # where the end-to-end time goes: the host's synthesize() assigning the device witness cell by cell
            k, n_sets = 20, 3
            nn = pkg.block_capacity(k, n_sets)
            hpt = np.random.default_rng(SEED + 3).integers(0, 256, (nn, 16), dtype=np.uint8)
            hk = np.random.default_rng(SEED + 4).integers(0, 256, 16, dtype=np.uint8)
            res = {}
            for label, kw in (("blocks_per_s", {}), ("bulk_assign_blocks_per_s", {"bulk_assign": True}),
                              ("streaming_values_only_blocks_per_s", {"streaming": True})):
                t0 = time.perf_counter()
                hc = pkg.HostCircuit.aes(ctx, k, n_sets, hk, hpt, **kw)
                res[label] = nn / (time.perf_counter() - t0)
                if not kw:
                    res["regions_per_s"] = hc.num_regions * res[label] / nn
                hc.close()
            t0 = time.perf_counter()
            hc = pkg.HostCircuit.aes_columns(ctx, k, n_sets, hk, hpt)
            res["whole_columns_blocks_per_s"] = nn / (time.perf_counter() - t0)
            hc.close()
            res["circuit"] = "FixedAes128Config<20,3>, %d blocks (full)" % nn
            res["what"] = "C++ stand-in for the halo2 front end (libaesw_host.so, halo2_lite.hpp), NOT halo2 and not the reference's synthesize()"
            res["note"] = ("C++ host mirror (libaesw_host.so): table + schedule_key + encrypt() per block, one thread, packed device "
                           "witness generation included (negligible); halo2's own layouter and Fr arithmetic are slower than this stand-in")
            extras["host_synthesize"] = res
        except Exception as e:
            extras["host_synthesize"] = {"error": str(e)}
        try:  # configs[3] / [4] are 2^24 blocks: generate them on this GPU in 2^20-block chunks and check EVERY block's constraints
            nn, chunks = 1 << 20, 16
            g24 = torch.Generator(device="cuda").manual_seed(SEED + 24)
            w24 = ctx.alloc_columns(nn, pkg.LAYOUT_PACKED, key_slab=True) if use_arena else ctx.alloc_witness(nn, pkg.LAYOUT_PACKED, key_slab=True, n_keys=nn)
            tot = [0] * 6
            t_gen = t_chk = 0.0
            for ci in range(chunks):
                dpt24 = torch.randint(0, 256, (nn, 16), dtype=torch.uint8, device="cuda", generator=g24)
                dk24 = torch.randint(0, 256, (nn, 16), dtype=torch.uint8, device="cuda", generator=g24)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                ctx.encrypt_witness(dpt24, dk24, layout=pkg.LAYOUT_PACKED, out=w24, key_slab=True)
                ev[1].record()
                rep24 = ctx.check_witness(dpt24, dk24, w24, w24.key, layout=pkg.LAYOUT_PACKED, sync=False)
                ev[2].record()
                torch.cuda.synchronize()
                t_gen += ev[0].elapsed_time(ev[1])
                t_chk += ev[1].elapsed_time(ev[2])
                tot = [a_ + int(b_) for a_, b_ in zip(tot, rep24.cpu().tolist()[:6])]
            if use_arena:
                ctx.free_columns(w24)
            del w24
            extras["device_check_2p24"] = {
                "blocks": tot[0], "keys": tot[1], "lookup_failures": tot[2], "copy_failures": tot[3], "gate_failures": tot[4], "input_failures": tot[5],
                "satisfied": not any(tot[2:]), "generate_ms": t_gen, "check_ms": t_chk, "check_blocks_per_s": tot[0] / (t_chk * 1e-3),
                "check_read_GBps": tot[0] * BYTES_PBK / (t_chk * 1e-3) / 1e9,
                "note": "2^24 blocks with per-block keys (the size of BASELINE configs[3] / [4]) generated in 16 launches of 2^20 and every block's "
                        "1 760 lookup rows, 2 592 copies, gate and literal rows checked on the device (aesw_check_witness_device)"}
            if not extras["device_check_2p24"]["satisfied"]:
                line["error"] = "device check failed over 2^24 blocks"
        except Exception as e:
            extras["device_check_2p24"] = {"error": str(e)}
        try:  # byte cells -> 32-byte Fr cells
            cells = torch.randint(0, 256, (1 << 26,), dtype=torch.uint8, device="cuda")
            out = torch.empty((1 << 26, 32), dtype=torch.uint8, device="cuda")
            ctx.expand_fr(cells, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.expand_fr(cells, out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            extras["expand_fr"] = {"cells": 1 << 26, "launch_ms": ms, "written_GBps": (1 << 26) * 32 / (ms * 1e-3) / 1e9}
            del cells, out
        except Exception as e:
            extras["expand_fr"] = {"error": str(e)}
        try:  # the key-schedule kernel on its own: 936 B written + 16 B read per key
            nk = 1 << 20
            dkeys = torch.randint(0, 256, (nk, 16), dtype=torch.uint8, device="cuda")
            karena = [ctx.alloc_columns(nk, pkg.LAYOUT_PACKED, key_only=True) for _ in range(2)] if use_arena else None
            kouts = [ka.key for ka in karena] if karena else [None, None]  # two output sets (2 x 0.98 GB > the Infinity Cache)
            ctx.key_schedule_witness(dkeys, layout=pkg.LAYOUT_PACKED, want_rk=False, out=kouts[0])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(10):
                kwit = ctx.key_schedule_witness(dkeys, layout=pkg.LAYOUT_PACKED, want_rk=False, out=kouts[i & 1])
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            for ka in karena or []:
                ctx.free_columns(ka)
            extras["key_schedule"] = {"keys": nk, "launch_ms": ms, "keys_per_s": nk / (ms * 1e-3),
                                      "achieved_GBps": 952 * nk / (ms * 1e-3) / 1e9, "frac": 952 * nk / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            del dkeys, kwit
        except Exception as e:
            extras["key_schedule"] = {"error": str(e)}
        line["extra"] = extras
    if rank == 0 and dist is None and not a.no_cpu:
        try:
            line["cpu_baseline"] = cpu_baseline(pbk)
        except Exception as e:  # the headline must reach the driver whatever happens to a reported baseline
            line["cpu_baseline"] = {"error": str(e)}
    elif rank == 0:
        line["cpu_baseline"] = None
    if dist is not None:
        # optional exchange step, timed separately (never part of `value`): gather every rank's columns on rank 0
        # through the C ABI's RCCL gather.  A per-phase watchdog reports a stalled exchange and exits NON-ZERO.
        dog = Watchdog(rank, line)
        strides = [pkg.column_stride(layout, c) for c in range(3)]
        failed = []  # phases of the N>1 tail that raised on THIS rank: the process then exits non-zero (after printing)

        def timed_gather(wset, nblk, cabi=a.gather_path == "cabi"):
            gcols = [wset.x, wset.y, wset.z] if a.backend == "nccl" else [c.cpu() for c in (wset.x, wset.y, wset.z)]
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            full = pkg.sharding.gather_columns(gcols, [nblk] * world, strides, dst=0, ctx=ctx if a.backend == "nccl" else None,
                                               force_torch=not cabi)
            torch.cuda.synchronize()
            dist.barrier()
            dt = time.perf_counter() - t0
            del full
            return dt

        # Order of the tail: configs[4] first (every rank for itself: barriers and two small all-reduces are its only collectives),
        # then the exchange steps whose RCCL leg has never run with real ranks -- if one of those stalls, the watchdog's
        # line already carries c4.
        if not a.no_extras:
            # BASELINE configs[4] at N > 1: every rank streams its own shard to its own host over its own PCIe link (DESIGN 7:
            # the right delivery path -- no gather), kernel + async D2H overlapped with a cheap consumer; ranks timed between
            # barriers, MAX over ranks.  Consumer to match: region.assign_advice, src/aes128.rs:176-192.
            try:
                import numpy as np
                import oracle_lib
                torch.cuda.empty_cache()
                n4 = 1 << a.c4_rank_log2_blocks
                dog.arm("c4 set-up", 180.0)
                hkey4 = torch.from_numpy(np.random.default_rng(SEED + 5).integers(0, 256, 16, dtype=np.uint8)).cuda()
                ctx.schedule_key(hkey4, layout=pkg.LAYOUT_PACKED, key_slab=False)
                torch.cuda.synchronize()
                pt4 = np.random.default_rng(SEED + 40 + rank).integers(0, 256, (n4, 16), dtype=np.uint8)
                key4 = hkey4.cpu().numpy()
                ctx.encrypt_witness_stream(pt4[:1 << 16], None, lambda *args: 0, layout=pkg.LAYOUT_PACKED)  # buffers sized, untimed
                c4n = {}
                for name, lay, olay in (("packed", pkg.LAYOUT_PACKED, oracle_lib.PACKED), ("values", pkg.LAYOUT_VALUES, oracle_lib.VALUES)):
                    dog.arm("c4 streaming (%s)" % name, 300.0)
                    torch.cuda.synchronize()
                    dist.barrier()
                    t0 = time.perf_counter()
                    r4 = c4_stream(pkg, ctx, pt4, key4, lay, olay, verify=False, check_stream=False)
                    torch.cuda.synchronize()
                    t_rank = time.perf_counter() - t0
                    dist.barrier()
                    dev4 = "cuda" if a.backend == "nccl" else "cpu"
                    tmax = torch.tensor([t_rank, r4["kernel_s"], r4["d2h_s"], r4["consumer_s"], r4["wait_s"]], dtype=torch.float64, device=dev4)
                    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                    chk4 = {"blocks": 0, "satisfied": True}
                    if lay != pkg.LAYOUT_VALUES:
                        # untimed: the same shard streamed once more with "stream_check": every block certified on the device on its way out
                        ctx.set_option("stream_check", 1)
                        try:
                            ctx.encrypt_witness_stream(pt4, None, lambda *args: 0, layout=lay)
                            chk4 = ctx.last_stream_check()
                        finally:
                            ctx.set_option("stream_check", 0)
                    bad4 = r4["mismatches"] + (0 if (chk4["satisfied"] and chk4["blocks"] in (0, n4)) else 1)
                    tsum = torch.tensor([float(bad4), r4["d2h_GBps"], float(chk4["blocks"])], dtype=torch.float64, device=dev4)
                    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
                    per = sum(pkg.column_stride(lay, c) for c in range(3))
                    c4n[name] = {"blocks_per_rank": n4, "blocks_total": n4 * world, "seconds_max_over_ranks": float(tmax[0]),
                                 "blocks_per_s_to_hosts": n4 * world / float(tmax[0]),
                                 "GBps_to_hosts_all_ranks": n4 * world * per / float(tmax[0]) / 1e9,
                                 "d2h_GBps_per_rank_mean": float(tsum[1]) / world,
                                 "kernel_s_max": float(tmax[1]), "d2h_s_max": float(tmax[2]), "consumer_s_max": float(tmax[3]),
                                 "wait_s_max": float(tmax[4]), "mismatches": int(tsum[0]), "stream_check_blocks_all_ranks": int(tsum[2]),
                                 "rank0": {k: r4[k] for k in ("seconds", "kernel_s", "d2h_s", "d2h_GBps", "consumer_s", "wait_s", "chunks")}}
                dog.disarm()
                if rank == 0:
                    c4n["note"] = ("every rank: aesw_encrypt_witness_stream over its own 2^%d-block shard into its own page-locked buffers "
                                   "(scheduled key); the consumer checks pt ^ rk0 on every chunk (no oracle inside the timed part); time = "
                                   "barrier .. stream end, MAX over ranks" % a.c4_rank_log2_blocks)
                    line["c4"] = c4n
                del pt4
            except Exception as e:
                dog.disarm()
                failed.append("c4")
                if rank == 0:
                    line["c4"] = {"error": str(e)}
        try:
            dog.arm("gather warm-up (RCCL communicator + peer channel set-up)", 120.0)
            timed_gather(runner.sets[0], n)
            dog.arm("gather", 60.0)
            dt = timed_gather(runner.sets[0], n)
            dog.disarm()
            if rank == 0:
                line["gather"] = {"seconds": dt, "GBps_into_root": (world - 1) * n * sum(strides) / dt / 1e9,
                                  "path": pkg.sharding.last_gather_path,
                                  "note": "per-rank column ranges gathered on rank 0, outside `value`"}
        except Exception as e:
            dog.disarm()
            failed.append("gather")
            if rank == 0:
                line["gather"] = {"error": str(e)}
        if not a.no_extras:
            # BASELINE configs[3]: 2^24 blocks over 8 GPUs = 2^21 per GPU, columns gathered on GPU 0 (never `value`)
            try:
                runner.close()
                del runner
                torch.cuda.empty_cache()
                n3 = 1 << (a.c3_log2_blocks or 21)
                dog.arm("c3 generation", 120.0)
                r3 = Runner(pkg, ctx, torch, n3, False, layout, False, SEED + 11 + rank, arena=use_arena)
                steps3 = 10
                w3, ms3, _ = r3.run(steps3, 2, not a.no_graph, barrier)
                t3 = torch.tensor([w3, ms3], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
                dist.all_reduce(t3, op=dist.ReduceOp.MAX)
                dog.arm("c3 gather", 120.0)
                dt3 = timed_gather(r3.sets[0], n3)
                dog.disarm()
                if rank == 0:
                    line["c3"] = {"workload": "2^%d blocks per GPU x %d GPUs, one shared key, %s columns" % (a.c3_log2_blocks or 21, world, a.layout),
                                  "blocks_total": n3 * world, "blocks_per_s": n3 * world * steps3 / float(t3[0]),
                                  "launch_ms": float(t3[1]),
                                  "achieved_GBps_per_gpu": BYTES_SHARED * n3 / (float(t3[1]) * 1e-3) / 1e9,
                                  "gather_seconds": dt3, "gather_GBps_into_root": (world - 1) * n3 * sum(strides) / dt3 / 1e9,
                                  "gather_path": pkg.sharding.last_gather_path,
                                  "note": "generation and gather timed separately; the gather is bound by the root's xGMI ingest"}
                r3.close()
                del r3
            except Exception as e:
                dog.disarm()
                failed.append("c3")
                if rank == 0:
                    line["c3"] = {"error": str(e)}
        if not a.no_extras and a.backend == "nccl" and a.gather_path != "cabi":
            # LAST, because it is the one step that has never run with real ranks: the C ABI's own RCCL gather (what a host without
            # torch calls) over 2^20-block columns.  Everything measured so far is already in the line: a stall
            # here ends the job through the watchdog with that line printed and gather_cabi = {error, fatal: false} (status 0: an
            # optional extra must not void the scaling measurement it rides on); an exception is recorded the same way.
            try:
                dog.arm("gather_cabi warm-up (second RCCL communicator beside torch's: ncclCommInitRank + peer channels)", 120.0, optional_key="gather_cabi")
                ng = 1 << 20
                wg = ctx.alloc_witness(ng, layout)
                timed_gather(wg, ng, cabi=True)
                dog.arm("gather_cabi", 60.0, optional_key="gather_cabi")
                dtg = timed_gather(wg, ng, cabi=True)
                dog.disarm()
                if rank == 0:
                    line["gather_cabi"] = {"seconds": dtg, "GBps_into_root": (world - 1) * ng * sum(strides) / dtg / 1e9,
                                           "path": pkg.sharding.last_gather_path, "blocks_per_rank": ng,
                                           "note": "aesw_gather_columns_device: ncclSend / ncclRecv in one group on the caller's stream"}
                del wg
            except Exception as e:
                dog.disarm()
                if rank == 0:
                    line["gather_cabi"] = {"error": str(e), "fatal": False}
        dog.arm("shutdown", 60.0)
        try:
            # a phase that raised on ANY rank fails the whole job: agree on it before leaving
            ft = torch.tensor([float(len(failed))], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(ft, op=dist.ReduceOp.SUM)
            any_failed = float(ft[0]) > 0
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            any_failed = True
        dog.disarm()  # cancels the last timer thread
        try:
            ctx.close()
        except Exception:
            pass
        if rank == 0:
            if any_failed:
                line["error"] = "N>1 tail failed: %s" % (", ".join(failed) or "on another rank")
            print(json.dumps(line), flush=True)
        if any_failed:
            sys.exit(4)
        return
    if rank == 0:
        print(json.dumps(line), flush=True)
    # leave nothing behind: the context's streams, events, cached arenas and page-locked buffers go with it; the library's
    # copy threads are joined inside every call; no watchdog timer exists at N = 1
    try:
        ctx.close()
    except Exception:
        pass
    if line.get("error"):
        sys.exit(5)


if __name__ == "__main__":
    main()
