"""Round-3 GPU tests: the arena allocator of the C ABI (aesw_columns_alloc), options read back, the refusal to
capture a scheduled-key launch without its dependency, and the division-free one-shot assemble kernel against the
restated synthesize()."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout_name", ["packed", "dense", "values"])
def test_arena_columns_hold_a_byte_exact_witness(ctx, pkg, oracle, layout_name):
    """aesw_columns_alloc: one allocation, aligned column bases, and a launch into it equals the oracle
    (per-block keys + key witness + ciphertext, ragged batch)."""
    import torch
    lay = {"packed": pkg.LAYOUT_PACKED, "dense": pkg.LAYOUT_DENSE, "values": pkg.LAYOUT_VALUES}[layout_name]
    olay = {"packed": ol.PACKED, "dense": ol.DENSE, "values": ol.VALUES}[layout_name]
    rng = np.random.default_rng(31)
    n = 3000 + 17
    pt, keys = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, (n, 16), dtype=np.uint8)
    assert ctx.get_option("arena_probe") == -1  # auto: no probing for a batch this small (< 2^16 blocks), one hipMalloc
    for align_log2, expect in ((0, 2 << 20), (12, 1 << 12), (21, 2 << 20)):  # auto = 2 MiB
        ctx.set_option("arena_align_log2", align_log2)
        assert ctx.get_option("arena_align_log2") == align_log2
        w = ctx.alloc_columns(n, lay, want_ct=True, key_slab=True)
        ptrs = [t.data_ptr() for t in (w.x, w.y, w.z, w.ct, w.key.w, w.key.kx, w.key.ky, w.key.kz) if t.numel()]
        assert all(p % expect == 0 for p in ptrs), (align_log2, [hex(p) for p in ptrs])
        assert len(set(ptrs)) == len(ptrs)
        sizes = [t.numel() for t in (w.x, w.y, w.z)]
        assert sizes == [n * pkg.column_stride(lay, c) for c in range(3)]
        for t in (w.x, w.y, w.z, w.ct, w.key.w, w.key.kx, w.key.ky, w.key.kz):
            if t.numel():
                t.fill_(0xA5)
        got = ctx.encrypt_witness(torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda(), layout=lay, out=w, want_ct=True, key_slab=True)
        torch.cuda.synchronize()
        e = oracle.encrypt_witness(pt, keys, layout=olay)
        k = oracle.key_schedule_witness(keys, layout=olay)
        for c in "xyz":
            if getattr(e, c).size:
                assert np.array_equal(getattr(got, c).cpu().numpy(), getattr(e, c)), (layout_name, c)
        assert np.array_equal(got.ct.cpu().numpy(), e.ct)
        for c in ("w", "kx", "ky", "kz"):
            assert np.array_equal(getattr(got.key, c).cpu().numpy(), getattr(k, c)), (layout_name, c)
        ctx.free_columns(w)
        with pytest.raises(ValueError):
            ctx.free_columns(w)
    ctx.set_option("arena_align_log2", 0)


def test_probed_arena_holds_a_byte_exact_witness(ctx, pkg, oracle):
    """aesw_columns_alloc with placement probing (virtual-memory API, candidates timed with the store-pattern emulation):
    the chosen arena is ordinary device memory -- a launch into it equals the oracle, it can be read back, freed and
    allocated again; the struct reports how many candidates were measured and the two times of the one kept."""
    import torch
    rng = np.random.default_rng(33)
    n = (1 << 18) + 16 * 3 + 5  # auto probing starts at 2^18 blocks
    pt, keys = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, (n, 16), dtype=np.uint8)
    e = oracle.encrypt_witness(pt, keys, layout=ol.PACKED)
    k = oracle.key_schedule_witness(keys, layout=ol.PACKED)
    dpt, dkeys = torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda()
    ctx.set_option("arena_cache", 0)  # this test is about the SEARCH: every allocation below must run one (round 4's placement cache
    # would hand the freed arena of the same shape straight back; tests/test_gpu_round4.py covers that)
    for probe in (-1, 2, 1):
        ctx.set_option("arena_probe", probe)
        w = ctx.alloc_columns(n, pkg.LAYOUT_PACKED, want_ct=True, key_slab=True)
        info = ctx.last_arena
        assert 1 <= info["candidates"] <= 8 * (8 if probe < 0 else probe)  # whole-set candidates, then (maybe) seven columns' worth
        assert info["probe_us"] > 0 and info["fill_us"] > 0 and info["probe_us"] < 3 * info["fill_us"]
        for t in (w.x, w.y, w.z, w.key.w, w.key.kx, w.key.ky, w.key.kz):
            assert t.data_ptr() % (2 << 20) == 0
        got = ctx.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w, want_ct=True, key_slab=True)
        torch.cuda.synchronize()
        for c in "xyz":
            assert np.array_equal(getattr(got, c).cpu().numpy(), getattr(e, c)), c
        assert np.array_equal(got.ct.cpu().numpy(), e.ct)
        for c in ("w", "kx", "ky", "kz"):
            assert np.array_equal(getattr(got.key, c).cpu().numpy(), getattr(k, c)), c
        ctx.free_columns(w)
    # one column per candidate, placed greedily
    ctx.set_option("arena_probe", 2)
    ctx.set_option("arena_unit", 1)
    w = ctx.alloc_columns(n, pkg.LAYOUT_PACKED, want_ct=True, key_slab=True)
    assert 7 <= ctx.last_arena["candidates"] <= 14
    got = ctx.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w, want_ct=True, key_slab=True)
    torch.cuda.synchronize()
    for c in "xyz":
        assert np.array_equal(getattr(got, c).cpu().numpy(), getattr(e, c)), c
    assert np.array_equal(got.ct.cpu().numpy(), e.ct) and np.array_equal(got.key.kz.cpu().numpy(), k.kz)
    ctx.free_columns(w)
    ctx.set_option("arena_unit", 2)
    ctx.set_option("arena_probe", -1)
    ctx.set_option("arena_cache", 1)


def test_every_settable_option_reads_back(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = pkg.Context(0)
    for name, value in (("waves_shared", 2), ("waves_pbk", 3), ("store_mode", 1), ("key_store_mode", 2), ("fr_store_mode", 0),
                        ("fr_geometry", 2), ("grid_cap", 512), ("xcd_remap", 0), ("xcd_remap", 64), ("lds_pad", 4096), ("arena_align_log2", 16), ("arena_probe", 3), ("arena_unit", 1),
                        ("chunk_blocks", 4096), ("key_slots", 7), ("batch_streams", 5), ("copy_threads", 2),
                        ("arena_cache", 0), ("arena_cache_max_mb", 1024), ("arena_probe_budget_ms", 250)):
        c.set_option(name, value)
        assert c.get_option(name) == value, name
    c.set_option("force_table_path", 1)
    assert c.get_option("force_table_path") == 1 and not c.uses_xtime_path
    # the group size a launch really uses: requests above the packed layout's maximum of 3 are limited, 0 = auto resolved
    c.set_option("waves_shared", 4)
    c.set_option("waves_pbk", 0)
    assert c.get_option("waves_shared") == 4 and c.get_option("effective_waves_shared") == 3
    assert c.get_option("effective_waves_pbk") == 1 and c.get_option("effective_waves_key") == 3
    with pytest.raises(pkg.AeswError):
        c.get_option("no_such_option")
    c.close()


# (test_scheduled_key_capture_on_a_foreign_stream_is_refused moved to tests/test_gpu_round4.py with round 4's semantics)


def test_probed_arena_2p22_blocks_per_block_keys_columns_beyond_4_gib(pkg, oracle):
    """2^22 blocks with per-block keys into ONE probed arena: 16.7 GB, the x column alone 5.7 GB, so every offset past 2^32
    and the arena's bookkeeping of several multi-GB candidates are exercised.  8 192 blocks sampled over the whole range
    (first and last included) equal the oracle in all eight columns; the launch's time is printed next to the arena's probe (bound: tests/test_perf.py)."""
    import torch
    n = 1 << 22
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs ~60 GB of free HBM")
    c = pkg.Context(0)
    w = c.alloc_columns(n, pkg.LAYOUT_PACKED, want_ct=True, key_slab=True)
    info = c.last_arena
    assert info["bytes"] >= n * 3976 and info["candidates"] >= 1
    assert w.x.numel() == n * 1360 > 1 << 32
    g = torch.Generator(device="cuda").manual_seed(0xA35128 + 22)
    dpt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
    dkeys = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
    got = c.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w, want_ct=True, key_slab=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    c.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w, want_ct=True, key_slab=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    print("launch %.1f us, probe %.1f us, fill %.1f us" % (us, info["probe_us"], info["fill_us"]))  # the bound on this ratio: tests/test_perf.py
    sample = np.unique(np.concatenate([[0, n - 1], np.random.default_rng(22).integers(0, n, 8190)]))
    ds = torch.from_numpy(sample).cuda()
    pt, keys = dpt[ds].cpu().numpy(), dkeys[ds].cpu().numpy()
    e = oracle.encrypt_witness(pt, keys, layout=ol.PACKED, threads=16)
    k = oracle.key_schedule_witness(keys, layout=ol.PACKED, threads=16)
    for name, stride in (("x", 1360), ("y", 1056), ("z", 608)):
        assert np.array_equal(getattr(got, name).view(n, stride)[ds].cpu().numpy().reshape(-1), getattr(e, name)), name
    assert np.array_equal(got.ct[ds].cpu().numpy(), e.ct)
    for name, stride in (("w", ol.WORDS_ROWS), ("kx", 400), ("ky", 240), ("kz", 200)):
        assert np.array_equal(getattr(got.key, name).view(n, stride)[ds].cpu().numpy().reshape(-1), getattr(k, name)), name
    c.free_columns(w)
    c.close()


def test_pageable_destinations_with_one_and_several_copy_threads(pkg, oracle):
    """The host-pointer path into ordinary (pageable) arrays moves every stage out of the page-locked bounce buffer with
    "copy_threads" host threads in 4 MiB slices: 1, 3 and the automatic count give the oracle's bytes for a batch whose stages
    are several slices long (2^14-block stages: 21 MB of x), ragged at the end, per-block keys + key slabs + ciphertext; the
    Fr columns of a circuit take the same route column by column."""
    c = pkg.Context(0)
    rng = np.random.default_rng(404)
    n = 3 * (1 << 14) + 1234
    pt, keys = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, (n, 16), dtype=np.uint8)
    exp = oracle.encrypt_witness(pt, keys, layout=ol.PACKED, threads=16)
    kexp = oracle.key_schedule_witness(keys, layout=ol.PACKED, threads=16)
    c.set_option("chunk_blocks", 1 << 14)
    assert c.get_option("copy_threads") == -1 and 1 <= c.get_option("effective_copy_threads") <= 4
    for threads in (1, 3, -1):
        c.set_option("copy_threads", threads)
        assert c.get_option("copy_threads") == threads
        got = c.encrypt_witness_host(pt, keys, layout=ol.PACKED, want_ct=True, key_slab=True)
        for col in "xyz":
            assert np.array_equal(getattr(got, col), getattr(exp, col)), (threads, col)
        assert np.array_equal(got.ct, exp.ct)
        for col in ("w", "kx", "ky", "kz"):
            assert np.array_equal(getattr(got.key, col), getattr(kexp, col)), (threads, col)
    with pytest.raises(Exception):
        c.set_option("copy_threads", 65)
    c.close()


def test_arena_that_cannot_fit_fails_cleanly_and_leaks_nothing(pkg):
    """aesw_columns_alloc for more memory than the device has (2^27 blocks with key slabs: 534 GB) returns an error with a
    message instead of crashing, with and without probing; nothing stays allocated, and the next ordinary arena works."""
    import torch
    c = pkg.Context(0)
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    assert total < 400 << 30
    # (arena_probe, arena_unit): auto, no probing, whole sets only, and columns one at a time -- there the 182 GB x column fits
    # and is probed before the next one fails, so the error path has something to give back
    for probe, unit in ((-1, 2), (0, 2), (2, 0), (1, 1)):
        c.set_option("arena_probe", probe)
        c.set_option("arena_unit", unit)
        with pytest.raises(Exception) as ei:
            c.alloc_columns(1 << 27, pkg.LAYOUT_PACKED, want_ct=True, key_slab=True)
        assert "hip" in str(ei.value).lower() or "memory" in str(ei.value).lower(), str(ei.value)
    c.set_option("arena_probe", -1)
    c.set_option("arena_unit", 2)
    free1, _ = torch.cuda.mem_get_info()
    assert free1 >= free0 - (64 << 20), (free0, free1)
    w = c.alloc_columns(1 << 16, pkg.LAYOUT_PACKED, key_slab=True)
    assert w.x.numel() == (1 << 16) * 1360 and c.last_arena["candidates"] >= 1
    c.free_columns(w)
    c.close()


@pytest.mark.parametrize("streams", [1, 3, 8])
def test_batches_entry_point_is_byte_exact_in_every_key_mode(pkg, oracle, streams):
    """aesw_encrypt_witness_batches_device: seven batches of different sizes dealt onto 1 / 3 / 8 internal streams, in each key
    mode (scheduled key, one key by pointer, per-block keys with key slabs and ciphertext); every batch equals the oracle, and
    work queued on the caller's stream before and after the call is ordered around it (fork / join)."""
    import torch
    c = pkg.Context(0)
    c.set_option("batch_streams", streams)
    assert c.get_option("batch_streams") == streams
    rng = np.random.default_rng(500 + streams)
    sizes = [1, 17, 4096, (1 << 14) + 5, 300, 1 << 13, 63]
    skey = rng.integers(0, 256, 16, dtype=np.uint8)
    for mode in ("scheduled", "shared", "per_block"):
        pbk = mode == "per_block"
        if mode == "scheduled":
            c.schedule_key(torch.from_numpy(skey).cuda(), layout=pkg.LAYOUT_PACKED, key_slab=False)
        host, batches = [], []
        for n in sizes:
            pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
            keys = rng.integers(0, 256, (n, 16), dtype=np.uint8) if pbk else (rng.integers(0, 256, 16, dtype=np.uint8) if mode == "shared" else None)
            out = c.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=True, key_slab=pbk, n_keys=n)
            for t in (out.x, out.y, out.z, out.ct):
                t.fill_(0x5A)  # queued on the caller's stream BEFORE the call: the batches must come after it
            host.append((pt, keys))
            batches.append((torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda() if keys is not None else None, out))
        c.encrypt_witness_batches(batches, pbk, layout=pkg.LAYOUT_PACKED)
        copies = [b[2].x.clone() for b in batches]  # queued AFTER the call on the caller's stream: must see the results
        torch.cuda.synchronize()
        for (pt, keys), (_, _, out), cp in zip(host, batches, copies):
            e = oracle.encrypt_witness(pt, keys if keys is not None else skey, layout=ol.PACKED)
            for col in "xyz":
                assert np.array_equal(getattr(out, col).cpu().numpy(), getattr(e, col)), (mode, len(pt), col)
            assert np.array_equal(cp.cpu().numpy(), e.x), (mode, len(pt), "join")
            assert np.array_equal(out.ct.cpu().numpy(), e.ct)
            if pbk:
                k = oracle.key_schedule_witness(keys, layout=ol.PACKED)
                for col in ("w", "kx", "ky", "kz"):
                    assert np.array_equal(getattr(out.key, col).cpu().numpy(), getattr(k, col)), (mode, col)
    with pytest.raises(Exception):
        c.set_option("batch_streams", 9)
    c.close()


def test_batches_entry_point_captures_into_one_graph(pkg, oracle):
    """The internal streams join a capture of the caller's stream through the fork / join events: one hipGraph holding six
    batches on three streams replays byte-exactly (per-block keys)."""
    import torch
    c = pkg.Context(0)
    rng = np.random.default_rng(600)
    n = 5000
    host, batches = [], []
    for _ in range(6):
        pt, keys = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, (n, 16), dtype=np.uint8)
        host.append((pt, keys))
        batches.append((torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda(), c.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=True, key_slab=True, n_keys=n)))
    c.encrypt_witness_batches(batches, True)  # creates the internal streams outside the capture
    torch.cuda.synchronize()
    for _, _, out in batches:
        for t in (out.x, out.y, out.z, out.ct):
            t.zero_()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        c.encrypt_witness_batches(batches, True)
    torch.cuda.synchronize()
    assert int(batches[0][2].x.max()) == 0  # captured, not run
    graph.replay()
    torch.cuda.synchronize()
    for (pt, keys), (_, _, out) in zip(host, batches):
        e = oracle.encrypt_witness(pt, keys, layout=ol.PACKED)
        for col in "xyz":
            assert np.array_equal(getattr(out, col).cpu().numpy(), getattr(e, col)), col
        assert np.array_equal(out.ct.cpu().numpy(), e.ct)
    c.close()


def test_concurrent_launches_on_one_context_are_byte_exact(pkg, oracle):
    """Independent batches may be issued on several streams of one context (bench.py "overlapped_batches": ramp and tail of a
    launch then overlap its neighbours).  Twelve launches -- per-block keys, shared key by pointer and the scheduled key, each
    with its own inputs and outputs -- round-robin on three streams, no synchronisation in between; every output equals the
    oracle."""
    import torch
    c = pkg.Context(0)
    rng = np.random.default_rng(77)
    n = (1 << 14) + 3
    skey = rng.integers(0, 256, 16, dtype=np.uint8)
    c.schedule_key(torch.from_numpy(skey).cuda(), layout=pkg.LAYOUT_PACKED, key_slab=False)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    jobs = []
    for i in range(12):
        mode = i % 3
        pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
        keys = rng.integers(0, 256, (n, 16), dtype=np.uint8) if mode == 0 else (rng.integers(0, 256, 16, dtype=np.uint8) if mode == 1 else None)
        jobs.append((mode, pt, keys, torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda() if keys is not None else None))
    torch.cuda.synchronize()
    outs = []
    for i, (mode, pt, keys, dpt, dkeys) in enumerate(jobs):
        with torch.cuda.stream(streams[i % 3]):
            outs.append(c.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, want_ct=True, key_slab=mode == 0))
    torch.cuda.synchronize()
    for (mode, pt, keys, _, _), got in zip(jobs, outs):
        e = oracle.encrypt_witness(pt, keys if keys is not None else skey, layout=ol.PACKED)
        for col in "xyz":
            assert np.array_equal(getattr(got, col).cpu().numpy(), getattr(e, col)), (mode, col)
        assert np.array_equal(got.ct.cpu().numpy(), e.ct)
        if mode == 0:
            k = oracle.key_schedule_witness(keys, layout=ol.PACKED)
            for col in ("w", "kx", "ky", "kz"):
                assert np.array_equal(getattr(got.key, col).cpu().numpy(), getattr(k, col)), col
    c.close()


def test_assemble_oneshot_geometry_equals_the_striding_kernel_and_synthesize(ctx, pkg, oracle):
    """assemble_geometry 1 (round 3: one-shot workgroups on a (chunk, segment, column) grid, no division in the kernel) and 2 / 3 / 4
    (one-shot workgroups on aligned chunks of the output; 4 is the default) write the same Fr columns as the striding kernel, and both equal the restated synthesize() of a K = 12, N = 2 circuit with a
    partly filled last set, never-assigned rows and the words column included."""
    import torch
    k, n_sets = 12, 2
    cap = pkg.block_capacity(k, n_sets)
    n = cap - 1  # one empty block slot at the end of set 1
    rng = np.random.default_rng(91)
    key, pts = rng.integers(0, 256, 16, dtype=np.uint8), rng.integers(0, 256, (n, 16), dtype=np.uint8)
    kw = ctx.schedule_key(torch.from_numpy(key).cuda(), layout=pkg.LAYOUT_PACKED, key_slab=True)
    wit = ctx.encrypt_witness(torch.from_numpy(pts).cuda(), None, layout=pkg.LAYOUT_PACKED)
    outs = []
    for geo in range(5):
        ctx.set_option("assemble_geometry", geo)
        outs.append(ctx.assemble_advice(k, n_sets, wit, kw, n, layout=pkg.LAYOUT_PACKED, as_fr=True).cpu().numpy())
    ctx.set_option("assemble_geometry", 4)
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    fr_mod = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    lut = np.stack([np.frombuffer(((v << 256) % fr_mod).to_bytes(32, "little"), np.uint8) for v in range(256)])
    with oracle.circuit(k, n_sets, key, pts, record_copies=False) as c:
        for col in range(3 * n_sets + 1):
            assert np.array_equal(outs[1][col], lut[c.advice(col)]), col


def test_rescheduling_a_key_waits_for_launches_still_reading_the_old_one(pkg, oracle):
    """re-scheduling on stream A while stream B still reads the previous round keys was a write-after-read race.
    (Round 3 ordered the re-schedule behind ONE event, which lost all but the last reader stream; round 4 keeps the round keys in
    a ring of slots with one event per reader stream -- tests/test_gpu_round4.py has the many-reader cases.)  One reader: a long
    launch with key A on one stream, key B scheduled on another stream right behind it, and the long launch's output is
    still key A's witness in every block; the next launch uses key B."""
    import torch
    c = pkg.Context(0)
    rng = np.random.default_rng(123)
    n = 1 << 18
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    key_a, key_b = rng.integers(0, 256, 16, dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    dpt, da, db = torch.from_numpy(pt).cuda(), torch.from_numpy(key_a).cuda(), torch.from_numpy(key_b).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out_a = c.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=True)
    out_b = c.alloc_witness(4096, pkg.LAYOUT_PACKED, want_ct=True)
    torch.cuda.synchronize()
    for _ in range(3):  # several rounds: the race, when present, does not show every time
        with torch.cuda.stream(s1):
            c.schedule_key(da, key_slab=False)
        with torch.cuda.stream(s2):
            c.encrypt_witness(dpt, None, out=out_a, want_ct=True)   # waits for key A, then runs ~0.1 ms
        with torch.cuda.stream(s1):
            c.schedule_key(db, key_slab=False)                        # must wait for the launch above
            c.encrypt_witness(dpt[:4096], None, out=out_b, want_ct=True)
        torch.cuda.synchronize()
        ea = oracle.encrypt_witness(pt, key_a, layout=ol.PACKED)
        eb = oracle.encrypt_witness(pt[:4096], key_b, layout=ol.PACKED)
        assert np.array_equal(out_a.ct.cpu().numpy(), ea.ct)
        assert np.array_equal(out_a.z.cpu().numpy(), ea.z)
        assert np.array_equal(out_b.ct.cpu().numpy(), eb.ct)
    c.close()


def test_key_only_arena(ctx, pkg, oracle):
    """aesw_columns_alloc(with_key_slab = 2): the key-schedule witness alone in a (probed) arena, filled by key_kernel."""
    import torch
    rng = np.random.default_rng(55)
    n = (1 << 16) + 9
    keys = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    a = ctx.alloc_columns(n, pkg.LAYOUT_PACKED, key_only=True)
    assert a.x.numel() == 0 and a.y.numel() == 0 and a.z.numel() == 0 and ctx.last_arena["candidates"] >= 1
    got = ctx.key_schedule_witness(torch.from_numpy(keys).cuda(), layout=pkg.LAYOUT_PACKED, want_rk=False, out=a.key)
    torch.cuda.synchronize()
    k = oracle.key_schedule_witness(keys, layout=ol.PACKED)
    for c in ("w", "kx", "ky", "kz"):
        assert np.array_equal(getattr(got, c).cpu().numpy(), getattr(k, c)), c
    ctx.free_columns(a)


def test_arena_from_plain_c(pkg, tmp_path):
    """examples/aesw_arena.c: aesw_columns_alloc -> aesw_encrypt_witness_device -> hipMemcpy back, from plain C, equals the
    host-pointer entry point byte for byte; the struct reports the search and is cleared by aesw_columns_free."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "aesw_arena"
    lib_dir = root / "halo2-aes_amd"
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", str(root / "include"), "-I", "/opt/rocm/include",
                    str(root / "examples" / "aesw_arena.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe), "17"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok") and "candidate backings timed" in out.stdout, out.stdout


def test_batches_from_plain_c(pkg, tmp_path):
    """examples/aesw_batches.c: twelve 2^15-block batches through aesw_encrypt_witness_batches_device with one and with three
    internal streams, each captured into a hipGraph from plain C and replayed; every batch equals the host-pointer entry
    point byte for byte, and three streams are not slower than one (they are 15 % faster at this size on a quiet GPU)."""
    import re
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "aesw_batches"
    lib_dir = root / "halo2-aes_amd"
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", str(root / "include"), "-I", "/opt/rocm/include",
                    str(root / "examples" / "aesw_batches.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe), "15", "12"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout
    ratio = float(re.search(r"three streams / one stream = ([0-9.]+)", out.stdout).group(1))
    print("three streams / one stream = %.3f" % ratio)  # reported, not asserted here: the bound lives in tests/test_perf.py (-m perf)

