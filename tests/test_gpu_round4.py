"""Round 4, GPU: the scheduled key under any number of reader streams (VERDICT r03 weak 1 / ADVICE r03 medium).

Reference semantics: FixedAes128Config::schedule_key stores `self.keys = Some(..)` (src/aes128.rs:143-152) and every later
encrypt() reads that field (src/aes128.rs:170): the key changes atomically BETWEEN encrypt calls.  On the device the calls are
asynchronous and may name different streams; include/aesw.h states what is guaranteed, these tests hold it to that."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

THREADS = 16


def _same(w, e, cols="xyz", what=""):
    for col in cols:
        got = getattr(w, col).cpu().numpy().reshape(-1)
        assert np.array_equal(got, np.asarray(getattr(e, col)).reshape(-1)), "%s column %s" % (what, col)


@pytest.mark.parametrize("key_slots", [1, 4])
def test_reschedule_behind_two_reader_streams(pkg, oracle, key_slots):
    """The judge's scenario: a 2^20-block scheduled-key launch on stream A, a 4 096-block one on stream B, a re-schedule on
    stream C.  With ONE round-key slot the re-schedule must wait for A AND B (round 3 waited for the last recorded reader only,
    so A's later workgroups read the new key); with the default ring it takes another slot and waits for nobody.  Every output
    equals the oracle under the OLD key, the launch that follows under the NEW one."""
    import torch
    c = pkg.Context(0)
    c.set_option("key_slots", key_slots)
    assert c.get_option("key_slots") == key_slots
    rng = np.random.default_rng(2024)
    n_a, n_b = 1 << 20, 4096
    pt = rng.integers(0, 256, (n_a, 16), dtype=np.uint8)
    keys = [rng.integers(0, 256, 16, dtype=np.uint8) for _ in range(4)]
    dpt = torch.from_numpy(pt).cuda()
    dkeys = [torch.from_numpy(k).cuda() for k in keys]
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    out_a = c.alloc_witness(n_a, pkg.LAYOUT_PACKED, want_ct=True)
    out_b = c.alloc_witness(n_b, pkg.LAYOUT_PACKED, want_ct=True)
    out_c = c.alloc_witness(n_b, pkg.LAYOUT_PACKED, want_ct=True)
    expect = {}
    with torch.cuda.stream(sc):
        c.schedule_key(dkeys[0], key_slab=False)
    torch.cuda.synchronize()
    waits0 = c.get_option("key_reader_waits")
    for rnd in range(3):  # the race, when present, does not show every time
        old, new = keys[rnd], keys[rnd + 1]
        with torch.cuda.stream(sa):
            c.encrypt_witness(dpt, None, out=out_a, want_ct=True)            # ~0.5 ms, reads the old key throughout
        with torch.cuda.stream(sb):
            c.encrypt_witness(dpt[-n_b:], None, out=out_b, want_ct=True)     # short, recorded LAST
        with torch.cuda.stream(sc):
            c.schedule_key(dkeys[rnd + 1], key_slab=False)
            c.encrypt_witness(dpt[:n_b], None, out=out_c, want_ct=True)      # the new key
        torch.cuda.synchronize()
        if rnd not in expect:
            expect[rnd] = oracle.encrypt_witness(pt, old, layout=ol.PACKED, threads=THREADS)
        e_a = expect[rnd]
        _same(out_a, e_a, "xyz", "round %d, long launch under the old key" % rnd)
        assert np.array_equal(out_a.ct.cpu().numpy(), e_a.ct)
        e_b = oracle.encrypt_witness(pt[-n_b:], old, layout=ol.PACKED)
        _same(out_b, e_b, "xyz", "round %d, short launch under the old key" % rnd)
        e_c = oracle.encrypt_witness(pt[:n_b], new, layout=ol.PACKED)
        _same(out_c, e_c, "xyz", "round %d, launch behind the re-schedule" % rnd)
        for w in (out_a, out_b, out_c):
            for t in (w.x, w.y, w.z, w.ct):
                t.zero_()
        torch.cuda.synchronize()
    waits = c.get_option("key_reader_waits") - waits0
    if key_slots == 1:
        # every re-schedule found the readers of streams A, B (and from the second round on C) on its one slot
        assert waits >= 2 * 3, waits
    else:
        assert waits == 0, waits  # three re-schedules, four slots: nobody's slot was taken
    c.close()


@pytest.mark.parametrize("key_slots", [1, 4])
def test_batches_with_the_scheduled_key_then_reschedule_on_a_foreign_stream(pkg, oracle, key_slots):
    """aesw_encrypt_witness_batches_device deals scheduled-key launches onto three internal streams; a re-schedule issued
    right behind it on a stream that is NOT the caller's (so not joined with them) must still come after all of them."""
    import torch
    c = pkg.Context(0)
    c.set_option("key_slots", key_slots)
    c.set_option("batch_streams", 3)
    rng = np.random.default_rng(99)
    n, count = 1 << 16, 6
    key_a, key_b = rng.integers(0, 256, 16, dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    pts = [rng.integers(0, 256, (n, 16), dtype=np.uint8) for _ in range(count)]
    dpts = [torch.from_numpy(p).cuda() for p in pts]
    da, db = torch.from_numpy(key_a).cuda(), torch.from_numpy(key_b).cuda()
    outs = [c.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=True) for _ in range(count)]
    out_new = c.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=True)
    caller, foreign = torch.cuda.Stream(), torch.cuda.Stream()
    exp = [oracle.encrypt_witness(p, key_a, layout=ol.PACKED, threads=THREADS) for p in pts]
    exp_new = oracle.encrypt_witness(pts[0], key_b, layout=ol.PACKED, threads=THREADS)
    for rnd in range(3):
        with torch.cuda.stream(caller):
            c.schedule_key(da, key_slab=False)
            c.encrypt_witness_batches([(dpts[i], None, outs[i]) for i in range(count)], per_block_keys=False)
        with torch.cuda.stream(foreign):
            c.schedule_key(db, key_slab=False)
            c.encrypt_witness(dpts[0], None, out=out_new, want_ct=True)
        torch.cuda.synchronize()
        for i in range(count):
            _same(outs[i], exp[i], "xyz", "round %d batch %d (old key)" % (rnd, i))
            assert np.array_equal(outs[i].ct.cpu().numpy(), exp[i].ct)
        _same(out_new, exp_new, "xyz", "round %d, new key" % rnd)
        for w in outs + [out_new]:
            for t in (w.x, w.y, w.z, w.ct):
                t.zero_()
        torch.cuda.synchronize()  # (zero_ runs on torch's current stream, the next round's launches on others)
    c.close()


def test_more_reader_streams_than_a_slot_tracks_are_folded_not_lost(pkg, oracle):
    """A slot keeps one event per distinct reader stream, sixteen at most; further streams are folded into an existing
    entry.  Twenty streams read one key (ring of one slot), the first one with a long launch; the re-schedule waits for all."""
    import torch
    c = pkg.Context(0)
    c.set_option("key_slots", 1)
    rng = np.random.default_rng(5)
    n_long, n_short, streams = 1 << 19, 1024, 20
    pt = rng.integers(0, 256, (n_long, 16), dtype=np.uint8)
    key_a, key_b = rng.integers(0, 256, 16, dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    dpt, da, db = torch.from_numpy(pt).cuda(), torch.from_numpy(key_a).cuda(), torch.from_numpy(key_b).cuda()
    ss = [torch.cuda.Stream() for _ in range(streams)]
    out_long = c.alloc_witness(n_long, pkg.LAYOUT_PACKED, want_ct=True)
    out_short = [c.alloc_witness(n_short, pkg.LAYOUT_PACKED) for _ in range(streams - 1)]
    out_new = c.alloc_witness(n_short, pkg.LAYOUT_PACKED)
    c.schedule_key(da, key_slab=False)
    torch.cuda.synchronize()
    with torch.cuda.stream(ss[0]):
        c.encrypt_witness(dpt, None, out=out_long, want_ct=True)
    for j in range(1, streams):
        with torch.cuda.stream(ss[j]):
            c.encrypt_witness(dpt[j * n_short:(j + 1) * n_short], None, out=out_short[j - 1])
    resched = torch.cuda.Stream()
    with torch.cuda.stream(resched):
        c.schedule_key(db, key_slab=False)
        c.encrypt_witness(dpt[:n_short], None, out=out_new)
    torch.cuda.synchronize()
    e = oracle.encrypt_witness(pt, key_a, layout=ol.PACKED, threads=THREADS)
    _same(out_long, e, "xyz", "long launch")
    sx, sy, sz = (pkg.column_stride(pkg.LAYOUT_PACKED, i) for i in range(3))
    for j in range(1, streams):
        lo, hi = j * n_short, (j + 1) * n_short
        w = out_short[j - 1]
        assert np.array_equal(w.x.cpu().numpy(), e.x[lo * sx:hi * sx]), j
        assert np.array_equal(w.y.cpu().numpy(), e.y[lo * sy:hi * sy]), j
        assert np.array_equal(w.z.cpu().numpy(), e.z[lo * sz:hi * sz]), j
    _same(out_new, oracle.encrypt_witness(pt[:n_short], key_b, layout=ol.PACKED), "xyz", "new key")
    c.close()


def test_a_captured_launch_keeps_the_key_it_was_captured_with(pkg, oracle):
    """include/aesw.h: a slot read by a captured launch is pinned.  Capture a scheduled-key launch, re-schedule more often
    than the ring has slots (with launches in between), replay: the graph still encrypts with the key current at capture, an
    un-captured launch with the newest."""
    import torch
    c = pkg.Context(0)
    c.set_option("key_slots", 2)
    rng = np.random.default_rng(31)
    n = 3000
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = [rng.integers(0, 256, 16, dtype=np.uint8) for _ in range(6)]
    dpt = torch.from_numpy(pt).cuda()
    dkeys = [torch.from_numpy(k).cuda() for k in keys]
    out_g, out_e = c.alloc_witness(n, pkg.LAYOUT_PACKED), c.alloc_witness(n, pkg.LAYOUT_PACKED)
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        c.schedule_key(dkeys[0], key_slab=False)
    cap.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        c.encrypt_witness(dpt, None, out=out_g)
    assert c.get_option("key_slots_pinned") == 1
    for k in range(1, 6):
        c.schedule_key(dkeys[k], key_slab=False)
        c.encrypt_witness(dpt, None, out=out_e)
        graph.replay()
        torch.cuda.synchronize()
        _same(out_g, oracle.encrypt_witness(pt, keys[0], layout=ol.PACKED), "xyz", "graph replay after %d re-schedules" % k)
        _same(out_e, oracle.encrypt_witness(pt, keys[k], layout=ol.PACKED), "xyz", "eager launch, key %d" % k)
        for t in (out_g.x, out_g.y, out_g.z):
            t.zero_()
    assert c.get_option("key_slots_pinned") == 1 and c.get_option("key_slots_allocated") <= 1 + 2 + 1
    c.close()


def test_a_captured_schedule_owns_its_slot(pkg, oracle):
    """Schedule + encrypt captured into one graph (the reference's call shape, benches/aes128.rs:50-53, as a replayable unit):
    un-captured re-schedules around its replays never touch the slot the graph writes and reads."""
    import torch
    c = pkg.Context(0)
    c.set_option("key_slots", 1)
    rng = np.random.default_rng(32)
    n = 2048
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    key_g, key_e = rng.integers(0, 256, 16, dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    dpt, dg, de = torch.from_numpy(pt).cuda(), torch.from_numpy(key_g).cuda(), torch.from_numpy(key_e).cuda()
    out_g, out_e = c.alloc_witness(n, pkg.LAYOUT_PACKED), c.alloc_witness(n, pkg.LAYOUT_PACKED)
    cap = torch.cuda.Stream()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        c.schedule_key(dg, key_slab=False)
        c.encrypt_witness(dpt, None, out=out_g)
    e_g = oracle.encrypt_witness(pt, key_g, layout=ol.PACKED)
    e_e = oracle.encrypt_witness(pt, key_e, layout=ol.PACKED)
    for _ in range(3):
        c.schedule_key(de, key_slab=False)
        c.encrypt_witness(dpt, None, out=out_e)
        graph.replay()
        torch.cuda.synchronize()
        _same(out_g, e_g, "xyz", "graph")
        _same(out_e, e_e, "xyz", "eager")
        for t in (out_g.x, out_g.y, out_g.z, out_e.x, out_e.y, out_e.z):
            t.zero_()
    c.close()


def test_scheduled_key_capture_on_a_foreign_stream(pkg, oracle):
    """A launch captured on a stream other than its key's cannot depend on the key kernel.  While that kernel may still be
    running the call is refused (AESW_ERR_INVALID_ARG, nothing dropped silently); once it has finished there is nothing to
    depend on and the capture goes through and replays byte-exact."""
    import torch
    c = pkg.Context(0)
    rng = np.random.default_rng(77)
    n = 500
    pt, key = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    dpt, dkey = torch.from_numpy(pt).cuda(), torch.from_numpy(key).cuda()
    out = c.alloc_witness(n, pkg.LAYOUT_PACKED)
    key_stream, other = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.from_numpy(rng.integers(0, 256, (1 << 20, 16), dtype=np.uint8)).cuda()
    sink = c.alloc_witness(1 << 20, pkg.LAYOUT_PACKED)
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(key_stream):
        for _ in range(16):                          # ~0.5 ms each: the stream stays busy while the host gets to the capture
            c.encrypt_witness(big, dkey, out=sink)
        c.schedule_key(dkey, key_slab=False)        # queued behind them: not finished when the capture below starts
    # (torch.cuda.graph() synchronises the device on entry, which would let the key kernel finish: begin the capture by hand)
    with torch.cuda.stream(other):
        graph.capture_begin()
        try:
            with pytest.raises(pkg.AeswError) as ei:
                c.encrypt_witness(dpt, None, out=out)
        finally:
            graph.capture_end()                      # the refusal issued nothing: the (empty) capture is still valid
    assert ei.value.status == pkg.api.ERR_INVALID_ARG and "captured" in str(ei.value)
    torch.cuda.synchronize()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=other):
        c.encrypt_witness(dpt, None, out=out)
    for t in (out.x, out.y, out.z):
        t.fill_(0)
    graph2.replay()
    torch.cuda.synchronize()
    _same(out, oracle.encrypt_witness(pt, key, layout=ol.PACKED), "xyz", "foreign-stream capture after a synchronise")
    c.close()


def _bench_mod():
    import importlib.util
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_mod", root / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("pbk", [True, False])
def test_bench_parity_gate_sees_one_wrong_byte(ctx, pkg, pbk):
    """bench.py's parity gate: the sample of the timed launches' output is compared with the oracle in every
    column; it passes on what the kernel wrote and counts exactly the bytes that are then flipped inside sampled blocks --
    first block, last block, key columns included."""
    import torch
    b = _bench_mod()
    n = 1 << 14
    r = b.Runner(pkg, ctx, torch, n, pbk, pkg.LAYOUT_PACKED, pbk, 4242, arena=False)
    r.prepare(3, 1, True)
    r.timed()
    g = b.parity_gate(r, blocks=2048)
    assert g["mismatches"] == 0 and g["blocks"] >= 1024 and g["sets_checked"] == min(r.nsets, 3)
    assert g["columns"] == (["kx", "ky", "kz", "w", "x", "y", "z"] if pbk else ["x", "y", "z"])
    r.sets[0].x[5] ^= 1                      # block 0
    r.sets[1].z[-1] ^= 0x80                  # the last block, another set
    if pbk:
        r.sets[0].key.kz[-3] ^= 2
    torch.cuda.synchronize()
    g2 = b.parity_gate(r, blocks=2048)
    # the sampled comparison counts the flipped bytes; the device check over ALL blocks objects too (one more entry)
    assert g2["mismatches"] == (3 if pbk else 2) + 1 and len(g2["where"]) == (3 if pbk else 2) + 1
    dc = g2["device_check"]
    assert not dc["satisfied"] and dc["lookup_failures"] + dc["copy_failures"] >= (3 if pbk else 2) and g["device_check"]["satisfied"]
    assert g["device_check"]["blocks"] == min(r.nsets, 3) * n
    # a flipped byte in a block the sample does NOT hold is still caught -- by the device check alone
    r.sets[0].x[5] ^= 1
    r.sets[1].z[-1] ^= 0x80
    if pbk:
        r.sets[0].key.kz[-3] ^= 2
    torch.cuda.synchronize()
    assert b.parity_gate(r, blocks=2048)["mismatches"] == 0
    sampled = set(np.unique(np.concatenate([np.arange(256), np.arange(n - 256, n), np.random.default_rng(b.SEED + 99).integers(0, n, 2048 - 512)])).tolist())
    victim = next(i for i in range(300, n) if i not in sampled)
    r.sets[0].y[victim * 1056 + 7] ^= 4
    torch.cuda.synchronize()
    g3 = b.parity_gate(r, blocks=2048)
    assert g3["mismatches"] == 1 and "device check" in g3["where"][0] and not g3["device_check"]["satisfied"]
    r.close()


def test_placement_cache_hands_a_freed_arena_back_without_a_search(pkg, oracle):
    """VERDICT r03 next 3 (consumer: one synthesize() per proof, three passes, benches/aes128.rs:80-107).  A probed arena that is
    freed keeps its backing inside the context; the next aesw_columns_alloc of the same shape returns the same columns with
    candidates == 0, and a launch into them is byte-exact.  Another shape searches again; "arena_cache" 0 releases what is
    cached; "arena_probe_budget_ms" bounds a search to the candidates it can time in that long (at least one)."""
    import time
    import torch
    c = pkg.Context(0)
    assert c.get_option("arena_cache") == 1 and c.get_option("arena_probe_budget_ms") == 3000
    rng = np.random.default_rng(808)
    n = (1 << 17) + 48
    pt, keys = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, (n, 16), dtype=np.uint8)
    dpt, dkeys = torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    w = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=True)
    first = dict(c.last_arena)
    assert first["candidates"] >= 1
    ptrs = [t.data_ptr() for t in (w.x, w.y, w.z, w.key.w, w.key.kx, w.key.ky, w.key.kz)]
    c.free_columns(w)
    assert c.get_option("arena_cached_bytes") == first["bytes"]
    t0 = time.perf_counter()
    w2 = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=True)
    dt = time.perf_counter() - t0
    again = dict(c.last_arena)
    assert again["candidates"] == 0 and again["probe_us"] == first["probe_us"] and again["bytes"] == first["bytes"]
    assert [t.data_ptr() for t in (w2.x, w2.y, w2.z, w2.key.w, w2.key.kx, w2.key.ky, w2.key.kz)] == ptrs
    assert c.get_option("arena_cache_hits") == 1 and c.get_option("arena_cached_bytes") == 0
    assert dt < 0.020, dt  # a gross bound (the call is a table lookup); the 3 % launch-time bound is tests/test_perf.py
    for t in (w2.x, w2.y, w2.z, w2.key.w, w2.key.kx, w2.key.ky, w2.key.kz):
        t.fill_(0xEE)
    c.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w2, key_slab=True)
    torch.cuda.synchronize()
    e = oracle.encrypt_witness(pt, keys, layout=ol.PACKED, threads=THREADS)
    k = oracle.key_schedule_witness(keys, layout=ol.PACKED, threads=THREADS)
    _same(w2, e, "xyz", "launch into the cached arena")
    for col in ("w", "kx", "ky", "kz"):
        assert np.array_equal(getattr(w2.key, col).cpu().numpy(), getattr(k, col)), col
    # another shape (no key slabs) is a new search and does not touch the entry of the first shape
    c.free_columns(w2)
    w3 = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=False)
    assert c.last_arena["candidates"] >= 1 and c.get_option("arena_cached_bytes") == first["bytes"]
    c.free_columns(w3)
    assert c.get_option("arena_cached_bytes") > first["bytes"]
    # switching the cache off releases everything it holds
    c.set_option("arena_cache", 0)
    assert c.get_option("arena_cached_bytes") == 0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 >= free0 - (64 << 20), (free0, free1)
    w4 = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=True)
    assert c.last_arena["candidates"] >= 1
    c.free_columns(w4)
    assert c.get_option("arena_cached_bytes") == 0
    # a search bounded to (almost) nothing still places the arena: exactly one candidate
    c.set_option("arena_probe_budget_ms", 1)
    w5 = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=True)
    assert c.last_arena["candidates"] == 1, c.last_arena
    c.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w5, key_slab=True)
    torch.cuda.synchronize()
    _same(w5, e, "xyz", "launch into a budget-bounded arena")
    c.free_columns(w5)
    c.close()


# ---- byte -> Fr cells, every cell ------------------------------------------------

FR_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001  # bn256::Fr (halo2curves 0.6.1, Cargo.lock:779-781)


def _fr_lut():
    """Fp::from(u64) for a byte (src/utils.rs:23, src/aes128.rs:187): Montgomery form v * 2^256 mod r, 32 bytes little-endian."""
    return np.stack([np.frombuffer(((v << 256) % FR_MOD).to_bytes(32, "little"), np.uint8) for v in range(256)])


@pytest.mark.parametrize("n_cells", [1, 255, 4097, (1 << 20) + 37])
def test_expand_fr_every_cell_in_every_geometry_and_store_flavour(pkg, n_cells):
    """aesw_expand_fr_device against the host LUT, EVERY cell, for the three geometries ("fr_geometry": striding workgroups with
    the LUT in LDS, one-shot 4 KiB, one-shot 16 KiB) x the three store flavours ("fr_store_mode"), at sizes of one cell, one
    short of a workgroup, one past a 4 KiB multiple, and ten workgroups and more with a ragged tail; bytes behind the last cell
    stay untouched."""
    import torch
    c = pkg.Context(0)
    lut = _fr_lut()
    rng = np.random.default_rng(n_cells)
    cells = rng.integers(0, 256, n_cells, dtype=np.uint8)
    cells[: min(n_cells, 256)] = np.arange(min(n_cells, 256), dtype=np.uint8)  # every byte value where there is room
    expect = lut[cells]
    dcells = torch.from_numpy(cells).cuda()
    guard = 4096
    for geo in (0, 1, 2):
        for mode in (0, 1, 2):
            c.set_option("fr_geometry", geo)
            c.set_option("fr_store_mode", mode)
            buf = torch.full((n_cells * 32 + guard,), 0xCC, dtype=torch.uint8, device="cuda")
            out = buf[: n_cells * 32].view(n_cells, 32)
            c.expand_fr(dcells, out)
            torch.cuda.synchronize()
            got = buf.cpu().numpy()
            assert np.array_equal(got[: n_cells * 32].reshape(n_cells, 32), expect), (geo, mode)
            assert (got[n_cells * 32:] == 0xCC).all(), (geo, mode, "wrote past the last cell")
    assert lut[1].tobytes() == bytes.fromhex("fbffff4f1c3496ac29cd609f9576fc362e4679786fa36e662fdf079ac1770a0e")  # Fr::one() = R
    c.close()


def _small_k_expectation(oracle, k, n_sets, key):
    """The advice matrix of a circuit too small for the reference to run (2^K < 1 760 rows: no block fits, and below K = 9 not
    even the 400 key rows do): what aesw_assemble_advice_* defines there is the key slab clipped at 2^K rows, everything else
    0 -- built here from the oracle's DENSE key slab."""
    rows = 1 << k
    kd = oracle.key_schedule_witness(key.reshape(1, 16), layout=ol.DENSE)
    m = np.zeros((3 * n_sets + 1, rows), np.uint8)
    for col, src in ((0, kd.kx), (1, kd.ky), (2, kd.kz)):
        m[col, : min(rows, 400)] = src[: min(rows, 400)]
    m[3 * n_sets, : min(rows, 96)] = kd.w[: min(rows, 96)]
    return m


@pytest.mark.parametrize("k,n_sets,spare", [(7, 2, 0), (8, 2, 0), (9, 1, 0), (11, 2, 0), (16, 3, 20)])
def test_assemble_geometries_at_the_fallback_boundary_and_with_a_partly_filled_last_set(ctx, pkg, oracle, k, n_sets, spare):
    """The five "assemble_geometry" forms write identical Fr columns and byte columns on both sides of K = 8 (below it the
    aligned forms 2-4 fall back to the striding kernel, aesw_layout.h assemble_kernel_choice), at K = 11 (the smallest circuit
    the reference's capacity rule gives a block: one, in set 1) and for K = 16, N = 3 with twenty empty block slots at the end
    of the last set; every cell equals the restated synthesize() (K >= 11) or the clipped key slab (K < 11)."""
    import torch
    lut = _fr_lut()
    cap = pkg.block_capacity(k, n_sets)
    n = cap - spare
    assert n >= 0 and (k != 11 or n == 1) and (k != 16 or n == 46 + 48 + 48 - 20)
    rng = np.random.default_rng(1000 + k)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (max(n, 1), 16), dtype=np.uint8)
    kw = ctx.schedule_key(torch.from_numpy(key).cuda(), layout=pkg.LAYOUT_PACKED, key_slab=True)
    wit = ctx.encrypt_witness(torch.from_numpy(pts).cuda(), None, layout=pkg.LAYOUT_PACKED)
    if k >= 11:
        with oracle.circuit(k, n_sets, key, pts[:n], record_copies=False) as circ:
            assert circ.status == 0
            expect = np.stack([circ.advice(col) for col in range(3 * n_sets + 1)])
    else:
        expect = _small_k_expectation(oracle, k, n_sets, key)
    try:
        for geo in range(5):
            ctx.set_option("assemble_geometry", geo)
            fr = ctx.assemble_advice(k, n_sets, wit, kw, n, layout=pkg.LAYOUT_PACKED, as_fr=True).cpu().numpy()
            by = ctx.assemble_advice(k, n_sets, wit, kw, n, layout=pkg.LAYOUT_PACKED, as_fr=False).cpu().numpy()
            assert np.array_equal(by, expect), (geo, "bytes")
            assert np.array_equal(fr, lut[expect]), (geo, "Fr cells")
    finally:
        ctx.set_option("assemble_geometry", 4)


# ---- SURVEY 8(e) / BASELINE configs[3]: the eight-rank shape of the C ABI's gather, without eight GPUs -----------------

@pytest.mark.parametrize("root", [0, 5])
def test_gather_pieces_pair_up_across_eight_ranks(pkg, tmp_path, root):
    """VERDICT r03 next 5(a).  aesw_gather_columns_device run as EVERY one of eight ranks (one process each, one after the other,
    on the one GPU) against the recording stand-in for librccl: ragged block counts with one rank that has nothing, a
    max_message that cuts every non-empty range of every column into three pieces or more.  What must hold for the real
    exchange not to hang or scramble: per peer the root's receives and the peer's sends are the SAME list of sizes in the same
    order (RCCL pairs the k-th send to a peer with the k-th receive from it), the receives land back to back at the peer's
    block offset of each column, nobody posts anything for the empty rank or for the root itself, every rank opens exactly one
    group, all traffic is ncclUint8."""
    import os
    import subprocess
    from pathlib import Path
    ROOT = Path(__file__).resolve().parent.parent
    mock_dir, lib_dir = ROOT / "tests" / "mock_rccl", ROOT / "halo2-aes_amd"
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", str(tmp_path / "librccl.so"), str(mock_dir / "mock_rccl.c")], check=True)
    exe = tmp_path / "gather_driver"
    subprocess.run(["gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
                    str(mock_dir / "gather_driver.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    world, counts, strides, maxmsg = 8, [9, 4, 0, 7, 1, 12, 3, 5], [1360, 1056, 608], 200
    offs = [sum(counts[:r]) for r in range(world)]
    assert all(c == 0 or c * min(strides) >= 3 * maxmsg for c in counts)

    def run(rank):
        log = tmp_path / ("log_%d_%d" % (root, rank))
        env = dict(os.environ, MOCK_RCCL_LOG=str(log), GATHER_ROOT=str(root),
                   LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        out = subprocess.run([str(exe), str(world), str(rank), str(maxmsg)] + [str(c) for c in counts], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True, timeout=120)
        assert out.returncode == 0 and "ok" in out.stdout, out.stdout
        base = {}
        for line in out.stdout.splitlines():
            if line.startswith("col"):
                _, c, _, s, _, r = line.split()
                base[int(c)] = (int(s), int(r))
        calls = [l.split() for l in log.read_text().splitlines()]
        assert calls[0][0] == "id" and calls[1][:3] == ["init", str(world), str(rank)] and calls[-1][0] == "destroy"
        body = calls[2:-1]
        assert [c[0] for c in body].count("group_start") == 1 and body[0][0] == "group_start" and body[-1][0] == "group_end"
        assert not any("badtype" in c[0] for c in body)
        return base, body[1:-1]

    logs = {r: run(r) for r in range(world)}
    rbase, rcalls = logs[root]
    assert all(c[0] == "recv" for c in rcalls)  # the root's own range is a device-to-device copy, not a message
    recv_by_peer = {p: [(int(c[2]), int(c[3])) for c in rcalls if int(c[1]) == p] for p in range(world)}
    assert recv_by_peer[root] == [] and recv_by_peer[2] == []
    for p in range(world):
        base, calls = logs[p]
        if p == root:
            continue
        assert all(c[0] == "send" and int(c[1]) == root for c in calls), (p, calls[:3])
        sends = [(int(c[2]), int(c[3])) for c in calls]
        recvs = recv_by_peer[p]
        assert [m for _, m in sends] == [m for _, m in recvs], p                       # same sizes, same order: they pair up
        assert sum(m for _, m in sends) == counts[p] * sum(strides)
        assert all(0 < m <= maxmsg for _, m in sends)
        if counts[p] == 0:
            assert sends == [] and recvs == []
            continue
        # walk both lists column by column: contiguous pieces from the send base / at the peer's block offset on the root
        i = 0
        for c, s in enumerate(strides):
            nbytes, o = counts[p] * s, 0
            npieces = 0
            while o < nbytes:
                (sa, sm), (ra, rm) = sends[i], recvs[i]
                assert sa == base[c][0] + o and ra == rbase[c][1] + offs[p] * s + o and sm == rm == min(maxmsg, nbytes - o), (p, c, o)
                o += sm
                i += 1
                npieces += 1
            assert npieces >= 3
        assert i == len(sends)


@pytest.mark.parametrize("split", [2, 3, 4])
def test_split_small_is_byte_exact(pkg, oracle, split):
    """Option "split_small" (round 4's experiment, off by default: profiles/r04_study/split_small.md): a lone shared- or
    scheduled-key batch of 2^15 .. 2^17 blocks dealt as `split` sub-ranges of whole 48-block groups onto the internal streams.
    Same bytes as one launch, ragged batch, ciphertext included, work queued before and after on the caller's stream ordered
    around it; sizes outside the window and per-block keys are left alone."""
    import torch
    c = pkg.Context(0)
    c.set_option("split_small", split)
    assert c.get_option("split_small") == split
    rng = np.random.default_rng(60 + split)
    n = (1 << 15) + 1234
    pt, key = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    dpt, dkey = torch.from_numpy(pt).cuda(), torch.from_numpy(key).cuda()
    e = oracle.encrypt_witness(pt, key, layout=ol.PACKED, threads=THREADS)
    s = torch.cuda.Stream()
    for mode in ("scheduled", "shared"):
        out = c.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=True)
        with torch.cuda.stream(s):
            if mode == "scheduled":
                c.schedule_key(dkey, key_slab=False)
            for t in (out.x, out.y, out.z, out.ct):
                t.fill_(0x77)                      # queued BEFORE the call on the caller's stream
            c.encrypt_witness(dpt, None if mode == "scheduled" else dkey, out=out, want_ct=True)
            x_after = out.x.clone()                # queued AFTER it
        torch.cuda.synchronize()
        _same(out, e, "xyz", "%s key, split %d" % (mode, split))
        assert np.array_equal(out.ct.cpu().numpy(), e.ct) and np.array_equal(x_after.cpu().numpy(), e.x)
    # outside the window (and with per-block keys) nothing is split: same results through the ordinary path
    small = 5000
    out = c.encrypt_witness(dpt[:small], dkey, want_ct=True)
    keys = rng.integers(0, 256, (1 << 15, 16), dtype=np.uint8)
    outk = c.encrypt_witness(dpt[:1 << 15], torch.from_numpy(keys).cuda(), want_ct=True)
    torch.cuda.synchronize()
    _same(out, oracle.encrypt_witness(pt[:small], key, layout=ol.PACKED), "xyz", "below the window")
    _same(outk, oracle.encrypt_witness(pt[:1 << 15], keys, layout=ol.PACKED, threads=THREADS), "xyz", "per-block keys")
    c.close()


def test_batches_with_the_scheduled_key_capture_into_one_graph(pkg, oracle):
    """The batch entry point (and "split_small") under capture with the SCHEDULED key: the internal streams are forked from the
    capture on the key's own stream, so the launches on them need no dependency of their own (round 3 refused this shape:
    "captured on a stream other than the key's").  Replays are byte-exact and keep the key they were captured with."""
    import torch
    c = pkg.Context(0)
    rng = np.random.default_rng(71)
    sizes = [3000, 1 << 15, 777, (1 << 15) + 48, 48]
    key, key2 = rng.integers(0, 256, 16, dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    pts = [rng.integers(0, 256, (m, 16), dtype=np.uint8) for m in sizes]
    outs = [c.alloc_witness(m, pkg.LAYOUT_PACKED) for m in sizes]
    dpts = [torch.from_numpy(p).cuda() for p in pts]
    c.set_option("split_small", 3)  # the two 2^15-block batches are dealt out once more: nested use of the same streams is excluded
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        c.schedule_key(torch.from_numpy(key).cuda(), key_slab=False)
        c.encrypt_witness_batches([(dpts[i], None, outs[i]) for i in range(len(sizes))], per_block_keys=False)  # creates the streams
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        c.encrypt_witness_batches([(dpts[i], None, outs[i]) for i in range(len(sizes))], per_block_keys=False)
        c.encrypt_witness(dpts[1], None, out=outs[1])  # a lone launch in the window: split into three
    c.schedule_key(torch.from_numpy(key2).cuda(), key_slab=False)  # later keys do not reach into the graph
    for _ in range(2):
        for o in outs:
            for t in (o.x, o.y, o.z):
                t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for p, o in zip(pts, outs):
            _same(o, oracle.encrypt_witness(p, key, layout=ol.PACKED, threads=THREADS), "xyz", "%d blocks" % len(p))
    c.close()


# ---- MockProver::assert_satisfied on the device (aesw_check_witness_device) ---------------------------------------------

def _lane_model(pkg):
    import ctypes as C
    import __graft_entry__ as ge
    L = C.CDLL(str(ge.build_lane_model()))
    L.lane_model_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64] + [C.c_void_p] * 9
    return L


def _model_report(L, pkg, layout, pt, keys, pbk, cols, kcols, ct):
    import ctypes as C
    tab = np.concatenate(pkg.reference_tables()).astype(np.uint8)
    rep = np.zeros(7, np.uint64)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    keep = [np.ascontiguousarray(a) if a is not None else None for a in (pt, keys, *cols, ct, *kcols)]
    rc = L.lane_model_check(p(tab), layout, p(keep[0]), p(keep[1]), 1 if pbk else 0, pt.shape[0], p(keep[2]), p(keep[3]), p(keep[4]), p(keep[5]),
                            p(keep[6]), p(keep[7]), p(keep[8]), p(keep[9]), p(rep))
    assert rc == 0
    f = int(rep[6])
    first = None if f == 2 ** 64 - 1 else (f >> 20, bool((f >> 19) & 1), (f >> 16) & 7, f & 0xFFFF)
    return {"blocks": int(rep[0]), "keys": int(rep[1]), "lookup_failures": int(rep[2]), "copy_failures": int(rep[3]),
            "gate_failures": int(rep[4]), "input_failures": int(rep[5]), "first": first, "satisfied": not any(int(v) for v in rep[2:6])}


@pytest.mark.parametrize("layout_name", ["packed", "dense"])
@pytest.mark.parametrize("pbk", [True, False])
def test_device_checker_accepts_the_product_and_counts_what_the_cpu_model_counts(pkg, oracle, layout_name, pbk):
    """aesw_check_witness_device over the product's own output: satisfied.  Then 300 random single-byte changes anywhere in the
    seven columns, the plaintext and the ciphertext: the device report (every count and the first failure) equals the report of
    the same source run on the CPU (tests/lane_model), which tests/test_check_model.py holds to the oracle's verifier."""
    import torch
    lay = pkg.LAYOUT_PACKED if layout_name == "packed" else pkg.LAYOUT_DENSE
    c = pkg.Context(0)
    rng = np.random.default_rng(17 + pbk)
    n = 5000 + 13
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (n, 16) if pbk else 16, dtype=np.uint8)
    dpt, dkeys = torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda()
    if pbk:
        w = c.encrypt_witness(dpt, dkeys, layout=lay, want_ct=True, key_slab=True)
        kw = w.key
    else:
        kw = c.schedule_key(dkeys, layout=lay, key_slab=True)
        w = c.encrypt_witness(dpt, None, layout=lay, want_ct=True)
    rep = c.check_witness(dpt, dkeys, w, kw, layout=lay, ct=w.ct)
    assert rep["satisfied"] and rep["first"] is None and rep["blocks"] == n and rep["keys"] == (n if pbk else 1), rep
    assert c.check_witness(dpt, None if not pbk else dkeys, w, kw, layout=lay)["satisfied"]
    L = _lane_model(pkg)
    targets = [w.x, w.y, w.z, kw.w, kw.kx, kw.ky, kw.kz, dpt.view(-1), w.ct.view(-1)]
    for _ in range(300):
        t = targets[int(rng.integers(0, len(targets)))]
        t[int(rng.integers(0, t.numel()))] ^= int(rng.integers(1, 256))
    torch.cuda.synchronize()
    got = c.check_witness(dpt, dkeys, w, kw, layout=lay, ct=w.ct)
    host = [t.cpu().numpy() for t in (w.x, w.y, w.z)], [t.cpu().numpy() for t in (kw.w, kw.kx, kw.ky, kw.kz)]
    exp = _model_report(L, pkg, 1 if layout_name == "packed" else 0, dpt.cpu().numpy(), keys, pbk, host[0], host[1], w.ct.cpu().numpy())
    assert got == exp, (got, exp)
    assert not got["satisfied"] and got["lookup_failures"] + got["copy_failures"] + got["input_failures"] >= 250
    # arguments the entry point refuses
    with pytest.raises(pkg.AeswError):
        c.check_witness(dpt, dkeys, w, kw, layout=pkg.LAYOUT_VALUES)
    c.close()


def test_device_checker_over_the_headline_batch_and_inside_a_graph(pkg):
    """2^20 blocks with per-block keys, the product's packed output: every one of 2^20 x (1 360 + 400 rows, 1 952 + 640 copies,
    96 gate rows, 48 literal rows) holds; captured into a hipGraph behind the launch that produces the witness, replayed."""
    import torch
    c = pkg.Context(0)
    n = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(5)
    dpt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
    dkeys = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda", generator=g)
    w = c.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=True, key_slab=True, n_keys=n)
    c.encrypt_witness(dpt, dkeys, out=w, want_ct=True, key_slab=True)
    rep = c.check_witness(dpt, dkeys, w, w.key, ct=w.ct)   # also builds the check table outside the capture
    assert rep["satisfied"] and rep["blocks"] == n and rep["keys"] == n, rep
    cap = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        c.encrypt_witness(dpt, dkeys, out=w, want_ct=True, key_slab=True)
        dev_rep = c.check_witness(dpt, dkeys, w, w.key, ct=w.ct, sync=False)
    for t in (w.x, w.y, w.z, w.key.kz):
        t.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    v = dev_rep.cpu().tolist()
    assert v[0] == n and v[1] == n and v[2:6] == [0, 0, 0, 0] and v[6] == -1, v
    print("witness launch + full check of 2^20 blocks: %.3f ms" % e0.elapsed_time(e1))
    w.z[123456 * 608 + 17] ^= 0x40
    rep = c.check_witness(dpt, dkeys, w, w.key, ct=w.ct)
    assert not rep["satisfied"] and rep["first"][0] == 123456 and not rep["first"][1], rep
    c.close()


def test_check_from_plain_c(pkg, tmp_path):
    """examples/aesw_check.c: generate 2^17 blocks with per-block keys into an arena, check every constraint on the device, change
    one byte, get the block / kind / row back -- from plain C, through include/aesw.h only."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "aesw_check"
    lib_dir = root / "halo2-aes_amd"
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", str(root / "include"), "-I", "/opt/rocm/include",
                    str(root / "examples" / "aesw_check.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe), "17"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok") and "first: block %d, lookup, row 40" % (((1 << 17) + 5) // 2) in out.stdout, out.stdout


@pytest.mark.parametrize("pbk", [True, False])
def test_host_pointer_checker_equals_the_device_one(pkg, oracle, pbk):
    """aesw_check_witness (host buffers, staged upload) over several stages with a ragged last one reports what
    aesw_check_witness_device reports for the same bytes: counts, and the first failure with batch-wide unit numbers; the shared
    key slab is checked once, not once per stage."""
    import torch
    c = pkg.Context(0)
    c.set_option("chunk_blocks", 1000)
    rng = np.random.default_rng(90 + pbk)
    n = 3 * 1000 + 77
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (n, 16) if pbk else 16, dtype=np.uint8)
    w = oracle.encrypt_witness(pt, keys, layout=ol.PACKED)
    k = oracle.key_schedule_witness(keys, layout=ol.PACKED)
    cols, kcols = [w.x, w.y, w.z], [k.w, k.kx, k.ky, k.kz]
    rep = c.check_witness_host(pt, keys, cols, kcols, ct=w.ct)
    assert rep["satisfied"] and rep["blocks"] == n and rep["keys"] == (n if pbk else 1) and rep["first"] is None, rep
    # failures in the third stage, in the last (ragged) stage and in a key slab
    w.z[2500 * 608 + 9] ^= 1
    w.x[(n - 1) * 1360 + 700] ^= 2
    k.kx[(1200 * 400 if pbk else 0) + 44] ^= 4
    if not pbk:
        k.w[20] ^= 8   # the round-constant row of round 1: the gate
    rep = c.check_witness_host(pt, keys, cols, kcols, ct=w.ct)
    dev = c.check_witness(torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda(),
                          pkg.Witness(*[torch.from_numpy(a).cuda() for a in cols], None, None),
                          pkg.KeyWitness(*[torch.from_numpy(a).cuda() for a in kcols], None), ct=torch.from_numpy(w.ct).cuda())
    assert rep == dev and not rep["satisfied"], (rep, dev)
    assert rep["first"][0] == (1200 if pbk else 0) and rep["first"][1], rep
    if not pbk:
        assert rep["gate_failures"] == 1
    # an empty batch is satisfied (nothing to check), on both forms
    empty = c.check_witness_host(pt[:0], keys[:0] if pbk else keys, [a[:0] for a in cols], kcols if not pbk else [a[:0] for a in kcols])
    assert empty["satisfied"] and empty["blocks"] == 0 and empty["first"] is None
    e_dev = c.check_witness(torch.from_numpy(pt[:0].copy()).cuda(), torch.from_numpy(keys).cuda()[:0] if pbk else torch.from_numpy(keys).cuda(),
                            pkg.Witness(*[torch.empty(0, dtype=torch.uint8, device="cuda")] * 3, None, None),
                            pkg.KeyWitness(*[torch.from_numpy(a).cuda() for a in kcols], None))
    assert e_dev["satisfied"] and e_dev["blocks"] == 0
    c.close()


def test_gather_moves_the_right_bytes_between_five_processes(pkg, tmp_path):
    """aesw_gather_columns_device run FOR REAL by five processes sharing the one GPU (with the test runner that makes six GPU users,
    the box's limit), against a functional stand-in for librccl
    (tests/mock_rccl/shm_rccl.c: a message is a file in a tmpfs directory; NCCL's pairing rule -- k-th send to a peer with the k-th
    receive from it, equal sizes -- is enforced).  Ragged counts, one empty rank, a max_message that cuts every range into pieces,
    root 0 and root 3: every byte of the gathered columns on the root is the byte its owner put there.  What this cannot show is
    xGMI itself; what it does show is that the C ABI's exchange is correct as a distributed program, not only as a call log."""
    import os
    import shutil
    import subprocess
    from pathlib import Path
    ROOT = Path(__file__).resolve().parent.parent
    mock_dir, lib_dir = ROOT / "tests" / "mock_rccl", ROOT / "halo2-aes_amd"
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-o", str(tmp_path / "librccl.so"),
                    str(mock_dir / "shm_rccl.c"), "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    exe = tmp_path / "gather_driver"
    subprocess.run(["gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
                    str(mock_dir / "gather_driver.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    world, counts, maxmsg = 5, [700, 301, 0, 1234, 917], 100000
    for root in (0, 3):
        box = Path("/dev/shm") / ("aesw_gather_%d_%d" % (os.getpid(), root))
        box.mkdir()
        try:
            env = dict(os.environ, SHM_RCCL_DIR=str(box), GATHER_ROOT=str(root), GATHER_DATA="1",
                       LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
            procs = [subprocess.Popen([str(exe), str(world), str(r), str(maxmsg)] + [str(c) for c in counts], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True) for r in range(world)]
            outs = [p.communicate(timeout=180)[0] for p in procs]
            for r, (p, o) in enumerate(zip(procs, outs)):
                assert p.returncode == 0 and "ok" in o, (root, r, o)
            assert "data ok" in outs[root], outs[root]
            assert not list(box.iterdir()), "messages nobody received: %r" % [f.name for f in box.iterdir()]
        finally:
            shutil.rmtree(box, ignore_errors=True)
    # the stand-in is not lenient: a peer that cuts its range differently from the root (max_message must agree) fails the exchange
    box = Path("/dev/shm") / ("aesw_gather_%d_bad" % os.getpid())
    box.mkdir()
    try:
        env = dict(os.environ, SHM_RCCL_DIR=str(box), GATHER_DATA="1", LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        procs = [subprocess.Popen([str(exe), "2", str(r), str(mm), "100", "100"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for r, mm in ((0, 9000), (1, 7000))]
        outs = [p.communicate(timeout=180)[0] for p in procs]
        assert procs[0].returncode != 0 and "cut the range differently" in outs[0], outs
    finally:
        shutil.rmtree(box, ignore_errors=True)


@pytest.mark.parametrize("mode", ["scheduled", "shared", "per_block"])
def test_stream_check_certifies_every_chunk(pkg, oracle, mode):
    """Option "stream_check" (BASELINE configs[4]: the stream to the host, now certified on the way): every chunk of
    aesw_encrypt_witness_stream is checked on the device behind its kernel; aesw_last_stream_check sums the chunks.  Several
    chunks with a ragged last one, the three key modes, PACKED and DENSE: the consumer still receives the oracle's bytes, the report
    counts every block (and key slab) exactly once and is satisfied; VALUES streams are not checked; off by default."""
    c = pkg.Context(0)
    c.set_option("chunk_blocks", 2048)
    rng = np.random.default_rng(40)
    n = 4 * 2048 + 333
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (n, 16) if mode == "per_block" else 16, dtype=np.uint8)
    if mode == "scheduled":
        import torch
        c.schedule_key(torch.from_numpy(keys).cuda(), key_slab=False)
        torch.cuda.synchronize()
    arg_keys = None if mode == "scheduled" else keys
    got = {}

    def consume(first, count, x, y, z):
        got[first] = (count, x.copy(), y.copy(), z.copy())
        return 0

    c.encrypt_witness_stream(pt, arg_keys, consume, layout=pkg.LAYOUT_PACKED)
    assert c.last_stream_check()["blocks"] == 0  # the option is off by default
    c.set_option("stream_check", 1)
    for layout, olay in ((pkg.LAYOUT_PACKED, ol.PACKED), (pkg.LAYOUT_DENSE, ol.DENSE)):
        got.clear()
        c.encrypt_witness_stream(pt, arg_keys, consume, layout=layout)
        rep = c.last_stream_check()
        assert rep["satisfied"] and rep["first"] is None and rep["blocks"] == n and rep["keys"] == (n if mode == "per_block" else 1), (layout, rep)
        e = oracle.encrypt_witness(pt, keys, layout=olay, threads=THREADS)
        sx, sy, sz = (pkg.column_stride(layout, i) for i in range(3))
        assert sorted(got) == list(range(0, n, 2048))
        for first, (count, x, y, z) in got.items():
            assert np.array_equal(x, e.x[first * sx:(first + count) * sx]) and np.array_equal(y, e.y[first * sy:(first + count) * sy])
            assert np.array_equal(z, e.z[first * sz:(first + count) * sz])
    c.encrypt_witness_stream(pt, arg_keys, lambda *a: 0, layout=pkg.LAYOUT_VALUES)
    assert c.last_stream_check()["blocks"] == 0
    # the check sees what travels: two cells of one block overwritten between kernel and check ("stream_poison", diagnostic) are
    # reported with the block's BATCH-WIDE index, whichever chunk it lies in -- the first, a middle one, the ragged last one
    for victim in (7, 2 * 2048 + 100, n - 1):
        c.set_option("stream_poison", victim + 1)
        c.encrypt_witness_stream(pt, arg_keys, lambda *a: 0, layout=pkg.LAYOUT_PACKED)
        rep = c.last_stream_check()
        assert not rep["satisfied"] and rep["blocks"] == n and rep["first"][0] == victim and not rep["first"][1], (victim, rep)
        assert 1 <= rep["lookup_failures"] <= 3 and rep["copy_failures"] >= 1
    c.set_option("stream_poison", 0)
    # a context whose tables differ from the ones the circuit is checked with cannot happen (they are one and the same), so provoke a
    # failure the other way: a force_table_path context is still satisfied (another kernel path, same constraints)
    c.set_option("force_table_path", 1)
    c.encrypt_witness_stream(pt, arg_keys, lambda *a: 0, layout=pkg.LAYOUT_PACKED)
    assert c.last_stream_check()["satisfied"] and c.last_stream_check()["blocks"] == n
    c.close()
