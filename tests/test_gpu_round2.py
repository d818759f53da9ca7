"""Round-2 GPU tests: hipGraph capture through the C ABI from plain C (BASELINE configs[4], device side), the
2^24-block streaming run with a verifying consumer (configs[4], host side), Fr advice columns delivered to the host
(SURVEY 8(f)-1/2), the single-rank path of the RCCL gather and the scheduled-key stream ordering."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
FR_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _fr_lut():
    """byte v -> the 32 little-endian bytes of v * 2^256 mod r (what Fp::from(u64) stores, src/utils.rs:23)."""
    return np.stack([np.frombuffer(((v << 256) % FR_MOD).to_bytes(32, "little"), np.uint8) for v in range(256)])


def test_graph_capture_from_plain_c(pkg, oracle, tmp_path):
    """examples/aesw_graph.c: hipStreamBeginCapture -> 6 x (per-block-key + scheduled-key launch) -> instantiate ->
    replay; the replayed output is byte-exact against the oracle, on the golden vectors' inputs and on a ragged batch."""
    exe = tmp_path / "aesw_graph"
    lib_dir = ROOT / "halo2-aes_amd"
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
                    str(ROOT / "examples" / "aesw_graph.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    g = np.load(ROOT / "tests" / "golden" / "slab_vectors.npz")
    rng = np.random.default_rng(5)
    cases = [(g["pt"], g["keys"], g["keys"][4].copy()),
             (rng.integers(0, 256, (5000 + 37, 16), dtype=np.uint8), rng.integers(0, 256, (5000 + 37, 16), dtype=np.uint8),
              rng.integers(0, 256, 16, dtype=np.uint8))]
    for ci, (pt, keys, shared) in enumerate(cases):
        n = pt.shape[0]
        fin, fout = tmp_path / ("in%d.bin" % ci), tmp_path / ("out%d.bin" % ci)
        fin.write_bytes(pt.tobytes() + keys.tobytes() + shared.tobytes())
        out = subprocess.run([str(exe), str(fin), str(fout), str(n), "6", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert out.returncode == 0 and "ok" in out.stdout, out.stdout
        got = np.frombuffer(fout.read_bytes(), np.uint8)
        e = oracle.encrypt_witness(pt, keys, layout=ol.PACKED)
        k = oracle.key_schedule_witness(keys, layout=ol.PACKED)
        s = oracle.encrypt_witness(pt, shared, layout=ol.PACKED)
        expect = np.concatenate([e.x, e.y, e.z, k.w, k.kx, k.ky, k.kz, s.x, s.y, s.z])
        assert got.size == expect.size
        assert np.array_equal(got, expect), "case %d: first difference at byte %d" % (ci, int(np.nonzero(got != expect)[0][0]))
        if ci == 0:  # and against the committed fixtures directly
            assert np.array_equal(got[:e.x.size], g["packed_x"]) and np.array_equal(got[-s.z.size:], g["packed_shared_z"])


def test_graph_capture_through_torch_streams(ctx, pkg, oracle):
    """The same through the Python binding: a torch-captured graph of C-ABI launches replays byte-exact, including the
    first launch of a (layout, key mode) this context has never run before the capture."""
    import torch
    rng = np.random.default_rng(8)
    n = 1000
    pt, keys = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, (n, 16), dtype=np.uint8)
    dpt, dkeys = torch.from_numpy(pt).cuda(), torch.from_numpy(keys).cuda()
    fresh = pkg.Context(0)  # nothing launched on it yet
    out = fresh.alloc_witness(n, pkg.LAYOUT_DENSE, key_slab=True, n_keys=n)
    cap = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        fresh.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_DENSE, out=out)
    for t in (out.x, out.y, out.z):
        t.fill_(0xEE)
    graph.replay()
    torch.cuda.synchronize()
    e = oracle.encrypt_witness(pt, keys, layout=ol.DENSE)
    for c in "xyz":
        assert np.array_equal(getattr(out, c).cpu().numpy(), getattr(e, c)), c
    fresh.close()


def test_scheduled_key_is_ordered_across_streams(ctx, pkg, oracle):
    """aesw_schedule_key_device on stream A, the encrypt launch on stream B without any host synchronisation: the
    launch waits for the round keys (event), so the witness is that of the NEW key."""
    import torch
    rng = np.random.default_rng(13)
    n = 4096
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    dpt = torch.from_numpy(pt).cuda()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for trial in range(4):
        key = rng.integers(0, 256, 16, dtype=np.uint8)
        dkey = torch.from_numpy(key).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(a):
            # keep stream A busy first so that the key kernel is still pending when B's launch is enqueued
            junk = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
            for _ in range(4):
                junk.fill_(trial)
            ctx.schedule_key(dkey, layout=pkg.LAYOUT_PACKED, key_slab=False)
        with torch.cuda.stream(b):
            w = ctx.encrypt_witness(dpt, None, layout=pkg.LAYOUT_PACKED)
        torch.cuda.synchronize()
        e = oracle.encrypt_witness(pt, key, layout=ol.PACKED)
        assert np.array_equal(w.z.cpu().numpy(), e.z), "trial %d: encrypt ran with stale round keys" % trial


def test_comm_single_rank_gather(ctx, pkg):
    """aesw_comm with one rank needs no RCCL: the gather is the root's own device copy on the caller's stream
    (ordered behind the producing kernel), or nothing at all when the data already sits in place."""
    import torch
    n = 3000
    rng = np.random.default_rng(3)
    dpt = torch.from_numpy(rng.integers(0, 256, (n, 16), dtype=np.uint8)).cuda()
    dkey = torch.from_numpy(rng.integers(0, 256, 16, dtype=np.uint8)).cuda()
    comm = pkg.Comm(ctx, 1, 0)
    strides = [pkg.column_stride(pkg.LAYOUT_PACKED, c) for c in range(3)]
    w = ctx.encrypt_witness(dpt, dkey, layout=pkg.LAYOUT_PACKED)
    full = comm.gather_columns([w.x, w.y, w.z], [n], strides, root=0)   # no synchronize in between: same stream
    torch.cuda.synchronize()
    for f, c in zip(full, (w.x, w.y, w.z)):
        assert f.data_ptr() != c.data_ptr() and torch.equal(f, c)
    same = comm.gather_columns([w.x, w.y, w.z], [n], strides, root=0, out=[w.x, w.y, w.z])  # in place: nothing to move
    torch.cuda.synchronize()
    assert all(s.data_ptr() == c.data_ptr() for s, c in zip(same, (w.x, w.y, w.z)))
    with pytest.raises(pkg.AeswError):
        comm.gather_columns([w.x, w.y, w.z], [n], strides, root=1)
    comm.close()
    with pytest.raises(pkg.AeswError):
        pkg.Comm(ctx, 2, 5)


def test_fr_columns_to_host_equal_synthesize(ctx, pkg, oracle):
    """aesw_assemble_advice_stream: every advice column of a K=15, N=3 circuit arrives on the host as bn256::Fr cells
    equal to the restated synthesize()'s cell values mapped through v -> v * 2^256 mod r, and as bytes."""
    import torch
    k, n_sets = 15, 3
    n = pkg.block_capacity(k, n_sets)
    rng = np.random.default_rng(17)
    pt, key = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    pt[3] = 0xFF
    kw = ctx.schedule_key(torch.from_numpy(key).cuda(), layout=pkg.LAYOUT_PACKED, key_slab=True)
    wit = ctx.encrypt_witness(torch.from_numpy(pt).cuda(), None, layout=pkg.LAYOUT_PACKED)
    lut = _fr_lut()
    with oracle.circuit(k, n_sets, key, pt, record_copies=False) as c:
        expect = [c.advice(j) for j in range(c.num_advice)]
    seen = []

    def check_fr(col, cells):
        assert cells.shape == (1 << k, 32)
        if not np.array_equal(cells, lut[expect[col]]):
            return 1
        seen.append(col)
        return 0

    ctx.assemble_advice_stream(k, n_sets, wit, kw, n, check_fr, layout=pkg.LAYOUT_PACKED, as_fr=True)
    assert seen == list(range(3 * n_sets + 1))
    st = ctx.last_stream_stats()
    assert st["chunks"] == 3 * n_sets + 1 and st["bytes_to_host"] == (3 * n_sets + 1) * (32 << k) and st["kernel_ns"] > 0 and st["d2h_ns"] > 0
    seen.clear()

    def check_bytes(col, cells):
        seen.append(col)
        return 0 if np.array_equal(cells, expect[col]) else 1

    ctx.assemble_advice_stream(k, n_sets, wit, kw, n, check_bytes, layout=pkg.LAYOUT_PACKED, as_fr=False)
    assert seen == list(range(3 * n_sets + 1))
    with pytest.raises(pkg.AeswError) as e:   # a consumer that refuses a column aborts the stream
        ctx.assemble_advice_stream(k, n_sets, wit, kw, n, lambda col, cells: 1 if col == 2 else 0, layout=pkg.LAYOUT_PACKED)
    assert e.value.status == 7
    with pytest.raises(pkg.AeswError) as e:   # the reference panics when the blocks do not fit
        ctx.assemble_advice_stream(k, n_sets, wit, kw, n + 1, check_bytes, layout=pkg.LAYOUT_PACKED)
    assert e.value.status == 5


@pytest.mark.parametrize("layout_name", ["values", "packed"])
def test_stream_2p24_blocks_with_verifying_consumer(ctx, pkg, oracle, layout_name):
    """BASELINE configs[4] on one GPU: 2^24 blocks through aesw_encrypt_witness_stream (kernel of chunk i+1 and its
    D2H overlap the consumer of chunk i).  The consumer checks size-independent properties on EVERY chunk (x rows 0..15
    are the plaintext; z rows 16..31 are plaintext ^ round key 0) and compares sampled chunks byte for byte with the oracle."""
    import torch
    layout = pkg.LAYOUT_VALUES if layout_name == "values" else pkg.LAYOUT_PACKED
    olayout = ol.VALUES if layout_name == "values" else ol.PACKED
    n = 1 << 24
    rng = np.random.default_rng(0xA35128 + 4)
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    ctx.schedule_key(torch.from_numpy(key).cuda(), layout=pkg.LAYOUT_PACKED, key_slab=False)
    torch.cuda.synchronize()
    strides = [pkg.column_stride(layout, c) for c in range(3)]
    chunk = ctx.get_option("chunk_blocks")
    n_chunks = -(-n // chunk)
    sampled = set(int(v) for v in np.linspace(0, n_chunks - 1, 6).astype(int))
    state = {"blocks": 0, "chunks": 0, "next": 0, "checked": 0}

    def consume(first, count, x, y, z):
        assert first == state["next"]
        state["next"] = first + count
        state["blocks"] += count
        state["chunks"] += 1
        p = pt[first:first + count]
        zz = z.reshape(count, strides[2])
        if not np.array_equal(zz[:, :16], p ^ key):          # rows 16..31 of z: s = pt ^ rk0 (src/aes128.rs:194-198)
            return 1
        if strides[0]:
            xx = x.reshape(count, strides[0])
            if not (np.array_equal(xx[:, :16], p) and np.array_equal(xx[:, 16:32], p)):   # :176-198
                return 1
        if first // chunk in sampled:
            m = min(count, 256)
            e = oracle.encrypt_witness(p[:m], key, layout=olayout)
            for got, exp, s in ((x, e.x, strides[0]), (y, e.y, strides[1]), (z, e.z, strides[2])):
                if s and not np.array_equal(got[:m * s], exp):
                    return 2
            state["checked"] += m
        return 0

    ctx.encrypt_witness_stream(pt, None, consume, layout=layout)
    assert state["blocks"] == n and state["chunks"] == n_chunks and state["checked"] >= 6 * 256
    st = ctx.last_stream_stats()
    assert st["chunks"] == n_chunks and st["bytes_to_host"] == n * sum(strides)
    assert 0 < st["kernel_ns"] < st["wall_ns"] and st["consumer_ns"] > 0


def test_assemble_advice_host_all_buffer_kinds(ctx, pkg, oracle):
    """aesw_assemble_advice_host into a pageable array, an aesw_host_alloc buffer and a caller-owned array pinned with
    aesw_host_register: the whole advice matrix equals the restated synthesize() (bytes, and Fr cells through the LUT)."""
    import torch
    k, n_sets = 14, 2
    n = pkg.block_capacity(k, n_sets)
    rng = np.random.default_rng(23)
    pt, key = rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)
    kw = ctx.schedule_key(torch.from_numpy(key).cuda(), layout=pkg.LAYOUT_PACKED, key_slab=True)
    wit = ctx.encrypt_witness(torch.from_numpy(pt).cuda(), None, layout=pkg.LAYOUT_PACKED)
    with oracle.circuit(k, n_sets, key, pt, record_copies=False) as c:
        expect = np.stack([c.advice(j) for j in range(c.num_advice)])
    lut = _fr_lut()
    ncols = 3 * n_sets + 1
    for as_fr in (False, True):
        nbytes = (ncols << k) * (32 if as_fr else 1)
        want = lut[expect].reshape(-1) if as_fr else expect.reshape(-1)
        pageable = np.full(nbytes, 0xEE, np.uint8)
        pinned = pkg.api.host_alloc(nbytes)
        registered = np.full(nbytes + 4096, 0xEE, np.uint8)[:nbytes]
        pkg.api.host_register(registered)
        try:
            for name, buf in (("pageable", pageable), ("aesw_host_alloc", pinned), ("aesw_host_register", registered)):
                ctx.assemble_advice_host(k, n_sets, wit, kw, n, buf, layout=pkg.LAYOUT_PACKED, as_fr=as_fr)
                assert np.array_equal(buf, want), "%s as_fr=%s" % (name, as_fr)
        finally:
            pkg.api.host_unregister(registered)
            pkg.api.host_free(pinned)
    with pytest.raises(ValueError):
        ctx.assemble_advice_host(k, n_sets, wit, kw, n, np.zeros(10, np.uint8))


def test_gather_call_sequence_for_three_ranks(pkg, tmp_path):
    """The multi-rank leg of aesw_gather_columns_device against a RECORDING stand-in for librccl (tests/mock_rccl/; no
    multi-GPU box is available to the builder): as root of 3 ranks it posts, inside ONE group, receives from ranks 1 and 2
    at their block offsets in pieces of at most max_message bytes and nothing for itself; as rank 2 it sends its range of
    every column to the root in the same pieces.  ncclUint8 everywhere."""
    import os
    mock_dir = ROOT / "tests" / "mock_rccl"
    lib_dir = ROOT / "halo2-aes_amd"
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", str(tmp_path / "librccl.so"), str(mock_dir / "mock_rccl.c")], check=True)
    exe = tmp_path / "gather_driver"
    subprocess.run(["gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
                    str(mock_dir / "gather_driver.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    counts, strides, maxmsg = [5, 3, 7], [1360, 1056, 608], 4000
    offs = [0, 5, 8]

    def run(rank):
        log = tmp_path / ("log%d" % rank)
        env = dict(os.environ, MOCK_RCCL_LOG=str(log), LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        out = subprocess.run([str(exe), "3", str(rank), str(maxmsg)] + [str(c) for c in counts], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True)
        assert out.returncode == 0 and "ok" in out.stdout, out.stdout
        base = {}
        for line in out.stdout.splitlines():
            if line.startswith("col"):
                _, c, _, s, _, r = line.split()
                base[int(c)] = (int(s), int(r))
        calls = [l.split() for l in log.read_text().splitlines()]
        return base, calls

    def pieces(nbytes):
        return [(o, min(maxmsg, nbytes - o)) for o in range(0, nbytes, maxmsg)]

    base, calls = run(0)
    assert calls[0][0] == "id" and calls[1][:3] == ["init", "3", "0"] and calls[1][3] == str(0x5a)
    body = calls[2:-1]
    assert body[0][0] == "group_start" and body[-1][0] == "group_end" and calls[-1][0] == "destroy"
    expect = []
    for c, s in enumerate(strides):
        for peer in (1, 2):
            for o, m in pieces(counts[peer] * s):
                expect.append(["recv", str(peer), str(base[c][1] + offs[peer] * s + o), str(m)])
    assert body[1:-1] == expect
    base, calls = run(2)
    body = calls[2:-1]
    expect = [["send", "0", str(base[c][0] + o), str(m)] for c, s in enumerate(strides) for o, m in pieces(counts[2] * s)]
    assert body[0][0] == "group_start" and body[-1][0] == "group_end" and body[1:-1] == expect
