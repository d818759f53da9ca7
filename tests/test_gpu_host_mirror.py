"""The host side of the boundary: the C++ mirror of the reference's interface
(halo2-aes_amd/host/) runs the reference's own circuits with every value
closure reading the device witness.  These tests read like the reference's:
MockProver::run(K, &circuit) then assert_satisfied() -- and, beyond that, the
whole advice matrix must equal what the oracle's restated synthesize() assigns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same_assembly(h, o, n_sets=None):
    assert h.num_advice == o.num_advice and h.num_selectors == o.num_selectors and h.num_rows == o.num_rows
    assert h.num_regions == o.num_regions
    assert h.num_copies == o.num_copies
    for c in range(h.num_advice):
        assert np.array_equal(h.advice_assigned(c), o.advice_assigned(c)), "assigned cells differ in advice column %d" % c
        assert np.array_equal(h.advice(c), o.advice(c)), "advice column %d differs" % c
    for s in range(h.num_selectors):
        assert np.array_equal(h.selector(s), o.selector(s)), "selector %d differs" % s
    assert np.array_equal(h.fixed(), o.fixed())


def test_correct_encryption(ctx, pkg, oracle):
    """src/aes128.rs:409-418: K=20, FixedAes128Config<K,3>, key = plaintext = 0, 1000 encryptions."""
    key = np.zeros(16, np.uint8)
    pts = np.zeros((1000, 16), np.uint8)
    with pkg.HostCircuit.aes(ctx, 20, 3, key, pts) as mock:
        rc, msg = mock.verify()        # mock.assert_satisfied()
        assert rc == 0, msg
        assert mock.ciphertext(0).tobytes().hex() == "66e94bd4ef8a2c3b884cfa59ca342b2e"
        # every value closure ran exactly once (the shape pass evaluates none)
        assigned = sum(int(mock.advice_assigned(c).sum()) for c in range(mock.num_advice))
        assert mock.closure_calls == assigned
        with oracle.circuit(20, 3, key, pts) as o:
            _same_assembly(mock, o)
        t = oracle.lookup_table()
        for c in range(4):
            assert np.array_equal(mock.table(c), t[c])
        # a corrupted witness byte is caught by the lookups / copy constraints
        mock.poke(2, 400 + 1360 * 7 + 20, int(mock.advice(2)[400 + 1360 * 7 + 20]) ^ 1)
        assert mock.verify()[0] == 8


def test_constraints_key_schedule(ctx, pkg, oracle):
    """src/key_schedule.rs:385-392: K=17, zero key."""
    key = np.zeros(16, np.uint8)
    with pkg.HostCircuit.key_schedule(ctx, 17, key) as mock:
        rc, msg = mock.verify()
        assert rc == 0, msg
        with oracle.key_circuit(17, key) as o:
            _same_assembly(mock, o)
        # src/key_schedule.rs:337-345 EXPANDED: the range-check rows hold the round keys
        kx = mock.advice(0)
        assert kx[40 * 9 + 24:40 * 10].tobytes().hex() == "b4ef5bcb3e92e21123e951cf6f8f188e"


def test_random_circuit_two_sets(ctx, pkg, oracle):
    rng = np.random.default_rng(0xA35128)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (60, 16), dtype=np.uint8)
    pts[5] = key ^ 0xFF   # S_BOX[0xff] on the first round's sbox rows
    with pkg.HostCircuit.aes(ctx, 16, 2, key, pts) as mock:
        rc, msg = mock.verify()
        assert rc == 0, msg
        with oracle.circuit(16, 2, key, pts) as o:
            _same_assembly(mock, o)
            for b in (0, 5, 45, 46, 59):
                assert np.array_equal(mock.ciphertext(b), o.ciphertext(b))


def test_dense_witness_is_still_an_option(ctx, pkg, oracle):
    """assign_mode 4: the mirror fed by the AESW_LAYOUT_DENSE witness (the default is PACKED since round 2) builds the
    same circuit, and a tampered cell is still caught through either layout."""
    rng = np.random.default_rng(77)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (40, 16), dtype=np.uint8)
    with pkg.HostCircuit.aes(ctx, 16, 2, key, pts, dense=True) as mock, oracle.circuit(16, 2, key, pts) as o:
        rc, msg = mock.verify()
        assert rc == 0, msg
        _same_assembly(mock, o)
        mock.poke(2, 400 + 31, int(mock.advice(2)[400 + 31]) ^ 1)   # z of the first block's last AddRoundKey row
        rc, msg = mock.verify()
        assert rc == 8 and msg


def test_values_only_witness_fills_the_whole_circuit(ctx, pkg, oracle):
    """AESW_LAYOUT_VALUES hands the host only the S-box / mul / xor outputs (1 056 B per block).  Running the
    reference's regions on it -- copy_advice() carrying every other value, as in the reference -- must assign
    the very same circuit as the restated synthesize(): every advice cell, selector, copy and the fixed column."""
    rng = np.random.default_rng(0xA35128 + 9)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (60, 16), dtype=np.uint8)
    pts[7] = key ^ 0xFF   # S_BOX[0xff]
    with pkg.HostCircuit.aes(ctx, 16, 2, key, pts, values_only=True) as mock:
        rc, msg = mock.verify()
        assert rc == 0, msg
        with oracle.circuit(16, 2, key, pts) as o:
            _same_assembly(mock, o)
            for b in (0, 7, 45, 46, 59):
                assert np.array_equal(mock.ciphertext(b), o.ciphertext(b))
    # the reference's own integration test, on the values-only witness
    with pkg.HostCircuit.aes(ctx, 20, 3, np.zeros(16, np.uint8), np.zeros((1000, 16), np.uint8), values_only=True) as mock:
        assert mock.verify() == (0, "")
        assert mock.ciphertext(999).tobytes().hex() == "66e94bd4ef8a2c3b884cfa59ca342b2e"


def test_streaming_assign_overlaps_device_and_host(ctx, pkg, oracle):
    """BASELINE configs[4] shape on the host side: synthesize() assigns chunk i (values-only witness, the reference's
    regions) while the device produces chunk i+1 and copies it over.  Same circuit as the restated synthesize(),
    chunk boundaries in the middle of column sets included; panics and errors raised inside a chunk still surface."""
    rng = np.random.default_rng(0xA35128 + 10)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (180, 16), dtype=np.uint8)
    ctx.set_option("chunk_blocks", 64)   # three chunks: 64 + 64 + 52; set 0 holds 46 blocks at K = 16
    try:
        with pkg.HostCircuit.aes(ctx, 16, 4, key, pts, streaming=True) as mock:
            assert mock.verify() == (0, "")
            with oracle.circuit(16, 4, key, pts) as o:
                _same_assembly(mock, o)
                for b in (0, 63, 64, 127, 128, 179):
                    assert np.array_equal(mock.ciphertext(b), o.ciphertext(b))
        with pytest.raises(pkg.AeswError) as e:   # capacity panic raised inside the second chunk
            pkg.HostCircuit.aes(ctx, 16, 1, key, pts[:100], streaming=True)
        assert e.value.status == 5 and "AES calls too many" in str(e.value)
        # the context is still usable afterwards
        with pkg.HostCircuit.aes(ctx, 16, 1, key, pts[:40], streaming=True) as mock:
            assert mock.verify() == (0, "")
    finally:
        ctx.set_option("chunk_blocks", 1 << 15)


def test_whole_circuit_from_columns_and_keygen_data(ctx, pkg, oracle):
    """No region is run: advice columns are the device witness placed by aesw_block_placement, selectors / fixed column /
    table / equality constraints are the library's input-independent keygen data.  The result must be the very circuit the
    restated synthesize() builds -- cells, assigned masks, selectors, fixed column, and the copies in the same order -- and
    MockProver must accept it and reject a corrupted byte."""
    rng = np.random.default_rng(0xA35128 + 12)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (130, 16), dtype=np.uint8)
    pts[3] = key ^ 0xFF
    with pkg.HostCircuit.aes_columns(ctx, 16, 3, key, pts) as mock, oracle.circuit(16, 3, key, pts) as o:
        assert mock.verify() == (0, "")
        assert mock.num_regions == 0 and mock.closure_calls == 0
        assert mock.num_advice == o.num_advice and mock.num_selectors == o.num_selectors and mock.num_copies == o.num_copies
        for c in range(mock.num_advice):
            assert np.array_equal(mock.advice_assigned(c), o.advice_assigned(c)), c
            assert np.array_equal(mock.advice(c), o.advice(c)), c
        for s_ in range(mock.num_selectors):
            assert np.array_equal(mock.selector(s_), o.selector(s_)), s_
        assert np.array_equal(mock.fixed(), o.fixed())
        oc = o.copies()           # (source col, source row, copy col, copy row)
        mc = mock.copies()        # (copy col, copy row, original col, original row)
        assert np.array_equal(mc[:, [2, 3, 0, 1]], oc)
        t = oracle.lookup_table()
        for c in range(4):
            assert np.array_equal(mock.table(c), t[c])
        for b in (0, 3, 45, 46, 129):
            assert np.array_equal(mock.ciphertext(b), o.ciphertext(b))
        mock.poke(1, 400 + 1360 * 5 + 40, int(mock.advice(1)[400 + 1360 * 5 + 40]) ^ 4)
        assert mock.verify()[0] == 8
    with pytest.raises(pkg.AeswError) as e:
        pkg.HostCircuit.aes_columns(ctx, 16, 1, key, pts)   # 46 blocks fit one set at K = 16
    assert e.value.status == 5
    # the reference's own integration test, K = 20, N = 3, 1 000 blocks
    with pkg.HostCircuit.aes_columns(ctx, 20, 3, np.zeros(16, np.uint8), np.zeros((1000, 16), np.uint8)) as mock:
        assert mock.verify() == (0, "")
        assert mock.ciphertext(0).tobytes().hex() == "66e94bd4ef8a2c3b884cfa59ca342b2e"


def test_bulk_assign_equals_per_region(ctx, pkg, oracle):
    """one 1 360-row region per block (after the first) == the reference's 1 360 one-row regions:
    same advice cells, selectors, fixed column and the same SET of equality constraints; far fewer regions."""
    rng = np.random.default_rng(31)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    pts = rng.integers(0, 256, (70, 16), dtype=np.uint8)
    with pkg.HostCircuit.aes(ctx, 16, 2, key, pts) as a, pkg.HostCircuit.aes(ctx, 16, 2, key, pts, bulk_assign=True) as b:
        assert b.verify() == (0, "")
        for c in range(a.num_advice):
            assert np.array_equal(a.advice(c), b.advice(c)) and np.array_equal(a.advice_assigned(c), b.advice_assigned(c))
        for s in range(a.num_selectors):
            assert np.array_equal(a.selector(s), b.selector(s))
        assert np.array_equal(a.fixed(), b.fixed())
        ca, cb = a.copies(), b.copies()
        assert ca.shape == cb.shape
        key_of = lambda m: np.lexsort((m[:, 3], m[:, 2], m[:, 1], m[:, 0]))
        assert np.array_equal(ca[key_of(ca)], cb[key_of(cb)])
        # table(1) + key schedule(421) + first block(1345) + 69 bulk blocks, against 1345 regions per block
        assert a.num_regions == 421 + 70 * 1345 and b.num_regions == 421 + 1345 + 69
        for blk in (0, 1, 45, 46, 69):
            assert np.array_equal(a.ciphertext(blk), b.ciphertext(blk))
        b.poke(1, 400 + 1360 * 3 + 40, int(b.advice(1)[400 + 1360 * 3 + 40]) ^ 0x80)
        assert b.verify()[0] == 8
    with oracle.circuit(16, 2, key, pts) as o, pkg.HostCircuit.aes(ctx, 16, 2, key, pts, bulk_assign=True) as b:
        for c in range(b.num_advice):
            assert np.array_equal(b.advice(c), o.advice(c))


def test_keygen_pass_evaluates_no_closure(ctx, pkg):
    """keygen_vk / keygen_pk ignore value closures: nothing is read, selectors are still laid out."""
    pts = np.zeros((3, 16), np.uint8)
    with pkg.HostCircuit.aes(ctx, 13, 1, np.zeros(16, np.uint8), pts, with_witnesses=False) as mock:
        assert mock.closure_calls == 0
        assert int(mock.advice_assigned(0).sum()) == 0
        enc, key, _, _ = pkg.selector_tags()
        xor_sel = mock.selector(1)
        assert np.array_equal(xor_sel[:400], (key == 2).astype(np.uint8))
        assert np.array_equal(xor_sel[400:1760], (enc == 2).astype(np.uint8))


def test_capacity_panic(ctx, pkg):
    """src/aes128.rs:159-162 panic!("AES calls too many ..."): K=12, N=1 holds one block."""
    key = np.zeros(16, np.uint8)
    with pkg.HostCircuit.aes(ctx, 12, 1, key, np.zeros((1, 16), np.uint8)) as mock:
        assert mock.verify()[0] == 0
    with pytest.raises(pkg.AeswError) as e:
        pkg.HostCircuit.aes(ctx, 12, 1, key, np.zeros((2, 16), np.uint8))
    assert e.value.status == 5 and "AES calls too many" in str(e.value)


def test_keys_should_be_scheduled(ctx, pkg):
    """src/aes128.rs:170 expect("Keys should be scheduled")."""
    with pytest.raises(pkg.AeswError) as e:
        pkg.HostCircuit.aes(ctx, 13, 1, np.zeros(16, np.uint8), np.zeros((1, 16), np.uint8), skip_schedule_key=True)
    assert e.value.status == 6 and "Keys should be scheduled" in str(e.value)


@pytest.mark.parametrize("layout_name", ["dense", "packed"])
def test_assembled_advice_columns_equal_synthesize(ctx, pkg, oracle, layout_name):
    """aesw_assemble_advice_device: the full advice columns (bytes, and 32-byte Fr cells) of a K=15, N=3
    circuit straight from the device == what the restated synthesize() assigns, zeros elsewhere."""
    import torch
    layout = pkg.LAYOUT_DENSE if layout_name == "dense" else pkg.LAYOUT_PACKED
    rng = np.random.default_rng(77)
    key = rng.integers(0, 256, 16, dtype=np.uint8)
    k, n_sets = 15, 3
    n = pkg.block_capacity(k, n_sets) - 1            # 22 + 24 + 24 - 1: the last set stays partly empty
    pts = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    kw = ctx.schedule_key(torch.from_numpy(key).cuda(), layout=layout)
    w = ctx.encrypt_witness(torch.from_numpy(pts).cuda(), None, layout=layout)
    cols = ctx.assemble_advice(k, n_sets, w, kw, n, layout=layout)
    fr = ctx.assemble_advice(k, n_sets, w, kw, n, layout=layout, as_fr=True)
    torch.cuda.synchronize()
    cols, fr = cols.cpu().numpy(), fr.cpu().numpy()
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    lut = np.stack([np.frombuffer(((v << 256) % r).to_bytes(32, "little"), np.uint8) for v in range(256)])
    with oracle.circuit(k, n_sets, key, pts, record_copies=False) as o:
        assert o.status == 0
        for c in range(3 * n_sets + 1):
            exp = o.advice(c)
            assert np.array_equal(cols[c], exp), "advice column %d" % c
            assert np.array_equal(fr[c], lut[exp]), "Fr cells of advice column %d" % c
    with pytest.raises(pkg.AeswError) as e:
        ctx.assemble_advice(k, n_sets, w, kw, n + 2, layout=layout)
    assert e.value.status == 5
