#!/usr/bin/env python3
"""Soak: many byte-exact comparisons against the oracle at sizes where timing-dependent faults show
(store-data hazards, LDS ordering), random layout / key mode / size / launch options per iteration; every third iteration
the five assemble geometries on a random circuit shape, every fourth three launches in flight on three streams.
Round 4: a random "key_slots" ring and the arena cache on or off per context, the device checker (aesw_check_witness_device) over
every witness next to the oracle comparison -- it must say "satisfied", and "not satisfied" after one random byte is changed --, and
every fifth iteration a re-schedule behind two reader streams (the scenario of tools/keyrace.py) against the oracle."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import oracle_lib as ol  # noqa: E402

ge.build()
pkg = ge.load_package()
orc = ol.Oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
threads = min(16, os.cpu_count() or 8)
t0 = time.time()
for it in range(iters):
    layout = int(rng.integers(0, 3))
    keymode = int(rng.integers(0, 3))
    n = int(rng.integers(1 << 16, 1 << 18)) + int(rng.integers(0, 64))
    ctx = pkg.Context(0)
    ctx.set_option("waves_shared", int(rng.integers(0, 5)))
    ctx.set_option("waves_pbk", int(rng.integers(0, 5)))
    ctx.set_option("store_mode", int(rng.integers(0, 3)))
    if rng.integers(0, 4) == 0:
        ctx.set_option("grid_cap", int(rng.integers(1, 2048)))
    ctx.set_option("xcd_remap", int(rng.choice([0, 1, 1, 2, 7, 64, 1000, 100000])))
    ctx.set_option("key_slots", int(rng.choice([1, 1, 2, 4, 7])))
    ctx.set_option("arena_cache", int(rng.integers(0, 2)))
    arena = bool(rng.integers(0, 2))  # round 3: output columns in a probed arena (virtual-memory backed) or in plain tensors
    ctx.set_option("arena_probe", int(rng.integers(1, 4)))
    ctx.set_option("arena_unit", int(rng.integers(0, 3)))
    pt = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    keys = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    dpt = torch.from_numpy(pt).cuda()
    out = ctx.alloc_columns(n, layout, want_ct=True, key_slab=keymode == 2) if arena else None
    if keymode == 0:
        ctx.schedule_key(torch.from_numpy(keys[0]).cuda(), layout=layout, key_slab=False)
        got = ctx.encrypt_witness(dpt, None, layout=layout, want_ct=True, out=out)
        kh = keys[0]
    elif keymode == 1:
        got = ctx.encrypt_witness(dpt, torch.from_numpy(keys[0]).cuda(), layout=layout, want_ct=True, out=out)
        kh = keys[0]
    else:
        got = ctx.encrypt_witness(dpt, torch.from_numpy(keys).cuda(), layout=layout, want_ct=True, key_slab=True, out=out)
        kh = keys
    torch.cuda.synchronize()
    exp = orc.encrypt_witness(pt, kh, layout=layout, threads=threads)
    for c in "xyz":
        a, e = getattr(got, c).cpu().numpy(), getattr(exp, c)
        if not np.array_equal(a, e):
            bad = np.nonzero(a != e)[0]
            print("MISMATCH iter %d layout %d keymode %d n %d column %s: %d bytes, first %d" % (it, layout, keymode, n, c, bad.size, bad[0]))
            sys.exit(1)
    if not np.array_equal(got.ct.cpu().numpy(), exp.ct):
        print("MISMATCH ct iter", it)
        sys.exit(1)
    if keymode == 2:
        kexp = orc.key_schedule_witness(keys, layout=layout, threads=threads)
        for c in ("w", "kx", "ky", "kz"):
            if not np.array_equal(getattr(got.key, c).cpu().numpy(), getattr(kexp, c)):
                print("MISMATCH key slab", c, "iter", it)
                sys.exit(1)
    if layout != pkg.LAYOUT_VALUES:
        # round 4: the reference's own criterion (constraint satisfaction) over every block, on the device
        kwit = got.key if keymode == 2 else ctx.schedule_key(torch.from_numpy(keys[0]).cuda(), layout=layout, key_slab=True)
        dk = torch.from_numpy(kh).cuda()
        rep = ctx.check_witness(dpt, dk, got, kwit, layout=layout, ct=got.ct)
        if not rep["satisfied"] or rep["blocks"] != n:
            print("DEVICE CHECK FAILED on a witness equal to the oracle's: iter %d %r" % (it, rep))
            sys.exit(1)
        col = [got.x, got.y, got.z][int(rng.integers(0, 3))]
        pos = int(rng.integers(0, col.numel()))
        col[pos] ^= int(rng.integers(1, 256))
        rep = ctx.check_witness(dpt, dk, got, kwit, layout=layout, ct=got.ct)
        if layout == pkg.LAYOUT_PACKED and rep["satisfied"]:
            print("DEVICE CHECK MISSED a changed byte: iter %d position %d" % (it, pos))  # (DENSE holds never-assigned cells: a change there is legal)
            sys.exit(1)
    if it % 5 == 2:
        # round 4: a long scheduled-key launch on A, a short one on B, a re-schedule on C (tools/keyrace.py) -- against the oracle
        sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        ka, kb = torch.from_numpy(keys[3]).cuda(), torch.from_numpy(keys[4]).cuda()
        m = min(n, 1 << 17)
        oa, ob, oc = (ctx.alloc_witness(q, pkg.LAYOUT_PACKED, want_ct=True) for q in (m, 512, 512))
        with torch.cuda.stream(sc):
            ctx.schedule_key(ka, key_slab=False)
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            ctx.encrypt_witness(dpt[:m], None, out=oa, want_ct=True)
        with torch.cuda.stream(sb):
            ctx.encrypt_witness(dpt[:512], None, out=ob, want_ct=True)
        with torch.cuda.stream(sc):
            ctx.schedule_key(kb, key_slab=False)
            ctx.encrypt_witness(dpt[:512], None, out=oc, want_ct=True)
        torch.cuda.synchronize()
        ea = orc.encrypt_witness(pt[:m], keys[3], layout=ol.PACKED, threads=threads)
        ec = orc.encrypt_witness(pt[:512], keys[4], layout=ol.PACKED)
        if not (np.array_equal(oa.z.cpu().numpy(), ea.z) and np.array_equal(oa.ct.cpu().numpy(), ea.ct) and
                np.array_equal(ob.ct.cpu().numpy(), ea.ct[:512]) and np.array_equal(oc.ct.cpu().numpy(), ec.ct)):
            print("MISMATCH re-schedule behind two reader streams: iter %d key_slots %d" % (it, ctx.get_option("key_slots")))
            sys.exit(1)
    if it % 3 == 0 and pkg.block_capacity(11, 2) > 0:
        # round 3: the Fr form of assemble in its four geometries on a random circuit shape (partly filled last set), all equal,
        # and equal to the restated synthesize() through the byte -> Fr table
        k, n_sets = int(rng.integers(11, 17)), int(rng.integers(2, 5))
        cap = pkg.block_capacity(k, n_sets)
        nb = int(rng.integers(max(1, cap - 3), cap + 1))
        lay = pkg.LAYOUT_PACKED if rng.integers(0, 2) else pkg.LAYOUT_DENSE
        akey = torch.from_numpy(keys[1]).cuda()
        kw = ctx.schedule_key(akey, layout=lay, key_slab=True)
        wit = ctx.encrypt_witness(dpt[:nb], None, layout=lay)
        outs = []
        for geo in range(5):
            ctx.set_option("assemble_geometry", geo)
            outs.append(ctx.assemble_advice(k, n_sets, wit, kw, nb, layout=lay, as_fr=True))
        torch.cuda.synchronize()
        for geo in range(1, 5):
            if not torch.equal(outs[0], outs[geo]):
                print("MISMATCH assemble geometry %d vs 0: iter %d K %d N %d layout %d blocks %d" % (geo, it, k, n_sets, lay, nb))
                sys.exit(1)
        if k <= 14:
            if "fr_lut" not in globals():
                fr_mod = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
                fr_lut = np.stack([np.frombuffer(((v << 256) % fr_mod).to_bytes(32, "little"), np.uint8) for v in range(256)])
            a3 = outs[4].cpu().numpy()
            with orc.circuit(k, n_sets, keys[1], pt[:nb], record_copies=False) as c:
                for col in range(3 * n_sets + 1):
                    if not np.array_equal(a3[col], fr_lut[c.advice(col)]):
                        print("MISMATCH assemble vs synthesize: iter %d K %d N %d layout %d column %d" % (it, k, n_sets, lay, col))
                        sys.exit(1)
    if it % 4 == 1:
        # round 3: three launches with different inputs in flight on three streams of this context
        streams = [torch.cuda.Stream() for _ in range(3)]
        m = min(n, 1 << 16)
        ins = [torch.from_numpy(np.roll(pt[:m], j + 1, axis=0)).cuda() for j in range(3)]
        dk = torch.from_numpy(keys[2]).cuda()
        torch.cuda.synchronize()
        res = []
        for j in range(3):
            with torch.cuda.stream(streams[j]):
                res.append(ctx.encrypt_witness(ins[j], dk, layout=layout, want_ct=True))
        torch.cuda.synchronize()
        for j in range(3):
            e = orc.encrypt_witness(np.roll(pt[:m], j + 1, axis=0), keys[2], layout=layout, threads=threads)
            for c in "xyz":
                if not np.array_equal(getattr(res[j], c).cpu().numpy(), getattr(e, c)):
                    print("MISMATCH concurrent launch %d column %s iter %d" % (j, c, it))
                    sys.exit(1)
    ctx.close()
    if it % 10 == 9:
        print("iter %d ok (%.0f s)" % (it + 1, time.time() - t0), flush=True)
print("soak ok: %d iterations" % iters)
