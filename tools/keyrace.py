#!/usr/bin/env python3
"""Does a build of the library lose a reader when a key is re-scheduled?   usage: keyrace.py [LIB.so] [ROUNDS] [KEY_SLOTS]

The scenario of VERDICT r03 weak 1, raw ctypes so that older builds can be driven too: a 2^20-block scheduled-key launch on
stream A, a 4 096-block one on stream B, a re-schedule on stream C.  The long launch's ciphertexts are compared with the same
library's own output for the same inputs with the OLD key passed as a shared key on a quiet device; blocks that differ were
encrypted with round keys the re-schedule had already overwritten.  Round 3's library (one `key_last_use` event): C waits for B
only.  Round 4: one event per reader stream and a ring of slots; KEY_SLOTS = 1 forces the re-schedule onto the slot in use."""
import ctypes as C
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
ge.build()
pkg = ge.load_package()
path = Path(sys.argv[1]).resolve() if len(sys.argv) > 1 and sys.argv[1] != "-" else None
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
key_slots = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = C.CDLL(str(path)) if path else pkg.load_library()
for name in ("aesw_create", "aesw_schedule_key_device", "aesw_encrypt_witness_device", "aesw_set_option", "aesw_destroy"):
    res, args = pkg.api.SYMBOLS[name]
    getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
t = [x.copy() for x in pkg.reference_tables()]
h = C.c_void_p()
assert lib.aesw_create(C.byref(h), 0, *[x.ctypes.data_as(C.c_void_p) for x in t]) == 0
if key_slots:
    assert lib.aesw_set_option(h, b"key_slots", key_slots) == 0, "this build has no key_slots option"
L = pkg.LAYOUT_PACKED
n_a, n_b = 1 << 20, 4096
g = torch.Generator(device="cpu").manual_seed(7)
pt = torch.randint(0, 256, (n_a, 16), dtype=torch.uint8, generator=g).cuda()
keys = [torch.randint(0, 256, (16,), dtype=torch.uint8, generator=g).cuda() for _ in range(rounds + 1)]


def cols(n):
    return [torch.empty(n * pkg.column_stride(L, c), dtype=torch.uint8, device="cuda") for c in range(3)] + \
           [torch.empty((n, 16), dtype=torch.uint8, device="cuda")]


def enc(p, key, out, stream):
    n = p.shape[0]
    rc = lib.aesw_encrypt_witness_device(h, p.data_ptr(), key.data_ptr() if key is not None else None, 0, n, L, out[0].data_ptr(),
                                         out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), None, stream.cuda_stream)
    assert rc == 0, rc


oa, ob, oc, ref = cols(n_a), cols(n_b), cols(n_b), cols(n_a)
sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
assert lib.aesw_schedule_key_device(h, keys[0].data_ptr(), L, None, sc.cuda_stream) == 0
torch.cuda.synchronize()
bad_rounds, bad_blocks = 0, 0
for r in range(rounds):
    enc(pt, None, oa, sa)
    enc(pt[-n_b:], None, ob, sb)
    assert lib.aesw_schedule_key_device(h, keys[r + 1].data_ptr(), L, None, sc.cuda_stream) == 0
    enc(pt[:n_b], None, oc, sc)
    torch.cuda.synchronize()
    enc(pt, keys[r], ref, sc)   # the old key as a shared key, nothing else running
    torch.cuda.synchronize()
    diff = int((oa[3] != ref[3]).any(dim=1).sum())
    same_cols = all(bool(torch.equal(oa[i], ref[i])) for i in range(3))
    if diff or not same_cols:
        bad_rounds += 1
        bad_blocks += diff
        first = int((oa[3] != ref[3]).any(dim=1).nonzero()[0]) if diff else -1
        print("round %2d: %7d of %d blocks of the long launch were encrypted with the NEW key (first: block %d)" % (r, diff, n_a, first))
print("%s key_slots=%s: %d of %d rounds lost a reader, %d blocks in all" % (path.name if path else "in-tree libaesw.so", key_slots or "default", bad_rounds, rounds, bad_blocks))
lib.aesw_destroy(h)
