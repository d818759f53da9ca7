#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

  reference_constants.json  DATA lifted from the reference's own files (needs
      /root/reference): the three byte tables of src/constant.rs:1-47, the row
      constants (:113-114), ROUND_CONSTANT (src/utils.rs:28), the EXPANDED
      zero-key round keys of src/key_schedule.rs:337-345 and the xor KAT of
      src/utils.rs:40-47.  Only values are stored, no source text.
  slab_vectors.npz          inputs and expected witness slabs produced by the
      CPU oracle (oracle/aesw_oracle.c) for a handful of blocks: the reference's
      all-zero vector, FIPS-197 App. B and C.1, a block whose first S-box input
      is 0xff (reaches the reference's S_BOX[255]==23) and seeded random blocks.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import re
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT / "tests"))


def lift_reference_constants():
    src = (REF / "src" / "constant.rs").read_text()

    def table(name):
        m = re.search(r"pub const %s: \[u8; 256\] = \[(.*?)\];" % name, src, re.S)
        vals = [int(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
        assert len(vals) == 256
        return vals

    def const(name):
        return int(re.search(r"const %s: u64 = (\d+);" % name, src).group(1))

    utils = (REF / "src" / "utils.rs").read_text()
    rc = [int(v) for v in re.search(r"ROUND_CONSTANT: \[u64; 10\] = \[(.*?)\];", utils).group(1).split(",")]
    ks = (REF / "src" / "key_schedule.rs").read_text()
    exp = re.findall(r'"([0-9a-f]{8})"', re.search(r"const EXPANDED: \[&str; 44\] = \[(.*?)\];", ks, re.S).group(1))
    assert len(exp) == 44
    return {
        "source": "tkmct/halo2-aes @ 2024_08_07: src/constant.rs:1-47,113-114; src/utils.rs:28,40-47; src/key_schedule.rs:337-345",
        "S_BOX": table("S_BOX"), "MUL_BY_2": table("MUL_BY_2"), "MUL_BY_3": table("MUL_BY_3"),
        "AES_ROWS": const("AES_ROWS"), "KEY_SCHEDULE_ROWS": const("KEY_SCHEDULE_ROWS"),
        "ROUND_CONSTANT": rc, "EXPANDED_ZERO_KEY": exp,
        "xor_kat": {"x": 5, "y": 12, "z": 9},
    }


def slab_vectors():
    import oracle_lib as ol
    orc = ol.Oracle()
    rng = np.random.default_rng(0xA35128)
    pt = [bytes(16), bytes.fromhex("3243f6a8885a308d313198a2e0370734"), bytes.fromhex("00112233445566778899aabbccddeeff"),
          bytes([0xFF] * 16)]
    key = [bytes(16), bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c"), bytes.fromhex("000102030405060708090a0b0c0d0e0f"),
           bytes(16)]
    pt += [rng.integers(0, 256, 16, dtype=np.uint8).tobytes() for _ in range(13)]
    key += [rng.integers(0, 256, 16, dtype=np.uint8).tobytes() for _ in range(13)]
    pt = np.frombuffer(b"".join(pt), np.uint8).reshape(-1, 16).copy()
    key = np.frombuffer(b"".join(key), np.uint8).reshape(-1, 16).copy()
    out = {"pt": pt, "keys": key}
    for layout, name in ((ol.DENSE, "dense"), (ol.PACKED, "packed"), (ol.VALUES, "values")):
        w = orc.encrypt_witness(pt, key, layout=layout)
        k = orc.key_schedule_witness(key, layout=layout)
        for c in "xyz":
            out["%s_%s" % (name, c)] = getattr(w, c)
        out["%s_ct" % name] = w.ct
        for c in ("w", "kx", "ky", "kz", "rk"):
            out["%s_%s" % (name, c)] = getattr(k, c)
        # the same plaintexts under ONE shared key (key[4])
        ws = orc.encrypt_witness(pt, key[4], layout=layout)
        for c in "xyz":
            out["%s_shared_%s" % (name, c)] = getattr(ws, c)
    return out


def main():
    if REF.exists():
        (HERE / "reference_constants.json").write_text(json.dumps(lift_reference_constants(), indent=0) + "\n")
        print("wrote reference_constants.json")
    else:
        print("no /root/reference here: reference_constants.json left as committed")
    np.savez_compressed(HERE / "slab_vectors.npz", **slab_vectors())
    print("wrote slab_vectors.npz")


if __name__ == "__main__":
    main()
