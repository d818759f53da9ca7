"""Static checks on the gfx950 code object inside libaesw.so (CPU test: hipcc cross-compiles, llvm-objdump disassembles).

Why: the flush stores are `global_store_dwordx4 ... sc1` issued through inline asm (csrc/aesw_kernels.hip gstore /
gstore_at).  hipcc neither counts nor pads an inline-asm store, so the "VMEM store with more than 64 bits of data, then
a write of its data VGPRs" hazard is covered by a hand-placed `s_nop 1` inside the same asm statement.  Without it the
2^20-block test (and only that one) produced wrong bytes -- a silent-corruption class whose only guard used
to be a dynamic stress test.  This lint asserts, for every kernel in the library:
  * every `global_store_dwordx4 ... sc1` and every store in the SGPR-base form `global_store_dwordx4 v, v[..], s[..]` (the
    form gstore_at emits for all three flavours) is IMMEDIATELY followed by `s_nop >= 1`, every such dwordx2 store by an
    `s_nop` (the compiler emits neither sc1 nor SGPR-base stores by itself in this library: these are the inline-asm ones);
  * no scratch (`.private_segment_fixed_size == 0`), no VGPR spills, `.vgpr_count <= 256` (8 waves per CU);
and it compares every kernel's register / LDS / spill / store-instruction counts with the tracked table
`profiles/isa_resources.json`, so that a silent jump (104 -> 201 VGPRs happened once between two commits) shows up as a
diff a reviewer sees.  Regenerate the table on purpose with  AESW_UPDATE_ISA_JSON=1 python -m pytest tests/test_isa_lint.py
"""
import json
import os
import re
import shutil
import subprocess
from pathlib import Path

import pytest
import yaml

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
TABLE = ROOT / "profiles" / "isa_resources.json"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

needs_llvm = pytest.mark.skipif(not (LLVM / "llvm-objdump").exists() or shutil.which("objcopy") is None or shutil.which("c++filt") is None,
                                reason="needs the ROCm LLVM tools, objcopy and c++filt")


@pytest.fixture(scope="module")
def code_object(pkg, tmp_path_factory):
    """The gfx950 code object of the built library, its disassembly and its kernel metadata."""
    lib = ROOT / "halo2-aes_amd" / "libaesw.so"
    d = tmp_path_factory.mktemp("isa")
    fat, co = d / "fat.bin", d / "k.co"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=" + TARGET, "--input=" + str(fat),
                    "--output=" + str(co), "--unbundle"], check=True)
    asm = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], stdout=subprocess.PIPE, text=True, check=True).stdout
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], stdout=subprocess.PIPE, text=True, check=True).stdout
    meta = yaml.safe_load(notes[notes.index("---"):notes.index("...", notes.index("---"))])
    names = [k[".name"] for k in meta["amdhsa.kernels"]]
    dem = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
    # per-function instruction lists: "<mangled>:" labels, then "\tmnemonic operands // addr: encoding"
    funcs, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith("\t"):
            cur.append(line.split("//")[0].strip())
    return {"meta": {k[".name"]: k for k in meta["amdhsa.kernels"]}, "demangled": dict(zip(names, dem)), "funcs": funcs}


def _short(demangled):
    """aesw::encrypt_kernel<1, true, 0, true, 2>(aesw::EncParams) -> encrypt_kernel<1,true,0,true,2>"""
    s = re.sub(r"\(.*\)$", "", demangled.replace("void ", "").replace("aesw::", ""))
    return s.replace(", ", ",")


@needs_llvm
def test_every_inline_asm_store_is_padded(code_object):
    bad, n4, n2 = [], 0, 0
    for name, ins in code_object["funcs"].items():
        if name not in code_object["meta"]:
            continue
        if _short(code_object["demangled"][name]).startswith("check_kernel<"):
            continue  # no inline-asm store in it: its one store (the report's unit counts, a uniform pointer) is the compiler's own, hazards included
        for i, text in enumerate(ins):
            # inline-asm stores: every sc1 store, and every store in the SGPR-base form (gstore_at emits all three flavours so)
            m = re.match(r"^global_store_dwordx([24])\b.*(\bsc1\b|, s\[\d+:\d+\])", text)
            if not m:
                continue
            wide = m.group(1) == "4"
            n4 += wide
            n2 += not wide
            nxt = ins[i + 1] if i + 1 < len(ins) else ""
            mn = re.match(r"^s_nop (\d+)$", nxt)
            if not mn or (wide and int(mn.group(1)) < 1):
                bad.append("%s: instruction %d `%s` is followed by `%s`" % (code_object["demangled"][name], i, text, nxt))
    assert n4 > 1000, "the sc1 stores of the witness kernels were not found: has the store flavour changed? (%d, %d)" % (n4, n2)
    assert not bad, "\n".join(bad[:20])


@needs_llvm
def test_no_scratch_no_vgpr_spills_and_the_tracked_resource_table(code_object):
    table, over = {}, []
    for name, k in code_object["meta"].items():
        ins = code_object["funcs"].get(name, [])
        short = _short(code_object["demangled"][name])
        assert k[".private_segment_fixed_size"] == 0, "%s uses %d B of scratch" % (short, k[".private_segment_fixed_size"])
        assert k.get(".vgpr_spill_count", 0) == 0, "%s spills VGPRs" % short
        # 256 unified registers = two waves per SIMD, what the 7 one-wave groups (packed) / two 3-wave groups per CU need.
        # The DENSE instantiations (a demoted option: 35 % of what they write are zeros) are allowed the whole file: they
        # run one wave per SIMD (two 2-wave groups per CU) and are launch-bounded accordingly in the source.
        limit = 512 if short.startswith(("encrypt_kernel<0,", "key_kernel<0,")) else 256
        if k[".vgpr_count"] > limit:
            over.append((short, k[".vgpr_count"]))
        table[short] = {
            "vgpr": k[".vgpr_count"], "agpr": k.get(".agpr_count", 0), "sgpr": k[".sgpr_count"],
            "sgpr_spill": k.get(".sgpr_spill_count", 0), "static_lds": k[".group_segment_fixed_size"],
            "instructions": len(ins),
            "stores_x4_sc1": sum(1 for t in ins if re.match(r"^global_store_dwordx4\b.*\bsc1\b", t)),
            "stores_x4_other": sum(1 for t in ins if re.match(r"^global_store_dwordx4\b", t) and " sc1" not in t),
            "ds_read_b128": sum(1 for t in ins if t.startswith("ds_read_b128")),
            "v_perm_b32": sum(1 for t in ins if t.startswith("v_perm_b32")),
            "readlane_writelane": sum(1 for t in ins if t.startswith(("v_readlane_b32", "v_writelane_b32"))),
        }
    assert any(k.startswith("encrypt_kernel<1,true,0,true,2>") for k in table), sorted(table)[:5]
    assert not over, "more unified registers than the launch geometry allows: %r" % over
    table = dict(sorted(table.items()))
    if os.environ.get("AESW_UPDATE_ISA_JSON"):
        TABLE.write_text(json.dumps(table, indent=1) + "\n")
    assert TABLE.exists(), "profiles/isa_resources.json is missing: run with AESW_UPDATE_ISA_JSON=1 and commit it"
    tracked = json.loads(TABLE.read_text())
    drift = []
    for name in sorted(set(table) | set(tracked)):
        a, b = tracked.get(name), table.get(name)
        if a != b:
            if a and b:
                what = ", ".join("%s %s -> %s" % (f, a.get(f), b.get(f)) for f in b if a.get(f) != b.get(f))
            else:
                what = "only in the %s" % ("tracked table" if a else "built library")
            drift.append("%s: %s" % (name, what))
    assert not drift, ("the built kernels differ from profiles/isa_resources.json (regenerate it with AESW_UPDATE_ISA_JSON=1 "
                       "and commit the diff if the change is intended):\n" + "\n".join(drift[:30]))


@needs_llvm
def test_headline_kernel_shape(code_object):
    """The instantiation bench.py's headline launches: a hand-written CDNA4 kernel, not a byte loop."""
    table = json.loads(TABLE.read_text())
    k = table["encrypt_kernel<1,true,0,true,1>"]  # packed, xtime path, per-block keys + key witness, nontemporal stores
    assert k["stores_x4_other"] >= 100 and k["stores_x4_sc1"] == 0 and k["ds_read_b128"] >= 100 and k["v_perm_b32"] >= 300
    assert k["vgpr"] <= 256 and k["static_lds"] == 0
    # the three store flavours are the same code but for the cache-policy modifier (profiles/r03_study/README.md 2)
    for other in ("encrypt_kernel<1,true,0,true,0>", "encrypt_kernel<1,true,0,true,2>"):
        o = table[other]
        assert (o["vgpr"], o["sgpr_spill"]) == (k["vgpr"], k["sgpr_spill"]), other
        assert abs(o["instructions"] - k["instructions"]) <= k["instructions"] // 100, other  # the key-slab flush pads its sc1 stores itself
