"""Builds the in-tree native artefacts.

  halo2-aes_amd/libaesw.so          the product: HIP kernels + C ABI (hipcc, gfx950)
  oracle/libaesw_oracle.so          test infrastructure: CPU oracle (gcc)
  tests/lane_model/liblane_model.so test infrastructure: CPU run of aesw_lane.h (g++)

hipcc cross-compiles gfx950 code objects without a GPU.  The .so files are
git-ignored but travel to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libaesw.so"
ORACLE_LIB = ROOT / "oracle" / "libaesw_oracle.so"
LANE_LIB = ROOT / "tests" / "lane_model" / "liblane_model.so"


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def _run(cmd, cwd=None):
    proc = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(map(str, cmd)), proc.stdout))
    return proc.stdout


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def build_product(force: bool = False) -> Path:
    srcs = [CSRC / "aesw_kernels.hip", CSRC / "aesw_api.cpp"]
    deps = srcs + [CSRC / "aesw_lane.h", CSRC / "aesw_layout.h", CSRC / "aesw_internal.h", ROOT / "include" / "aesw.h"]
    if not force and _newer(LIB, deps):
        return LIB
    tmp = LIB.with_suffix(".so.tmp")
    _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
          "-o", str(tmp)] + [str(s) for s in srcs])
    os.replace(tmp, LIB)
    return LIB


def build_oracle(force: bool = False) -> Path:
    d = ROOT / "oracle"
    if not force and _newer(ORACLE_LIB, [d / "aesw_oracle.c", d / "aesw_oracle.h"]):
        return ORACLE_LIB
    _run(["make", "-C", str(d), "-B", "libaesw_oracle.so"])
    return ORACLE_LIB


def build_lane_model(force: bool = False) -> Path:
    src = ROOT / "tests" / "lane_model" / "lane_model.cpp"
    deps = [src, CSRC / "aesw_lane.h", CSRC / "aesw_layout.h"]
    if not force and _newer(LANE_LIB, deps):
        return LANE_LIB
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", str(LANE_LIB), str(src)])
    return LANE_LIB


def build_all(force: bool = False):
    return build_product(force), build_oracle(force), build_lane_model(force)
