"""Builds the product libraries in-tree.

  halo2-aes_amd/libaesw.so        HIP kernels + the C ABI of include/aesw.h (hipcc --offload-arch=gfx950)
  halo2-aes_amd/libaesw_host.so   the C++ mirror of the reference's host interface (include/aesw_host.h): plain g++,
                                  no device code, linked against libaesw.so -- it only calls the C ABI

hipcc cross-compiles gfx950 code objects without a GPU.  The .so is git-ignored
but travels to the GPU box with the snapshot.  (The test-only artefacts are
built by __graft_entry__.build(), not from inside the product package.)
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
HOST_LIB = PKG / "libaesw_host.so"


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


def build_host(force: bool = False) -> Path:
    """libaesw_host.so: host code only (g++), NEEDED libaesw.so found next to it ($ORIGIN)."""
    host = PKG / "host"
    srcs = [host / "host_capi.cpp"]
    deps = srcs + [ROOT / "include" / "aesw.h", ROOT / "include" / "aesw_host.h", host / "halo2_lite.hpp", host / "aes_gadget.hpp", LIB]
    if not force and _newer(HOST_LIB, deps):
        return HOST_LIB
    import fcntl
    with open(PKG / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _newer(HOST_LIB, deps):
                return HOST_LIB
            tmp = HOST_LIB.with_suffix(".so.tmp%d" % os.getpid())
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", str(tmp)] + [str(s) for s in srcs] +
                 ["-L" + str(PKG), "-laesw", "-Wl,-rpath,$ORIGIN"])
            os.replace(tmp, HOST_LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return HOST_LIB


def build_product(force: bool = False) -> Path:
    srcs = [CSRC / "aesw_kernels.hip", CSRC / "aesw_api.cpp", CSRC / "aesw_arena.cpp", CSRC / "aesw_comm.cpp"]
    deps = srcs + [CSRC / "aesw_lane.h", CSRC / "aesw_layout.h", CSRC / "aesw_check.h", CSRC / "aesw_internal.h", CSRC / "aesw_ctx.h", ROOT / "include" / "aesw.h"]
    if not force and _newer(LIB, deps):
        return LIB
    # several ranks may get here at once (torchrun): serialise on a lock file, build under a
    # private name, publish with an atomic rename
    import fcntl
    with open(PKG / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _newer(LIB, deps):
                return LIB  # another process built it while we waited
            tmp = LIB.with_suffix(".so.tmp%d" % os.getpid())
            _run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread",
                  "-o", str(tmp)] + [str(s) for s in srcs])
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB
