"""Timing bounds, kept OUT of the parity suite: `pytest -m perf` on a GPU box.  These tests carry the
`perf` marker only -- `-m gpu` does not select them, and without a GPU they skip -- so a noisy lease can never turn the
byte-exactness run red or hide later tests behind `-x`.  Byte comparisons live in tests/test_gpu_*.py."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.perf
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("timing bounds need a GPU")
    return torch


def _launch_us(torch, c, pkg, dpt, dkeys, w, reps=7, per=10):
    """Microseconds per launch: `per` launches between two events, median of `reps` such measurements (a single launch's event
    time wanders by 3 % on one and the same memory)."""
    for _ in range(3):
        c.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w, key_slab=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per):
            c.encrypt_witness(dpt, dkeys, layout=pkg.LAYOUT_PACKED, out=w, key_slab=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / per)
    return sorted(ts)[len(ts) // 2]


def test_a_launch_costs_about_what_the_arena_probe_measured(gpu, pkg):
    """(was tests/test_gpu_round3.py:176) a 2^20-block launch into a probed arena takes less than twice the probe's store-pattern
    emulation over the same columns."""
    torch = gpu
    c = pkg.Context(0)
    n = 1 << 20
    w = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=True)
    info = c.last_arena
    dpt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda")
    dkeys = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda")
    us = _launch_us(torch, c, pkg, dpt, dkeys, w)
    assert us < 2.0 * info["probe_us"], (us, info)
    c.close()


def test_an_arena_from_the_placement_cache_is_immediate_and_as_fast_as_the_first(gpu, pkg):
    """alloc -> free -> alloc of the same shape returns in < 20 ms without building a candidate, and a
    launch into it runs within 3 % of a launch into the arena the search placed (it IS that memory)."""
    import time
    torch = gpu
    c = pkg.Context(0)
    n = 1 << 20
    dpt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda")
    dkeys = torch.randint(0, 256, (n, 16), dtype=torch.uint8, device="cuda")
    w = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=True)
    first = dict(c.last_arena)
    us_first = _launch_us(torch, c, pkg, dpt, dkeys, w)
    c.free_columns(w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w2 = c.alloc_columns(n, pkg.LAYOUT_PACKED, key_slab=True)
    dt = time.perf_counter() - t0
    again = dict(c.last_arena)
    assert again["candidates"] == 0 and dt < 0.020, (dt, again)
    us_again = _launch_us(torch, c, pkg, dpt, dkeys, w2)
    assert abs(us_again - us_first) < 0.03 * us_first, (us_first, us_again)
    assert us_again < 1.10 * first["probe_us"] + 5.0, (us_again, first)
    c.close()


def test_three_batch_streams_do_not_cost_more_than_one(gpu, pkg, tmp_path):
    """(was tests/test_gpu_round3.py:463) examples/aesw_batches.c: the batches entry point on three internal streams against one."""
    exe = tmp_path / "aesw_batches"
    lib_dir = ROOT / "halo2-aes_amd"
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
                    str(ROOT / "examples" / "aesw_batches.c"), "-o", str(exe), "-L", str(lib_dir), "-laesw", "-L", "/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + str(lib_dir), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe), "15", "12"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    ratio = float(re.search(r"three streams / one stream = ([0-9.]+)", out.stdout).group(1))
    assert ratio < 1.10, out.stdout
