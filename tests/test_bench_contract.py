"""bench.py pieces that need no GPU: argument defaults, the CPU-baseline leg and
the constants the roofline is computed from."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_and_constants(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.workload == "c1" and a.layout == "packed"
    assert b.BYTES_SHARED == 3024 + 16 and b.BYTES_PBK == 3024 + 936 + 32
    assert b.HBM_PEAK_GBPS == 8000.0


def test_cpu_baseline_leg(pkg):
    b = _bench()
    r = b.cpu_baseline(n_target_seconds=0.05)
    assert r["kind"] == "port" and r["cores"] == 1 and r["unit"] == "blocks/s" and r["value"] > 0
    assert "sample" in r and r["all_cores"]["cores"] >= 1
