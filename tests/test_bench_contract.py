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
    # the headline is BASELINE configs[2]: 2^20 blocks, per-block keys + key-schedule witness, packed columns
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.workload == "c2" and a.layout == "packed"
    assert a.replays >= 5 and a.log2_blocks is None
    assert b.BYTES_SHARED == 3024 + 16 and b.BYTES_PBK == 3024 + 936 + 32
    assert b.HBM_PEAK_GBPS == 8000.0


def test_cpu_baseline_leg(pkg):
    b = _bench()
    for pbk in (True, False):
        r = b.cpu_baseline(pbk, n_target_seconds=0.05)
        assert r["kind"] == "port" and r["cores"] == 1 and r["unit"] == "blocks/s" and r["value"] > 0
        assert ("per-block keys" in r["sample"]) == pbk and r["all_cores"]["cores"] >= 1


def test_watchdog_exits_non_zero(tmp_path):
    """A stalled phase of the N>1 tail must reach the driver as a failure: the watchdog prints the line with the phase
    name on rank 0 and exits with status 3 (never 0)."""
    import subprocess
    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('bench_mod', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "d = b.Watchdog(0, {'metric': 'x'}); d.arm('gather', 0.2); time.sleep(5); sys.exit(0)\n" % (str(ROOT), str(ROOT / "bench.py")))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, timeout=60)
    assert out.returncode == 3
    assert "phase 'gather' timed out" in out.stdout


def test_watchdog_of_an_optional_phase_records_the_stall_and_exits_zero():
    """The C ABI's own RCCL gather runs LAST in the N > 1 tail and has never run with real ranks: if it stalls, the line -- which
    already holds the scaling measurement -- is printed with gather_cabi = {error, fatal: false} and the job ends with status 0."""
    import json
    import subprocess
    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('bench_mod', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "d = b.Watchdog(0, {'metric': 'x', 'value': 1.0}); d.arm('gather_cabi', 0.2, optional_key='gather_cabi'); time.sleep(5); sys.exit(9)\n" % (str(ROOT), str(ROOT / "bench.py")))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, timeout=60)
    assert out.returncode == 0
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and "error" not in line and line["gather_cabi"] == {"error": "phase 'gather_cabi' timed out", "fatal": False}


def test_main_never_rebinds_its_long_lived_names():
    """bench.py's main() is one long function; a loop variable that shadows the argparse namespace (`a`), the line, the context
    ... breaks code hundreds of lines further down and loses the headline (it happened once, caught on the GPU box).  No for /
    comprehension / with / except target and no later assignment in main() may rebind these names."""
    import ast
    tree = ast.parse((ROOT / "bench.py").read_text())
    main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    protected = {"a", "line", "ctx", "pkg", "rank", "world", "use_arena", "layout", "pbk", "extras"}
    first_assign = {}
    bad = []
    for node in ast.walk(main):
        targets = []
        if isinstance(node, (ast.For, ast.comprehension)):
            targets = [node.target]
        elif isinstance(node, ast.With):
            targets = [i.optional_vars for i in node.items if i.optional_vars is not None]
        elif isinstance(node, ast.ExceptHandler) and node.name in protected:
            bad.append((node.name, node.lineno))
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                for n in ast.walk(t):
                    if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store) and n.id in protected:
                        first_assign.setdefault(n.id, []).append(n.lineno)
        for t in targets:
            for n in ast.walk(t):
                if isinstance(n, ast.Name) and n.id in protected:
                    bad.append((n.id, n.lineno))
    assert not bad, bad
    for name, lines in first_assign.items():
        assert len(lines) == 1 or name in ("ctx",), (name, lines)  # assigned exactly once
