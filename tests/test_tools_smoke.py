"""tools/ is measurement code outside the product and outside the other tests: this keeps it from rotting.
CPU only: every tools/*.py parses, and what it takes from the package, from bench.py and from the option table still exists;
every tools/*.hip and tools/*.c (and the plain-C examples) passes a syntax-only compile against the current headers."""
import ast
import importlib.util
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
TOOLS = ROOT / "tools"


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_python_tools_parse_and_use_names_that_exist(pkg):
    bench = _bench()
    api_src = (ROOT / "halo2-aes_amd" / "csrc" / "aesw_api.cpp").read_text()
    options = set(re.findall(r'std::strcmp\(name, "([a-z0-9_]+)"\)', api_src))
    assert {"store_mode", "key_slots", "arena_cache", "split_small"} <= options
    ctx_methods = set(dir(pkg.Context))
    scripts = sorted(TOOLS.glob("*.py")) + sorted((ROOT / "examples").glob("*.py"))
    assert len(scripts) >= 20
    problems = []
    for path in scripts:
        tree = ast.parse(path.read_text(), filename=str(path))  # SyntaxError fails the test with the file name
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name):
                if node.value.id == "bench" and not hasattr(bench, node.attr):
                    problems.append("%s:%d bench.%s does not exist" % (path.name, node.lineno, node.attr))
                if node.value.id == "pkg" and not hasattr(pkg, node.attr):
                    problems.append("%s:%d pkg.%s does not exist" % (path.name, node.lineno, node.attr))
                if node.value.id in ("ctx", "c") and node.attr.startswith(("alloc_", "encrypt_", "schedule_", "assemble_", "key_schedule", "expand_", "free_")) \
                        and node.attr not in ctx_methods:
                    problems.append("%s:%d Context.%s does not exist" % (path.name, node.lineno, node.attr))
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in ("set_option", "get_option") \
                    and node.args and isinstance(node.args[0], ast.Constant) and isinstance(node.args[0].value, str):
                if node.args[0].value not in options:
                    problems.append("%s:%d option %r is not in aesw_set_option / aesw_get_option" % (path.name, node.lineno, node.args[0].value))
    assert not problems, "\n".join(problems)


def _syntax_only(path: Path):
    inc = ["-I", str(ROOT / "include"), "-I", str(ROOT / "halo2-aes_amd" / "csrc"), "-I", "/opt/rocm/include"]
    if path.suffix == ".hip":
        cmd = ["hipcc", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-Wno-unused-command-line-argument"] + inc + [str(path)]
    else:
        cmd = ["gcc", "-std=c11", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-D_GNU_SOURCE"] + inc + [str(path)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    return path.name, out.returncode, out.stdout[-2000:]


def test_native_tools_and_examples_compile_against_the_current_headers():
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not found")
    srcs = sorted(TOOLS.glob("*.hip")) + sorted(TOOLS.glob("*.c")) + sorted((ROOT / "examples").glob("*.c")) + \
        sorted((ROOT / "tests" / "mock_rccl").glob("*.c"))
    assert len(srcs) >= 15
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(_syntax_only, srcs))
    bad = ["%s (rc %d):\n%s" % r for r in results if r[1] != 0]
    assert not bad, "\n\n".join(bad)
