"""bench.py end to end on the GPU at small sizes: the N = 1 line with every extra, and the N > 1 tail (gather, configs[3],
configs[4] per rank) rehearsed with two ranks on the one GPU (gloo) -- the part of the bench the driver otherwise runs for the
first time on an 8-GPU node.  Three processes at most touch the card at once."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, (out.returncode, out.stdout[-2000:], out.stderr[-3000:])
    return json.loads(lines[-1])


def test_bench_single_gpu_line(pkg):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "1", "--log2-blocks", "16", "--c4-log2-blocks", "17",
                          "--no-cpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=str(ROOT))
    d = _line(out)
    assert d["metric"] == "AES-128 block witnesses/sec" and d["unit"] == "blocks/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["dtype"] == "u8" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["value"] > 1e8 and 0.05 < d["roofline"]["frac"] < 1.0 and d["roofline"]["bound"] == "hbm"
    assert d["roofline"]["algorithmic_bytes_per_block"] == 3992 and "configs[2]" in d["config"]["workload"]
    assert d["config"]["columns"].startswith("one probed arena") and len(d["config"]["arena_probe"]) == d["config"]["output_ring_sets"]
    for a in d["config"]["arena_probe"]:
        assert a["candidates"] >= 1 and a["probe_us"] > 0 and a["fill_us"] > 0
    # the line certifies itself: a sample of what the timed graph wrote, compared with the oracle (BASELINE.md "parity gate")
    g = d["parity_gate"]
    assert g["mismatches"] == 0 and g["blocks"] >= 3500 and g["sets_checked"] == d["config"]["output_ring_sets"] and "error" not in d
    assert g["columns"] == ["kx", "ky", "kz", "w", "x", "y", "z"]
    dc = g["device_check"]  # every block of every set against every constraint (the reference's MockProver criterion, on the device)
    assert dc["satisfied"] and dc["blocks"] == d["config"]["output_ring_sets"] << 16 and dc["keys"] == dc["blocks"] and dc["lookup_failures"] == 0
    d24 = d["extra"]["device_check_2p24"]
    assert "error" not in d24 and d24["satisfied"] and d24["blocks"] == 1 << 24 and d24["keys"] == 1 << 24
    assert d["config"]["arena_setup_s"] > 0
    hp = d["extra"]["headline_plain_tensors"]
    assert "error" not in hp and hp["parity_gate"]["mismatches"] == 0 and 0.05 < hp["frac"] < 1.0
    assert "NOT halo2" in d["extra"]["host_synthesize"]["what"]
    ex = d["extra"]
    for k in ("pcie_inclusive", "c4", "fr_columns_to_host", "c1_packed", "c1_packed_2p20", "c1_values", "c2_dense", "host_synthesize", "expand_fr",
              "key_schedule"):
        assert k in ex and "error" not in ex[k], (k, ex.get(k))
    for lay in ("packed", "values"):
        c4 = ex["c4"][lay]
        assert c4["mismatches"] == 0 and c4["sampled_chunks_verified_after_the_stream"] >= 1 and c4["blocks"] == 1 << 17
        sc = c4["stream_check"]  # packed: every block of the stream checked on the device on its way out; values-only: not checkable
        assert sc["satisfied"] and sc["blocks"] == ((1 << 17) if lay == "packed" else 0)


def test_bench_two_ranks_on_one_gpu_rehearses_the_multi_gpu_tail(pkg):
    env = dict(os.environ, AESW_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
                          "--log2-blocks", "15", "--c3-log2-blocks", "15", "--c4-rank-log2-blocks", "17"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=str(ROOT), env=env)
    d = _line(out)
    assert d["n_gpus"] == 2 and "error" not in d
    assert d["parity_gate"]["mismatches"] == 0 and d["parity_gate"]["ranks_with_mismatches"] == 0 and d["parity_gate"]["blocks_all_ranks"] >= 7000
    assert "error" not in d["gather"] and d["gather"]["path"].startswith("torch.distributed point-to-point")
    assert "error" not in d["c3"] and d["c3"]["blocks_total"] == 2 << 15
    c4 = d["c4"]
    assert "error" not in c4
    for lay in ("packed", "values"):
        assert c4[lay]["blocks_total"] == 2 << 17 and c4[lay]["mismatches"] == 0 and c4[lay]["blocks_per_s_to_hosts"] > 0
        assert c4[lay]["stream_check_blocks_all_ranks"] == ((2 << 17) if lay == "packed" else 0)  # every rank's shard certified on the device
        assert c4[lay]["seconds_max_over_ranks"] >= c4[lay]["rank0"]["seconds"] * 0.5
