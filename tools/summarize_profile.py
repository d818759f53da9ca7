#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile.sh) into the committed summary:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.json and an entry in
profiles/traffic.json (HBM bytes per launch from the PMC passes, corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950, WRITE_SIZE as is; both in KiB)."""
import csv
import glob
import json
import shutil
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
key = sys.argv[2] if len(sys.argv) > 2 else "c1_packed"
src = ROOT / "gpurun_out" / ("prof_" + tag)
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def one(pattern):
    files = glob.glob(str(src / pattern), recursive=True)
    # a tag profiled twice leaves both runs' files (named by pid) in the directory: take the latest
    return max(files, key=lambda f: Path(f).stat().st_mtime) if files else None


summary = {"tag": tag, "workload": key}
ks = one("trace/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, dst / ("%s_%s_kernel_stats.csv" % (tag, key)))
    for r in csv.DictReader(open(ks)):
        if "encrypt_kernel" in r["Name"]:
            summary["kernel"] = r["Name"]
            summary["calls"] = int(r["Calls"])
            summary["avg_ns"] = float(r["AverageNs"])
            summary["min_ns"] = float(r["MinNs"])
            summary["max_ns"] = float(r["MaxNs"])
kt = one("trace/**/*kernel_trace.csv")
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if "encrypt_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-100:]]
    summary["timed_region_us"] = {"n": len(d), "mean": statistics.mean(d), "median": statistics.median(d), "min": min(d), "max": max(d)}
    summary["lds_block_bytes"] = rows[-1].get("LDS_Block_Size")
    # the kernel trace's VGPR_Count column is the ARCHITECTURAL VGPR allocation of the dispatch (104 for the headline
    # kernel since round 1); the code object's unified register count (.vgpr_count: VGPRs + AGPRs, what bounds residency)
    # is tracked in profiles/isa_resources.json by tests/test_isa_lint.py and copied in below
    summary["vgpr_count_column_of_kernel_trace"] = rows[-1].get("VGPR_Count")
    try:
        import re
        isa = json.loads((ROOT / "profiles" / "isa_resources.json").read_text())
        short = re.sub(r"\(.*\)$", "", summary.get("kernel", "").replace("void ", "").replace("aesw::", "")).replace(", ", ",")
        summary["code_object"] = isa.get(short)
    except Exception:
        pass
    summary["sgpr"] = rows[-1].get("SGPR_Count")


def counter(pass_dir):
    f = one(pass_dir + "/**/*counter_collection.csv")
    out = {}
    if not f:
        return out
    acc = {}
    for r in csv.DictReader(open(f)):
        if "encrypt_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = statistics.mean(v)
    return out


w, f, sq = counter("pmc_write"), counter("pmc_fetch"), counter("pmc_sq")
if "WRITE_SIZE" in w and "FETCH_SIZE" in f:
    wr = w["WRITE_SIZE"] * 1024
    rd = f["FETCH_SIZE"] * 1024 * 2   # gfx950: FETCH_SIZE reports half of a coalesced stream
    summary["pmc"] = {"WRITE_SIZE_KiB": w["WRITE_SIZE"], "FETCH_SIZE_KiB": f["FETCH_SIZE"],
                      "hbm_write_bytes": wr, "hbm_read_bytes_corrected": rd, "traffic_bytes_per_launch": wr + rd}
    tj = dst / "traffic.json"
    t = json.loads(tj.read_text()) if tj.exists() else {}
    t[key] = wr + rd
    tj.write_text(json.dumps(t, indent=1) + "\n")
if sq:
    summary["sq"] = sq
bl = src / "bench_line.json"
if bl.exists() and bl.read_text().strip():
    summary["bench_line_under_rocprof"] = json.loads(bl.read_text())
(dst / ("%s_%s_summary.json" % (tag, key))).write_text(json.dumps(summary, indent=1) + "\n")
print(json.dumps(summary, indent=1))
