#!/usr/bin/env python3
"""Launch timeline of encrypt_kernel: a private -DAESW_TRACE build records, per wave, the 100 MHz wall
clock at start / tables ready / first flush (before, after) / round 5 / last flush issued / all stores
acknowledged, and the XCC + hardware id.  usage: trace.py [LOG2N] [name=value options ...]"""
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
pkg = ge.load_package()
b = ge._load_build()
lib_path = ROOT / "tools" / "libaesw_trace.so"
csrc, host = ROOT / "halo2-aes_amd" / "csrc", ROOT / "halo2-aes_amd" / "host"
srcs = [csrc / "aesw_kernels.hip", csrc / "aesw_api.cpp", csrc / "aesw_arena.cpp", host / "host_capi.cpp"]
if not b._newer(lib_path, srcs + [csrc / "aesw_lane.h", csrc / "aesw_layout.h", csrc / "aesw_internal.h"]):
    b._run([b.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DAESW_TRACE",
            "-o", str(lib_path)] + [str(s) for s in srcs])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
pkg.api._lib = pkg.api.load_library(lib_path)

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << log2n
ctx = pkg.Context(0)
waves = 4
for opt in sys.argv[2:]:
    k, v = opt.split("=")
    ctx.set_option(k, int(v))
    if k == "waves_shared":
        waves = int(v)
nw = ((n + 15) // 16 + 3) // 4 * 4
trace = torch.zeros((nw, 8), dtype=torch.int64, device="cuda")
g = torch.Generator(device="cpu").manual_seed(7)
pt = torch.randint(0, 256, (n, 16), dtype=torch.uint8, generator=g).cuda()
key = torch.randint(0, 256, (16,), dtype=torch.uint8, generator=g).cuda()
ctx.schedule_key(key, layout=pkg.LAYOUT_PACKED, key_slab=False)
sets = [ctx.alloc_witness(n, pkg.LAYOUT_PACKED, want_ct=False) for _ in range(4)]
for i in range(12):  # back to back, like the bench; the last launch is the one traced
    ctx.set_option("trace_ptr", trace.data_ptr() if i == 11 else 0)
    ctx.encrypt_witness(pt, None, layout=pkg.LAYOUT_PACKED, out=sets[i % 4])
torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64)
t0 = t[t[:, 0] != 0][:, 0].min()
t = t[t[:, 2] != 0]
us = (t[:, :7] - t0) / 100.0
us[t[:, 0] == 0, 0:2] = np.nan  # start / tables exist only for a wave's first unit
names = ["start", "tables", "flush1 begin", "flush1 issued", "round5", "last flush issued", "stores acked"]
print("units traced: %d (waves/group %d); kernel span %.2f us" % (len(t), waves, (t[:, 6].max() - t[:, 2].min()) / 100.0 + 1.2))
for i, nm in enumerate(names):
    c = us[:, i][~np.isnan(us[:, i])]
    print("%-18s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f" %
          (nm, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
life = us[:, 6] - us[:, 2]
print("wave lifetime      min %7.2f  median %7.2f  max %7.2f" % (life.min(), np.median(life), life.max()))
# generations: units whose start is later than the earliest finish
first_end = us[:, 6].min()
print("units begun before the first one ended: %d; first end %.2f us" % ((us[:, 2] < first_end).sum(), first_end))
# completion profile: bytes acknowledged over time (each wave = 16 blocks x 3024 B at its ack time, coarse)
edges = np.arange(0, us[:, 6].max() + 2, 2.0)
hist_s, _ = np.histogram(us[:, 2], bins=edges)
hist_e, _ = np.histogram(us[:, 6], bins=edges)
print("t(us)  starts  ends")
for e, s_, e_ in zip(edges[:-1], hist_s, hist_e):
    print("%5.0f  %6d  %5d" % (e, s_, e_))
xcc = (t[:, 7] >> 32) & 0xf
print("units per XCC:", np.bincount(xcc.astype(np.int64), minlength=8).tolist())
print("last ack per XCC (us):", [round(float(us[xcc == x, 6].max()), 2) for x in range(8)])
tag = "_".join(o.replace("=", "") for o in sys.argv[2:])
hw = t[:, 7] & 0xffffffff
cu_key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 0xf)
print("distinct CUs used: %d; waves per used CU: %s" % (len(np.unique(cu_key)), np.bincount(np.unique(cu_key, return_counts=True)[1]).tolist()))
np.save(str(ROOT / "gpurun_out" / ("trace_%d%s.npy" % (log2n, "_" + tag if tag else ""))), t)
