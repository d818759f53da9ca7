import os, sys, statistics
os.environ.setdefault('AESW_DIAGNOSTIC', '1')  # allows store_mode=3 (flush left out)
sys.path.insert(0,".")
import torch, __graft_entry__ as ge
ge.build(); pkg=ge.load_package()
import bench
ctx=pkg.Context(0)
for o in sys.argv[1:]:
    k,v=o.split('='); ctx.set_option(k,int(v))
for n in (64, 8192, 65536):
    r=bench.Runner(pkg,ctx,torch,n,False,pkg.LAYOUT_PACKED,False,5)
    ms=[r.run(400,5,True)[1]*1e3 for _ in range(5)]
    print(n, "blocks: %.2f us per launch"%statistics.median(ms))
