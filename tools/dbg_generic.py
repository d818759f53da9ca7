import sys, numpy as np, torch
sys.path.insert(0,"."); sys.path.insert(0,"tests")
import __graft_entry__ as ge
ge.build(); pkg=ge.load_package()
import oracle_lib as ol
rng=np.random.default_rng(0xA35128); n=64
pt=rng.integers(0,256,(n,16),dtype=np.uint8); keys=rng.integers(0,256,(n,16),dtype=np.uint8)
sb,m2,m3=pkg.reference_tables()
def run(tables,name,pbk):
    c=pkg.Context(0,tables=tables); o=ol.Oracle(tables=tables)
    k=keys if pbk else keys[0].copy()
    got=c.encrypt_witness(torch.from_numpy(pt).cuda(),torch.from_numpy(k).cuda(),layout=0); torch.cuda.synchronize()
    exp=o.encrypt_witness(pt,k,layout=0)
    for col in "xyz":
        a=getattr(got,col).cpu().numpy(); e=getattr(exp,col); d=np.nonzero(a!=e)[0]
        print(name,"pbk",pbk,"xt",c.uses_xtime_path,col,len(d),d[:12], a[d[:6]], e[d[:6]])
    c.close()
for pbk in (0,1):
    run((sb,m2,m3),"ref",pbk)
    t2=m2.copy(); t2[0]=1
    run((sb,t2,m3),"m2[0]=1",pbk)
    t3=m3.copy(); t3[0]=1
    run((sb,m2,t3),"m3[0]=1",pbk)
    r=np.random.default_rng(5)
    run((r.permutation(256).astype(np.uint8),m2,m3),"rand sbox",pbk)
    run((sb,r.integers(0,256,256,dtype=np.uint8),m3),"rand m2",pbk)
    run((sb,m2,r.integers(0,256,256,dtype=np.uint8)),"rand m3",pbk)
