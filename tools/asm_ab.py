#!/usr/bin/env python3
"""aesw_assemble_advice_device (K=20, N=5: 537 MB of Fr cells, 16.8 MB of bytes) of the in-tree library against another build
(tools/libaesw_r01.so) in one process."""
import ctypes as C, sys, statistics
sys.path.insert(0,'/root/repo')
import torch
import __graft_entry__ as ge
ge.build(); pkg=ge.load_package()
old=C.CDLL('/root/repo/tools/libaesw_r01.so')
for name in ("aesw_create","aesw_schedule_key_device","aesw_encrypt_witness_device","aesw_assemble_advice_device","aesw_set_option"):
    res,args=pkg.api.SYMBOLS[name]; getattr(old,name).restype,getattr(old,name).argtypes=res,args
new=pkg.load_library()
k,n_sets=20,5
nn=pkg.block_capacity(k,n_sets)
ctx=pkg.Context(0)
apt=torch.randint(0,256,(nn,16),dtype=torch.uint8,device='cuda'); akey=torch.randint(0,256,(16,),dtype=torch.uint8,device='cuda')
kw=ctx.schedule_key(akey,layout=pkg.LAYOUT_PACKED,key_slab=True)
wit=ctx.encrypt_witness(apt,None,layout=pkg.LAYOUT_PACKED)
out=torch.empty((16<<20)*32,dtype=torch.uint8,device='cuda')
t=[x.copy() for x in pkg.reference_tables()]
h2=C.c_void_p(); assert old.aesw_create(C.byref(h2),0,*[x.ctypes.data_as(C.c_void_p) for x in t])==0
ks=pkg.api.KeySlab(*[x.data_ptr() for x in kw[:4]])
torch.cuda.synchronize()
for as_fr in (1,0):
  for name,lib,h in (("new",new,ctx._h),("old",old,h2)):
    ts=[]
    for rep in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            rc=lib.aesw_assemble_advice_device(h,k,n_sets,nn,pkg.LAYOUT_PACKED,wit.x.data_ptr(),wit.y.data_ptr(),wit.z.data_ptr(),C.byref(ks),as_fr,out.data_ptr(),None)
            assert rc==0
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/4*1e3)
    med=statistics.median(ts); nbytes=(16<<20)*(32 if as_fr else 1)
    print(name,'as_fr',as_fr,'%.1f us %.0f GB/s'%(med,nbytes/med/1e3))
