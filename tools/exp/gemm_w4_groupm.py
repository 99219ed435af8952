import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionrag_amd import ops
from microbench import timeit
DEV="cuda"
g=torch.Generator().manual_seed(1)
for name,M,N,K in (("QKV",35552,9216,3072),("to_out",35552,3072,3072),("FF1",35552,12288,3072),("FF2",35552,3072,12288)):
    x=torch.randn(M,K,generator=g).to(DEV,torch.bfloat16); w=(torch.randn(N,K,generator=g)*K**-0.5).to(DEV,torch.bfloat16); b=torch.randn(N,generator=g).to(DEV,torch.bfloat16)
    out=torch.empty(M,N,device=DEV,dtype=torch.bfloat16)
    res={}
    for rnd in range(3):
        for gm in (1,2,4,8,16,32):
            def run():
                ops.TUNING["gemm"]=gm<<8
                try: ops.linear(x,w,b,out=out)
                finally: ops.TUNING["gemm"]=0
            res.setdefault(gm,[]).append(timeit(run,iters=8,warm=2))
    print(name, {gm: round(sorted(v)[1]*1e3,3) for gm,v in res.items()}, flush=True)
