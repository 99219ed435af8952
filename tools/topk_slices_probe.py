import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from motionrag_amd import ops
from microbench import timeit
for N in (100000, 1000000, 4000000):
    db = torch.randn(N, 768, device="cuda")
    q = torch.randn(1, 768, device="cuda")
    plan = ops.TopkPlan(db, 1, 12)
    plan.queries.copy_(q)
    ts = sorted(timeit(lambda: plan.run(), iters=20, warm=3) for _ in range(5))
    print(f"N={N} Q=1 plan: {ts[2]*1e6:.1f} us  {N*768*4/ts[2]/1e12:.2f} TB/s", flush=True)
    del db, plan
