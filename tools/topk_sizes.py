"""developer tool: fan-out (order mfma) vs scan (order chain16) kernel by table size, 256 and 64 queries -- where `order = auto` should switch"""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch, microbench as mb
from motionrag_amd import ops
for Q in (256, 64, 16):
    for N in (1000, 2000, 4000, 10000, 20000, 32768, 100000):
        db = torch.randn(N, 768, device="cuda"); q = torch.randn(Q, 768, device="cuda")
        a = mb.timeit(lambda: ops.topk(db, q, 12, order="mfma"), iters=20, warm=3)
        b = mb.timeit(lambda: ops.topk(db, q, 12, order="chain16"), iters=20, warm=3)
        print(f"N={N:7d} Q={Q:4d}: mfma {a*1e6:8.1f} us   chain16 {b*1e6:8.1f} us", flush=True)
