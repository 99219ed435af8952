"""developer tool: the fan-out kernel with a forced query tile (tools/topk_variants.sh tk_tn{1,2,4,8} ... -DMRAG_TOPK_FORCE_TN=t) by table size"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch, microbench as mb
from motionrag_amd import ops
out = []
for Q in (256, 64):
    for N in (4000, 10000, 20000, 32768, 65536, 100000, 300000):
        db = torch.randn(N, 768, device="cuda"); q = torch.randn(Q, 768, device="cuda")
        dt = mb.timeit(lambda: ops.topk(db, q, 12, order="mfma"), iters=20, warm=3)
        out.append(f"{N}x{Q}:{dt*1e6:7.1f}")
print(" ".join(out))
'''
for n in sys.argv[1:]:
    env = dict(os.environ, MRAG_HIP_LIB=os.path.join(ROOT, "tools", f"lib_{n}.so"), MRAG_HIP_LIB_ANY_SOURCE="1")
    p = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True)
    print(f"{n:10s} {p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]}", flush=True)
