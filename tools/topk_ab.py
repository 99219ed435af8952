"""developer tool: fan-out top-k kernel variants interleaved on one box (tools/topk_variants.sh builds them):  python tools/topk_ab.py name1 name2 ...  [--rounds 3]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch, microbench as mb
from motionrag_amd import ops
out = []
for N, Q in ((1000000, 256), (1000000, 64), (100000, 256), (10000, 256)):
    db = torch.randn(N, 768, device="cuda"); q = torch.randn(Q, 768, device="cuda")
    dt = mb.timeit(lambda: ops.topk(db, q, 12, order="mfma"), iters=8, warm=2)
    out.append(f"{N}x{Q}: {dt*1e6:8.1f} us {2.0*N*Q*768/dt/1e12:6.1f} TF")
out.append(f"fp32 MFMA probe {mb.mfma_f32_sustained():.1f} TF")
print(" | ".join(out))
'''
names = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = 3
for r in range(rounds):
    for n in names:
        env = dict(os.environ)
        if n != "product":
            env.update(MRAG_HIP_LIB=os.path.join(ROOT, "tools", f"lib_{n}.so"), MRAG_HIP_LIB_ANY_SOURCE="1")
        p = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True)
        print(f"{n:24s} {p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]}", flush=True)
