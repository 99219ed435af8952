import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from motionrag_amd import ops
from microbench import timeit
B, H = 2, 48
res = {}
for S in (16896, 17664, 17776, 18432):
    qkv = torch.randn(B, S, 3, H, 64, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, device="cuda", dtype=torch.bfloat16)
    ts = sorted(timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out), iters=10, warm=3) for _ in range(5))
    fl = 4.0 * B * H * S * S * 64
    res[S] = ts[2]
    print(f"S={S}: {ts[2]*1e3:.3f} ms  {fl/ts[2]/1e12:.0f} TFLOP/s  q-tiles/bh {S/192:.2f}  rounds {96*(S//192)/768:.2f}", flush=True)
