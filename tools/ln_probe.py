#!/usr/bin/env python3
"""LayerNorm throughput at the UNets' row widths (developer probe)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

for rows, D in ((28 * 9216, 320), (28 * 2304, 640), (28 * 576, 1280), (2 * 17776, 3072), (32 * 9216, 320)):
    x = torch.randn(rows, D, device="cuda").to(torch.bfloat16)
    w = torch.ones(D, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    dt = timeit(lambda: ops.layernorm(x, w, w, 1e-5, out=y), iters=20, warm=3)
    print(f"layernorm [{rows},{D}]: {dt*1e6:.1f} us  {2*x.numel()*2/dt/1e9:.0f} GB/s")
