#!/usr/bin/env python3
"""Same-box cost of the fused epilogues on the FF1 / QKV shapes: plain, bias, bias + GELU (developer probe)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
M, K = 2 * 17776, 3072
for N in (12288, 9216):
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    for rep in range(2):
        for name, fn in (("plain", lambda: ops.linear(x, w, out=out)), ("bias", lambda: ops.linear(x, w, b, out=out)),
                         ("bias+gelu_tanh", lambda: ops.linear(x, w, b, out=out, epilogue=ops.EPI_GELU_TANH)),
                         ("bias+silu", lambda: ops.linear(x, w, b, out=out, epilogue=ops.EPI_SILU))):
            dt = timeit(fn, iters=20, warm=3)
            print(f"N={N} {name:16s}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.1f} TF/s")
