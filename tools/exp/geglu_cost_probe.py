#!/usr/bin/env python3
"""developer probe: what the erf-GELU gate costs in the UNets' level-0 GEGLU projection ([258 048, 320] -> 2 x 1280): erf gate vs the cheaper tanh gate vs a plain
GEMM of the same operand shapes (which writes twice the bytes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

M = 28 * 9216
for K, N, m in ((320, 2560, M), (640, 5120, M // 4), (1280, 10240, M // 16)):
    x = torch.randn(m, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    wg, bg = ops.geglu_interleave(w, b)
    for rep in range(2):
        for name, fn in (("geglu erf", lambda: ops.linear(x, wg, bg, epilogue=ops.EPI_GEGLU)), ("geglu tanh", lambda: ops.linear(x, wg, bg, epilogue=ops.EPI_GEGLU, geglu_tanh=True)),
                         ("plain + bias", lambda: ops.linear(x, w, b))):
            dt = timeit(fn, iters=20, warm=3)
            print(f"K={K} N={N} M={m} {name:14s}: {dt*1e3:.3f} ms  {2.0*m*N*K/dt/1e12:.0f} TF/s")
