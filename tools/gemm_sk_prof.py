#!/usr/bin/env python3
"""rocprofv3 target: the DiT's to_out and FF2 GEMMs with the stream-K tail (main launch + tail launch) and as the plain grid -- per-kernel durations"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
M = 2 * 17776
g = torch.Generator().manual_seed(1)
for N, K in ((3072, 3072), (3072, 12288)):
    x = torch.randn(M, K, generator=g).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for tune in (ops.GEMM_TUNE_STREAMK, 0):
        ops.TUNING["gemm"] = tune
        for _ in range(10):
            ops.linear(x, w, out=out)
        torch.cuda.synchronize()
ops.TUNING["gemm"] = 0
