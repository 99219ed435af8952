#!/usr/bin/env python3
"""developer probe: MRAG_GEMM_TUNE_WARM_NEXT (each workgroup touches the first K-tile lines of the workgroup that follows it on its XCD) on the UNets' short-K
GEMMs and on the DiT shapes"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

for (M, N, K) in ((258048, 960, 320), (258048, 2560, 320), (64512, 1920, 640), (35552, 9216, 3072), (35552, 3072, 12288)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    ref = None
    for rep in range(2):
        for flag in (0, 8):
            ops.TUNING["gemm"] = flag
            y = ops.linear(x, w, b)
            ref = y if ref is None else ref
            assert torch.equal(y, ref)
            dt = timeit(lambda: ops.linear(x, w, b), iters=20, warm=3)
            print(f"M={M} N={N} K={K} warm_next={flag >> 3}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TF/s")
ops.TUNING["gemm"] = 0
