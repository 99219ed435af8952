#!/usr/bin/env python3
"""developer probe: the software-pipelined 64x64-wave-tile loop against the compiler-scheduled one (MRAG_GEMM_TUNE_PLAIN_LOOP = 8) on the small-M GEMMs that run
128x128 tiles; results must be bit-equal"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

for (M, N, K, what) in ((452, 12288, 4096, "T5 qkv"), (452, 4096, 4096, "T5 o"), (452, 4096, 10240, "T5 wo"), (250, 1024, 1024, "CAMA"), (250, 4096, 1024, "CAMA ff1"),
                        (16, 2304, 768, "gte query qkv"), (2700, 512, 13824, "VAE L0 conv as GEMM"), (1568 * 2, 768, 768, "VideoMAE proj")):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    ref = None
    for rep in range(2):
        for flag in (8, 0):
            ops.TUNING["gemm"] = flag
            y = ops.linear(x, w)
            ref = y if ref is None else ref
            assert torch.equal(y, ref), (what, flag)
            dt = timeit(lambda: ops.linear(x, w), iters=50, warm=5)
            print(f"{what:22s} M={M} N={N} K={K} loop={'plain' if flag else 'pipelined'}: {dt*1e6:.1f} us  weights {N*K*2/dt/1e12:.2f} TB/s")
ops.TUNING["gemm"] = 0
