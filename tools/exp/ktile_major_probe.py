#!/usr/bin/env python3
"""developer probe: weights stored K-tile-major ([K / 64][N][64]: a workgroup's W tile of one K-tile is contiguous) against the nn.Linear layout on the
weight-bandwidth-bound small-M GEMMs; results must be bit-equal"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

for (M, N, K, what) in ((452, 12288, 4096, "T5 qkv"), (452, 4096, 4096, "T5 o"), (452, 4096, 10240, "T5 wo"), (452, 20480, 4096, "T5 wi"), (250, 4096, 1024, "CAMA ff1"),
                        (35552, 3072, 3072, "DiT to_out"), (35552, 9216, 3072, "DiT qkv")):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    wt = w.view(N, K // 64, 64).permute(1, 0, 2).contiguous().view(N, K)
    ops.TUNING["gemm"] = 0
    ref = ops.linear(x, w)
    for rep in range(2):
        for flag, wm in ((0, w), (8, wt)):
            ops.TUNING["gemm"] = flag
            assert torch.equal(ops.linear(x, wm), ref), (what, flag)
            dt = timeit(lambda: ops.linear(x, wm), iters=30, warm=5)
            print(f"{what:12s} M={M} N={N} K={K} layout={'k-tile-major' if flag else 'row-major   '}: {dt*1e6:.1f} us  weights {N*K*2/dt/1e12:.2f} TB/s  {2.0*M*N*K/dt/1e12:.0f} TF/s")
ops.TUNING["gemm"] = 0
