#!/usr/bin/env python3
"""256x256 GEMM tile on four waves (MRAG_GEMM_CFG=3, EPI_NONE only) against the shipped 8-wave kernel: equality + timing (developer probe)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
for name, N, K in (("qkv", 9216, 3072), ("to_out", 3072, 3072), ("ff1", 12288, 3072), ("ff2", 3072, 12288)):
    M = 2 * 17776
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    outs = {}
    for rep in range(2):
        for cfg in ("0", "3"):
            os.environ["MRAG_GEMM_CFG"] = cfg
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            dt = timeit(lambda: ops.linear(x, w, out=out), iters=20, warm=3)
            outs[cfg] = out
            print(f"{name:7s} cfg={cfg}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.1f} TF/s")
    print("   equal:", torch.equal(outs["0"], outs["3"]), (outs["0"].float() - outs["3"].float()).abs().max().item())
