#!/usr/bin/env python3
"""Interleaved same-box A/B: the UNets' GEGLU projections (value * gelu(gate) epilogue) as persistent workgroups with next-tile prefetch (shipped) against one
tile per workgroup (MRAG_GEMM_TUNE_NO_PERSIST), at the three resolutions of the SVD / DynamiCrafter UNets (14 frames x 2 CFG samples)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(2)
for name, M, C in (("level 0", 28 * 9216, 320), ("level 1", 28 * 2304, 640), ("level 2", 28 * 576, 1280)):
    inner = 4 * C
    x = torch.randn(M, C, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(2 * inner, C, generator=g) * C ** -0.5).to(DEV, torch.bfloat16)
    b = torch.randn(2 * inner, generator=g).to(DEV, torch.bfloat16)
    wi, bi = ops.geglu_interleave(w, b)
    out = torch.empty(M, inner, device=DEV, dtype=torch.bfloat16)

    def run(t):
        def f():
            ops.TUNING["gemm"] = t
            try:
                return ops.linear(x, wi, bi, out=out, epilogue=ops.EPI_GEGLU)
            finally:
                ops.TUNING["gemm"] = 0
        return f
    a = run(0)().clone()
    bb = run(ops.GEMM_TUNE_NO_PERSIST)()
    print(f"  {name}: persistent == one-tile-per-workgroup: {torch.equal(a, bb)}")
    res = {"persistent + prefetch": [], "one tile per workgroup": []}
    for rnd in range(int(os.environ.get("ROUNDS", "5"))):
        res["persistent + prefetch"].append(timeit(run(0), iters=10, warm=2))
        res["one tile per workgroup"].append(timeit(run(ops.GEMM_TUNE_NO_PERSIST), iters=10, warm=2))
    fl = 2.0 * M * 2 * inner * C
    for n, ts in res.items():
        ts = sorted(ts)
        print(f"GEGLU {name} [{M} x {2*inner} x {C}] {n:24s}: min {ts[0]*1e3:.3f} ms  median {ts[len(ts)//2]*1e3:.3f} ms -> {fl/ts[len(ts)//2]/1e12:.0f} TFLOP/s")
