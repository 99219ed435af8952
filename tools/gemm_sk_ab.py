#!/usr/bin/env python3
"""Interleaved A/B on ONE device (cdna guide rule 24): the DiT's four linears with the opt-in stream-K tail (MRAG_GEMM_TUNE_STREAMK) against the plain tile grid
(the shipped default) and against hipBLASLt (torch.nn.functional.linear: the vendor ceiling, plain GEMM only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
M = 2 * 17776
SHAPES = (("to_out (gate*x+resid)", 3072, 3072, "gate"), ("FF1 + GELU", 12288, 3072, "gelu"), ("FF2 (gate*x+resid)", 3072, 12288, "gate"), ("QKV plain", 9216, 3072, "none"))
g = torch.Generator().manual_seed(1)
for name, N, K, epi in SHAPES:
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV, torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    kw = {}
    if epi == "gate":
        r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
        g0, g1 = (torch.randn(2, N, generator=g).to(DEV, torch.bfloat16) for _ in range(2))
        kw = dict(epilogue=ops.EPI_GATE_RESID, resid=r, gate0=g0, gate1=g1, rows_per_batch=17776, split=226, gate_stride=N)
    elif epi == "gelu":
        kw = dict(epilogue=ops.EPI_GELU_TANH)
    fl = 2.0 * M * N * K
    res = {"stream-K": [], "plain grid": [], "hipBLASLt (no epilogue)": []}
    for rnd in range(int(os.environ.get("ROUNDS", "4"))):
        for label in res:
            if label.startswith("hip"):
                t = timeit(lambda: torch.nn.functional.linear(x, w, b), iters=8, warm=2)
            else:
                ops.TUNING["gemm"] = ops.GEMM_TUNE_STREAMK if label == "stream-K" else 0
                t = timeit(lambda: ops.linear(x, w, b, out=out, **kw), iters=8, warm=2)
                ops.TUNING["gemm"] = 0
            res[label].append(t)
    for label, ts in res.items():
        ts = sorted(ts)
        print(f"{name} [{M} x {N} x {K}] {label}: min {ts[0]*1e3:.3f} ms  median {ts[len(ts)//2]*1e3:.3f} ms -> {fl/ts[len(ts)//2]/1e12:.0f} TFLOP/s")
