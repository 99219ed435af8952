#!/usr/bin/env python3
"""Interleaved same-box table (cdna guide rule 24): the DiT's four linears as this repo's plain GEMM, with the fused epilogue the DiT runs, and as hipBLASLt's plain
GEMM (torch.nn.functional.linear, bias only) -- what the epilogues cost, and where the plain kernel stands against the vendor's"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
M, S = 2 * 17776, 17776
g = torch.Generator().manual_seed(1)
for name, N, K, epi in (("QKV", 9216, 3072, "qkrope"), ("to_out", 3072, 3072, "gate"), ("FF1", 12288, 3072, "gelu"), ("FF2", 3072, 12288, "gate")):
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV, torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    fused = None
    if epi == "gate":
        r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
        g0, g1 = (torch.randn(2, N, generator=g).to(DEV, torch.bfloat16) for _ in range(2))
        fused = lambda: ops.linear(x, w, b, out=out, epilogue=ops.EPI_GATE_RESID, resid=r, gate0=g0, gate1=g1, rows_per_batch=S, split=226, gate_stride=N)
    elif epi == "gelu":
        fused = lambda: ops.linear(x, w, b, out=out, epilogue=ops.EPI_GELU_TANH)
    else:
        qg, qb, kg, kb = (torch.randn(64, generator=g).to(DEV, torch.bfloat16) for _ in range(4))
        ang = torch.rand(S - 226, 64, generator=g) * 6.28
        cos, sin = torch.cos(ang).to(DEV), torch.sin(ang).to(DEV)
        x3 = x.view(2, S, K)
        fused = lambda: ops.qkv_linear_qknorm_rope(x3, w, b, 48, qg, qb, kg, kb, cos, sin, 226, q_premul=0.18, out=out.view(2, S, N))
    cases = (("plain (bias)", lambda: ops.linear(x, w, b, out=out)), (f"fused {epi}", fused), ("hipBLASLt plain (bias)", lambda: torch.nn.functional.linear(x, w, b)))
    res = {n: [] for n, _ in cases}
    for rnd in range(int(os.environ.get("ROUNDS", "4"))):
        for n, fn in cases:
            res[n].append(timeit(fn, iters=8, warm=2))
    fl = 2.0 * M * N * K
    for n, ts in res.items():
        ts = sorted(ts)
        print(f"{name:7s} [{M} x {N} x {K}] {n:30s}: min {ts[0]*1e3:.3f} ms  median {ts[len(ts)//2]*1e3:.3f} ms -> {fl/ts[len(ts)//2]/1e12:.0f} TFLOP/s")
