#!/usr/bin/env python3
"""(needs tools/exp/gemm_pingpong_and_w4_agpr_experiment.patch applied) Interleaved same-box A/B (plain GEMM + bias, the DiT's four shapes): ping-pong 8-wave loop (shipped), round-2 lockstep loop, the 4-wave AGPR-pinned experiment
(tuning cfg 3) and hipBLASLt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
M = 2 * 17776
g = torch.Generator().manual_seed(1)
for name, N, K in (("QKV", 9216, 3072), ("to_out", 3072, 3072), ("FF1", 12288, 3072), ("FF2", 3072, 12288)):
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)

    def tuned(t):
        def run():
            ops.TUNING["gemm"] = t
            try:
                return ops.linear(x, w, out=out)
            finally:
                ops.TUNING["gemm"] = 0
        return run
    cases = (("ping-pong 8 waves", tuned(0)), ("lockstep 8 waves (round 2)", tuned(ops.GEMM_TUNE_LOCKSTEP)), ("4 waves, AGPR accumulators", tuned(3 << 4)),
             ("hipBLASLt", lambda: torch.nn.functional.linear(x, w)))
    ref = tuned(ops.GEMM_TUNE_LOCKSTEP)().clone()
    for n, fn in cases[:3]:
        o = fn()
        print(f"  {n}: equal to the lockstep result: {torch.equal(o, ref)}  max |diff| {(o.float() - ref.float()).abs().max().item():.4g}")
    res = {n: [] for n, _ in cases}
    for rnd in range(int(os.environ.get("ROUNDS", "4"))):
        for n, fn in cases:
            res[n].append(timeit(fn, iters=8, warm=2))
    fl = 2.0 * M * N * K
    for n, ts in res.items():
        ts = sorted(ts)
        print(f"{name:7s} [{M} x {N} x {K}] {n:28s}: min {ts[0]*1e3:.3f} ms  median {ts[len(ts)//2]*1e3:.3f} ms -> {fl/ts[len(ts)//2]/1e12:.0f} TFLOP/s")
