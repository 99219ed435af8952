#!/usr/bin/env python3
"""Interleaved A/B of the two joint-attention kernel families on ONE device (cdna guide rule 24): attn16.hip (16x16x32, lazy max) against
attn_flash.hip (32x32x16) at the BASELINE shape B=2, H=48, S=17 776 and at the DynamiCrafter level-0 shape B=32, H=5, S=9 216."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
for name, (B, H, S) in (("dit joint", (2, 48, 17776)), ("dc level0", (32, 5, 9216))):
    qkv = torch.randn(B, S, 3, H, 64, device=DEV).to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, device=DEV, dtype=torch.bfloat16)
    fl = 4.0 * B * H * S * S * 64
    res = {0: [], ops.ATTN_TUNE_LEGACY: []}
    for rnd in range(int(os.environ.get("ROUNDS", "4"))):
        for tune in (0, ops.ATTN_TUNE_LEGACY):
            ops.TUNING["attn"] = tune
            res[tune].append(timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out), iters=8, warm=2))
    ops.TUNING["attn"] = 0
    for tune, label in ((0, "attn16 16x16x32"), (ops.ATTN_TUNE_LEGACY, "legacy 32x32x16")):
        ts = sorted(res[tune])
        print(f"{name} {label}: min {ts[0]*1e3:.3f} ms  median {ts[len(ts)//2]*1e3:.3f} ms  -> {fl/ts[0]/1e12:.0f} / {fl/ts[len(ts)//2]/1e12:.0f} TFLOP/s ({fl/ts[len(ts)//2]/2.5e15*100:.1f} % of 2.5 PF)")
