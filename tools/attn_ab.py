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
    # the attn32 family and the attn16 workgroup-shape variants this script compared in rounds 2-3 are archived in tools/exp/ (ABI 8)
    variants = ((0, "attn16 QB3 NW4 x3/CU (shipped)"), (ops.ATTN_TUNE_LEGACY, "legacy 32x32x16 (round-1 algorithm)"))
    res = {t: [] for t, _ in variants}
    for rnd in range(int(os.environ.get("ROUNDS", "4"))):
        for tune, _ in variants:
            ops.TUNING["attn"] = tune
            ops.TUNING["attn_no_split"] = True      # all variants without the key-split tail (the QB4 forms have none)
            res[tune].append(timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out), iters=8, warm=2))
    ops.TUNING["attn_no_split"] = True
    ref = None
    for tune, label in variants:          # every variant computes the same attention: compare with the 32x32x16 kernel
        ops.TUNING["attn"] = tune
        o = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]).float()
        if tune == ops.ATTN_TUNE_LEGACY:
            ref = o
    for tune, label in variants:
        ops.TUNING["attn"] = tune
        o = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]).float()
        d = (o - ref).abs().max().item()
        print(f"  {label}: max |diff| vs legacy {d:.4g}" + ("  <-- WRONG" if not d < 0.01 else ""))
    ops.TUNING["attn"] = 0
    ops.TUNING["attn_no_split"] = False
    for tune, label in variants:
        ts = sorted(res[tune])
        print(f"{name} {label}: min {ts[0]*1e3:.3f} ms  median {ts[len(ts)//2]*1e3:.3f} ms  -> {fl/ts[0]/1e12:.0f} / {fl/ts[len(ts)//2]/1e12:.0f} TFLOP/s ({fl/ts[len(ts)//2]/2.5e15*100:.1f} % of 2.5 PF)")

# fp8 path (config #5) against the bf16 kernel at the DynamiCrafter level-0 shape, interleaved
B, H, S = 32, 5, 9216
qkv = torch.randn(B, S, 3, H, 64, device=DEV).to(torch.bfloat16)
out = torch.empty(B, S, H * 64, device=DEV, dtype=torch.bfloat16)
fl = 4.0 * B * H * S * S * 64
r = {False: [], True: []}
for rnd in range(3):
    for f8 in (False, True):
        r[f8].append(timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out, fp8=f8), iters=8, warm=2))
for f8 in (False, True):
    ts = sorted(r[f8])
    print(f"dc level0 {'fp8 e4m3 (amax + quantise + attention)' if f8 else 'bf16 attn16'}: min {ts[0]*1e3:.3f} ms  median {ts[1]*1e3:.3f} ms -> {fl/ts[1]/1e12:.0f} TFLOP/s")
