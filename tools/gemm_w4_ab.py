#!/usr/bin/env python3
"""Interleaved same-box A/B (cdna guide rule 24) of the 256x256 GEMM tile: the 8-wave loop, the hand-scheduled 4-wave loop (tuning cfg 3) and hipBLASLt,
on the DiT's four shapes and the UNets' short-K ones; the 4-wave result must be bit-equal to the 8-wave one (same K order per output)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(1)
SHAPES = [("QKV", 35552, 9216, 3072, "none"), ("to_out", 35552, 3072, 3072, "gate"), ("FF1", 35552, 12288, 3072, "gelu"), ("FF2", 35552, 3072, 12288, "gate")]
if os.environ.get("UNET"):
    SHAPES += [("L0 GEGLU", 258048, 2560, 320, "geglu"), ("L1 GEGLU", 64512, 5120, 640, "geglu"), ("L2 GEGLU", 16128, 10240, 1280, "geglu"), ("odd GEGLU", 100037, 2560, 320, "geglu"),
               ("L0 qkv", 258048, 960, 320, "none"), ("L1 to_out", 64512, 640, 640, "resid"), ("L2 ff2", 16128, 1280, 5120, "resid"), ("odd", 100037, 2564, 320, "resid"),
               ("odd gate", 70001, 1028, 192, "gate"), ("odd M gate", 35552 + 77, 3072, 3072, "gate"), ("odd M gelu", 70001, 1024, 640, "gelu"), ("odd M resid", 64512 + 130, 640, 640, "resid"), ("k64", 70000, 1024, 64, "gelu"), ("k128", 70000, 1024, 128, "none")]
S = 17776
for name, M, N, K, epi in SHAPES:
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV, torch.bfloat16)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16) if epi in ("gate", "resid") else None
    if epi == "geglu":
        wi, bi = ops.geglu_interleave(w, b)
        out2 = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
    nb = -(-M // min(S, M))
    g0, g1 = (torch.randn(nb, N, generator=g).to(DEV, torch.bfloat16) for _ in range(2))

    def tuned(t):
        def run():
            ops.TUNING["gemm"] = t
            try:
                if epi == "gate":
                    return ops.linear(x, w, b, out=out, epilogue=ops.EPI_GATE_RESID, resid=r, gate0=g0, gate1=g1, rows_per_batch=min(S, M), split=226, gate_stride=N)
                if epi == "resid":
                    return ops.linear(x, w, b, out=out, epilogue=ops.EPI_RESID, resid=r)
                if epi == "geglu":
                    return ops.linear(x, wi, bi, out=out2, epilogue=ops.EPI_GEGLU)
                if epi == "gelu":
                    return ops.linear(x, w, b, out=out, epilogue=ops.EPI_GELU_TANH)
                return ops.linear(x, w, b, out=out)
            finally:
                ops.TUNING["gemm"] = 0
        return run
    cases = (("8 waves", tuned(ops.GEMM_TUNE_NO_W4)), ("4 waves, persistent", tuned(3 << 4)), ("hipBLASLt", lambda: torch.nn.functional.linear(x, w, b)))
    ref = tuned(ops.GEMM_TUNE_NO_W4)().clone()
    o = tuned(3 << 4)()
    print(f"  4 waves equal to 8 waves: {torch.equal(o, ref)}  max |diff| {(o.float() - ref.float()).abs().max().item():.4g}", flush=True)
    res = {n: [] for n, _ in cases}
    for rnd in range(int(os.environ.get("ROUNDS", "4"))):
        for n, fn in cases:
            res[n].append(timeit(fn, iters=8, warm=2))
    fl = 2.0 * M * N * K
    for n, ts in res.items():
        ts = sorted(ts)
        print(f"{name:9s} [{M} x {N} x {K}] {epi:5s} {n:26s}: min {ts[0]*1e3:.3f} ms  median {ts[len(ts)//2]*1e3:.3f} ms -> {fl/ts[len(ts)//2]/1e12:.0f} TFLOP/s", flush=True)
