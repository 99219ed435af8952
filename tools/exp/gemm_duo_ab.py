#!/usr/bin/env python3
"""Interleaved same-box A/B of the UNets' linears (K = 320 ... 1280): the shipped tile choice against the 256x128 four-wave tile that puts TWO workgroups on a CU
(tuning cfg 3 of tools/exp/gemm_duo_two_per_cu_experiment.patch -- apply it first; without it cfg 3 falls through to the shipped choice), with a correctness check of the new tile against the shipped one (same accumulation order per output: bit-equal expected for the same epilogue)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(1)
# (name, M, N, K, epilogue): SVD step at 576x1024, 14 frames, CFG batch 2 -> rows per level 258 048 / 64 512 / 16 128
CASES = (("L0 GEGLU", 258048, 2560, 320, "geglu"), ("L0 FF2", 258048, 320, 1280, "resid"), ("L0 to_out", 258048, 320, 320, "resid"), ("L0 qkv", 258048, 960, 320, "none"),
         ("L1 GEGLU", 64512, 5120, 640, "geglu"), ("L1 FF2", 64512, 640, 2560, "resid"), ("L1 to_out", 64512, 640, 640, "resid"), ("L1 qkv", 64512, 1920, 640, "none"),
         ("L2 GEGLU", 16128, 10240, 1280, "geglu"), ("L2 FF2", 16128, 1280, 5120, "resid"), ("L2 to_out", 16128, 1280, 1280, "resid"),
         ("odd rows", 100000 + 37, 2560, 320, "geglu"), ("odd rows resid", 100000 + 37, 1280, 320, "resid"))
for name, M, N, K, epi in CASES:
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV, torch.bfloat16)
    if epi == "geglu":
        wi, bi = ops.geglu_interleave(w, b)
        out = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
        fn = lambda: ops.linear(x, wi, bi, out=out, epilogue=ops.EPI_GEGLU)
    elif epi == "resid":
        r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        fn = lambda: ops.linear(x, w, b, out=out, epilogue=ops.EPI_RESID, resid=r)
    else:
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        fn = lambda: ops.linear(x, w, b, out=out)

    def run(cfg, extra=0):
        old = ops.TUNING["gemm"]
        ops.TUNING["gemm"] = (old & ~0xf0) | (cfg << 4) | extra
        try:
            fn()
        finally:
            ops.TUNING["gemm"] = old
    run(0); ref = out.clone()
    run(3); new = out.clone()
    torch.cuda.synchronize()
    diff = (ref.float() - new.float()).abs().max().item()
    res = {0: [], 3: []}
    for rnd in range(int(os.environ.get("ROUNDS", "4"))):
        for cfg in (0, 3):
            res[cfg].append(timeit(lambda: run(cfg), iters=10, warm=2))
    if os.environ.get("STAGGER"):
        for mode in (1, 2, 3):
            for n in (1, 2, 4):
                ex = (mode << 16) | (n << 18)
                t = sorted(timeit(lambda: run(3, ex), iters=10, warm=2) for _ in range(3))[1]
                print(f"    stagger mode {mode} x{n}: {t*1e6:8.1f} us")
    fl = 2.0 * M * N * K
    t0, t3 = sorted(res[0])[len(res[0]) // 2], sorted(res[3])[len(res[3]) // 2]
    print(f"{name:15s} [{M} x {N} x {K}] {epi:6s}: shipped {t0*1e6:8.1f} us ({fl/t0/1e12:4.0f} TF)   256x128 duo {t3*1e6:8.1f} us ({fl/t3/1e12:4.0f} TF)   {100*(t0/t3-1):+5.1f} %   max|diff| {diff:.4g}", flush=True)
