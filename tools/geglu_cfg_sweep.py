"""developer tool: the UNets' feed-forward projections under the GEMM's developer tile knobs (`tuning` bits 4-7 / NO_W4): which tile configuration a short-K,
epilogue-heavy GEMM wants.   python tools/geglu_cfg_sweep.py"""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch, microbench as mb
from motionrag_amd import ops
DEV = "cuda"
CFGS = (("shipped", 0), ("256x256 16 waves", 1 << 4), ("128x128 4 waves x2", 2 << 4), ("256x256 8 waves", 1 << 16))
SHAPES = (("geglu", 258048, 2560, 320), ("geglu", 64512, 5120, 640), ("geglu", 16128, 10240, 1280), ("plain", 258048, 960, 320), ("resid", 258048, 320, 1280),
          ("resid", 64512, 640, 2560), ("plain", 64512, 1920, 640))
for kind, M, N, K in SHAPES:
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=DEV).to(torch.bfloat16)
    r = torch.randn(M, N, device=DEV).to(torch.bfloat16) if kind == "resid" else None
    line = [f"{kind:6s} [{M} x {N} x {K}]"]
    ref = None
    for name, t in CFGS:
        ops.TUNING["gemm"] = t
        try:
            fn = {"geglu": lambda: ops.linear(x, w, b, epilogue=ops.EPI_GEGLU), "plain": lambda: ops.linear(x, w, b),
                  "resid": lambda: ops.linear(x, w, b, epilogue=ops.EPI_RESID, resid=r)}[kind]
            with ops.dispatched() as d:
                y = fn()
            dt = mb.timeit(fn, iters=10, warm=2)
            same = "" if ref is None else (" =" if torch.equal(y, ref) else " !=")
            ref = y if ref is None else ref
            line.append(f"{name}: {dt*1e6:7.1f} us {2.0*M*N*K/dt/1e12:5.0f} TF [{','.join(d.counts)}]{same}")
        except Exception as e:  # noqa: BLE001
            line.append(f"{name}: {type(e).__name__}")
    ops.TUNING["gemm"] = 0
    print(" | ".join(line), flush=True)
