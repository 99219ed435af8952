"""developer tool: the few-row GEMM (gemm_skinny_kernel, M <= 256) against the 128x128 tile it replaces (tuning bit MRAG_GEMM_TUNE_NO_SKINNY), CAMA's / the query
embedder's shapes, with the error of both against an fp32 product.   python tools/skinny_sweep.py"""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch, microbench as mb
from motionrag_amd import ops
DEV = "cuda"
NO_SKINNY = 1 << 17
SHAPES = [(250, 768, 1024, "none"), (250, 2304, 1024, "none"), (250, 1024, 768, "resid"), (250, 4096, 1024, "gelu"), (250, 1024, 4096, "resid"),
          (251, 3072, 1024, "none"), (251, 1024, 1024, "resid"), (251, 4096, 1024, "gelu"), (251, 1024, 4096, "resid"),
          (25, 768, 1024, "none"), (25, 4096, 1024, "gelu"), (25, 1024, 4096, "resid"), (16, 2304, 768, "none"), (16, 3072, 768, "gelu"), (16, 768, 3072, "resid"),
          (64, 4096, 4096, "none"), (128, 4096, 4096, "none"), (256, 4096, 4096, "none"), (256, 10240, 4096, "none"), (256, 4096, 10240, "none")]
for M, N, K, kind in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV, torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    kw = {"none": {}, "gelu": dict(epilogue=ops.EPI_GELU_ERF), "resid": dict(epilogue=ops.EPI_RESID, resid=r)}[kind]
    ref = x.float() @ w.float().T + b.float()
    ref = torch.nn.functional.gelu(ref) if kind == "gelu" else ref + r.float() if kind == "resid" else ref
    out = []
    for name, t in (("few-row", 0), ("128x128", NO_SKINNY)) + ((("few-row 8 waves", 1 << 18),) if K >= 2048 else ()):
        ops.TUNING["gemm"] = t
        with ops.dispatched() as d:
            y = ops.linear(x, w, b, **kw)
        dt = mb.timeit(lambda: ops.linear(x, w, b, **kw), iters=50, warm=5)
        err = ((y.float() - ref).norm() / ref.norm()).item()
        out.append(f"{name} {dt*1e6:6.1f} us [{','.join(d.counts)}] rel {err:.1e}")
    ops.TUNING["gemm"] = 0
    print(f"[{M:4d} x {N:5d} x {K:5d}] {kind:5s} | " + " | ".join(out) + f" | weights {N*K*2/1e6:5.1f} MB", flush=True)
