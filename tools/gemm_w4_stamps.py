#!/usr/bin/env python3
"""developer tool (diagnostic build, tools/build_diag.sh): where a workgroup of the persistent four-wave GEMM spends its cycles per tile"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd._lib import GemmArgs  # noqa: E402

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("MRAG_DIAG_LIB", "libmrag_diag.so")))
L.mrag_gemm_bf16.argtypes = [ctypes.c_void_p, ctypes.POINTER(GemmArgs)]
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(35552, 9216, 3072), (35552, 3072, 12288), (258048, 960, 320)]
EPI = os.environ.get("EPI", "none")
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    buf = torch.zeros(1024 * 8 * 8 + 4 * 8 * 4 * 8, dtype=torch.int64, device="cuda")
    assert L.mrag_debug_set_gemm_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    a = GemmArgs()
    a.A, a.W, a.C, a.bias, a.M, a.N, a.K, a.lda, a.ldw, a.ldc = x.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr(), M, N, K, K, K, N
    a.tuning = int(os.environ.get("CFG", "3")) << 4
    if EPI in ("gate", "resid"):
        r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        g0, g1 = (torch.randn(4, N, device="cuda").to(torch.bfloat16) for _ in range(2))
        a.resid, a.ldr = r.data_ptr(), N
        a.epilogue = 4 if EPI == "gate" else 3
        if EPI == "gate":
            a.gate0, a.gate1, a.rows_per_batch, a.split, a.gate_stride = g0.data_ptr(), g1.data_ptr(), 17776, 226, N
    elif EPI == "gelu":
        a.epilogue = 1
    for _ in range(3):
        assert L.mrag_gemm_bf16(None, ctypes.byref(a)) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        assert L.mrag_gemm_bf16(None, ctypes.byref(a)) == 0
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    st = buf[:256 * 64].view(256, 8, 8)[:, :4].cpu().double()
    nt = st[..., 4].clamp(min=1)
    names = ["zero accumulators + loop top", f"K loop ({K // 64} K-tiles)", "vmcnt(0) + drain in front of the epilogue", "epilogue (reads, math, stores)"]
    print(f"M={M} N={N} K={K} epilogue {EPI}: {ms:.3f} ms, tiles per workgroup {nt.mean().item():.2f}")
    for i, n in enumerate(names):
        print(f"   {n:44s} {(st[..., i] / nt).mean().item():9.0f} cycles per tile")
    tot = st[..., 5].mean().item()
    print(f"   whole workgroup {tot:9.0f} cycles = {tot / (ms * 1e3):.0f} cycles/us (s_memtime clock 100 MHz x?)  sum of phases {(st[..., :4].sum(-1)).mean().item():9.0f}")
