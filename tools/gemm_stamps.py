#!/usr/bin/env python3
"""developer tool: where does a K-tile of the 256x256 GEMM main loop spend its cycles? (diagnostic build, tools/build_diag.sh)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd._lib import GemmArgs  # noqa: E402

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmrag_diag.so"))
L.mrag_gemm_bf16.argtypes = [ctypes.c_void_p, ctypes.POINTER(GemmArgs)]
for (M, N, K) in ((35552, 9216, 3072), (35552, 3072, 12288)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    buf = torch.zeros(1024 * 8 * 8, dtype=torch.int64, device="cuda")
    assert L.mrag_debug_set_gemm_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    a = GemmArgs()
    a.A, a.W, a.C, a.M, a.N, a.K, a.lda, a.ldw, a.ldc = x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, K, N
    for _ in range(3):
        assert L.mrag_gemm_bf16(None, ctypes.byref(a)) == 0
    torch.cuda.synchronize()
    st = buf.view(1024, 8, 8).cpu().double()
    nk = st[..., 4].clamp(min=1)
    print(f"M={M} N={N} K={K}")
    for i, n in enumerate(["vmcnt(0) wait (DMA)", "barrier", "first fragments (LDS latency + DMA issue)", "MFMA body (64 MFMAs)"]):
        print(f"   {n:44s} {(st[..., i] / nk).mean().item():8.0f} cycles/K-tile")
