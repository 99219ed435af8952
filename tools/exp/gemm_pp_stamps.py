#!/usr/bin/env python3
"""(needs tools/exp/gemm_pingpong_and_w4_agpr_experiment.patch applied) developer tool (diagnostic build, tools/build_diag.sh): where does a 32-deep k-step of the ping-pong GEMM main loop spend its cycles?"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd._lib import GemmArgs  # noqa: E402

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libmrag_diag.so"))
L.mrag_gemm_bf16.argtypes = [ctypes.c_void_p, ctypes.POINTER(GemmArgs)]
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(35552, 9216, 3072), (35552, 3072, 12288)]
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    buf = torch.zeros(1024 * 8 * 8 + 4 * 8 * 4 * 8, dtype=torch.int64, device="cuda")
    assert L.mrag_debug_set_gemm_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    a = GemmArgs()
    a.A, a.W, a.C, a.M, a.N, a.K, a.lda, a.ldw, a.ldc = x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, K, N
    for _ in range(5):
        assert L.mrag_gemm_bf16(None, ctypes.byref(a)) == 0
    torch.cuda.synchronize()
    st = buf[:65536].view(1024, 8, 8).cpu().double()
    ns = st[..., 5].clamp(min=1)
    print(f"M={M} N={N} K={K}: cycles per 32-deep k-step, mean over 1024 workgroups; group 0 = waves 0-3, group 1 = waves 4-7")
    names = ["LOAD: issue 12 reads + 4 DMA pieces, fragments landed", "      s_waitcnt vmcnt (stage s + 1 landed)", "      barrier at the end of LOAD", "MFMA: issue 32 MFMAs", "      barrier at the end of MFMA"]
    for i, n in enumerate(names):
        v = st[..., i] / ns
        print(f"   {n:58s} group0 {v[:, :4].mean().item():7.0f}   group1 {v[:, 4:].mean().item():7.0f}")
    tot = (st[..., :5].sum(-1) / ns)
    print(f"   {'sum per k-step (matrix floor: 2 x 32 x 16 = 1024)':58s} group0 {tot[:, :4].mean().item():7.0f}   group1 {tot[:, 4:].mean().item():7.0f}")
