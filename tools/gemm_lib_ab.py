#!/usr/bin/env python3
"""developer tool: interleaved timing of mrag_gemm_bf16 from several library builds (tools/build_variant.sh) on one box:
    python3 tools/gemm_lib_ab.py MxNxK epilogue lib_a.so lib_b.so ...      (epilogue: none | gelu | resid | gate)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd._lib import GemmArgs  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1].split("x"))
EPI = sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
libs = {n: ctypes.CDLL(os.path.join(here, n)) for n in sys.argv[3:]}
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
g0, g1 = (torch.randn(4, N, device="cuda").to(torch.bfloat16) for _ in range(2))
a = GemmArgs()
a.A, a.W, a.C, a.bias, a.M, a.N, a.K, a.lda, a.ldw, a.ldc = x.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr(), M, N, K, K, K, N
if EPI in ("gate", "resid"):
    a.resid, a.ldr, a.epilogue = r.data_ptr(), N, 4 if EPI == "gate" else 3
    if EPI == "gate":
        a.gate0, a.gate1, a.rows_per_batch, a.split, a.gate_stride = g0.data_ptr(), g1.data_ptr(), 17776, 226, N
elif EPI == "gelu":
    a.epilogue = 1
res = {n: [] for n in libs}
outs = {}
for rnd in range(5):
    for n, L in libs.items():
        L.mrag_gemm_bf16.argtypes = [ctypes.c_void_p, ctypes.POINTER(GemmArgs)]
        for _ in range(2):
            assert L.mrag_gemm_bf16(None, ctypes.byref(a)) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.mrag_gemm_bf16(None, ctypes.byref(a))
        e1.record(); torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 10)
        outs[n] = out.clone()
ref = next(iter(outs.values()))
for n, ts in res.items():
    ts = sorted(ts)
    print(f"{sys.argv[1]} {EPI:5s} {n:18s}: min {ts[0]:.4f} ms  median {ts[len(ts)//2]:.4f} ms   equal to the first library: {torch.equal(outs[n], ref)}")
