#!/usr/bin/env python3
"""developer tool: where does a K-tile of the 256x256 GEMM main loop spend its cycles? (diagnostic build, tools/build_diag.sh)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd._lib import GemmArgs  # noqa: E402

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("MRAG_DIAG_LIB", "libmrag_diag.so")))
L.mrag_gemm_bf16.argtypes = [ctypes.c_void_p, ctypes.POINTER(GemmArgs)]
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(35552, 9216, 3072), (35552, 3072, 12288)]      # e.g. 258048x960x320
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    buf = torch.zeros(1024 * 8 * 8 + 4 * 8 * 4 * 8, dtype=torch.int64, device="cuda")
    assert L.mrag_debug_set_gemm_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    a = GemmArgs()
    a.A, a.W, a.C, a.M, a.N, a.K, a.lda, a.ldw, a.ldc = x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, K, N
    for _ in range(3):
        assert L.mrag_gemm_bf16(None, ctypes.byref(a)) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        assert L.mrag_gemm_bf16(None, ctypes.byref(a)) == 0
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    st = buf[:65536].view(1024, 8, 8).cpu().double()
    tr = buf[65536:].view(4, 8, 4, 8).cpu()
    nk = st[..., 4].clamp(min=1)
    print(f"M={M} N={N} K={K}")
    for i, n in enumerate(["wait for k-step 0 fragments", "32 MFMAs (k-step 0) + wait k-step 1 fragments", "vmcnt(0) + barrier + issue next reads", "32 MFMAs (k-step 1) + DMA issue"]):
        print(f"   {n:44s} {(st[..., i] / nk).mean().item():8.0f} cycles/K-tile")
    print(f"   {'   of which vmcnt(0) (own DMA pieces)':44s} {(st[..., 5] / nk).mean().item():8.0f}   max over the WG's waves {(st[..., 5] / nk).max(dim=1).values.mean().item():8.0f}")
    pro, epi = st[..., 6].mean().item(), st[..., 7].mean().item()
    loop = sum((st[..., i]).mean().item() for i in range(4))
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"   prologue {pro:8.0f}  loop {loop:8.0f}  epilogue {epi:8.0f} cycles per workgroup; kernel {ms:.3f} ms; {nwg} WGs / 256 CUs -> implied s_memtime clock "
          f"{(pro + loop + epi) * nwg / 256 / (ms * 1e-3) / 1e9:.2f} GHz")
    if os.environ.get("MRAG_TRACE"):
        for blk in range(2):
            t0 = int(tr[blk, :, 0, 0].min())
            print(f"  block {blk}: per wave [simd] then per K-tile: start | frag0 ready | ks0 done | vmcnt done | barrier passed+reads issued | ks1 done")
            for w in range(8):
                simd = (int(tr[blk, w, 0, 6]) >> 4) & 3
                line = f"   w{w} simd{simd}: "
                for it in range(4):
                    line += " ".join(f"{int(tr[blk, w, it, k]) - t0:6d}" for k in range(6)) + "  ||  "
                print(line)
