#!/usr/bin/env python3
"""developer tool: interleaved timing of mrag_attn_fwd_bf16 from several library builds (tools/build_variant.sh) in ONE process on one box:
    python3 tools/attn_lib_ab.py lib_a.so lib_b.so ...        (ROUNDS=5; shapes: the DiT joint attention and the DynamiCrafter level-0 one)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd._lib import AttnArgs  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
libs = {n: ctypes.CDLL(os.path.join(here, n)) for n in sys.argv[1:]}
for L in libs.values():
    L.mrag_attn_fwd_bf16.argtypes = [ctypes.c_void_p, ctypes.POINTER(AttnArgs)]
    L.mrag_attn_workspace_bytes.argtypes = [ctypes.c_int32] * 4
    L.mrag_attn_workspace_bytes.restype = ctypes.c_int64
for name, (B, H, S) in (("dit joint", (2, 48, 17776)), ("dit shipped 17f", (2, 48, 6976)), ("dc level0", (32, 5, 9216))):
    qkv = torch.randn(B, S, 3, H, 64, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, device="cuda", dtype=torch.bfloat16)
    a = AttnArgs()
    a.Q, a.K, a.V, a.O = qkv[:, :, 0].data_ptr(), qkv[:, :, 1].data_ptr(), qkv[:, :, 2].data_ptr(), out.data_ptr()
    for pre in ("q", "k", "v"):
        setattr(a, pre + "_sb", S * 3 * H * 64); setattr(a, pre + "_ss", 3 * H * 64); setattr(a, pre + "_sh", 64)
    a.o_sb, a.o_ss = S * H * 64, H * 64
    a.B, a.H, a.Sq, a.Skv, a.kv_batch_div, a.scale, a.out_scale = B, H, S, S, 1, 0.125, 1.0
    nb = max(L.mrag_attn_workspace_bytes(B, H, S, S) for L in libs.values())
    ws = torch.zeros(max(nb, 16), dtype=torch.uint8, device="cuda")
    a.workspace, a.workspace_bytes = ws.data_ptr(), nb
    fl = 4.0 * B * H * S * S * 64
    res, outs = {n: [] for n in libs}, {}
    for rnd in range(int(os.environ.get("ROUNDS", "5"))):
        for n, L in libs.items():
            for _ in range(2):
                assert L.mrag_attn_fwd_bf16(None, ctypes.byref(a)) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                L.mrag_attn_fwd_bf16(None, ctypes.byref(a))
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 8)
            outs[n] = out.float().clone()
    ref = next(iter(outs.values()))
    for n, ts in res.items():
        ts = sorted(ts)
        md = ts[len(ts) // 2]
        print(f"{name:16s} {n:28s}: min {ts[0]:.3f} ms  median {md:.3f} ms -> {fl / md / 1e9:.0f} TFLOP/s ({fl / md / 1e9 / 2500 * 100:.1f} % of 2.5 PF)   "
              f"max |diff| vs first library {float((outs[n] - ref).abs().max()):.3g}", flush=True)
