#!/usr/bin/env python3
"""developer tool: where does a tile of the long-sequence attention loop spend its cycles? (diagnostic build, tools/build_diag.sh)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd._lib import AttnArgs  # noqa: E402

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmrag_diag.so"))
L.mrag_attn_fwd_bf16.argtypes = [ctypes.c_void_p, ctypes.POINTER(AttnArgs)]
B, H, S = 2, 48, 17776
qkv = torch.randn(B, S, 3, H, 64, device="cuda").to(torch.bfloat16)
out = torch.empty(B, S, H * 64, device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(2048 * 8 * 8, dtype=torch.int64, device="cuda")
assert L.mrag_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
a = AttnArgs()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
a.q_sb, a.q_ss, a.q_sh = q.stride(0), q.stride(1), q.stride(2)
a.k_sb, a.k_ss, a.k_sh = a.q_sb, a.q_ss, a.q_sh
a.v_sb, a.v_ss, a.v_sh = a.q_sb, a.q_ss, a.q_sh
a.o_sb, a.o_ss = out.stride(0), out.stride(1)
a.B, a.H, a.Sq, a.Skv, a.kv_batch_div, a.scale, a.out_scale = B, H, S, S, 1, 0.125, 1.0
for _ in range(3):
    assert L.mrag_attn_fwd_bf16(None, ctypes.byref(a)) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); L.mrag_attn_fwd_bf16(None, ctypes.byref(a)); e1.record(); torch.cuda.synchronize()
print(f"diag kernel: {e0.elapsed_time(e1):.3f} ms (stamps cost ~10%)")
st = buf.view(2048, 8, 8).cpu().double()
nt = st[..., 6].clamp(min=1)
names = ["barrier(early)", "Kread+QK issue", "max (MFMA result + chain)", "barrier(late)+issue", "exp/cvt + V read + PV issue", "whole tile"]
for grp, sl in (("early waves 0-3", slice(0, 4)), ("late waves 4-7", slice(4, 8))):
    print(grp)
    for i, n in enumerate(names):
        print(f"   {n:32s} {(st[:, sl, i] / nt[:, sl]).mean().item():8.0f} cycles/tile")
    raw = buf.view(2048, 8, 8)[:, sl, 7].cpu()
    print(f"   {'  of which vmcnt wait':32s} {((raw >> 32).double() / nt[:, sl]).mean().item():8.0f}")
    print(f"   {'  of which s_barrier wait':32s} {((raw & 0xffffffff).double() / nt[:, sl]).mean().item():8.0f}")
