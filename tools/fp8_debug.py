#!/usr/bin/env python3
"""developer check of mrag_attn_fwd_fp8's workspace contents (amax, Q8, K8, V8 layouts) against a torch restatement"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionrag_amd import ops, _lib
DEV = "cuda"
B, H, Sq, Skv = 1, 2, 512, 512
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B, S, H, 64, generator=g).to(torch.bfloat16) for S in (Sq, Skv, Skv))
q = q * 1.5; v = v * 2
out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), fp8=True)
torch.cuda.synchronize()
ws = ops._attn_workspace(torch.device("cuda", 0), 1, "fp8").cpu()
al = lambda x: (x + 255) & ~255
bh = B * H
amax = ws[:bh * 16].view(torch.float32).view(bh, 4)
print("amax gpu", amax[:, :3].tolist())
print("amax ref", [[q[0, :, h].abs().max().item(), k[0, :, h].abs().max().item(), v[0, :, h].abs().max().item()] for h in range(H)])
o = al(bh * 16)
q8 = ws[o:o + bh * Sq * 64].view(bh, Sq, 64); o += al(bh * Sq * 64)
k8 = ws[o:o + bh * Skv * 64].view(bh, Skv, 64); o += al(bh * Skv * 64)
v8 = ws[o:o + bh * Skv * 64].view(bh, Skv // 64, 64, 64)
def fit(a):
    e = math.floor(math.log2(448.0 / a))
    if a * 2.0 ** e > 448: e -= 1
    if a * 2.0 ** (e + 1) <= 448: e += 1
    return e
c = 0.125 * 1.4426950408889634
for h in range(H):
    qq, kk, vv = q[0, :, h].float(), k[0, :, h].float(), v[0, :, h].float()
    y, ek, ev = fit(qq.abs().max().item() * c), fit(kk.abs().max().item()), fit(vv.abs().max().item())
    q_ref = (qq * (c * 2.0 ** y)).to(torch.float8_e4m3fn).view(torch.uint8)
    print(h, "y ek ev", y, ek, ev, "| Q8 equal:", (q8[h] == q_ref).float().mean().item(), "nan bytes", ((q8[h] & 0x7f) == 0x7f).sum().item())
    k_ref = (kk * 2.0 ** ek).to(torch.float8_e4m3fn).view(torch.uint8)          # [Skv, 64]
    rows = torch.arange(Skv)
    sw = ((rows % 64) >> 2) & 3
    k_sw = torch.empty_like(k_ref)
    for ch in range(4):
        for r in range(Skv):
            pass
    # de-swizzle the GPU image: chunk position p holds logical chunk p ^ sw
    kg = k8[h].view(Skv, 4, 16)
    kd = torch.stack([kg[torch.arange(Skv), (ch ^ sw)] for ch in range(4)], dim=1).reshape(Skv, 64)
    print(h, "K8 equal:", (kd == k_ref).float().mean().item(), "nan bytes", ((k8[h] & 0x7f) == 0x7f).sum().item())
    v_ref = (vv * 2.0 ** ev).to(torch.float8_e4m3fn).view(torch.uint8)          # [Skv, 64 d]
    ok = tot = 0
    for tile in range(Skv // 64):
        img = v8[h, tile]                                                         # [64 d][64 pos]
        for key in range(64):
            kk_ = key & 31
            slot = ((kk_ >> 2) & 1) * 32 + (key >> 5) * 16 + (kk_ & 3) + 4 * (kk_ >> 3)
            d = torch.arange(64)
            pos = (((slot >> 4) ^ ((d >> 2) & 3)) << 4) | (slot & 15)
            ok += (img[d, pos] == v_ref[tile * 64 + key]).sum().item(); tot += 64
    print(h, "V8 equal:", ok / tot, "nan bytes", ((v8[h] & 0x7f) == 0x7f).sum().item())
of = out.float().cpu()
print("out nan fraction per head", [torch.isnan(of[..., 64 * h:64 * h + 64]).float().mean().item() for h in range(H)])
s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * 0.125
want = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float()).reshape(B, Sq, H * 64)
fin = torch.isfinite(of)
print("rel err on finite", ((of - want)[fin].norm() / want[fin].norm()).item() if fin.any() else None)
print("nan rows (first head)", torch.isnan(of[0, :, 0]).nonzero().flatten()[:20].tolist())
