"""Motion-injected DynamiCrafter 3-D UNet denoiser + DDIM sampler on hand-written gfx950 kernels.

Host-side mirror of the vendored LVDM code the reference runs for its DynamiCrafter backbone (class names, constructor
arguments and state-dict keys kept):

    reference (src/projects/dynamicrafter/DynamiCrafter/lvdm/...)              here
    modules/attention.py:37-223   CrossAttention.efficient_forward               CrossAttention
    modules/attention.py:226-266  BasicTransformerBlock, :448-475 GEGLU / FF      BasicTransformerBlock, FeedForward
    modules/attention.py:269-332  SpatialTransformer, :335-445 TemporalTransformer SpatialTransformer, TemporalTransformer
    modules/networks/openaimodel3d.py:52-107,110-237,240-281                      Downsample, Upsample, ResBlock, TemporalConvBlock
    modules/networks/openaimodel3d.py:284-635  UNetModel                          UNetModel
    models/samplers/ddim.py:24-57,135-298 + models/utils_diffusion.py             DDIMSampler (tables on the host, update kernel on the GPU)

Layout: activations are channels-last rows `[(b t), H, W, C]` bf16 for the whole network (the reference's NCHW <->
`b (h w) c` <-> `(b h w) t c` rearranges disappear: spatial tokens are contiguous rows, temporal tokens are a strided
view handed to the attention kernel).  3x3 convolutions and the (3,1,1) temporal convolutions run as a row gather
(`mrag_im2col3x3_bf16` / `mrag_unfold_t3_bf16`) + the MFMA GEMM with fused bias / residual; GroupNorm(+SiLU, + timestep
embedding pre-add) is one two-pass kernel pair.  nn.Module is a weight container only; no torch arithmetic in forward.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import ops


def _cat0(ts):
    return torch.cat([t.detach() for t in ts], dim=0).contiguous()


class _Cache:
    """weights re-laid-out once per weight tensor: conv kernels as [Cout, (ky, kx, cin)] GEMM operands, fused K|V projections, GEGLU row
    interleaves.  An entry is valid only for the SAME tensor object (weak reference: `id()` of a collected module can be reused), at the
    same storage address, dtype and in-place version (`load_state_dict` copies in place)."""

    def __init__(self):
        self.d = {}

    def get(self, key, ref, build):
        """`ref`: the tensor -- or a tuple of ALL tensors (None entries allowed) -- that `build` reads: replacing or updating any one of
        them in place invalidates the entry"""
        import weakref
        refs = tuple(r for r in (ref if isinstance(ref, (tuple, list)) else (ref,)) if r is not None)
        tag = tuple((r.data_ptr(), r.dtype, r._version) for r in refs)
        ent = self.d.get(key)
        if ent is None or ent[0] != tag or any(w() is not r for w, r in zip(ent[1], refs)):
            ent = (tag, tuple(weakref.ref(r) for r in refs), build())
            self.d[key] = ent
        return ent[2]


_CACHE = _Cache()


class TembBank:
    """SiLU(emb) [N, E] together with EVERY residual block's projection of it.  Each ResBlock applies its own `Linear(emb_channels, C_out)` to the
    same SiLU(emb) (openaimodel3d.py:222; diffusers `time_emb_proj`): 25-44 GEMMs of 28-32 rows per step, each a 20 us launch on 3-10 workgroups.
    They do not depend on the activations, so the bank runs them as ONE GEMM against the row-concatenated weights at the start of the step and
    hands out column slices `[N, C_out]` (row stride = the concatenated width).  Same kernel (128 x 128 tiles), same K order per output
    element as the separate launches: identical bits."""

    def __init__(self, silu_emb: torch.Tensor, owner: nn.Module, linears):
        self.silu = silu_emb
        refs = tuple(t for lin in linears for t in (lin.weight, lin.bias))
        w, b, offs = _CACHE.get(("tembbank", id(owner)), refs, lambda: (
            torch.cat([lin.weight.detach() for lin in linears], dim=0).contiguous(),
            torch.cat([(lin.bias.detach() if lin.bias is not None else torch.zeros(lin.out_features, dtype=lin.weight.dtype, device=lin.weight.device))
                       for lin in linears], dim=0).contiguous(),
            {id(lin): (o, lin.out_features) for lin, o in zip(linears, np.cumsum([0] + [lin.out_features for lin in linears[:-1]]).tolist())}))
        self._all = ops.linear(silu_emb, w, b) if linears else None
        self._offs = offs

    def proj(self, lin: nn.Linear) -> torch.Tensor:
        ent = self._offs.get(id(lin))
        if ent is None or ent[0] % 8:                      # a block outside the bank (or an unaligned slice): its own launch
            return ops.linear(self.silu, lin.weight, lin.bias)
        return self._all[:, ent[0]:ent[0] + ent[1]]


def temb_proj(silu_emb, lin: nn.Linear) -> torch.Tensor:
    """`lin(SiLU(emb))`: a slice of the step's TembBank, or the plain projection when a block is driven with a bare tensor (unit tests)"""
    return silu_emb.proj(lin) if isinstance(silu_emb, TembBank) else ops.linear(silu_emb, lin.weight, lin.bias)


def _tanh_scalar(p: torch.Tensor) -> float:
    """tanh of a learnable scalar gate (attention.py:200-202, 216-218) as a host float, read back ONCE per weight version: a `.item()` per
    attention call is a host sync per layer and cannot be captured in a HIP graph"""
    return _CACHE.get(("tanh", id(p)), p, lambda: float(torch.tanh(p.detach().float()).item()))


def conv3x3(x: torch.Tensor, conv: nn.Conv2d, *, stride: int = 1, upsample: bool = False, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N, H, W, Cin] -> [N, Ho, Wo, Cout]: implicit GEMM over gathered rows; `resid` (same shape as the output) fused in the epilogue."""
    N, H, W, C = x.shape
    cout = conv.weight.shape[0]
    kp = ops._kpad(9 * C)

    def build():
        w = conv.weight.detach().permute(0, 2, 3, 1).reshape(cout, 9 * C)          # [Cout, Cin, ky, kx] -> [Cout, (ky, kx, cin)]
        if kp != 9 * C:
            w = torch.cat([w, torch.zeros(cout, kp - 9 * C, dtype=w.dtype, device=w.device)], dim=1)
        return w.contiguous()

    wk = _CACHE.get(("c3", id(conv)), conv.weight, build)
    if C % 64 == 0:                                                                 # implicit GEMM: the GEMM's DMA gathers the taps itself
        return ops.conv_implicit(x.contiguous(), wk, conv.bias, ops.CONV_3X3, stride=stride, upsample=upsample,
                                 resid=resid.contiguous() if resid is not None else None)
    rows = ops.im2col3x3(x, stride=stride, upsample=upsample)
    Hi, Wi = (2 * H, 2 * W) if upsample else (H, W)
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    if resid is not None:
        y = ops.linear(rows, wk, conv.bias, epilogue=ops.EPI_RESID, resid=resid.reshape(-1, cout))
    else:
        y = ops.linear(rows, wk, conv.bias)
    return y.view(N, Ho, Wo, cout)


def conv_t3(x: torch.Tensor, conv: nn.Conv3d, B: int, T: int, *, resid: Optional[torch.Tensor] = None, acc_scale: float = 1.0) -> torch.Tensor:
    """nn.Conv3d((3,1,1), padding (1,0,0)) on x [(b t), HW, C]; with `resid`: resid + acc_scale * conv(x)"""
    C = x.shape[-1]
    cout = conv.weight.shape[0]
    wk = _CACHE.get(("t3", id(conv)), conv.weight, lambda: conv.weight.detach()[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, 3 * C).contiguous())
    if C % 64 == 0:
        return ops.conv_implicit(x.contiguous(), wk, conv.bias, ops.CONV_T3, frames=T, resid=resid.contiguous() if resid is not None else None, acc_scale=acc_scale)
    rows = ops.unfold_t3(x, B, T)
    if resid is not None:
        return ops.linear(rows, wk, conv.bias, epilogue=ops.EPI_RESID, resid=resid.reshape(-1, cout), acc_scale=acc_scale).view(x.shape[0], x.shape[1], cout)
    return ops.linear(rows, wk, conv.bias).view(x.shape[0], x.shape[1], cout)


def _lin_w(m) -> torch.Tensor:
    """nn.Linear or 1x1 Conv1d / Conv2d weight as [out, in]"""
    w = m.weight
    return w if w.dim() == 2 else _CACHE.get(("w2", id(m)), w, lambda: w.detach().reshape(w.shape[0], w.shape[1]).contiguous())


# ------------------------------------------------------------------------------------------------------ attention
def set_attention_precision(module: nn.Module, precision: str = "bf16") -> nn.Module:
    """'bf16' (the reference's precision) or 'fp8': spatial self-attention (attention.py:189 with context None) of every CrossAttention under
    `module` runs on the e4m3 MFMA path where the shape allows (S % 128 == 0, S >= 512); everything else stays bf16."""
    if precision not in ("bf16", "fp8"):
        raise ValueError("precision must be 'bf16' or 'fp8'")
    for m in module.modules():
        if isinstance(m, CrossAttention):
            m.attention_precision = precision
    return module


class CrossAttention(nn.Module):
    """attention.py:37-223 (`efficient_forward` path: relative_position=False)."""
    attention_precision = "bf16"

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, relative_position=False, temporal_length=None,
                 video_length=None, image_cross_attention=False, image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False,
                 text_context_len=77, action_cross_attention=False, action_cross_attention_scale=1.0,
                 action_cross_attention_scale_learnable=False, mix_attention=False):
        super().__init__()
        if dim_head != 64 or relative_position or mix_attention:
            raise NotImplementedError("head_dim 64, no relative position, no mix attention (the shipped DynamiCrafter config)")
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.inner = heads, inner
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self.image_cross_attention, self.action_cross_attention = image_cross_attention, action_cross_attention
        self.image_cross_attention_scale, self.action_cross_attention_scale = image_cross_attention_scale, action_cross_attention_scale
        self.image_cross_attention_scale_learnable = image_cross_attention_scale_learnable
        self.action_cross_attention_scale_learnable = action_cross_attention_scale_learnable
        if image_cross_attention:
            self.to_k_ip = nn.Linear(context_dim, inner, bias=False)
            self.to_v_ip = nn.Linear(context_dim, inner, bias=False)
            if image_cross_attention_scale_learnable:
                self.register_parameter("alpha", nn.Parameter(torch.tensor(0.0)))
        if action_cross_attention:
            self.to_q_a = nn.Linear(inner, inner, bias=False)
            self.to_k_a = nn.Linear(context_dim, inner, bias=False)
            self.to_v_a = nn.Linear(context_dim, inner, bias=False)
            if action_cross_attention_scale_learnable:
                self.register_parameter("alpha_action", nn.Parameter(torch.tensor(0.0)))

    def _kv(self, key, wk, wv, ctx):
        w = _CACHE.get((key, id(self)), (wk.weight, wv.weight), lambda: _cat0([wk.weight, wv.weight]))
        kv = ops.linear(ctx.contiguous(), w)
        return kv[..., : self.inner].unflatten(-1, (self.heads, 64)), kv[..., self.inner:].unflatten(-1, (self.heads, 64))

    def forward(self, x: torch.Tensor, context: Optional[dict] = None, mask=None, *, resid: Optional[torch.Tensor] = None, kv_div: int = 1,
                temporal: Optional[tuple] = None) -> torch.Tensor:
        """x [Nb, L, C] rows.  `resid` is added to the projected output (the block's `+ x`).  `context` tensors may have a
        batch of Nb / kv_div (prompt / action repeated over frames: openaimodel3d.py:586-596).  `temporal=(b, t, hw)`: x rows
        are ordered (b, t, hw) and attention runs over t for every (b, hw) (TemporalTransformer, attention.py:399-402)."""
        if mask is not None:
            raise NotImplementedError                                               # attention.py:172-173
        H, inner = self.heads, self.inner
        Nb, L, _ = x.shape
        if context is None:                                                         # spatial / temporal self-attention  :175-183
            w = _CACHE.get(("qkv", id(self)), (self.to_q.weight, self.to_k.weight, self.to_v.weight), lambda: _cat0([self.to_q.weight, self.to_k.weight, self.to_v.weight]))
            qkv = ops.linear(x, w)
            if temporal is None:
                q5 = qkv.view(Nb, L, 3, H, 64)
                # BASELINE config "fp8 MFMA attention path": opt-in per module (set_attention_precision), long spatial sequences only
                out = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], fp8=self.attention_precision == "fp8" and ops.fp8_attention_supported(L, L))
            else:
                b, t, hw = temporal
                out = torch.empty(Nb, L, inner, dtype=torch.bfloat16, device=x.device)
                q6 = qkv.view(b, t, hw, 3, H, 64)
                o4 = out.view(b, t, hw, inner)
                for i in range(b):                                                  # batch = hw (stride one row), sequence = t
                    ops.attention(q6[i, :, :, 0].permute(1, 0, 2, 3), q6[i, :, :, 1].permute(1, 0, 2, 3), q6[i, :, :, 2].permute(1, 0, 2, 3),
                                  out=o4[i].permute(1, 0, 2))
        else:
            q = ops.linear(x, self.to_q.weight).view(Nb, L, H, 64)
            k, v = self._kv("kv", self.to_k, self.to_v, context["prompt"])
            out = ops.attention(q, k, v, kv_batch_div=Nb // k.shape[0])             # :189
            if self.image_cross_attention:                                          # :191-204
                k, v = self._kv("kv_ip", self.to_k_ip, self.to_v_ip, context["image"])
                s = self.image_cross_attention_scale * ((_tanh_scalar(self.alpha) + 1) if self.image_cross_attention_scale_learnable else 1.0)
                ops.attention(q, k, v, out=out, resid=out, kv_batch_div=Nb // k.shape[0], out_scale=float(s))
            if self.action_cross_attention:                                         # :206-220  q_a = to_q_a(out)
                q_a = ops.linear(out, self.to_q_a.weight).view(Nb, L, H, 64)
                k, v = self._kv("kv_a", self.to_k_a, self.to_v_a, context["action"])
                s = self.action_cross_attention_scale * ((_tanh_scalar(self.alpha_action) + 1) if self.action_cross_attention_scale_learnable else 1.0)
                ops.attention(q_a, k, v, out=out, resid=out, kv_batch_div=Nb // k.shape[0], out_scale=float(s))
        if resid is not None:
            return ops.linear(out, self.to_out[0].weight, self.to_out[0].bias, epilogue=ops.EPI_RESID, resid=resid)
        return ops.linear(out, self.to_out[0].weight, self.to_out[0].bias)          # :222-223

    efficient_forward = forward


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    """attention.py:458-475 with glu=True"""

    def __init__(self, dim, dim_out=None, mult=4, glu=True, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("the UNet uses gated_ff=True")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim if dim_out is None else dim_out))

    def forward(self, x, resid=None):
        pj = self.net[0].proj
        w, b = _CACHE.get(("geglu", id(pj)), (pj.weight, pj.bias), lambda: ops.geglu_interleave(pj.weight, pj.bias))
        h = ops.linear(x, w, b, epilogue=ops.EPI_GEGLU)                             # value * gelu(gate) in the GEMM epilogue
        if resid is not None:
            return ops.linear(h, self.net[2].weight, self.net[2].bias, epilogue=ops.EPI_RESID, resid=resid)
        return ops.linear(h, self.net[2].weight, self.net[2].bias)


class BasicTransformerBlock(nn.Module):
    """attention.py:226-266"""

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True, disable_self_attn=False,
                 attention_cls=None, video_length=None, image_cross_attention=False, image_cross_attention_scale=1.0,
                 image_cross_attention_scale_learnable=False, text_context_len=77, action_cross_attention=False,
                 action_cross_attention_scale_learnable=False):
        super().__init__()
        if disable_self_attn:
            raise NotImplementedError
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                                    image_cross_attention=image_cross_attention, image_cross_attention_scale=image_cross_attention_scale,
                                    image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                    action_cross_attention=action_cross_attention,
                                    action_cross_attention_scale_learnable=action_cross_attention_scale_learnable)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def forward(self, x, context=None, mask=None, temporal=None):
        ln = lambda n, v: ops.layernorm(v, n.weight, n.bias, n.eps)
        x = self.attn1(ln(self.norm1, x), None, resid=x, temporal=temporal)                     # :263
        x = self.attn2(ln(self.norm2, x), context, resid=x, temporal=temporal)                  # :264  (context None -> self-attention)
        return self.ff(ln(self.norm3, x), resid=x)                                              # :265


class SpatialTransformer(nn.Module):
    """attention.py:269-332 (use_linear=True)"""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_checkpoint=True, disable_self_attn=False,
                 use_linear=True, video_length=None, image_cross_attention=False, image_cross_attention_scale_learnable=False,
                 action_cross_attention=False, action_cross_attention_scale_learnable=False):
        super().__init__()
        if not use_linear:
            raise NotImplementedError("the shipped config sets use_linear: true")
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim, image_cross_attention=image_cross_attention,
                                  image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                  action_cross_attention=action_cross_attention,
                                  action_cross_attention_scale_learnable=action_cross_attention_scale_learnable) for _ in range(depth)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x: torch.Tensor, context=None) -> torch.Tensor:
        """x [(b t), H, W, C]"""
        N, H, W, C = x.shape
        rows = x.view(N, H * W, C)
        y = ops.groupnorm(rows, self.norm.weight, self.norm.bias, 32, self.norm.eps)
        y = ops.linear(y, self.proj_in.weight, self.proj_in.bias)
        for blk in self.transformer_blocks:
            y = blk(y, context=context)
        return ops.linear(y, self.proj_out.weight, self.proj_out.bias, epilogue=ops.EPI_RESID, resid=rows).view(N, H, W, C)   # + x_in :332


class TemporalTransformer(nn.Module):
    """attention.py:335-445 (only_self_att, no causal mask, no relative position)"""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_checkpoint=True, use_linear=False,
                 only_self_att=True, causal_attention=False, causal_block_size=1, relative_position=False, temporal_length=None,
                 action_cross_attention=False, action_cross_attention_scale_learnable=False):
        super().__init__()
        if causal_attention or relative_position or action_cross_attention or not only_self_att:
            raise NotImplementedError("shipped config: temporal self-attention only")
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.use_linear = use_linear
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv1d(in_channels, inner, kernel_size=1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=None) for _ in range(depth)])
        self.proj_out = nn.Linear(inner, in_channels) if use_linear else nn.Conv1d(inner, in_channels, kernel_size=1)

    def forward(self, x: torch.Tensor, b: int, context=None) -> torch.Tensor:
        """x [(b t), H, W, C]; GroupNorm statistics over (t, h, w) per sample (5-D input in the reference, :397-398)"""
        N, H, W, C = x.shape
        t, hw = N // b, H * W
        y = ops.groupnorm(x.view(b, t * hw, C), self.norm.weight, self.norm.bias, 32, self.norm.eps).view(N, hw, C)
        y = ops.linear(y, _lin_w(self.proj_in), self.proj_in.bias)
        for blk in self.transformer_blocks:
            y = blk(y, context=None, temporal=(b, t, hw))
        return ops.linear(y, _lin_w(self.proj_out), self.proj_out.bias, epilogue=ops.EPI_RESID, resid=x.view(N, hw, C)).view(N, H, W, C)


# ------------------------------------------------------------------------------------------------------ conv blocks
class TemporalConvBlock(nn.Module):
    """openaimodel3d.py:240-281"""

    def __init__(self, in_channels, out_channels=None, dropout=0.0, spatial_aware=False):
        super().__init__()
        if spatial_aware:
            raise NotImplementedError
        out_channels = in_channels if out_channels is None else out_channels
        mk = lambda cin, cout, drop: nn.Sequential(*([nn.GroupNorm(32, cin), nn.SiLU()] + ([nn.Dropout(dropout)] if drop else []) +
                                                     [nn.Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0))]))
        self.conv1, self.conv2 = mk(in_channels, out_channels, False), mk(out_channels, in_channels, True)
        self.conv3, self.conv4 = mk(out_channels, in_channels, True), mk(out_channels, in_channels, True)

    def forward(self, x: torch.Tensor, b: int) -> torch.Tensor:
        """x [(b t), HW, C]"""
        N, HW, C = x.shape
        t = N // b
        y = x
        for i, seq in enumerate((self.conv1, self.conv2, self.conv3, self.conv4)):
            gn = seq[0]
            y = ops.groupnorm(y.view(b, t * HW, y.shape[-1]), gn.weight, gn.bias, 32, gn.eps, silu=True).view(N, HW, -1)
            y = conv_t3(y, seq[-1], b, t, resid=x if i == 3 else None)               # identity + x  :281
        return y


class ResBlock(nn.Module):
    """openaimodel3d.py:110-237 (no up/down, use_scale_shift_norm=False)"""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, dims=2, use_checkpoint=False,
                 use_conv=False, up=False, down=False, use_temporal_conv=False, tempspatial_aware=False):
        super().__init__()
        if up or down or use_scale_shift_norm or dims != 2:
            raise NotImplementedError("shipped config: resblock_updown=False, use_scale_shift_norm=False")
        self.channels, self.out_channels = channels, out_channels or channels
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 3 if use_conv else 1, padding=1 if use_conv else 0)
        self.use_temporal_conv = use_temporal_conv
        if use_temporal_conv:
            self.temopral_conv = TemporalConvBlock(self.out_channels, self.out_channels, dropout=0.1, spatial_aware=tempspatial_aware)   # (sic)

    def forward(self, x: torch.Tensor, silu_emb: torch.Tensor, batch_size: Optional[int] = None) -> torch.Tensor:
        """x [(b t), H, W, C]; silu_emb = SiLU(emb) [(b t), emb_channels] (shared by every ResBlock of the step)"""
        N, H, W, C = x.shape
        gi, go = self.in_layers[0], self.out_layers[0]
        h = ops.groupnorm(x.view(N, H * W, C), gi.weight, gi.bias, 32, gi.eps, silu=True).view(N, H, W, C)
        h = conv3x3(h, self.in_layers[2])
        emb_out = temb_proj(silu_emb, self.emb_layers[1])                                                      # :222 (one batched GEMM per step: TembBank)
        h = ops.groupnorm(h.view(N, H * W, -1), go.weight, go.bias, 32, go.eps, silu=True, emb=emb_out).view(N, H, W, -1)   # h + emb_out -> GN -> SiLU
        if isinstance(self.skip_connection, nn.Identity):
            skip = x
        elif self.skip_connection.kernel_size == (1, 1):
            skip = ops.linear(x, _lin_w(self.skip_connection), self.skip_connection.bias)
        else:
            skip = conv3x3(x, self.skip_connection)
        h = conv3x3(h, self.out_layers[3], resid=skip)                                                          # skip + h  :231
        if self.use_temporal_conv and batch_size:
            h = self.temopral_conv(h.view(N, H * W, -1), batch_size).view(N, H, W, -1)
        return h


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2:
            raise NotImplementedError
        self.op = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x):
        return conv3x3(x, self.op, stride=2)


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2:
            raise NotImplementedError
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=padding)

    def forward(self, x):
        return conv3x3(x, self.conv, upsample=True)                                 # nearest x2 fused into the row gather


class TimestepEmbedSequential(nn.Sequential):
    """openaimodel3d.py:30-49"""

    def forward(self, x, silu_emb, context=None, batch_size=None):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, silu_emb, batch_size=batch_size)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            elif isinstance(layer, TemporalTransformer):
                x = layer(x, batch_size, context)
            elif isinstance(layer, nn.Conv2d):
                x = conv3x3(x, layer)
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    """openaimodel3d.py:284-635 for the shipped DynamiCrafter-1024 configuration (configs/dynamicrafter/MotionRAG_open.yml:206-238)."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0, channel_mult=(1, 2, 4, 8),
                 conv_resample=True, dims=2, context_dim=None, use_scale_shift_norm=False, resblock_updown=False, num_heads=-1,
                 num_head_channels=-1, transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False, tempspatial_aware=False,
                 temporal_attention=True, use_relative_position=True, use_causal_attention=False, temporal_length=None, use_fp16=False,
                 addition_attention=False, temporal_self_att_only=True, image_cross_attention=False, image_cross_attention_scale_learnable=False,
                 action_cross_attention=False, action_cross_attention_scale_learnable=False, temporal_action_cross_attention=False,
                 temporal_action_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False):
        super().__init__()
        if num_head_channels != 64 or not use_linear or use_relative_position or use_causal_attention or resblock_updown or temporal_action_cross_attention:
            raise NotImplementedError("built for the shipped config: head_dim 64, use_linear, no relative position / causal attention")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.addition_attention, self.fs_condition, self.default_fs = addition_attention, fs_condition, default_fs
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if fs_condition:
            self.fps_embedding = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        st = lambda ch: SpatialTransformer(ch, ch // 64, 64, depth=transformer_depth, context_dim=context_dim, use_linear=True,
                                           image_cross_attention=image_cross_attention, action_cross_attention=action_cross_attention,
                                           image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                           action_cross_attention_scale_learnable=action_cross_attention_scale_learnable)
        tt = lambda ch: TemporalTransformer(ch, ch // 64, 64, depth=transformer_depth, context_dim=context_dim, use_linear=True,
                                            only_self_att=temporal_self_att_only, temporal_length=temporal_length)
        res = lambda cin, cout: ResBlock(cin, ted, dropout, out_channels=cout, use_temporal_conv=temporal_conv, tempspatial_aware=tempspatial_aware)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        if addition_attention:
            self.init_attn = TimestepEmbedSequential(TemporalTransformer(model_channels, n_heads=8, d_head=64, depth=transformer_depth,
                                                                         context_dim=context_dim, only_self_att=temporal_self_att_only,
                                                                         temporal_length=temporal_length))      # Conv1d projections, inner 512
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers: List[nn.Module] = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(st(ch))
                    if temporal_attention:
                        layers.append(tt(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        mid: List[nn.Module] = [res(ch, ch), st(ch)]
        if temporal_attention:
            mid.append(tt(ch))
        mid.append(res(ch, ch))
        self.middle_block = TimestepEmbedSequential(*mid)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [res(ch + ich, mult * model_channels)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(st(ch))
                    if temporal_attention:
                        layers.append(tt(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))

    @torch.no_grad()
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context: Optional[Dict[str, torch.Tensor]] = None, features_adapter=None,
                fs: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        """x [b, c, t, h, w] -> [b, c_out, t, h, w] (reference layout at the boundary; channels-last rows inside)"""
        if features_adapter is not None:
            raise NotImplementedError
        b, c, t, hh, ww = x.shape
        dev = x.device
        mlp = lambda seq, e: ops.linear(ops.linear(e, seq[0].weight, seq[0].bias, epilogue=ops.EPI_SILU), seq[2].weight, seq[2].bias)
        emb = mlp(self.time_embed, ops.timestep_embedding(timesteps.to(dev, torch.float32), self.model_channels))      # :581-582
        ctx = {}
        if "image" in context:
            ctx["image"] = context["image"].to(torch.bfloat16).reshape(b * t, -1, context["image"].shape[-1])            # 'b (t l) c -> (b t) l c'
        if "prompt" in context:
            ctx["prompt"] = context["prompt"].to(torch.bfloat16)       # repeated over t through kv_batch_div (no copy)  :590
        if "action" in context:
            ctx["action"] = context["action"].to(torch.bfloat16)       # 'b l c -> (b t) l c'  :593
        if self.fs_condition:                                                                                          # :603-610
            if fs is None:
                fs = torch.tensor([self.default_fs] * b, dtype=torch.long, device=dev)
            emb = ops.add(emb, mlp(self.fps_embedding, ops.timestep_embedding(fs.to(dev, torch.float32), self.model_channels)))
        silu_emb = ops.silu(emb).repeat_interleave(t, dim=0).contiguous()                                               # every ResBlock uses SiLU(emb)
        if getattr(self, "_temb_linears", None) is None:
            self._temb_linears = [m.emb_layers[1] for m in self.modules() if isinstance(m, ResBlock)]
        silu_emb = TembBank(silu_emb, self, self._temb_linears)
        h = x.to(torch.bfloat16).permute(0, 2, 3, 4, 1).reshape(b * t, hh, ww, c).contiguous()                          # b c t h w -> (b t) h w c
        hs = []
        for i, module in enumerate(self.input_blocks):
            h = module(h, silu_emb, context=ctx, batch_size=b)
            if i == 0 and self.addition_attention:
                h = self.init_attn(h, silu_emb, context=ctx, batch_size=b)
            hs.append(h)
        h = self.middle_block(h, silu_emb, context=ctx, batch_size=b)
        for module in self.output_blocks:
            h = torch.cat([h, hs.pop()], dim=-1)                                                                        # channel concat (memory only)
            h = module(h, silu_emb, context=ctx, batch_size=b)
        N, H, W, C = h.shape
        go = self.out[0]
        y = conv3x3(ops.groupnorm(h.view(N, H * W, C), go.weight, go.bias, 32, go.eps, silu=True).view(N, H, W, C), self.out[2])
        return y.view(b, t, H, W, -1).permute(0, 4, 1, 2, 3).contiguous()


# ------------------------------------------------------------------------------------------------------ DDIM sampler
def make_alphas_cumprod(timesteps=1000, linear_start=0.00085, linear_end=0.012, rescale_betas_zero_snr=True) -> np.ndarray:
    """ddpm3d.py:134-148 + utils_diffusion.py:31-54,113-146 (host-side float64 tables)"""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    if rescale_betas_zero_snr:
        s = np.sqrt(np.cumprod(1.0 - betas))
        s0, sT = s[0].copy(), s[-1].copy()
        s = (s - sT) * (s0 / (s0 - sT))
        ab = s ** 2
        betas = 1 - np.concatenate([ab[0:1], ab[1:] / ab[:-1]])
    return np.cumprod(1.0 - betas)


class DDIMSampler:
    """samplers/ddim.py for the v-parameterised, dynamically rescaled model; `model(x, t, cond) -> v` is any callable returning bf16
    `[2b, ...]` for the CFG-concatenated batch (cond first).  Noise is supplied by the caller (CPU-seeded)."""

    def __init__(self, alphas_cumprod: Optional[np.ndarray] = None, use_dynamic_rescale=True, base_scale=0.3, turning_step=400):
        self.ac = make_alphas_cumprod() if alphas_cumprod is None else alphas_cumprod
        self.ac32 = self.ac.astype(np.float32)
        n = len(self.ac)
        self.use_dynamic_rescale = use_dynamic_rescale
        self.scale_arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(n, base_scale))).astype(np.float32)   # ddpm3d.py:535-541

    def make_schedule(self, ddim_num_steps: int, ddim_eta: float = 0.0):
        n = len(self.ac)
        self.ddim_timesteps = np.asarray(list(range(0, n, n // ddim_num_steps))) + 1                  # utils_diffusion.py:57-60,71 ('uniform')
        a = self.ac32[self.ddim_timesteps]
        a_prev = np.asarray([self.ac32[0]] + self.ac32[self.ddim_timesteps[:-1]].tolist(), dtype=np.float32)
        self.ddim_alphas, self.ddim_alphas_prev = a, a_prev
        self.ddim_sigmas = (ddim_eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))).astype(np.float32)
        sc = self.scale_arr[self.ddim_timesteps]
        self.ddim_scale_arr, self.ddim_scale_arr_prev = sc, np.concatenate([sc[0:1], sc[:-1]])
        return self.ddim_timesteps

    def step_coeffs(self, index: int):
        t = int(self.ddim_timesteps[index])
        a_t, a_prev, sigma = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index]), float(self.ddim_sigmas[index])
        rescale = float(self.ddim_scale_arr_prev[index] / self.ddim_scale_arr[index]) if self.use_dynamic_rescale else 1.0
        sa, sb = float(np.sqrt(self.ac32[t])), float(np.sqrt(np.float32(1.0) - self.ac32[t]))
        return t, sa, sb, rescale, float(np.sqrt(np.float32(a_prev))), float(np.sqrt(np.float32(1.0 - a_prev - sigma ** 2))), sigma

    @torch.no_grad()
    def sample(self, model, x_T: torch.Tensor, cond, uncond, S: int, eta: float = 1.0, unconditional_guidance_scale: float = 2.0,
               noises: Optional[List[torch.Tensor]] = None, callback=None) -> torch.Tensor:
        """ddim_sampling (:135-200) + p_sample_ddim (:203-298); x_T fp32 latents [b, 4, t, h, w] (updated in place)."""
        self.make_schedule(S, eta)
        x = x_T.to(torch.float32).contiguous()
        n_steps = len(self.ddim_timesteps)
        for i in range(n_steps):
            index = n_steps - 1 - i
            t, sa, sb, rescale, sqrt_aprev, dir_coef, sigma = self.step_coeffs(index)
            v = model(x, t, cond, uncond)
            ops.ddim_v_step_(v.contiguous(), x, noises[i] if noises is not None else None, unconditional_guidance_scale, sa, sb, rescale,
                             sqrt_aprev, dir_coef, sigma)
            if callback is not None:
                callback(i, t, x)
        return x


class DynamiCrafterDenoiser:
    """`LatentDiffusion.apply_model` + `DiffusionWrapper.forward` in 'hybrid' mode (ddpm3d.py:745-760,1378-1382) with the CFG batch of
    p_sample_ddim (samplers/ddim.py:219-237): x is concatenated with `c_concat` on channels, cond and uncond on the batch (cond FIRST)."""

    def __init__(self, unet: UNetModel):
        self.unet = unet

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, t: int, cond: dict, uncond: dict) -> torch.Tensor:
        b = x.shape[0]
        xb = x.to(torch.bfloat16)
        xin = torch.cat([torch.cat([xb, cond["c_concat"][0].to(torch.bfloat16)], dim=1), torch.cat([xb, uncond["c_concat"][0].to(torch.bfloat16)], dim=1)], dim=0)
        ctx = {k: torch.cat([cond["c_crossattn"][k], uncond["c_crossattn"][k]], dim=0) for k in cond["c_crossattn"]}
        ts = torch.full((2 * b,), float(t), dtype=torch.float32, device=x.device)
        fs = torch.cat([cond["fs"], uncond["fs"]], dim=0) if "fs" in cond else None
        return self.unet(xin, ts, context=ctx, fs=fs)
