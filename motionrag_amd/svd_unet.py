"""Motion-injected SVD spatio-temporal UNet denoise step on hand-written gfx950 kernels (BASELINE config "SVD-UNet
14x576x1024 single denoise step").

The reference builds this network from the third-party diffusers package (`UNetSpatioTemporalConditionModel.from_pretrained`,
src/projects/svd/module.py:38-47), installs `APAdapterAttnProcessor2_0` on every spatial `attn2`
(src/projects/svd/module.py:145-165, sites listed in configs/svd/MotionRAG_open.yml:115-131) and feeds the CAMA motion tokens through
`TupleTensor` (src/projects/svd/pipelines/pipeline.py:25-57,113-119).  This module is the host-side mirror of that model: the same
class / attribute names, hence the same state-dict keys as the diffusers checkpoint + `Motion-Adapter.ckpt`, the same
`attn_processors` / `set_attn_processor` protocol, the same `forward(sample, timestep, encoder_hidden_states, added_time_ids)`.

Layout: activations are channels-last rows `[(b f), H, W, C]` bf16 end to end; the package's NCHW <-> `(b f) (h w) c` <->
`(b h w) f c` rearranges disappear (temporal attention reads a strided view, GroupNorm over (f, h, w) is a row-count change).
nn.Module is a weight container; every arithmetic op is a libmrag_hip.so kernel (ops.*), no CPU or torch fallback.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
from torch import nn

from . import ops
from .attn_processor import Attention
from .dynamicrafter import _CACHE, TembBank, _cat0, conv3x3, conv_t3, temb_proj


def _ln(n: nn.LayerNorm, x):
    return ops.layernorm(x, n.weight, n.bias, n.eps)


def _sigmoid_scalar(p: nn.Parameter) -> float:
    """AlphaBlender's learned scalar, read back once per weight version (host sync only on the first call)"""
    return _CACHE.get(("mix", id(p)), p, lambda: float(torch.sigmoid(p.detach().float()).item()))


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim if out_dim is None else out_dim)

    def forward(self, x):
        return ops.linear(ops.linear(x, self.linear_1.weight, self.linear_1.bias, epilogue=ops.EPI_SILU), self.linear_2.weight, self.linear_2.bias)


class AlphaBlender(nn.Module):
    """merge_strategy 'learned_with_images' with image_only_indicator == 0 (the only mode the SVD pipeline drives)"""

    def __init__(self, alpha: float = 0.5):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha]))

    def forward(self, x_spatial, x_temporal):
        a = _sigmoid_scalar(self.mix_factor)
        return ops.axpby(x_spatial, x_temporal, a, 1.0 - a)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, silu_temb):
        """x [N, H, W, C]; silu_temb [N, temb] (SiLU already applied)"""
        N, H, W, C = x.shape
        h = ops.groupnorm(x.view(N, H * W, C), self.norm1.weight, self.norm1.bias, 32, self.norm1.eps, silu=True).view(N, H, W, C)
        h = conv3x3(h, self.conv1)
        t = temb_proj(silu_temb, self.time_emb_proj)                               # a slice of the step's one batched projection (TembBank)
        co = h.shape[-1]
        h = ops.groupnorm(h.view(N, H * W, co), self.norm2.weight, self.norm2.bias, 32, self.norm2.eps, silu=True, emb=t).view(N, H, W, co)
        if self.conv_shortcut is not None:
            w = _CACHE.get(("sc", id(self.conv_shortcut)), self.conv_shortcut.weight, lambda: self.conv_shortcut.weight.detach().reshape(co, C).contiguous())
            x = ops.linear(x, w, self.conv_shortcut.bias)
        return conv3x3(h, self.conv2, resid=x)


class TemporalResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps):
        super().__init__()
        if in_channels != out_channels:
            raise NotImplementedError("SpatioTemporalResBlock always builds the temporal block with in == out")
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x, silu_temb, b: int, branch_scale: float = 1.0):
        """x [(b f), HW, C]; GroupNorm statistics over (f, h, w) per sample (5-D input in the package).  Returns x + branch_scale * branch(x):
        the block's own `x + h` for 1, the AlphaBlender mix behind it (SpatioTemporalResBlock) for 1 - alpha"""
        N, HW, C = x.shape
        f = N // b
        h = ops.groupnorm(x.view(b, f * HW, C), self.norm1.weight, self.norm1.bias, 32, self.norm1.eps, silu=True).view(N, HW, C)
        h = conv_t3(h, self.conv1, b, f)
        t = temb_proj(silu_temb, self.time_emb_proj)                                             # [(b f), C] -> one vector per frame
        h = ops.add_bcast(h, t, HW)
        h = ops.groupnorm(h.view(b, f * HW, C), self.norm2.weight, self.norm2.bias, 32, self.norm2.eps, silu=True).view(N, HW, C)
        return conv_t3(h, self.conv2, b, f, resid=x, acc_scale=branch_scale)


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps=1e-6, merge_factor=0.5):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, out_channels, temb_channels, eps)
        self.time_mixer = AlphaBlender(merge_factor)

    def forward(self, x, silu_temb, b: int):
        s = self.spatial_res_block(x, silu_temb)
        N, H, W, C = s.shape
        # time_mixer(s, t) with t = s + c (the temporal block's residual form): a s + (1 - a)(s + c) = s + (1 - a) c -- the blend rides in the epilogue
        # of the temporal block's last convolution (one rounding of the mixed value instead of the package's three; no separate pass over the activation)
        a = _sigmoid_scalar(self.time_mixer.mix_factor)
        return self.temporal_res_block(s.view(N, H * W, C), silu_temb, b, branch_scale=1.0 - a).view(N, H, W, C)


class AttnProcessor2_0:
    """plain scaled-dot-product processor (diffusers' default) on the gfx950 kernels.  `temporal=(b, f, hw)`: rows are ordered
    (b, f, hw) and attention runs over f for every (b, hw).  `precision='fp8'`: long spatial self-attention on the e4m3 MFMA path."""

    def __init__(self, precision: str = "bf16"):
        if precision not in ("bf16", "fp8"):
            raise ValueError("precision must be 'bf16' or 'fp8'")
        self.precision = precision

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, temporal=None, block_residual=None, **_):
        residual = block_residual                                                       # the block's `x` of `x = attn(norm(x)) + x` (not a diffusers argument)
        x = hidden_states
        Nb, L, C = x.shape
        H = attn.heads
        if encoder_hidden_states is None:
            w = _CACHE.get(("qkv", id(attn)), (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight), lambda: _cat0([attn.to_q.weight, attn.to_k.weight, attn.to_v.weight]))
            qkv = ops.linear(x, w)
            if temporal is None:
                q5 = qkv.view(Nb, L, 3, H, 64)
                o = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], fp8=self.precision == "fp8" and ops.fp8_attention_supported(L, L))
            else:
                b, f, hw = temporal
                o = torch.empty(Nb, L, C, dtype=torch.bfloat16, device=x.device)
                q6, o4 = qkv.view(b, f, hw, 3, H, 64), o.view(b, f, hw, C)
                for i in range(b):
                    ops.attention(q6[i, :, :, 0].permute(1, 0, 2, 3), q6[i, :, :, 1].permute(1, 0, 2, 3), q6[i, :, :, 2].permute(1, 0, 2, 3),
                                  out=o4[i].permute(1, 0, 2))
        else:
            ctx = encoder_hidden_states.contiguous()
            q = ops.linear(x, attn.to_q.weight).view(Nb, L, H, 64)
            wkv = _CACHE.get(("kv", id(attn)), (attn.to_k.weight, attn.to_v.weight), lambda: _cat0([attn.to_k.weight, attn.to_v.weight]))
            kv = ops.linear(ctx, wkv)
            o = ops.attention(q, kv[..., :C].unflatten(-1, (H, 64)), kv[..., C:].unflatten(-1, (H, 64)), kv_batch_div=Nb // ctx.shape[0])
        if residual is not None:                                                        # block's `attn(norm(x)) + x` in the projection's epilogue
            return ops.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, epilogue=ops.EPI_RESID, resid=residual)
        return ops.linear(o, attn.to_out[0].weight, attn.to_out[0].bias)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim if dim_out is None else dim_out)])

    def forward(self, x, resid=None):
        pj = self.net[0].proj
        w, b = _CACHE.get(("geglu", id(pj)), (pj.weight, pj.bias), lambda: ops.geglu_interleave(pj.weight, pj.bias))
        h = ops.linear(x, w, b, epilogue=ops.EPI_GEGLU)                             # value * gelu(gate) in the GEMM epilogue
        if resid is not None:
            return ops.linear(h, self.net[2].weight, self.net[2].bias, epilogue=ops.EPI_RESID, resid=resid)
        return ops.linear(h, self.net[2].weight, self.net[2].bias)


def _attn(dim, heads, head_dim, cross=None):
    if head_dim != 64:
        raise NotImplementedError("head_dim 64 (every SVD attention site)")
    return Attention(dim, cross_attention_dim=cross, heads=heads, dim_head=head_dim, bias=False, out_bias=True, processor=AttnProcessor2_0())


def _attn_plus(attn, normed, x, **kw):
    """attn(normed) + x with the add in the output projection's epilogue when the site's processor takes `block_residual` (the two processors of this
    package); any other processor object installed through the diffusers protocol gets the plain call and a separate add"""
    import inspect
    proc = attn.processor
    ok = getattr(proc, "_takes_block_residual", None)
    if ok is None:
        ok = "block_residual" in inspect.signature(proc.__call__).parameters
        try:
            proc._takes_block_residual = ok
        except AttributeError:
            pass
    if ok:
        return attn(normed, block_residual=x, **kw)
    return ops.add(attn(normed, **kw), x)


class BasicTransformerBlock(nn.Module):
    """spatial block; `attn2` is the motion-adapter site (its processor is replaced by APAdapterAttnProcessor2_0)"""

    def __init__(self, dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn1 = _attn(dim, heads, head_dim)
        self.attn2 = _attn(dim, heads, head_dim, cross_attention_dim)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states):
        x = _attn_plus(self.attn1, _ln(self.norm1, x), x)
        x = _attn_plus(self.attn2, _ln(self.norm2, x), x, encoder_hidden_states=encoder_hidden_states)
        return self.ff(_ln(self.norm3, x), resid=x)


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, time_mix_inner_dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        if dim != time_mix_inner_dim:
            raise NotImplementedError("is_res = False")
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = _attn(time_mix_inner_dim, heads, head_dim)
        self.norm2 = nn.LayerNorm(time_mix_inner_dim)
        self.attn2 = _attn(time_mix_inner_dim, heads, head_dim, cross_attention_dim)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, x, temporal, first_frame_context):
        """x [(b f), HW, C] in place of the package's [(b hw), f, C]; first_frame_context [b, 1, D]"""
        b, f, hw = temporal
        x = self.ff_in(_ln(self.norm_in, x), resid=x)
        x = _attn_plus(self.attn1, _ln(self.norm1, x), x, temporal=temporal)
        # attn2: every (b, hw) row attends to ONE context token, so softmax == 1 exactly and the branch is
        # to_out(to_v(context)) whatever the query; the package builds `time_context` with rows ordered (hw, b) while the
        # hidden rows are ordered (b, hw), i.e. row n = b*hw + s reads the context of batch n % B -- reproduced as is.
        if first_frame_context.shape[1] != 1:
            raise NotImplementedError("the SVD image embedding is one token")
        if hw % b != 0:
            raise NotImplementedError("hw % batch != 0")
        a = self.attn2
        vec = ops.linear(ops.linear(first_frame_context.reshape(b, -1).contiguous(), a.to_v.weight), a.to_out[0].weight, a.to_out[0].bias)   # [b, C]
        x = ops.add_bcast(x, vec, 1)                                                   # row r -> vec[r % b]; ((b f) hw + s) % b == (b hw + s) % b
        return self.ff(_ln(self.norm3, x), resid=x)


class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, num_attention_heads, attention_head_dim, in_channels, num_layers=1, cross_attention_dim=None):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim) for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList([TemporalBasicTransformerBlock(inner, inner, num_attention_heads, attention_head_dim, cross_attention_dim)
                                                          for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_mixer = AlphaBlender(0.5)
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states, b: int):
        """x [(b f), H, W, C]; encoder_hidden_states [(b f), 1, D] tensor or TupleTensor (image embedding, motion tokens)"""
        N, H, W, C = x.shape
        f, hw = N // b, H * W
        image_ctx = encoder_hidden_states[:]                                       # TupleTensor.__getitem__ -> first member (pipeline.py:42-43)
        first = image_ctx.view(b, f, -1, image_ctx.shape[-1])[:, 0]
        rows = x.view(N, hw, C)
        h = ops.groupnorm(rows, self.norm.weight, self.norm.bias, 32, self.norm.eps)
        h = ops.linear(h, self.proj_in.weight, self.proj_in.bias)
        idx = torch.arange(f, device=x.device, dtype=torch.float32).repeat(b)
        emb = self.time_pos_embed(ops.timestep_embedding(idx, self.in_channels))   # [(b f), C]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            h = blk(h, encoder_hidden_states)
            m = ops.add_bcast(h, emb, hw)
            m = tblk(m, (b, f, hw), first)
            h = self.time_mixer(h, m)
        return ops.linear(h, self.proj_out.weight, self.proj_out.bias, epilogue=ops.EPI_RESID, resid=rows).view(N, H, W, C)


class _Sampler2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class DownBlock(nn.Module):
    """CrossAttnDownBlockSpatioTemporal (cross=True) / DownBlockSpatioTemporal"""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, heads, cross_attention_dim, cross, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps=1e-6 if cross else 1e-5)
                                      for i in range(num_layers)])
        if cross:
            self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(heads, out_channels // heads, out_channels, 1, cross_attention_dim)
                                             for _ in range(num_layers)])
        self.has_cross = cross
        if add_downsample:
            self.downsamplers = nn.ModuleList([_Sampler2D(out_channels)])
        self.add_downsample = add_downsample

    def forward(self, x, silu_temb, ehs, b):
        outs = []
        for i, res in enumerate(self.resnets):
            x = res(x, silu_temb, b)
            if self.has_cross:
                x = self.attentions[i](x, ehs, b)
            outs.append(x)
        if self.add_downsample:
            x = conv3x3(x, self.downsamplers[0].conv, stride=2)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, channels, temb_channels, heads, cross_attention_dim):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(channels, channels, temb_channels, eps=1e-5) for _ in range(2)])
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(heads, channels // heads, channels, 1, cross_attention_dim)])

    def forward(self, x, silu_temb, ehs, b):
        x = self.resnets[0](x, silu_temb, b)
        x = self.attentions[0](x, ehs, b)
        return self.resnets[1](x, silu_temb, b)


class UpBlock(nn.Module):
    """CrossAttnUpBlockSpatioTemporal (cross=True) / UpBlockSpatioTemporal"""

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, heads, cross_attention_dim, cross, add_upsample):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            res.append(SpatioTemporalResBlock(rin + skip, out_channels, temb_channels, eps=1e-5))
        self.resnets = nn.ModuleList(res)
        if cross:
            self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(heads, out_channels // heads, out_channels, 1, cross_attention_dim)
                                             for _ in range(num_layers)])
        self.has_cross = cross
        if add_upsample:
            self.upsamplers = nn.ModuleList([_Sampler2D(out_channels)])
        self.add_upsample = add_upsample

    def forward(self, x, skips, silu_temb, ehs, b):
        for i, res in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=-1)                                # channel concat of channels-last rows (memory plumbing)
            x = res(x, silu_temb, b)
            if self.has_cross:
                x = self.attentions[i](x, ehs, b)
        if self.add_upsample:
            x = conv3x3(x, self.upsamplers[0].conv, upsample=True)
        return x


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNetSpatioTemporalConditionModel(nn.Module):
    """diffusers' SVD UNet (constructor arguments and defaults of the img2vid checkpoint)."""

    def __init__(self, in_channels=8, out_channels=4, block_out_channels: Sequence[int] = (320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024, num_attention_heads: Sequence[int] = (5, 10, 20, 20),
                 num_frames=25):
        super().__init__()
        boc, heads = list(block_out_channels), list(num_attention_heads)
        temb = boc[0] * 4
        self.block_out_channels, self.addition_time_embed_dim = boc, addition_time_embed_dim
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)
        n = len(boc)
        downs, out_ch = [], boc[0]
        for i in range(n):
            in_ch, out_ch = out_ch, boc[i]
            downs.append(DownBlock(in_ch, out_ch, temb, layers_per_block, heads[i], cross_attention_dim, cross=i < n - 1, add_downsample=i < n - 1))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb, heads[-1], cross_attention_dim)
        rboc, rheads = boc[::-1], heads[::-1]
        ups, out_ch = [], rboc[0]
        for i in range(n):
            prev, out_ch = out_ch, rboc[i]
            in_ch = rboc[min(i + 1, n - 1)]
            ups.append(UpBlock(in_ch, prev, out_ch, temb, layers_per_block + 1, rheads[i], cross_attention_dim, cross=i > 0, add_upsample=i < n - 1))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    # ---- diffusers attention-processor protocol, as used by src/projects/svd/module.py:147-165
    @property
    def attn_processors(self) -> Dict[str, object]:
        return {f"{name}.processor": m.processor for name, m in self.named_modules() if isinstance(m, Attention)}

    def set_attn_processor(self, processor):
        mods = {f"{name}.processor": m for name, m in self.named_modules() if isinstance(m, Attention)}
        if isinstance(processor, dict):
            if len(processor) != len(mods):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the number of attention layers: {len(mods)}.")
            for name, m in mods.items():
                m.set_processor(processor[name])
                if isinstance(processor[name], nn.Module):
                    m.add_module("processor", processor[name])                      # so `...attn2.processor.to_q_ip.0.weight` is a state-dict key
        else:
            for m in mods.values():
                m.set_processor(processor)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict: bool = True):
        """sample [B, F, C, H, W]; timestep scalar / [B]; encoder_hidden_states [B, 1, D] tensor or TupleTensor(image [B,1,D], motion [B,25,D]);
        added_time_ids [B, 3] -> sample [B, F, out, H, W]"""
        B, Fr, C, H, W = sample.shape
        dev = sample.device
        t = torch.as_tensor(timestep, dtype=torch.float32, device=dev).reshape(-1).expand(B).contiguous()
        emb = self.time_embedding(ops.timestep_embedding(t, self.block_out_channels[0]))
        tid = added_time_ids.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if tid.numel() * self.addition_time_embed_dim != B * self.add_embedding.linear_1.in_features:
            raise ValueError("Model expects an added time embedding vector of length %d" % self.add_embedding.linear_1.in_features)
        aug = self.add_embedding(ops.timestep_embedding(tid, self.addition_time_embed_dim).view(B, -1))
        emb = ops.add(emb, aug)
        silu_temb = ops.silu(emb).repeat_interleave(Fr, dim=0)                      # every resnet starts from SiLU(temb)
        if getattr(self, "_temb_linears", None) is None:
            self._temb_linears = [m.time_emb_proj for m in self.modules() if isinstance(m, (ResnetBlock2D, TemporalResnetBlock))]
        silu_temb = TembBank(silu_temb.contiguous(), self, self._temb_linears)
        ehs = encoder_hidden_states.repeat_interleave(Fr, dim=0)                    # TupleTensor forwards this to both members (pipeline.py:39-40)
        ehs = ehs.to(torch.bfloat16)
        x = sample.to(torch.bfloat16).permute(0, 1, 3, 4, 2).reshape(B * Fr, H, W, C).contiguous()
        x = conv3x3(x, self.conv_in)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, silu_temb, ehs, B)
            skips.extend(outs)
        x = self.mid_block(x, silu_temb, ehs, B)
        for blk in self.up_blocks:
            x = blk(x, skips, silu_temb, ehs, B)
        N, Hh, Ww, Cc = x.shape
        x = ops.groupnorm(x.view(N, Hh * Ww, Cc), self.conv_norm_out.weight, self.conv_norm_out.bias, 32, self.conv_norm_out.eps, silu=True).view(N, Hh, Ww, Cc)
        x = conv3x3(x, self.conv_out)
        out = x.view(B, Fr, Hh, Ww, -1).permute(0, 1, 4, 2, 3).contiguous()
        return UNetOutput(out) if return_dict else (out,)


# ------------------------------------------------------------------------------------------------ Euler / CFG step
class EulerDiscreteScheduler:
    """Karras-sigma Euler scheduler as the SVD pipeline drives it (v-prediction; c_skip / c_out / c_noise as in
    src/projects/svd/module.py:92-98).  Tables on the host, the update is one kernel (`mrag_cfg_euler_step_bf16`)."""

    def __init__(self, sigma_min: float = 0.002, sigma_max: float = 700.0, rho: float = 7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho
        self.sigmas = None

    def set_timesteps(self, num_inference_steps: int):
        import numpy as np
        ramp = np.linspace(0, 1, num_inference_steps)
        mi, ma = self.sigma_min ** (1 / self.rho), self.sigma_max ** (1 / self.rho)
        s = (ma + ramp * (mi - ma)) ** self.rho
        self.sigmas = np.concatenate([s, [0.0]])
        self.timesteps = 0.25 * np.log(s)                                          # c_noise
        self.init_noise_sigma = float((s.max() ** 2 + 1) ** 0.5)

    def input_scale(self, i: int) -> float:
        return float(1.0 / math.sqrt(self.sigmas[i] ** 2 + 1))

    def coeffs(self, i: int):
        s, sn = float(self.sigmas[i]), float(self.sigmas[i + 1])
        c_skip, c_out = 1.0 / (s * s + 1), -s / math.sqrt(s * s + 1)
        r = (sn - s) / s
        return 1.0 + (1.0 - c_skip) * r, -c_out * r

    def step_(self, v_pred, latents, i: int, guidance):
        c_x, c_v = self.coeffs(i)
        return ops.cfg_euler_step_(v_pred, latents, guidance, c_x, c_v)
