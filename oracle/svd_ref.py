"""CPU restatement (fp32, plain torch) of the SVD UNet denoise step that MotionRAG's SVD backbone runs.  TEST INFRASTRUCTURE
ONLY: imported by tests/, `__graft_entry__.smoke()` and bench.py's cpu_baseline leg, never by the product path.

PARITY UNPINNED.  The arithmetic lives in the third-party package `diffusers==0.32.2` (requirements.txt:10 of the reference;
not vendored, not installed here): `UNetSpatioTemporalConditionModel`, `SpatioTemporalResBlock`, `TemporalResnetBlock`,
`AlphaBlender`, `TransformerSpatioTemporalModel`, `BasicTransformerBlock`, `TemporalBasicTransformerBlock`,
`EulerDiscreteScheduler`, `StableVideoDiffusionPipeline`.  This file restates that package's published algorithm
 in NCHW layout with `torch.nn.functional`, driven by a diffusers-keyed state dict, and is anchored
on the reference's own call sites:
  * src/projects/svd/module.py:38-47 (model construction), :92-98 (c_skip / c_out / c_noise), :145-165 (adapter install);
  * src/projects/svd/pipelines/pipeline.py:25-57 (TupleTensor), :113-119 (`_encode_image`), :147-160 (`__call__`);
  * src/projects/condition/attn_processor.py:18-141 (`APAdapterAttnProcessor2_0`, restated in `adapter_cross_attention`).
The reference holds no test or golden vector at this boundary.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`: [cos | sin]"""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class SD:
    """prefix view over a state dict"""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        self.sd, self.p = sd, prefix

    def __call__(self, name: str) -> torch.Tensor:
        return self.sd[self.p + name].float()

    def has(self, name: str) -> bool:
        return (self.p + name) in self.sd

    def sub(self, name: str) -> "SD":
        return SD(self.sd, self.p + name + ".")


def linear(w: SD, x, bias=True):
    return F.linear(x, w("weight"), w("bias") if bias and w.has("bias") else None)


def time_mlp(w: SD, x):
    return linear(w.sub("linear_2"), F.silu(linear(w.sub("linear_1"), x)))


def attention(q, k, v, heads):
    B, L, C = q.shape
    sp = lambda t: t.view(t.shape[0], t.shape[1], heads, C // heads).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    return o.transpose(1, 2).reshape(B, L, C)


def plain_attention(w: SD, x, ctx, heads):
    ctx = x if ctx is None else ctx
    o = attention(linear(w.sub("to_q"), x), linear(w.sub("to_k"), ctx), linear(w.sub("to_v"), ctx), heads)
    return linear(w.sub("to_out.0"), o)


def adapter_cross_attention(w: SD, x, image_ctx, ip_ctx, heads, scale=1.0):
    """attn_processor.py:18-141: cross-attention to the image embedding, then the motion branch whose query is
    `to_q_ip(attention output)` (:93-100), keys / values repeated '(b r)' over the frames of a sample (:109-111), added with `scale`
    (:127; skipped when 0, :93-99), then `to_out` (:129-131).
    PINNED to the reference's own class: tests/golden/svd_attn_processor.npz (oracle/gen_golden_attn_processor.py)."""
    o = attention(linear(w.sub("to_q"), x), linear(w.sub("to_k"), image_ctx), linear(w.sub("to_v"), image_ctx), heads)
    if ip_ctx is not None and scale != 0:
        p = w.sub("processor")
        ipq = F.linear(o, p("to_q_ip.0.weight"))
        ipk, ipv = F.linear(ip_ctx, p("to_k_ip.0.weight")), F.linear(ip_ctx, p("to_v_ip.0.weight"))
        r = x.size(0) // ip_ctx.size(0)
        o = o + scale * attention(ipq, ipk.repeat_interleave(r, dim=0), ipv.repeat_interleave(r, dim=0), heads)
    return linear(w.sub("to_out.0"), o)


def adapter_processor_call(w: SD, hidden, image_ctx, ip_ctx, heads, scale=1.0, residual_connection=False, rescale_output_factor=1.0):
    """the rest of `APAdapterAttnProcessor2_0.__call__`: the 4-D `[b, c, h, w]` form (:48-50, :133-134), `residual_connection` (:136-137)
    and `rescale_output_factor` (:139)"""
    x = hidden
    if hidden.ndim == 4:
        b, c, hh, ww = hidden.shape
        x = hidden.view(b, c, hh * ww).transpose(1, 2)
    out = adapter_cross_attention(w, x, image_ctx, ip_ctx, heads, scale)
    if hidden.ndim == 4:
        out = out.transpose(-1, -2).reshape(b, c, hh, ww)
    if residual_connection:
        out = out + hidden
    return out / rescale_output_factor


def feed_forward(w: SD, x):
    h, gate = linear(w.sub("net.0.proj"), x).chunk(2, dim=-1)
    return linear(w.sub("net.2"), h * F.gelu(gate))


def layer_norm(w: SD, x):
    return F.layer_norm(x, (x.shape[-1],), w("weight"), w("bias"), 1e-5)


def alpha_blend(w: SD, x_spatial, x_temporal):
    """AlphaBlender 'learned_with_images' with image_only_indicator == 0 everywhere: alpha = sigmoid(mix_factor)"""
    a = torch.sigmoid(w("mix_factor"))
    return a * x_spatial + (1.0 - a) * x_temporal


def resnet2d(w: SD, x, temb, eps):
    h = F.conv2d(F.silu(F.group_norm(x, 32, w("norm1.weight"), w("norm1.bias"), eps)), w("conv1.weight"), w("conv1.bias"), padding=1)
    h = h + linear(w.sub("time_emb_proj"), F.silu(temb))[:, :, None, None]
    h = F.conv2d(F.silu(F.group_norm(h, 32, w("norm2.weight"), w("norm2.bias"), eps)), w("conv2.weight"), w("conv2.bias"), padding=1)
    if w.has("conv_shortcut.weight"):
        x = F.conv2d(x, w("conv_shortcut.weight"), w("conv_shortcut.bias"))
    return x + h


def temporal_resnet(w: SD, x, temb, eps):
    """x [b, c, f, h, w], temb [b, f, temb_ch]"""
    h = F.conv3d(F.silu(F.group_norm(x, 32, w("norm1.weight"), w("norm1.bias"), eps)), w("conv1.weight"), w("conv1.bias"), padding=(1, 0, 0))
    t = linear(w.sub("time_emb_proj"), F.silu(temb))                       # [b, f, c]
    h = h + t.permute(0, 2, 1)[:, :, :, None, None]
    h = F.conv3d(F.silu(F.group_norm(h, 32, w("norm2.weight"), w("norm2.bias"), eps)), w("conv2.weight"), w("conv2.bias"), padding=(1, 0, 0))
    return x + h


def st_resblock(w: SD, x, temb, frames, eps):
    """SpatioTemporalResBlock: x [(b f), c, h, w]"""
    h = resnet2d(w.sub("spatial_res_block"), x, temb, eps)
    bf, c, hh, ww = h.shape
    b = bf // frames
    h5 = h.view(b, frames, c, hh, ww).permute(0, 2, 1, 3, 4)
    t5 = temporal_resnet(w.sub("temporal_res_block"), h5, temb.view(b, frames, -1), eps)
    out = alpha_blend(w.sub("time_mixer"), h5, t5)
    return out.permute(0, 2, 1, 3, 4).reshape(bf, c, hh, ww)


def st_transformer(w: SD, x, image_ctx, ip_ctx, frames, heads, layers=1):
    """TransformerSpatioTemporalModel: x [(b f), c, h, w]; image_ctx [(b f), 1, D]; ip_ctx [(b f), 25, D] or None"""
    bf, c, hh, ww = x.shape
    b = bf // frames
    first = image_ctx.view(b, frames, -1, image_ctx.shape[-1])[:, 0]                                   # [b, 1, D]
    time_ctx = first[None].expand(hh * ww, b, 1, first.shape[-1]).reshape(hh * ww * b, 1, first.shape[-1])   # row n -> batch n % b (package order)
    h = F.group_norm(x, 32, w("norm.weight"), w("norm.bias"), 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(bf, hh * ww, c)
    h = linear(w.sub("proj_in"), h)
    idx = torch.arange(frames).repeat(b)
    emb = time_mlp(w.sub("time_pos_embed"), timestep_embedding(idx, c))[:, None, :]
    for i in range(layers):
        s, t = w.sub(f"transformer_blocks.{i}"), w.sub(f"temporal_transformer_blocks.{i}")
        h = h + plain_attention(s.sub("attn1"), layer_norm(s.sub("norm1"), h), None, heads)
        h = h + adapter_cross_attention(s.sub("attn2"), layer_norm(s.sub("norm2"), h), image_ctx, ip_ctx, heads)
        h = h + feed_forward(s.sub("ff"), layer_norm(s.sub("norm3"), h))
        m = h + emb
        L, C = m.shape[1], m.shape[2]
        m = m.view(b, frames, L, C).permute(0, 2, 1, 3).reshape(b * L, frames, C)
        m = m + feed_forward(t.sub("ff_in"), layer_norm(t.sub("norm_in"), m))
        m = m + plain_attention(t.sub("attn1"), layer_norm(t.sub("norm1"), m), None, heads)
        m = m + plain_attention(t.sub("attn2"), layer_norm(t.sub("norm2"), m), time_ctx, heads)
        m = m + feed_forward(t.sub("ff"), layer_norm(t.sub("norm3"), m))
        m = m.view(b, L, frames, C).permute(0, 2, 1, 3).reshape(bf, L, C)
        h = alpha_blend(w.sub("time_mixer"), h, m)
    h = linear(w.sub("proj_out"), h)
    return h.view(bf, hh, ww, c).permute(0, 3, 1, 2) + x


def unet_forward(sd: Dict[str, torch.Tensor], cfg: dict, sample, timestep, image_emb, added_time_ids, action_emb: Optional[torch.Tensor] = None):
    """sample [B, F, C, H, W]; image_emb [B, 1, D]; action_emb [B, 25, D] or None; returns [B, F, out, H, W]"""
    w = SD({k: v for k, v in sd.items()})
    boc: Sequence[int] = cfg["block_out_channels"]
    heads: Sequence[int] = cfg["num_attention_heads"]
    lpb = cfg.get("layers_per_block", 2)
    B, Fr = sample.shape[:2]
    sample = sample.float()
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    emb = time_mlp(w.sub("time_embedding"), timestep_embedding(t, boc[0]))
    add = timestep_embedding(added_time_ids.flatten().float(), cfg.get("addition_time_embed_dim", 256)).reshape(B, -1)
    emb = emb + time_mlp(w.sub("add_embedding"), add)
    x = sample.flatten(0, 1)
    emb = emb.repeat_interleave(Fr, dim=0)
    img = image_emb.float().repeat_interleave(Fr, dim=0)
    ip = action_emb.float().repeat_interleave(Fr, dim=0) if action_emb is not None else None
    x = F.conv2d(x, w("conv_in.weight"), w("conv_in.bias"), padding=1)
    skips = [x]
    n = len(boc)
    for i in range(n):
        blk = w.sub(f"down_blocks.{i}")
        cross = i < n - 1
        for j in range(lpb):
            x = st_resblock(blk.sub(f"resnets.{j}"), x, emb, Fr, 1e-6 if cross else 1e-5)
            if cross:
                x = st_transformer(blk.sub(f"attentions.{j}"), x, img, ip, Fr, heads[i])
            skips.append(x)
        if i < n - 1:
            x = F.conv2d(x, blk("downsamplers.0.conv.weight"), blk("downsamplers.0.conv.bias"), stride=2, padding=1)
            skips.append(x)
    mid = w.sub("mid_block")
    x = st_resblock(mid.sub("resnets.0"), x, emb, Fr, 1e-5)
    x = st_transformer(mid.sub("attentions.0"), x, img, ip, Fr, heads[-1])
    x = st_resblock(mid.sub("resnets.1"), x, emb, Fr, 1e-5)
    rheads = list(reversed(heads))
    for i in range(n):
        blk = w.sub(f"up_blocks.{i}")
        cross = i > 0
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = st_resblock(blk.sub(f"resnets.{j}"), x, emb, Fr, 1e-5)
            if cross:
                x = st_transformer(blk.sub(f"attentions.{j}"), x, img, ip, Fr, rheads[i])
        if i < n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, blk("upsamplers.0.conv.weight"), blk("upsamplers.0.conv.bias"), padding=1)
    x = F.silu(F.group_norm(x, 32, w("conv_norm_out.weight"), w("conv_norm_out.bias"), 1e-5))
    x = F.conv2d(x, w("conv_out.weight"), w("conv_out.bias"), padding=1)
    return x.view(B, Fr, *x.shape[1:])


# ------------------------------------------------------------------------------------------------ scheduler / CFG step
def karras_sigmas(num_steps: int, sigma_min: float = 0.002, sigma_max: float = 700.0, rho: float = 7.0) -> torch.Tensor:
    """EulerDiscreteScheduler(use_karras_sigmas=True) as configured for SVD (sigma_min 0.002, sigma_max 700), with the
    terminal 0 appended"""
    ramp = torch.linspace(0, 1, num_steps, dtype=torch.float64)
    mi, ma = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    s = (ma + ramp * (mi - ma)) ** rho
    return torch.cat([s, torch.zeros(1, dtype=torch.float64)])


def euler_cfg_step(v_uncond, v_cond, latents, sigma: float, sigma_next: float, guidance: torch.Tensor):
    """v-prediction Euler step with per-frame guidance: latents [B, F, C, H, W], guidance [F].
    denoised = c_skip x + c_out v with c_skip = 1/(sigma^2+1), c_out = -sigma/sqrt(sigma^2+1) (svd/module.py:92-98)."""
    v = v_uncond + guidance.view(1, -1, 1, 1, 1) * (v_cond - v_uncond)
    denoised = latents / (sigma ** 2 + 1) - v * sigma / math.sqrt(sigma ** 2 + 1)
    d = (latents - denoised) / sigma
    return latents + d * (sigma_next - sigma)


def euler_coeffs(sigma: float, sigma_next: float):
    """the same step as `x <- c_x x + c_v v` (what the GPU kernel takes)"""
    c_skip, c_out = 1.0 / (sigma ** 2 + 1), -sigma / math.sqrt(sigma ** 2 + 1)
    r = (sigma_next - sigma) / sigma
    return 1.0 + (1.0 - c_skip) * r, -c_out * r
