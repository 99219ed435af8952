"""ORACLE (test infrastructure only -- never imported by the product path).

CPU / fp32 restatement of DynamiCrafter's motion-injected cross attention
(src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/attention.py:171-223, `CrossAttention.efficient_forward`).
Pinned against the reference's own class by tests/golden/dc_cross_attention.npz (oracle/gen_golden.py G7).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _heads(t: torch.Tensor, h: int) -> torch.Tensor:
    b, l, w = t.shape
    return t.view(b, l, h, w // h).transpose(1, 2)


def _sdpa(q, k, v):
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(s, dim=-1), v)


def cross_attention(sd: SD, x: torch.Tensor, context: Optional[dict], heads: int, image_scale: float = 1.0, action_scale: float = 1.0):
    """attention.py:171-223: out = SDPA(q, k_text, v_text) [+ s_img SDPA(q, k_ip, v_ip)]
    [+ s_act SDPA(to_q_a(merge(out)), k_a, v_a)]; to_out.  context None -> spatial self-attention."""
    q = F.linear(x, sd["to_q.weight"])                                             # :177
    src = x if context is None else context["prompt"]                             # :178-183
    k = F.linear(src, sd["to_k.weight"])
    v = F.linear(src, sd["to_v.weight"])
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)                # :185-188
    out = _sdpa(q, k, v)                                                          # :189
    if "to_k_ip.weight" in sd and context is not None:                            # :191-204
        k_ip = _heads(F.linear(context["image"], sd["to_k_ip.weight"]), heads)
        v_ip = _heads(F.linear(context["image"], sd["to_v_ip.weight"]), heads)
        out = out + image_scale * _sdpa(q, k_ip, v_ip)
    if "to_q_a.weight" in sd and context is not None:                             # :206-220
        merged = out.transpose(1, 2).reshape(x.shape[0], -1, heads * out.shape[-1])
        q_a = _heads(F.linear(merged, sd["to_q_a.weight"]), heads)
        k_a = _heads(F.linear(context["action"], sd["to_k_a.weight"]), heads)
        v_a = _heads(F.linear(context["action"], sd["to_v_a.weight"]), heads)
        out = out + action_scale * _sdpa(q_a, k_a, v_a)
    out = out.transpose(1, 2).reshape(x.shape[0], -1, heads * out.shape[-1])      # :222
    return F.linear(out, sd["to_out.0.weight"], sd["to_out.0.bias"])              # :223


# ----------------------------------------------------------------------------------------------------------------------
# UNet blocks (restated on the reference's own NCHW / NCTHW layout; eval mode, dropout = identity).
# Pinned by tests/golden/dc_*.npz generated from the reference classes (oracle/gen_golden.py G8-G12).
# ----------------------------------------------------------------------------------------------------------------------
import numpy as np


def sub(sd: SD, prefix: str) -> SD:
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def _ln(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def feed_forward_geglu(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """attention.py:448-475: GEGLU proj -> x * gelu(gate) -> Linear"""
    a, gate = F.linear(x, sd["net.0.proj.weight"], sd["net.0.proj.bias"]).chunk(2, dim=-1)
    return F.linear(a * F.gelu(gate), sd["net.2.weight"], sd["net.2.bias"])


def basic_transformer_block(sd: SD, x: torch.Tensor, context: Optional[dict], heads: int, image_scale=1.0, action_scale=1.0):
    """attention.py:262-266: x = attn1(norm1(x)) + x; x = attn2(norm2(x), context) + x; x = ff(norm3(x)) + x"""
    x = cross_attention(sub(sd, "attn1"), _ln(x, sd, "norm1"), None, heads) + x
    x = cross_attention(sub(sd, "attn2"), _ln(x, sd, "norm2"), context, heads, image_scale, action_scale) + x
    return feed_forward_geglu(sub(sd, "ff"), _ln(x, sd, "norm3")) + x


def spatial_transformer(sd: SD, x: torch.Tensor, context: dict, heads: int, depth: int = 1) -> torch.Tensor:
    """attention.py:316-332 (use_linear=True): GroupNorm(32, eps 1e-6) -> b (h w) c -> proj_in -> blocks -> proj_out -> + x_in"""
    b, c, h, w = x.shape
    y = F.group_norm(x, 32, sd["norm.weight"], sd["norm.bias"], 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = F.linear(y, sd["proj_in.weight"], sd["proj_in.bias"])
    for i in range(depth):
        y = basic_transformer_block(sub(sd, f"transformer_blocks.{i}"), y, context, heads)
    y = F.linear(y, sd["proj_out.weight"], sd["proj_out.bias"])
    return y.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


def temporal_transformer(sd: SD, x: torch.Tensor, heads: int, depth: int = 1) -> torch.Tensor:
    """attention.py:395-445 (use_linear, only_self_att, no mask): GroupNorm over (c/g, t, h, w) -> (b h w) t c -> proj_in ->
    blocks with TWO temporal self-attentions (context None for attn1 and attn2) -> proj_out -> + x_in"""
    b, c, t, h, w = x.shape
    y = F.group_norm(x, 32, sd["norm.weight"], sd["norm.bias"], 1e-6)
    y = y.permute(0, 3, 4, 2, 1).reshape(b * h * w, t, c)
    w2 = lambda w_: w_.squeeze(-1) if w_.dim() == 3 else w_          # init_attn is built without use_linear: Conv1d k=1 == Linear
    y = F.linear(y, w2(sd["proj_in.weight"]), sd["proj_in.bias"])
    for i in range(depth):
        y = basic_transformer_block(sub(sd, f"transformer_blocks.{i}"), y, None, heads)
    y = F.linear(y, w2(sd["proj_out.weight"]), sd["proj_out.bias"])
    return y.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2) + x


def temporal_conv_block(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """openaimodel3d.py:274-281: identity + conv4(conv3(conv2(conv1(x)))), each GroupNorm(32) -> SiLU -> Conv3d (3,1,1)"""
    y = x
    for i, idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
        y = F.silu(F.group_norm(y, 32, sd[f"conv{i}.0.weight"], sd[f"conv{i}.0.bias"], 1e-5))
        y = F.conv3d(y, sd[f"conv{i}.{idx}.weight"], sd[f"conv{i}.{idx}.bias"], padding=(1, 0, 0))
    return x + y


def res_block(sd: SD, x: torch.Tensor, emb: torch.Tensor, batch_size: Optional[int] = None) -> torch.Tensor:
    """openaimodel3d.py:211-237 (no up/down, no scale-shift norm)"""
    h = F.conv2d(F.silu(F.group_norm(x.float(), 32, sd["in_layers.0.weight"], sd["in_layers.0.bias"], 1e-5)), sd["in_layers.2.weight"],
                 sd["in_layers.2.bias"], padding=1)
    emb_out = F.linear(F.silu(emb), sd["emb_layers.1.weight"], sd["emb_layers.1.bias"])
    h = h + emb_out[..., None, None]
    h = F.conv2d(F.silu(F.group_norm(h, 32, sd["out_layers.0.weight"], sd["out_layers.0.bias"], 1e-5)), sd["out_layers.3.weight"],
                 sd["out_layers.3.bias"], padding=1)
    skip = x if "skip_connection.weight" not in sd else F.conv2d(x, sd["skip_connection.weight"], sd["skip_connection.bias"])
    h = skip + h
    if "temopral_conv.conv1.0.weight" in sd and batch_size:
        bt, c, hh, ww = h.shape
        h5 = h.view(batch_size, bt // batch_size, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sub(sd, "temopral_conv"), h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
    return h


def downsample(sd: SD, x):
    """openaimodel3d.py:52-77 (use_conv): Conv2d 3x3 stride 2 pad 1"""
    return F.conv2d(x, sd["op.weight"], sd["op.bias"], stride=2, padding=1)


def upsample(sd: SD, x):
    """openaimodel3d.py:80-107: nearest x2 then Conv2d 3x3 pad 1"""
    return F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), sd["conv.weight"], sd["conv.bias"], padding=1)


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """utils_diffusion.py:8-29: cat([cos, sin])"""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


class UNetSpec:
    def __init__(self, in_channels=8, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
                 channel_mult=(1, 2, 4, 4), num_head_channels=64, context_dim=1024, temporal_length=16, init_attn_heads=8):
        self.in_channels, self.out_channels, self.model_channels = in_channels, out_channels, model_channels
        self.attention_resolutions, self.num_res_blocks, self.channel_mult = tuple(attention_resolutions), num_res_blocks, tuple(channel_mult)
        self.num_head_channels, self.context_dim, self.temporal_length, self.init_attn_heads = num_head_channels, context_dim, temporal_length, init_attn_heads

    def block_plan(self):
        """layer kinds per block, mirroring UNetModel.__init__ (openaimodel3d.py:395-577): lists of ('res'|'st'|'tt'|'down'|'up', channels)"""
        mc = self.model_channels
        inp, ch, ds, chans = [[("conv_in", mc)]], mc, 1, [mc]
        for level, mult in enumerate(self.channel_mult):
            for _ in range(self.num_res_blocks):
                layers = [("res", mult * mc)]
                ch = mult * mc
                if ds in self.attention_resolutions:
                    layers += [("st", ch), ("tt", ch)]
                inp.append(layers); chans.append(ch)
            if level != len(self.channel_mult) - 1:
                inp.append([("down", ch)]); chans.append(ch); ds *= 2
        mid = [("res", ch), ("st", ch), ("tt", ch), ("res", ch)]
        out = []
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(self.num_res_blocks + 1):
                chans.pop()
                layers = [("res", mult * mc)]
                ch = mult * mc
                if ds in self.attention_resolutions:
                    layers += [("st", ch), ("tt", ch)]
                if level and i == self.num_res_blocks:
                    layers.append(("up", ch)); ds //= 2
                out.append(layers)
        return inp, mid, out


def _run_layers(sd: SD, prefix: str, layers, h, emb, ctx, b, spec: UNetSpec):
    for li, (kind, ch) in enumerate(layers):
        s = sub(sd, f"{prefix}.{li}")
        if kind == "conv_in":
            h = F.conv2d(h, s["weight"], s["bias"], padding=1)
        elif kind == "res":
            h = res_block(s, h, emb, batch_size=b)
        elif kind == "st":
            h = spatial_transformer(s, h, ctx, ch // spec.num_head_channels)
        elif kind == "tt":
            bt, c, hh, ww = h.shape
            h5 = h.view(b, bt // b, c, hh, ww).permute(0, 2, 1, 3, 4)                     # (b f) c h w -> b c f h w  (:44-46)
            h5 = temporal_transformer(s, h5, ch // spec.num_head_channels)
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        elif kind == "down":
            h = downsample(s, h)
        elif kind == "up":
            h = upsample(s, h)
    return h


def unet_forward(sd: SD, spec: UNetSpec, x: torch.Tensor, timesteps: torch.Tensor, context: dict, fs: torch.Tensor) -> torch.Tensor:
    """UNetModel.forward (openaimodel3d.py:579-635), image + action cross-attention, fps conditioning, init temporal attention."""
    b, _, t, _, _ = x.shape
    mc = spec.model_channels
    lin = lambda v, p: F.linear(v, sd[p + ".weight"], sd[p + ".bias"])
    emb = lin(F.silu(lin(timestep_embedding(timesteps, mc), "time_embed.0")), "time_embed.2")
    ctx = {"image": context["image"].reshape(b * t, -1, context["image"].shape[-1]),                       # 'b (t l) c -> (b t) l c'
           "prompt": context["prompt"].repeat_interleave(t, dim=0), "action": context["action"].repeat_interleave(t, dim=0)}
    emb = emb.repeat_interleave(t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])                       # b c t h w -> (b t) c h w
    fs_embed = lin(F.silu(lin(timestep_embedding(fs, mc), "fps_embedding.0")), "fps_embedding.2")
    emb = emb + fs_embed.repeat_interleave(t, dim=0)
    inp, mid, out = spec.block_plan()
    hs = []
    for i, layers in enumerate(inp):
        h = _run_layers(sd, f"input_blocks.{i}", layers, h, emb, ctx, b, spec)
        if i == 0:                                                                                       # init_attn (:611-612)
            bt, c, hh, ww = h.shape
            h5 = h.view(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sub(sd, "init_attn.0"), h5, spec.init_attn_heads)
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        hs.append(h)
    h = _run_layers(sd, "middle_block", mid, h, emb, ctx, b, spec)
    for i, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, f"output_blocks.{i}", layers, h, emb, ctx, b, spec)
    y = F.conv2d(F.silu(F.group_norm(h, 32, sd["out.0.weight"], sd["out.0.bias"], 1e-5)), sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return y.view(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


# ---- diffusion schedule + DDIM sampler (models/ddpm3d.py:134-198,535-541; utils_diffusion.py:31-77,79-92,113-146; samplers/ddim.py)
def dc_schedule(timesteps=1000, linear_start=0.00085, linear_end=0.012, zero_snr=True):
    betas = (np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2)
    if zero_snr:
        abs_ = np.sqrt(np.cumprod(1.0 - betas))
        a0, aT = abs_[0].copy(), abs_[-1].copy()
        abs_ = (abs_ - aT) * (a0 / (a0 - aT))
        ab = abs_ ** 2
        alphas = np.concatenate([ab[0:1], ab[1:] / ab[:-1]])
        betas = 1 - alphas
    return np.cumprod(1.0 - betas)


def dc_ddim_timesteps(num_steps: int, n_train: int = 1000) -> np.ndarray:
    """'uniform': range(0, n_train, n_train // S) + 1   (S = 30 -> 31 entries)"""
    return np.asarray(list(range(0, n_train, n_train // num_steps))) + 1


def dc_ddim_params(alphacums: np.ndarray, ts: np.ndarray, eta: float):
    ac = torch.tensor(alphacums, dtype=torch.float32)
    alphas = ac[ts]
    alphas_prev = torch.asarray([ac[0].item()] + ac[ts[:-1]].tolist())
    sigmas = eta * torch.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def dc_scale_arr(n_train=1000, base_scale=0.3, turning_step=400) -> np.ndarray:
    return np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(n_train, base_scale))).astype(np.float32)


def dc_ddim_step(v_cond, v_uncond, x, noise, guidance, ac32, t, sigma, a_t, a_prev, scale_t, scale_prev):
    """samplers/ddim.py:244-296 (v-parameterisation, dynamic rescale); ac32 = fp32 alphas_cumprod table, t = model timestep"""
    v = v_uncond + guidance * (v_cond - v_uncond)
    sa, sb = torch.sqrt(ac32[t]), torch.sqrt(1.0 - ac32[t])
    e_t = sa * v + sb * x
    pred_x0 = (sa * x - sb * v) * (scale_prev / scale_t)
    dir_xt = torch.sqrt(1.0 - a_prev - sigma ** 2) * e_t
    return torch.sqrt(a_prev) * pred_x0 + dir_xt + sigma * noise, pred_x0


# ------------------------------------------------------------------------------------------------ generation glue (row a20)
def image_guided_synthesis_ref(unet_sd: SD, spec: "UNetSpec", proj_sd: SD, embedder, text, first_stage, condition_transformer, image: torch.Tensor,
                               prompts, ref_videos: torch.Tensor, num_frames: int, ddim_steps: int, guidance: float, fs: int, x_T: torch.Tensor,
                               noises, proj_heads: int = 2, proj_depth: int = 2) -> torch.Tensor:
    """scripts/evaluation/inference.py:174-305 + pipelines/pipeline.py:64-115 restated on the oracle's own UNet / DDIM restatement
    (condition_transformer branch, uncond_type 'empty_seq', conditioning_key 'hybrid', eta = 1).  Returns frames [b, t, c, h, w]."""
    from . import cama_ref
    b = image.shape[0]
    videos = image[:, :, None].expand(-1, -1, num_frames, -1, -1)                                     # pipeline.py:95
    img = videos[:, :, 0]                                                                              # inference.py:189
    proj = lambda e: cama_ref.resampler(proj_sd, e, proj_heads, proj_depth)
    cond = {"image": proj(embedder(img)),                                                               # :190-192
            "action": condition_transformer.predict({"ref_videos": ref_videos, "video": img[:, None].expand(-1, ref_videos.size(2), -1, -1, -1)}),   # :219-223
            "prompt": text(prompts)}                                                                    # :225-226
    x = videos.permute(0, 2, 1, 3, 4).reshape(b * num_frames, *videos.shape[1:2], *videos.shape[3:])    # get_latent_z :165-170
    z = first_stage.encode_first_stage(x)
    z = z.reshape(b, num_frames, *z.shape[1:]).permute(0, 2, 1, 3, 4)
    img_cat = z[:, :, :1].expand(-1, -1, z.shape[2], -1, -1)                                            # :235-237
    uc = {"prompt": text(b * [""]), "image": proj(embedder(torch.zeros_like(img))),                     # :239-251
          "action": condition_transformer.encode_vision(torch.zeros_like(ref_videos[:, 0:1]))[:, 0]}     # :257-259
    ac = dc_schedule()
    ts = dc_ddim_timesteps(ddim_steps)
    sig, al, alp = dc_ddim_params(ac, ts, 1.0)
    ac32 = torch.tensor(ac, dtype=torch.float32)
    sc_t = torch.from_numpy(dc_scale_arr())[ts]
    sc_prev = torch.cat([sc_t[0:1], sc_t[:-1]])
    xs = x_T.clone()
    fs_t = torch.tensor([fs] * (2 * b))
    for i in range(len(ts)):                                                                            # ddim.py:166-200, CFG batch cond FIRST :219-237
        index = len(ts) - 1 - i
        t = int(ts[index])
        xin = torch.cat([torch.cat([xs, img_cat], dim=1)] * 2, dim=0)
        ctx = {k: torch.cat([cond[k], uc[k]], dim=0) for k in cond}
        v = unet_forward(unet_sd, spec, xin, torch.full((2 * b,), t), ctx, fs_t)
        xs, _ = dc_ddim_step(v[:b], v[b:], xs, noises[i], guidance, ac32, t, sig[index], al[index], alp[index], sc_t[index], sc_prev[index])
    out = first_stage.decode_first_stage(xs)                                                            # :301  b c t h w
    return out.permute(0, 2, 1, 3, 4)                                                                   # pipeline.py:115
