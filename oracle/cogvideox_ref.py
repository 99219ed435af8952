"""ORACLE (test infrastructure only -- never imported by the product path).

CPU / fp32 restatement of the motion-injected CogVideoX-5B-I2V denoising step.

* `adapter_attn_processor` follows the in-tree reference `APAdapterCogVideoXAttnProcessor2_0.__call__`
  (src/projects/condition/attn_processor.py:176-283) line by line.
* Everything around it (CogVideoXBlock, CogVideoXLayerNormZero, patch embed, AdaLayerNorm tail,
  get_3d_rotary_pos_embed / apply_rotary_emb, CogVideoXDDIMScheduler) lives in the third-party package
  diffusers==0.32.2 (requirements.txt:10), which is NOT vendored in /root/reference and not installed
  here.  Those parts are restated from that package's published algorithm and
  anchored on the reference's call sites (src/projects/cogvideox/module.py:23-48,125-130;
  pipeline.py:46-57,80-89).  PARITY UNPINNED for the diffusers-owned arithmetic: the reference holds
  no test or golden vector for it.  The shared `to_q_ip(out)` adapter arithmetic is additionally pinned
  through DynamiCrafter's importable CrossAttention (tests/golden/dc_cross_attention.npz).

State-dict keys are the diffusers / MotionRAG checkpoint keys.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .cama_ref import layer_norm, sub

SD = Dict[str, torch.Tensor]


class DiTConfig:
    """THUDM/CogVideoX-5b-I2V transformer config (public model card values)."""

    def __init__(self, num_layers=42, heads=48, head_dim=64, in_channels=32, out_channels=16, time_embed_dim=512,
                 text_embed_dim=4096, max_text_len=226, patch=2, ip_dim=1024, norm_eps=1e-5, qk_eps=1e-6,
                 frames=13, height=60, width=90, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0):
        self.spatial_interpolation_scale, self.temporal_interpolation_scale = spatial_interpolation_scale, temporal_interpolation_scale
        self.num_layers, self.heads, self.head_dim = num_layers, heads, head_dim
        self.dim = heads * head_dim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.time_embed_dim, self.text_embed_dim, self.max_text_len = time_embed_dim, text_embed_dim, max_text_len
        self.patch, self.ip_dim, self.norm_eps, self.qk_eps = patch, ip_dim, norm_eps, qk_eps
        self.frames, self.height, self.width = frames, height, width

    @property
    def video_tokens(self):
        return self.frames * (self.height // self.patch) * (self.width // self.patch)


# ----------------------------------------------------------------------------------------------
# rotary embedding (diffusers get_3d_rotary_pos_embed / apply_rotary_emb, use_real=True)
# ----------------------------------------------------------------------------------------------
def rope_1d(dim: int, pos: torch.Tensor, theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    f = torch.outer(pos.float(), freqs)
    return f.cos().repeat_interleave(2, dim=1), f.sin().repeat_interleave(2, dim=1)


def rope_3d(head_dim: int, t: int, h: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """head-dim split t = d/4, h = w = 3d/8; positions arange (crop region == full grid); -> [t*h*w, d] fp32."""
    dt, dh, dw = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3
    ct, st = rope_1d(dt, torch.arange(t))
    ch, sh = rope_1d(dh, torch.arange(h))
    cw, sw = rope_1d(dw, torch.arange(w))

    def bc(a_t, a_h, a_w):
        a = torch.cat([a_t[:, None, None, :].expand(t, h, w, dt), a_h[None, :, None, :].expand(t, h, w, dh),
                       a_w[None, None, :, :].expand(t, h, w, dw)], dim=-1)
        return a.reshape(t * h * w, head_dim).contiguous()

    return bc(ct, ch, cw), bc(st, sh, sw)


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [b, h, s, d]; x_rotated = stack([-x_imag, x_real]); out = x*cos + x_rotated*sin (fp32)."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


# ----------------------------------------------------------------------------------------------
# the motion-injection attention processor (in-tree reference)
# ----------------------------------------------------------------------------------------------
def _sdpa(q, k, v):
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(s, dim=-1), v)


def adapter_attn_processor(attn: SD, proc: SD, hidden: torch.Tensor, enc: torch.Tensor, rope: Optional[Tuple[torch.Tensor, torch.Tensor]],
                           ip_hidden: torch.Tensor, heads: int, scale: float = 1.0, qk_eps: float = 1e-6):
    """attn_processor.py:176-283.  attn: to_q/to_k/to_v/to_out.0 (+bias), norm_q/norm_k; proc:
    to_{q,k,v}_ip.0.weight.  Returns (hidden_states, encoder_hidden_states)."""
    text_len = enc.size(1)
    x = torch.cat([enc, hidden], dim=1)                                           # :199
    b = x.size(0)
    q = F.linear(x, attn["to_q.weight"], attn.get("to_q.bias"))                   # :209-211
    k = F.linear(x, attn["to_k.weight"], attn.get("to_k.bias"))
    v = F.linear(x, attn["to_v.weight"], attn.get("to_v.bias"))
    hd = k.shape[-1] // heads
    q, k, v = (t.view(b, -1, heads, hd).transpose(1, 2) for t in (q, k, v))       # :216-218
    if "norm_q.weight" in attn:                                                   # :220-223
        q = layer_norm(q, attn["norm_q.weight"], attn.get("norm_q.bias"), qk_eps)
        k = layer_norm(k, attn["norm_k.weight"], attn.get("norm_k.bias"), qk_eps)
    if rope is not None:                                                          # :226-231
        q = q.clone(); k = k.clone()
        q[:, :, text_len:] = apply_rotary_emb(q[:, :, text_len:], *rope)
        k[:, :, text_len:] = apply_rotary_emb(k[:, :, text_len:], *rope)
    o = _sdpa(q, k, v)                                                            # :233-235
    o = o.transpose(1, 2).reshape(b, -1, heads * hd)                              # :237
    if scale != 0:                                                                # :243-249
        ip_q = F.linear(o, proc["to_q_ip.0.weight"])                              # :250  (text tokens included)
        ip_k = F.linear(ip_hidden, proc["to_k_ip.0.weight"])
        ip_v = F.linear(ip_hidden, proc["to_v_ip.0.weight"])
        r = b // ip_hidden.size(0)                                                # :254-256  'b ... -> (b r) ...'
        ip_k = ip_k.repeat_interleave(r, dim=0)
        ip_v = ip_v.repeat_interleave(r, dim=0)
        ip_q, ip_k, ip_v = (t.view(b, -1, heads, hd).transpose(1, 2) for t in (ip_q, ip_k, ip_v))
        ip = _sdpa(ip_q, ip_k, ip_v)                                              # :264-266
        ip = ip.transpose(1, 2).reshape(b, -1, heads * hd)
        o = o + scale * ip                                                        # :273
    o = F.linear(o, attn["to_out.0.weight"], attn.get("to_out.0.bias"))           # :276 (dropout p=0 :278)
    return o[:, text_len:], o[:, :text_len]                                       # :280-283


# ----------------------------------------------------------------------------------------------
# diffusers-owned parts (restated, unpinned)
# ----------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0): cat([cos, sin])."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([a.cos(), a.sin()], dim=-1)


def layer_norm_zero(sd: SD, h, e, temb, eps):
    """CogVideoXLayerNormZero: linear(silu(temb)).chunk(6) = shift, scale, gate, enc_shift, enc_scale, enc_gate."""
    sh, sc, g, esh, esc, eg = F.linear(F.silu(temb), sd["linear.weight"], sd["linear.bias"]).chunk(6, dim=1)
    nh = layer_norm(h, sd["norm.weight"], sd["norm.bias"], eps) * (1 + sc)[:, None] + sh[:, None]
    ne = layer_norm(e, sd["norm.weight"], sd["norm.bias"], eps) * (1 + esc)[:, None] + esh[:, None]
    return nh, ne, g[:, None], eg[:, None]


def block(sd: SD, cfg: DiTConfig, h, e, temb, rope, ip_hidden, ip_scale=1.0):
    """CogVideoXBlock.forward with attn1.processor = adapter_attn_processor."""
    text_len = e.size(1)
    nh, ne, g, eg = layer_norm_zero(sub(sd, "norm1"), h, e, temb, cfg.norm_eps)
    ah, ae = adapter_attn_processor(sub(sd, "attn1"), sub(sd, "attn1.processor"), nh, ne, rope, ip_hidden, cfg.heads, ip_scale, cfg.qk_eps)
    h = h + g * ah
    e = e + eg * ae
    nh, ne, g, eg = layer_norm_zero(sub(sd, "norm2"), h, e, temb, cfg.norm_eps)
    x = torch.cat([ne, nh], dim=1)
    x = F.gelu(F.linear(x, sd["ff.net.0.proj.weight"], sd["ff.net.0.proj.bias"]), approximate="tanh")
    x = F.linear(x, sd["ff.net.2.weight"], sd["ff.net.2.bias"])
    h = h + g * x[:, text_len:]
    e = e + eg * x[:, :text_len]
    return h, e


def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """diffusers get_1d_sincos_pos_embed_from_grid: [M] positions -> [M, embed_dim] = [sin | cos], float64"""
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_3d(embed_dim: int, spatial_size, temporal_size: int, spatial_interpolation_scale: float = 1.0,
              temporal_interpolation_scale: float = 1.0) -> np.ndarray:
    """diffusers get_3d_sincos_pos_embed (0.32.2), step by step as the package writes it (the MAE recipe): [T, H*W, D]; spatial_size = (W, H)"""
    d_spatial, d_temporal = 3 * embed_dim // 4, embed_dim // 4
    grid_h = np.arange(spatial_size[1], dtype=np.float32) / spatial_interpolation_scale
    grid_w = np.arange(spatial_size[0], dtype=np.float32) / spatial_interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0)                       # w goes first
    grid = grid.reshape([2, 1, spatial_size[1], spatial_size[0]])
    emb_h = sincos_1d(d_spatial // 2, grid[0])                                 # get_2d_sincos_pos_embed_from_grid
    emb_w = sincos_1d(d_spatial // 2, grid[1])
    pos_spatial = np.concatenate([emb_h, emb_w], axis=1)                       # [H*W, 3D/4]
    grid_t = np.arange(temporal_size, dtype=np.float32) / temporal_interpolation_scale
    pos_temporal = sincos_1d(d_temporal, grid_t)                               # [T, D/4]
    pos_spatial = np.repeat(pos_spatial[np.newaxis, :, :], temporal_size, axis=0)
    pos_temporal = np.repeat(pos_temporal[:, np.newaxis, :], spatial_size[0] * spatial_size[1], axis=1)
    return np.concatenate([pos_temporal, pos_spatial], axis=-1)


def patch_embed_positions(sd: SD, cfg: DiTConfig, frames: int, height: int, width: int) -> torch.Tensor:
    """CogVideoXPatchEmbed.forward's choice of table (diffusers 0.32.2, use_learned_positional_embeddings): the learned parameter at the model's
    sample geometry; otherwise `_get_positional_embeddings` -- a freshly generated sin-cos table with ZERO text rows (never a slice of the learned
    one); a different resolution raises.  [1, max_text + frames*height*width, D]"""
    if (height, width) != (cfg.height // cfg.patch, cfg.width // cfg.patch):
        raise ValueError("learned positional embeddings: the model's own resolution only")
    if frames == cfg.frames:
        return sd["patch_embed.pos_embedding"]
    video = torch.from_numpy(sincos_3d(cfg.dim, (width, height), frames, cfg.spatial_interpolation_scale, cfg.temporal_interpolation_scale)).flatten(0, 1)
    joint = torch.zeros(1, cfg.max_text_len + video.shape[0], cfg.dim)
    joint[:, cfg.max_text_len:] = video.float()
    return joint


def dit_forward(sd: SD, cfg: DiTConfig, latents: torch.Tensor, text: torch.Tensor, timestep: torch.Tensor, rope, ip_hidden,
                ip_scale: float = 1.0) -> torch.Tensor:
    """CogVideoXTransformer3DModel.forward (5B-I2V flavour: rotary + learned positional embedding).
    latents [B, F, C_in, H, W]; text [B, L, text_dim]; returns [B, F, C_out, H, W]."""
    B, Fr, C, H, W = latents.shape
    p = cfg.patch
    temb = timestep_embedding(timestep, cfg.dim)
    temb = F.linear(F.silu(F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                    sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    e = F.linear(text, sd["patch_embed.text_proj.weight"], sd["patch_embed.text_proj.bias"])
    x = F.conv2d(latents.reshape(B * Fr, C, H, W), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=p)
    x = x.view(B, Fr, cfg.dim, -1).transpose(2, 3).flatten(1, 2)     # [B, F*h*w, D]
    x = torch.cat([e, x], dim=1) + patch_embed_positions(sd, cfg, Fr, H // p, W // p).to(x.dtype)
    text_len = e.size(1)
    e, h = x[:, :text_len], x[:, text_len:]
    for i in range(cfg.num_layers):
        h, e = block(sub(sd, f"transformer_blocks.{i}"), cfg, h, e, temb, rope, ip_hidden, ip_scale)
    x = layer_norm(torch.cat([e, h], dim=1), sd["norm_final.weight"], sd["norm_final.bias"], cfg.norm_eps)[:, text_len:]
    shift, scale = F.linear(F.silu(temb), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"]).chunk(2, dim=1)
    x = layer_norm(x, sd["norm_out.norm.weight"], sd["norm_out.norm.bias"], cfg.norm_eps) * (1 + scale)[:, None] + shift[:, None]
    x = F.linear(x, sd["proj_out.weight"], sd["proj_out.bias"])
    x = x.reshape(B, Fr, H // p, W // p, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    return x


# ----------------------------------------------------------------------------------------------
# CogVideoXDDIMScheduler (v-prediction, trailing spacing, zero-terminal-SNR scaled-linear betas)
# ----------------------------------------------------------------------------------------------
def ddim_alphas_cumprod(n_train=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=1.0) -> np.ndarray:
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    s = np.sqrt(ac)                     # rescale_zero_terminal_snr on alphas_cumprod
    s0, sT = s[0], s[-1]
    s = (s - sT) * (s0 / (s0 - sT))
    return s ** 2


def ddim_timesteps(num_inference_steps: int, n_train=1000) -> np.ndarray:
    return (np.round(np.arange(n_train, 0, -n_train / num_inference_steps)) - 1).astype(np.int64)


def ddim_coeffs(ac: np.ndarray, t: int, num_inference_steps: int, n_train=1000):
    """returns (sqrt_alpha_t, sqrt_beta_t, a_t, b_t): x0 = sa*x - sb*v; x_prev = a*x + b*x0."""
    prev = t - n_train // num_inference_steps
    a_t = ac[t]
    a_prev = ac[prev] if prev >= 0 else 1.0
    a = ((1 - a_prev) / (1 - a_t)) ** 0.5
    b = a_prev ** 0.5 - a_t ** 0.5 * a
    return float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a), float(b)


def cfg_ddim_step(v_pred: torch.Tensor, latents: torch.Tensor, guidance: float, coeffs) -> torch.Tensor:
    """pipeline CFG (uncond first) + scheduler.step; fp32 math."""
    sa, sb, a, b = coeffs
    vu, vc = v_pred.float().chunk(2)
    v = vu + guidance * (vc - vu)
    x = latents.float()
    x0 = sa * x - sb * v
    return a * x + b * x0


# ----------------------------------------------------------------------------------------------
# random-init weights (N(0, 0.02), norms gamma = 1 + N), diffusers key layout
# ----------------------------------------------------------------------------------------------
def dpm_step(ac: np.ndarray, model_output: torch.Tensor, old_pred_original_sample: Optional[torch.Tensor], timestep: int, timestep_back: Optional[int],
             sample: torch.Tensor, num_inference_steps: int, noise_fn, n_train: int = 1000):
    """diffusers 0.32.2 `CogVideoXDPMScheduler.step` (v-prediction), statement by statement -- third-party, PARITY UNPINNED; the scheduler the reference's shipped
    config selects (configs/cogvideox/MotionRAG_open.yml:189-194 `scheduler: "dpm"`, cogvideox/module.py:28-35).  `noise_fn()` stands for
    `randn_tensor(sample.shape, generator=...)`: it is called once per step, and a second time on a second-order step (whose result uses the second draw).
    Returns (prev_sample, pred_original_sample)."""
    acp = torch.from_numpy(np.asarray(ac)).to(torch.float32)                 # the scheduler's alphas_cumprod tensor is fp32
    prev_timestep = timestep - n_train // num_inference_steps
    alpha_prod_t = acp[timestep]
    alpha_prod_t_prev = acp[prev_timestep] if prev_timestep >= 0 else torch.tensor(1.0)
    alpha_prod_t_back = acp[timestep_back] if timestep_back is not None else None
    beta_prod_t = 1 - alpha_prod_t
    pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
    # get_variables
    lamb = ((alpha_prod_t / (1 - alpha_prod_t)) ** 0.5).log()
    lamb_next = ((alpha_prod_t_prev / (1 - alpha_prod_t_prev)) ** 0.5).log()
    h = lamb_next - lamb
    r = None
    if alpha_prod_t_back is not None:
        lamb_previous = ((alpha_prod_t_back / (1 - alpha_prod_t_back)) ** 0.5).log()
        r = (lamb - lamb_previous) / h
    # get_mult
    mult1 = ((1 - alpha_prod_t_prev) / (1 - alpha_prod_t)) ** 0.5 * (-h).exp()
    mult2 = (-2 * h).expm1() * alpha_prod_t_prev ** 0.5
    mult_noise = (1 - alpha_prod_t_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
    noise = noise_fn()
    prev_sample = mult1 * sample - mult2 * pred_original_sample + mult_noise * noise
    if old_pred_original_sample is None or prev_timestep < 0:
        return prev_sample, pred_original_sample
    mult3, mult4 = 1 + 1 / (2 * r), 1 / (2 * r)
    denoised_d = mult3 * pred_original_sample - mult4 * old_pred_original_sample
    noise = noise_fn()
    return mult1 * sample - mult2 * denoised_d + mult_noise * noise, pred_original_sample


def random_dit_sd(cfg: DiTConfig, seed: int = 0, std: float = 0.02) -> SD:
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * std
    D = cfg.dim
    sd = {
        "patch_embed.proj.weight": r(D, cfg.in_channels, cfg.patch, cfg.patch), "patch_embed.proj.bias": r(D),
        "patch_embed.text_proj.weight": r(D, cfg.text_embed_dim), "patch_embed.text_proj.bias": r(D),
        "patch_embed.pos_embedding": torch.cat([torch.zeros(1, cfg.max_text_len, D), r(1, cfg.video_tokens, D)], dim=1),
        "time_embedding.linear_1.weight": r(cfg.time_embed_dim, D), "time_embedding.linear_1.bias": r(cfg.time_embed_dim),
        "time_embedding.linear_2.weight": r(cfg.time_embed_dim, cfg.time_embed_dim), "time_embedding.linear_2.bias": r(cfg.time_embed_dim),
        "norm_final.weight": 1 + r(D), "norm_final.bias": r(D),
        "norm_out.linear.weight": r(2 * D, cfg.time_embed_dim), "norm_out.linear.bias": r(2 * D),
        "norm_out.norm.weight": 1 + r(D), "norm_out.norm.bias": r(D),
        "proj_out.weight": r(cfg.patch * cfg.patch * cfg.out_channels, D), "proj_out.bias": r(cfg.patch * cfg.patch * cfg.out_channels),
    }
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[p + n + ".linear.weight"] = r(6 * D, cfg.time_embed_dim)
            sd[p + n + ".linear.bias"] = r(6 * D)
            sd[p + n + ".norm.weight"] = 1 + r(D)
            sd[p + n + ".norm.bias"] = r(D)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            sd[p + f"attn1.{n}.weight"] = r(D, D)
            sd[p + f"attn1.{n}.bias"] = r(D)
        for n in ("norm_q", "norm_k"):
            sd[p + f"attn1.{n}.weight"] = 1 + r(cfg.head_dim)
            sd[p + f"attn1.{n}.bias"] = r(cfg.head_dim)
        sd[p + "attn1.processor.to_q_ip.0.weight"] = r(D, D)
        sd[p + "attn1.processor.to_k_ip.0.weight"] = r(D, cfg.ip_dim)
        sd[p + "attn1.processor.to_v_ip.0.weight"] = r(D, cfg.ip_dim)
        sd[p + "ff.net.0.proj.weight"] = r(4 * D, D)
        sd[p + "ff.net.0.proj.bias"] = r(4 * D)
        sd[p + "ff.net.2.weight"] = r(D, 4 * D)
        sd[p + "ff.net.2.bias"] = r(D)
    return sd
