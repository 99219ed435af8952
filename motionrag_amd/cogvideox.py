"""Motion-injected CogVideoX-5B-I2V denoising loop on hand-written gfx950 kernels.

Mirrors, for the hot path only, what the reference reaches through diffusers==0.32.2 (not vendored;
SURVEY.md Appendix E) and its own thin subclasses:

    reference                                                     here
    diffusers CogVideoXTransformer3DModel / CogVideoXBlock         CogVideoXTransformer3DModel / CogVideoXBlock
      (installed via cogvideox/module.py:23-48, adapters :163-175)   (same state-dict keys, set_attn_processor)
    diffusers CogVideoXDDIMScheduler (module.py:28-35)             CogVideoXDDIMScheduler
    CogVideoXImageToVideoCTPipeline (cogvideox/pipeline.py:92-130) CogVideoXImageToVideoCTPipeline
      ._prepare_rotary_positional_embeddings :46-57                  (rope, action_emb) hand-off kept
      .prepare_action_embeddings :117-130                            -> ActionTransformer.predict
      .__call__ :80-89                                               denoise loop: CFG concat, DiT, DDIM

The DiT keeps the residual stream as ONE joint [text ; video] buffer [B, S, D] for all 42 blocks (the
reference's processor concatenates and splits per block, attn_processor.py:199,280-283): AdaLN-zero
modulation, gated residuals and the text/video split are row-range parameters of the LayerNorm and
GEMM-epilogue kernels.  T5 / VAE are third-party and outside the hot path: the pipeline
takes prompt embeddings and image latents, or user-supplied encoders.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from .attn_processor import APAdapterCogVideoXAttnProcessor2_0, Attention, joint_attention_core


class _LayerNormZero(nn.Module):
    """CogVideoXLayerNormZero weight container: linear (512 -> 6 D) + LayerNorm(D)."""

    def __init__(self, cond_dim: int, dim: int, eps: float):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, eps=eps)


class _GeluProj(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner)


class _FeedForward(nn.Module):
    """diffusers FeedForward('gelu-approximate') container: net.0.proj, net.2"""

    def __init__(self, dim, inner):
        super().__init__()
        self.net = nn.ModuleList([_GeluProj(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])


class CogVideoXBlock(nn.Module):
    def __init__(self, dim: int, heads: int, time_embed_dim: int, norm_eps: float, qk_eps: float = 1e-6):
        super().__init__()
        self.norm1 = _LayerNormZero(time_embed_dim, dim, norm_eps)
        self.attn1 = Attention(dim, heads=heads, dim_head=dim // heads, bias=True, out_bias=True, qk_norm="layer_norm", eps=qk_eps)
        self.norm2 = _LayerNormZero(time_embed_dim, dim, norm_eps)
        self.ff = _FeedForward(dim, 4 * dim)


class _PatchEmbed(nn.Module):
    def __init__(self, patch, in_channels, dim, text_dim, n_pos):
        super().__init__()
        self.proj = nn.Conv2d(in_channels, dim, kernel_size=patch, stride=patch)
        self.text_proj = nn.Linear(text_dim, dim)
        self.register_buffer("pos_embedding", torch.zeros(1, n_pos, dim), persistent=True)


class _TimeEmbedding(nn.Module):
    def __init__(self, dim, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(dim, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class _AdaLayerNorm(nn.Module):
    def __init__(self, cond_dim, dim, eps):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, eps=eps)


class _PlainProcessor:
    """blocks without a motion adapter: same fused path, adapter branch skipped"""

    def __init__(self):
        from .attn_processor import _FusedWeights
        self._fused = _FusedWeights()
        self.scale = [0.0]


class CogVideoXTransformer3DModel(nn.Module):
    """CogVideoX-5B-I2V DiT (rotary + learned positional embedding flavour)."""

    def __init__(self, num_layers=42, num_attention_heads=48, attention_head_dim=64, in_channels=32, out_channels=16, time_embed_dim=512,
                 text_embed_dim=4096, max_text_seq_length=226, patch_size=2, sample_frames=13, sample_height=60, sample_width=90,
                 norm_eps=1e-5, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0):
        super().__init__()
        if attention_head_dim != 64:
            raise NotImplementedError("head_dim 64 only")
        D = num_attention_heads * attention_head_dim
        self.cfg = dict(num_layers=num_layers, heads=num_attention_heads, dim=D, in_channels=in_channels, out_channels=out_channels,
                        time_embed_dim=time_embed_dim, text_embed_dim=text_embed_dim, max_text=max_text_seq_length, patch=patch_size,
                        norm_eps=norm_eps, sample=(sample_frames, sample_height // patch_size, sample_width // patch_size),
                        pos_scales=(spatial_interpolation_scale, temporal_interpolation_scale))
        self._pos_tables: Dict[tuple, torch.Tensor] = {}
        n_video = sample_frames * (sample_height // patch_size) * (sample_width // patch_size)
        self.patch_embed = _PatchEmbed(patch_size, in_channels, D, text_embed_dim, max_text_seq_length + n_video)
        self.time_embedding = _TimeEmbedding(D, time_embed_dim)
        self.transformer_blocks = nn.ModuleList(
            [CogVideoXBlock(D, num_attention_heads, time_embed_dim, norm_eps) for _ in range(num_layers)])
        self.norm_final = nn.LayerNorm(D, eps=norm_eps)
        self.norm_out = _AdaLayerNorm(time_embed_dim, D, norm_eps)
        self.proj_out = nn.Linear(D, patch_size * patch_size * out_channels)
        self._fused: Dict[str, torch.Tensor] = {}
        self._plain = _PlainProcessor()

    # ---- diffusers-style processor plumbing (cogvideox/module.py:163-175 uses exactly these two) ----
    @property
    def attn_processors(self):
        return {f"transformer_blocks.{i}.attn1.processor": blk.attn1.processor for i, blk in enumerate(self.transformer_blocks)}

    def set_attn_processor(self, processors):
        for i, blk in enumerate(self.transformer_blocks):
            name = f"transformer_blocks.{i}.attn1.processor"
            blk.attn1.set_processor(processors[name] if isinstance(processors, dict) else processors)

    def install_motion_adapters(self, cross_attention_dim: int, scale: float = 1.0):
        """set_attention_processors of cogvideox/module.py:163-175 with every attn1 adapted (42 sites)."""
        self.set_attn_processor({name: APAdapterCogVideoXAttnProcessor2_0(self.cfg["dim"], cross_attention_dim, scale=scale)
                                 for name in self.attn_processors})
        return self

    # ---- fused weights (built once) ----
    def _mod_weights(self):
        tag = tuple((w.data_ptr(), w.dtype, w._version) for w in (self.norm_out.linear.weight, self.transformer_blocks[0].norm1.linear.weight,
                                                                    self.patch_embed.proj.weight))
        if self._fused.get("tag") != tag:
            self._fused = {"tag": tag}
            ws, bs = [], []
            for blk in self.transformer_blocks:
                for n in (blk.norm1, blk.norm2):
                    ws.append(n.linear.weight.detach()); bs.append(n.linear.bias.detach())
            ws.append(self.norm_out.linear.weight.detach()); bs.append(self.norm_out.linear.bias.detach())
            self._fused["mod_w"] = torch.cat(ws, 0).contiguous()
            self._fused["mod_b"] = torch.cat(bs, 0).contiguous()
            self._fused["patch_w"] = self.patch_embed.proj.weight.detach().reshape(self.cfg["dim"], -1).contiguous()
        return self._fused["mod_w"], self._fused["mod_b"]

    def joint_pos_embedding(self, frames: int, height: int, width: int) -> torch.Tensor:
        """[max_text + frames * height * width, D] bf16: the positional rows added to the joint sequence (diffusers 0.32.2
        `CogVideoXPatchEmbed.forward`, parity unpinned).  At the model's own sample geometry that is the LEARNED table of the checkpoint.  For any
        other clip length diffusers does not slice the learned table: it REGENERATES a table with `_get_positional_embeddings` -- 3-D sin-cos video
        rows at the model's interpolation scales and ZERO text rows -- which is what the reference's shipped evaluation runs with (17 frames ->
        5 latent frames on a 13-frame model, configs/cogvideox/MotionRAG_open.yml:189-194).  A different resolution is refused, as diffusers
        refuses it for learned tables.  Host-built once per geometry, cached on the device."""
        cfg = self.cfg
        sf, sh, sw = cfg["sample"]
        if (height, width) != (sh, sw):
            raise ValueError(f"{height}x{width} patch grid: a model with learned positional embeddings runs at its own resolution only ({sh}x{sw})")
        if frames == sf:
            return self.patch_embed.pos_embedding[0]
        key = (frames, height, width, self.patch_embed.pos_embedding.device)
        if key not in self._pos_tables:
            video = get_3d_sincos_pos_embed(cfg["dim"], (width, height), frames, *cfg["pos_scales"])
            joint = torch.cat([torch.zeros(cfg["max_text"], cfg["dim"]), video], dim=0)
            self._pos_tables[key] = joint.to(self.patch_embed.pos_embedding.device, torch.bfloat16).contiguous()
        return self._pos_tables[key]

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, timestep: torch.Tensor,
                image_rotary_emb=None, image_latents: Optional[torch.Tensor] = None, batch: Optional[int] = None, sp=None) -> torch.Tensor:
        """hidden_states [Bl, F, C, H, W] bf16 (C = in_channels, or the noisy half when `image_latents`
        carries the other half); encoder_hidden_states [B, L, text_dim]; timestep [B] fp32;
        image_rotary_emb = ((cos, sin), action_emb) as handed over by the pipeline (pipeline.py:46-57).
        Batch entry b reads latent b % Bl (CFG duplication without materialising torch.cat([latents] * 2)).
        `sp` (dist.SequenceParallel): this rank computes rows [r0, r1) of the joint sequence of every batch entry; one K/V all-gather per
        block and one all-gather of the projected output rows; the returned tensor is complete on every rank."""
        cfg = self.cfg
        D, Hh, p = cfg["dim"], cfg["heads"], cfg["patch"]
        B = encoder_hidden_states.shape[0] if batch is None else batch
        Bl, F, C0, H, W = hidden_states.shape
        Lt = encoder_hidden_states.shape[1]
        Nv = F * (H // p) * (W // p)
        S = Lt + Nv
        if Lt != cfg["max_text"]:
            raise ValueError(f"{Lt} text tokens: the positional table has {cfg['max_text']} text rows (max_text_seq_length)")
        (rope, ip) = image_rotary_emb if (isinstance(image_rotary_emb, tuple) and isinstance(image_rotary_emb[0], tuple)) else (image_rotary_emb, None)
        mod_w, mod_b = self._mod_weights()

        # timestep embedding -> temb -> every block's AdaLN-zero modulation in one GEMM
        temb = ops.timestep_embedding(timestep.to(torch.float32), D)
        temb = ops.linear(temb, self.time_embedding.linear_1.weight, self.time_embedding.linear_1.bias, epilogue=ops.EPI_SILU)
        temb = ops.linear(temb, self.time_embedding.linear_2.weight, self.time_embedding.linear_2.bias, epilogue=ops.EPI_SILU)  # silu(temb)
        mod = ops.linear(temb, mod_w, mod_b)                        # [B, (2L*6 + 2) D]
        ms = mod.stride(0)

        # patch embed (Conv2d k=2 s=2 as a GEMM over patch rows) + text projection + positional embedding
        from .dist import SequenceParallel
        r0, r1, Ltl, t0, v0, v1 = (SequenceParallel(0, 1) if sp is None else sp).layout(S, Lt)   # this rank's rows: text [t0, t0 + Ltl) first, video [v0, v1)
        Sl = r1 - r0
        x = torch.empty(B, Sl, D, dtype=torch.bfloat16, device=hidden_states.device)
        pos = self.joint_pos_embedding(F, H // p, W // p)
        patches = ops.patchify(hidden_states, image_latents, B).view(B, Nv, -1)
        for b in range(B):
            if Ltl > 0:
                ops.linear(encoder_hidden_states[b, t0:t0 + Ltl], self.patch_embed.text_proj.weight, self.patch_embed.text_proj.bias, out=x[b, :Ltl],
                           epilogue=ops.EPI_RESID, resid=pos[t0:t0 + Ltl])
            if v1 > v0:
                ops.linear(patches[b, v0:v1], self._fused["patch_w"], self.patch_embed.proj.bias, out=x[b, Ltl:], epilogue=ops.EPI_RESID,
                           resid=pos[Lt + v0:Lt + v1])
        if sp is not None and rope is not None:
            rope = (rope[0][v0:v1], rope[1][v0:v1])                 # RoPE rows of the local video tokens
        S_full, Lt_full = S, Lt
        S, Lt = Sl, Ltl                                             # everything below is per-row: local row count / local split

        def chunk(layer_slot: int, j: int) -> torch.Tensor:
            off = (layer_slot * 6 + j) * D
            return mod[:, off:off + D]

        for i, blk in enumerate(self.transformer_blocks):
            proc = blk.attn1.processor if isinstance(blk.attn1.processor, APAdapterCogVideoXAttnProcessor2_0) else self._plain
            # norm1: shift, scale, gate, enc_shift, enc_scale, enc_gate = chunk(6)
            nh = ops.layernorm(x, blk.norm1.norm.weight, blk.norm1.norm.bias, cfg["norm_eps"], shift0=chunk(2 * i, 3), scale0=chunk(2 * i, 4),
                               shift1=chunk(2 * i, 0), scale1=chunk(2 * i, 1), rows_per_batch=S, split=Lt, mod_stride=ms)
            scale = proc.scale[0] if ip is not None else 0.0
            o = joint_attention_core(blk.attn1, proc, nh, Lt, rope, ip, scale, sp=sp)
            ops.linear(o, blk.attn1.to_out[0].weight, blk.attn1.to_out[0].bias, out=x, epilogue=ops.EPI_GATE_RESID, resid=x,
                       gate0=chunk(2 * i, 5), gate1=chunk(2 * i, 2), rows_per_batch=S, split=Lt, gate_stride=ms)
            nh = ops.layernorm(x, blk.norm2.norm.weight, blk.norm2.norm.bias, cfg["norm_eps"], shift0=chunk(2 * i + 1, 3),
                               scale0=chunk(2 * i + 1, 4), shift1=chunk(2 * i + 1, 0), scale1=chunk(2 * i + 1, 1), rows_per_batch=S,
                               split=Lt, mod_stride=ms, out=nh)
            f = ops.linear(nh, blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, epilogue=ops.EPI_GELU_TANH)
            ops.linear(f, blk.ff.net[2].weight, blk.ff.net[2].bias, out=x, epilogue=ops.EPI_GATE_RESID, resid=x,
                       gate0=chunk(2 * i + 1, 5), gate1=chunk(2 * i + 1, 2), rows_per_batch=S, split=Lt, gate_stride=ms)

        # tail: norm_final over the joint sequence, AdaLayerNorm (shift, scale = chunk(2)), proj_out, unpatchify
        y = ops.layernorm(x, self.norm_final.weight, self.norm_final.bias, cfg["norm_eps"])
        base = 2 * len(self.transformer_blocks) * 6 * D
        sh, sc = mod[:, base:base + D], mod[:, base + D:base + 2 * D]
        y = ops.layernorm(y, self.norm_out.norm.weight, self.norm_out.norm.bias, cfg["norm_eps"], shift0=sh, scale0=sc, shift1=sh, scale1=sc,
                          rows_per_batch=S, split=Lt, mod_stride=ms, out=y)
        out = ops.linear(y, self.proj_out.weight, self.proj_out.bias)
        if sp is not None:                                          # [Sl, B, C] per rank -> [S, B, C] -> [B, S, C]
            out = sp.all_gather(out.permute(1, 0, 2).contiguous()).permute(1, 0, 2)
        out = out[:, Lt_full:].contiguous()
        return ops.unpatchify(out, B, F, cfg["out_channels"], H, W)


# ------------------------------------------------------------------------------------------------------
def get_3d_sincos_pos_embed(embed_dim: int, spatial_size: Tuple[int, int], temporal_size: int, spatial_interpolation_scale: float = 1.0,
                            temporal_interpolation_scale: float = 1.0) -> torch.Tensor:
    """host-side table of diffusers' `get_3d_sincos_pos_embed` (0.32.2; parity unpinned): [T * H * W, D] fp32, spatial_size = (W, H).
    Row (t, y, x) = [ sincos(t / ts ; D/4) | sincos(x / ss ; 3D/8) | sincos(y / ss ; 3D/8) ] with sincos(p ; n) = [sin(p w_k), cos(p w_k)],
    w_k = 10000^(-k / (n/2)), k < n/2 -- the MAE layout, in which the slot diffusers NAMES `emb_h` is fed the x coordinate (its meshgrid
    puts w first); kept, since the checkpoint's learned table was initialised from exactly this."""
    if embed_dim % 4:
        raise ValueError("`embed_dim` must be divisible by 4")
    W, H = spatial_size
    T = temporal_size

    def sincos(pos: torch.Tensor, n: int) -> torch.Tensor:
        w = 10000.0 ** (-torch.arange(n // 2, dtype=torch.float64) / (n / 2.0))
        a = pos.to(torch.float64)[:, None] * w[None, :]
        return torch.cat([a.sin(), a.cos()], dim=1)
    ds, dt = 3 * embed_dim // 4, embed_dim // 4
    ex = sincos(torch.arange(W, dtype=torch.float32) / spatial_interpolation_scale, ds // 2)       # [W, ds/2]
    ey = sincos(torch.arange(H, dtype=torch.float32) / spatial_interpolation_scale, ds // 2)       # [H, ds/2]
    et = sincos(torch.arange(T, dtype=torch.float32) / temporal_interpolation_scale, dt)           # [T, dt]
    out = torch.cat([et[:, None, None, :].expand(T, H, W, dt), ex[None, None, :, :].expand(T, H, W, ds // 2),
                     ey[None, :, None, :].expand(T, H, W, ds // 2)], dim=-1)
    return out.reshape(T * H * W, embed_dim).to(torch.float32)


def get_3d_rotary_pos_embed(head_dim: int, t: int, h: int, w: int, theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """host-side table (diffusers get_3d_rotary_pos_embed, use_real=True, crop == full grid): [t*h*w, d] fp32."""
    def one(dim, n):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        f = torch.outer(torch.arange(n, dtype=torch.float32), freqs)
        return f.cos().repeat_interleave(2, dim=1), f.sin().repeat_interleave(2, dim=1)
    dt, dh, dw = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3
    (ct, st), (ch, sh), (cw, sw) = one(dt, t), one(dh, h), one(dw, w)

    def bc(a_t, a_h, a_w):
        return torch.cat([a_t[:, None, None, :].expand(t, h, w, dt), a_h[None, :, None, :].expand(t, h, w, dh),
                          a_w[None, None, :, :].expand(t, h, w, dw)], dim=-1).reshape(t * h * w, head_dim).contiguous()
    return bc(ct, ch, cw), bc(st, sh, sw)


class CogVideoXDDIMScheduler:
    """v-prediction DDIM, trailing spacing, zero-terminal-SNR scaled-linear betas (module.py:28-35 selects it;
    tables are host-side float64 scalars, the update itself is the mrag_cfg_ddim_step kernel)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, snr_shift_scale=1.0):
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
        ac = np.cumprod(1.0 - betas)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        s = np.sqrt(ac)
        s0, sT = s[0], s[-1]
        s = (s - sT) * (s0 / (s0 - sT))
        self.alphas_cumprod = s ** 2
        self.num_train_timesteps = num_train_timesteps
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps: int):
        n = self.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        self.timesteps = (np.round(np.arange(n, 0, -n / num_inference_steps)) - 1).astype(np.int64)
        return self.timesteps

    def coeffs(self, t: int):
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else 1.0
        a = ((1 - a_prev) / (1 - a_t)) ** 0.5
        b = a_prev ** 0.5 - a_t ** 0.5 * a
        return float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a), float(b)


class CogVideoXDPMScheduler(CogVideoXDDIMScheduler):
    """diffusers 0.32.2 `CogVideoXDPMScheduler` -- what the shipped config samples with (`scheduler: "dpm"`, 25 steps: configs/cogvideox/MotionRAG_open.yml:189-194,
    module.py:28-35): the same zero-terminal-SNR tables and trailing timesteps as the DDIM class, and the SDE form of DPM-Solver++(2M) as the step
    (`get_variables` / `get_mult` / `step`): first order on the first and the last step, second order in between, fresh Gaussian noise every step.
    The multipliers are host-side float64 scalars; the update is the `mrag_cfg_dpm_step_bf16` kernel."""

    def dpm_coeffs(self, t: int, t_back: Optional[int]):
        """(sqrt(a_t), sqrt(1 - a_t), mult1, mult2, mult3, mult4, mult_noise, second_order) for the step at timestep t whose predecessor in the schedule was t_back"""
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else 1.0
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            lam = lambda a: 0.5 * (np.log(np.float64(a)) - np.log(np.float64(1.0 - a)))          # log sqrt(a / (1 - a)); -inf at a = 0 (zero terminal SNR), +inf at a = 1
            h = lam(a_prev) - lam(a_t)
            m1 = float(np.sqrt((1.0 - a_prev) / (1.0 - a_t)) * np.exp(-h))
            m2 = float(np.expm1(-2.0 * h) * np.sqrt(a_prev))
            mn = float(np.sqrt(1.0 - a_prev) * np.sqrt(1.0 - np.exp(-2.0 * h)))
            second = t_back is not None and prev >= 0
            m3, m4 = 1.0, 0.0
            if second:
                r = (lam(a_t) - lam(float(self.alphas_cumprod[t_back]))) / h
                m3, m4 = float(1.0 + 1.0 / (2.0 * r)), float(1.0 / (2.0 * r))
        return float(np.sqrt(a_t)), float(np.sqrt(1.0 - a_t)), m1, m2, m3, m4, mn, second


def make_scheduler(name: str = "ddim"):
    """cogvideox/module.py:28-35: `eval_pipeline_call_kwargs.scheduler` -> scheduler object"""
    if name == "ddim":
        return CogVideoXDDIMScheduler()
    if name == "dpm":
        return CogVideoXDPMScheduler()
    raise ValueError(f"Unknown scheduler: {name}")


class CogVideoXPipelineOutput:
    """what diffusers' pipeline returns: `.frames`, and `output[0]` is `.frames` too (the reference indexes it: cogvideox/module.py:211)"""

    def __init__(self, frames):
        self.frames = frames

    def __getitem__(self, i):
        return (self.frames,)[i]

    def __iter__(self):
        return iter((self.frames,))


def _model_device(m) -> torch.device:
    return next(m.parameters()).device


class CogVideoXImageToVideoActionPipeline:
    """Stage-1 pipeline (src/projects/cogvideox/pipeline.py:13-89): motion tokens of the k retrieved reference clips from a frozen
    `action_embedder`, fused over k by `condition_fusion` (HIP kernel), projected by `action_proj_model`, injected through the rope hook.

    Mirrors the reference's constructor and call surface.  The diffusers base class it inherits there (CogVideoXImageToVideoPipeline 0.32.2,
    third-party) is restated in `__call__` / `denoise`; `tokenizer` + `text_encoder` (T5) and `vae` are third-party modules the caller supplies
    -- duck-typed: `text_encoder(tokenizer(...).input_ids)[0]` or, without a tokenizer, `text_encoder(list_of_str) -> [b, 226, 4096]`;
    `vae.encode(x[b, 3, 1, H, W])` -> `.latent_dist.sample(generator)` / `.sample()` / tensor, `vae.decode(z[b, 16, F, h, w])` -> `.sample` /
    tensor, `vae.config.scaling_factor` (0.7 for CogVideoX-5B-I2V)."""

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer: CogVideoXTransformer3DModel = None, scheduler=None,
                 action_embedder=None, action_proj_model=None, ref_fusion_type: str = "mean"):
        self.tokenizer, self.text_encoder, self.vae = tokenizer, text_encoder, vae
        self.transformer = transformer
        self.scheduler = scheduler if scheduler is not None else CogVideoXDPMScheduler()   # module.py:44: the motion-injected pipelines get the DPM scheduler
        self.action_embedder, self.action_proj_model = action_embedder, action_proj_model
        self.ref_fusion_type = ref_fusion_type
        self._rope_cache = {}

    @property
    def _execution_device(self):
        return _model_device(self.transformer)

    def set_progress_bar_config(self, **_):          # diffusers API used by module.py:279
        return None

    def _prepare_rotary_positional_embeddings(self, frames: int, gh: int, gw: int, device):
        """pipeline.py:46-57: returns ((cos, sin), action_emb)"""
        assert hasattr(self, "action_emb"), "action_emb is not set"
        key = (frames, gh, gw, str(device))
        if key not in self._rope_cache:
            cos, sin = get_3d_rotary_pos_embed(64, frames, gh, gw)
            self._rope_cache = {key: (cos.to(device), sin.to(device))}        # single slot: a pipeline works on one geometry at a time
        return self._rope_cache[key], self.action_emb

    def prepare_action_embeddings(self, ref_videos: torch.Tensor, metadata, do_classifier_free_guidance: bool = False, *args, **kwargs):
        """pipeline.py:59-78: [b, k, f, c, h, w] -> [b or 2b, t, c]"""
        from .cama import condition_fusion
        b, k = ref_videos.shape[:2]
        emb = self.action_embedder(ref_videos.reshape(b * k, *ref_videos.shape[2:]))
        emb = emb.view(b, k, *emb.shape[1:])
        emb = condition_fusion(emb, self.ref_fusion_type, weight=[m["ref_video_distance"] for m in metadata] if self.ref_fusion_type == "weight" else None)
        if do_classifier_free_guidance:
            uncond = self.action_embedder(torch.zeros_like(ref_videos[:, 0]))
            if emb.shape[1:] != uncond.shape[1:]:
                raise ValueError("ref_fusion_type 'concat' changes the token count: the unconditional branch cannot be concatenated (as in the reference)")
            emb = torch.cat([uncond.to(emb.dtype), emb], dim=0)
        return self.action_proj_model(emb)

    # ---- pieces of diffusers' CogVideoXImageToVideoPipeline.__call__ the loop needs ----
    def encode_prompt(self, prompt, negative_prompt, prompt_embeds=None, negative_prompt_embeds=None, max_sequence_length: int = 226):
        def enc(texts):
            if self.text_encoder is None:
                raise ValueError("pass prompt_embeds / negative_prompt_embeds or a text_encoder")
            if self.tokenizer is None:
                return self.text_encoder(texts)
            ids = self.tokenizer(texts, padding="max_length", max_length=max_sequence_length, truncation=True, add_special_tokens=True,
                                 return_tensors="pt").input_ids
            return self.text_encoder(ids.to(self._execution_device))[0]
        if prompt_embeds is None:
            prompt = [prompt] if isinstance(prompt, str) else list(prompt)
            prompt_embeds = enc(prompt)
        if negative_prompt_embeds is None:
            b = prompt_embeds.shape[0]
            neg = negative_prompt if negative_prompt is not None else ""
            neg = [neg] * b if isinstance(neg, str) else list(neg)
            negative_prompt_embeds = enc(neg)
        return prompt_embeds, negative_prompt_embeds

    def _vae_scale(self) -> float:
        return float(getattr(getattr(self.vae, "config", None), "scaling_factor", 0.7))

    def encode_image_latents(self, image: torch.Tensor, lat_frames: int, generator=None, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        """image [b, 3, H, W] in [0, 1] -> [b, F, 16, H/8, W/8]: VAE latents of the first frame, zero-padded to F latent frames (diffusers
        `prepare_latents`: every image is encoded and its posterior sampled on its own, in batch order, so the generator is consumed image by image).
        diffusers' `video_processor.preprocess` would resize to height x width first; a mismatching image is refused here instead."""
        if height is not None and tuple(image.shape[-2:]) != (height, width):
            raise ValueError(f"image is {tuple(image.shape[-2:])}, the call asks for {(height, width)}: resize it before the pipeline")
        x = (image.to(torch.float32) * 2.0 - 1.0).unsqueeze(2)                     # [0, 1] -> [-1, 1], one frame
        zs = []
        for one in x.to(self._execution_device).split(1):
            enc = self.vae.encode(one)
            if hasattr(enc, "latent_dist"):
                enc = enc.latent_dist.sample(generator)
            elif hasattr(enc, "sample") and callable(enc.sample):
                enc = enc.sample()
            zs.append(enc)
        z = torch.cat(zs, dim=0) * self._vae_scale()                                # [b, 16, 1, h, w]
        z = z.permute(0, 2, 1, 3, 4)                                                # [b, 1, 16, h, w]
        pad = torch.zeros(z.shape[0], lat_frames - 1, *z.shape[2:], dtype=z.dtype, device=z.device)
        return torch.cat([z, pad], dim=1)

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        z = latents.permute(0, 2, 1, 3, 4).to(torch.float32) / self._vae_scale()    # [b, 16, F, h, w]
        out = self.vae.decode(z)
        out = out.sample if hasattr(out, "sample") and not callable(out.sample) else out
        return out                                                                  # [b, 3, f, H, W] in [-1, 1]

    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, image_latents: torch.Tensor, prompt_embeds: torch.Tensor, action_emb: torch.Tensor,
                num_inference_steps: int = 50, guidance_scale: float = 6.0, callback=None, sp=None, cfgp=None, hip_graph: bool = False, generator=None,
                use_dynamic_cfg: bool = False) -> torch.Tensor:
        """the hot loop: latents [b, F, 16, h, w] bf16 (N(0,1) noise), prompt_embeds = cat([negative, positive])
        [2b, L, 4096], action_emb [2b, 25, 1024] (uncond first, module.py:329).
        `sp` (dist.SequenceParallel): the token sequence of the clip sharded over the ranks; `cfgp` (dist.CFGParallel): this rank runs ONE of
        the two guidance branches at batch b and the pair exchanges the velocity prediction (2.2 MB per step) before the shared update.
        With a `CogVideoXDPMScheduler` the update is the stochastic DPM step: its noise is drawn per step from `generator` exactly as the reference's
        `randn_tensor(sample.shape, generator, device, dtype)` does (bf16, on the generator's device; TWO draws on a second-order step, the second one used).
        `use_dynamic_cfg`: the cosine guidance schedule of diffusers' pipeline (off by default there and in the reference's calls).
        Ranks that share a clip (`sp`, `cfgp`) must pass generators in the SAME state: every one of them draws the clip's sampler noise itself."""
        self.action_emb = action_emb
        dpm = isinstance(self.scheduler, CogVideoXDPMScheduler)
        x0_prev = torch.zeros_like(latents) if dpm else None

        def draw():
            if generator is not None and generator.device.type != latents.device.type:
                return torch.randn(latents.shape, generator=generator, dtype=torch.bfloat16).to(latents.device)
            return torch.randn(latents.shape, generator=generator, dtype=torch.bfloat16, device=latents.device)
        b, F, C, h, w = latents.shape
        p = self.transformer.cfg["patch"]
        ts = self.scheduler.set_timesteps(num_inference_steps)
        B = 2 * b
        if cfgp is not None:
            sl = slice(cfgp.branch * b, (cfgp.branch + 1) * b)
            prompt_embeds, self.action_emb, B = prompt_embeds[sl].contiguous(), (action_emb[sl].contiguous() if action_emb is not None else None), b
        rope_ip = self._prepare_rotary_positional_embeddings(F, h // p, w // p, latents.device)
        graph = None
        if hip_graph:
            # The DiT forward of a step is the same ~420 launches with the same arguments every step (only the timestep VALUE and the latents' contents change):
            # capture it once per clip and replay it; the CFG + DDIM update stays an eager launch (its coefficients are host scalars).  One eager pass first, on a
            # side stream, builds the per-clip caches (fused weights, folded motion tokens, RoPE tables) outside the capture.  Bit-identical to the eager loop.
            if sp is not None or cfgp is not None:
                raise NotImplementedError("hip_graph: single-GPU loop only (the sharded tiers interleave collectives)")
            timestep = torch.full((B,), float(ts[0]), dtype=torch.float32, device=latents.device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.transformer(latents, prompt_embeds, timestep, image_rotary_emb=rope_ip, image_latents=image_latents, batch=B)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                v_static = self.transformer(latents, prompt_embeds, timestep, image_rotary_emb=rope_ip, image_latents=image_latents, batch=B)
        for i, t in enumerate(ts):
            if graph is not None:
                timestep.fill_(float(t))
                graph.replay()
                v = v_static
            else:
                timestep = torch.full((B,), float(t), dtype=torch.float32, device=latents.device)
                v = self.transformer(latents, prompt_embeds, timestep, image_rotary_emb=rope_ip, image_latents=image_latents, batch=B, sp=sp)
            if cfgp is not None:
                v = cfgp.gather_branches(v.view(1, *v.shape)).view(2 * b, *v.shape[1:])     # [uncond ; cond]
            g_t = guidance_scale
            if use_dynamic_cfg:
                g_t = 1.0 + guidance_scale * ((1.0 - math.cos(math.pi * ((num_inference_steps - float(t)) / num_inference_steps) ** 5.0)) / 2.0)
            if dpm:
                sa, sb, m1, m2, m3, m4, mn, second = self.scheduler.dpm_coeffs(int(t), int(ts[i - 1]) if i > 0 else None)
                noise = draw()
                if second:
                    noise = draw()
                ops.cfg_dpm_step_(v, latents, x0_prev, noise, g_t, sa, sb, m1, m2, m3, m4, mn, second)
            else:
                sa, sb, a_t, b_t = self.scheduler.coeffs(int(t))
                ops.cfg_ddim_step_(v, latents, g_t, sa, sb, a_t, b_t)
            if callback is not None:
                callback(i, int(t), latents)
        self.action_emb = action_emb
        return latents

    @torch.no_grad()
    def __call__(self, ref_videos: torch.Tensor = None, metadata=None, *args, image=None, prompt=None, negative_prompt=None, height: int = 480,
                 width: int = 720, num_frames: int = 49, num_inference_steps: int = 50, guidance_scale: float = 6.0, generator=None,
                 latents=None, prompt_embeds=None, negative_prompt_embeds=None, image_latents=None, output_type: str = "pil",
                 max_sequence_length: int = 226, return_dict: bool = True, use_dynamic_cfg: bool = False, **kwargs):
        """pipeline.py:80-89: motion tokens first (CFG on), then the body of diffusers' CogVideoXImageToVideoPipeline.__call__ with
        do_classifier_free_guidance (guidance_scale > 1 in every shipped config): `pipe(prompt=, image=, negative_prompt=, output_type='pt',
        ref_videos=, metadata=, num_frames=, num_inference_steps=, guidance_scale=, ...)` -> output with `.frames` [b, f, c, H, W] in [0, 1]
        ('pt'), or the final latents [b, F, 16, h, w] ('latent')."""
        if args:
            raise TypeError("pass the diffusers arguments by keyword (prompt=, image=, ...)")
        if guidance_scale <= 1.0:
            raise NotImplementedError("the motion-injection path runs with classifier-free guidance (uncond motion tokens first, module.py:329)")
        dev = self._execution_device
        self.action_emb = self.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=True, image=image, **kwargs)
        pos, neg = self.encode_prompt(prompt, negative_prompt, prompt_embeds, negative_prompt_embeds, max_sequence_length)
        pe = torch.cat([neg, pos], dim=0).to(dev, torch.bfloat16).contiguous()          # [negative ; positive]
        b = pos.shape[0]
        F = (num_frames - 1) // 4 + 1
        if image_latents is None:
            if self.vae is None:
                raise ValueError("pass image_latents or a vae")
            image_latents = self.encode_image_latents(image, F, generator, height, width)
        if latents is None:                                                              # randn_tensor(shape, generator, device, dtype = bf16): drawn IN bf16 on the
            gdev = generator.device if generator is not None else torch.device("cpu")    # generator's device (a CPU generator by default), then moved
            latents = torch.randn(b, F, self.transformer.cfg["out_channels"], height // 8, width // 8, generator=generator, dtype=torch.bfloat16,
                                  device=gdev).to(dev)                                   # init_noise_sigma == 1 for both schedulers
        latents = self.denoise(latents.to(dev, torch.bfloat16).contiguous(), image_latents.to(dev, torch.bfloat16).contiguous(), pe,
                               self.action_emb, num_inference_steps, guidance_scale, generator=generator, use_dynamic_cfg=use_dynamic_cfg)
        if output_type == "latent":
            frames = latents
        else:
            if self.vae is None:
                raise ValueError("output_type != 'latent' needs a vae")
            video = self.decode_latents(latents)                                         # [b, 3, f, H, W] in [-1, 1]
            frames = (video / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)                # video_processor.postprocess_video(output_type='pt')
            if output_type not in ("pt", "np"):
                raise NotImplementedError("output_type 'pt', 'np' or 'latent' (PIL conversion is host-side glue outside the hot path)")
            if output_type == "np":
                frames = frames.float().cpu().numpy()
        return CogVideoXPipelineOutput(frames) if return_dict else (frames,)


class CogVideoXImageToVideoPipeline(CogVideoXImageToVideoActionPipeline):
    """The pipeline WITHOUT motion injection -- diffusers' `CogVideoXImageToVideoPipeline` as the reference's baseline module runs it (`CogVideoX5B`,
    cogvideox/module.py:16-79, configs/cogvideox/baseline_open.yml; the "CogVideoX baseline" row of the reference's README): same prompt / image / loop / decode
    path, plain joint attention in every block, `pipe(prompt=, image=, negative_prompt=, output_type='pt', ...)`."""

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer: CogVideoXTransformer3DModel = None, scheduler=None):
        super().__init__(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, transformer=transformer, scheduler=scheduler)

    def prepare_action_embeddings(self, ref_videos=None, metadata=None, do_classifier_free_guidance: bool = False, *args, **kwargs):
        return None                                                                  # no motion tokens: `joint_attention_core` skips the adapter branch


class CogVideoXImageToVideoCTPipeline(CogVideoXImageToVideoActionPipeline):
    """Stage-2 pipeline (pipeline.py:92-130): motion tokens predicted by the CAMA `condition_transformer` from the references + the
    conditioning image.  Keyword construction mirrors the reference; `CogVideoXImageToVideoCTPipeline(transformer, scheduler,
    condition_transformer=...)` (round-1 positional form) is kept."""

    def __init__(self, *args, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None, condition_transformer=None):
        if args:                                                                     # (transformer, scheduler[, condition_transformer])
            transformer, scheduler = args[0], args[1]
            condition_transformer = args[2] if len(args) > 2 else condition_transformer
        super().__init__(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, transformer=transformer, scheduler=scheduler)
        self.condition_transformer = condition_transformer

    def prepare_action_embeddings(self, ref_videos: torch.Tensor, metadata=None, do_classifier_free_guidance: bool = False, *args, **kwargs):
        """pipeline.py:117-130"""
        image = kwargs.get("image").to(ref_videos.device, ref_videos.dtype)
        batch_ = {"ref_videos": ref_videos, "video": image[:, None].expand(-1, ref_videos.size(2), -1, -1, -1)}
        return self.condition_transformer.predict(batch_, do_classifier_free_guidance=do_classifier_free_guidance)


def set_attention_processors(transformer: CogVideoXTransformer3DModel, adapter_modules, cross_attention_dim: int, scale: float = 1.0) -> None:
    """cogvideox/module.py:163-175: install `APAdapterCogVideoXAttnProcessor2_0` on the processor names listed in `adapter_modules`
    (`transformer_blocks.{i}.attn1.processor`, configs/cogvideox/MotionRAG_open.yml), keep the others."""
    hidden = transformer.cfg["dim"]
    attn = {}
    for name, orig in transformer.attn_processors.items():
        attn[name] = APAdapterCogVideoXAttnProcessor2_0(hidden, cross_attention_dim, scale=scale) if name in adapter_modules else orig
    transformer.set_attn_processor(attn)


_FRAME_PICKERS = {
    "first": lambda n: slice(0, 16),
    "uniform": lambda n: torch.linspace(0, n - 1, 16).round().long(),
    None: lambda n: slice(None),
}


def eval_pipeline(pipe, image, positive_prompt, negative_prompt, dtype, ref_videos, metadata, *args, **kwargs) -> torch.Tensor:
    """CogVideoX5BAction.eval_pipeline (cogvideox/module.py:197-223): image in [-1, 1] -> video [b, 16 | all, c, H, W] in [-1, 1].

    `kwargs` is the YAML's `eval_pipeline_call_kwargs`.  Its `scheduler` entry never reaches the reference's pipeline call: module.py:28-35 pops it
    while the module is configured, validates it ('ddim' | 'dpm') and installs it on the BASELINE pipe only; the motion-injected Action / CT pipelines
    are built with `scheduler=self.scheduler`, which is always the DPM scheduler (module.py:44, 183, 255).  Same rule here: the entry is validated,
    installed on a baseline `CogVideoXImageToVideoPipeline`, and otherwise dropped -- an Action / CT pipeline samples with the scheduler it was built
    with (`scheduler=None` in its constructor builds the reference's DPM scheduler)."""
    chosen = make_scheduler(kwargs.pop("scheduler", "ddim"))                         # unknown names raise as the reference's configure step does
    if isinstance(pipe, CogVideoXImageToVideoPipeline):
        pipe.scheduler = chosen
    pick = kwargs.pop("sample_method", "first")
    if pick not in _FRAME_PICKERS:
        raise ValueError(f"Unknown sample method: {pick}")
    out = pipe(prompt=positive_prompt, image=image / 2 + 0.5,                  # [-1, 1] -> [0, 1]: the pipeline's image convention
               negative_prompt=negative_prompt, output_type="pt", ref_videos=ref_videos, metadata=metadata, *args, **kwargs)
    video = out[0]
    idx = _FRAME_PICKERS[pick](video.shape[1])
    video = video[:, idx.to(video.device) if isinstance(idx, torch.Tensor) else idx]
    return video * 2 - 1
