"""ORACLE (test infrastructure only -- never imported by the product path).

CPU / fp32 restatement of the CAMA hot path of MCG-NJU/MotionRAG, written functionally over plain
state dicts (same key names as the reference checkpoints).  Each function cites
the reference lines it follows.  Pinned against the reference's own importable code by
`oracle/gen_golden.py` -> `tests/golden/*.npz` (see tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def sub(sd: SD, prefix: str) -> SD:
    """state-dict slice: keys under `prefix.` with the prefix stripped"""
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


# ----------------------------------------------------------------------------------------------
# position embeddings / mask / fusion
# ----------------------------------------------------------------------------------------------
def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """src/projects/condition/position_embeddings.py:158-170: angle = pos / 10000^(2*(j//2)/d) in
    float64, sin on even columns, cos on odd columns, cast to fp32; shape [1, n_position, d_hid]."""
    j = np.arange(d_hid)
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    table = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.from_numpy(table.astype(np.float32)).unsqueeze(0)


def sinusoid_pe(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """position_embeddings.py:172-174: x + table[:, :L]"""
    assert x.size(-2) <= table.size(1)
    return x + table[:, : x.size(-2)].to(x.dtype)


def block_causal_mask(num_frames: int, frame_tokens: int) -> torch.Tensor:
    """src/projects/condition/module.py:131-135: bool [n*l, n*l], True = blocked; rows of frame i may
    see columns < (i+1)*l."""
    n = num_frames * frame_tokens
    row_frame = torch.arange(n) // frame_tokens
    col = torch.arange(n)
    return col[None, :] >= ((row_frame + 1) * frame_tokens)[:, None]


def condition_fusion(emb: torch.Tensor, fusion_type: str = "mean", weight: Optional[Iterable] = None) -> torch.Tensor:
    """src/projects/condition/utils.py:7-36 over [b, k, l, c]."""
    assert fusion_type in ("mean", "concat", "top1", "weight") and emb.dim() == 4
    if fusion_type == "mean":
        return emb.mean(dim=1)
    if fusion_type == "weight":
        d = torch.as_tensor(weight, dtype=emb.dtype)
        w = (1 - d) / (1 - d).sum(dim=1, keepdim=True)
        return (emb * w[..., None, None]).sum(dim=1)
    if fusion_type == "concat":
        b, k, l, c = emb.shape
        return emb.reshape(b, k * l, c)
    return emb[:, 0]


# ----------------------------------------------------------------------------------------------
# Perceiver resampler
# ----------------------------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: Optional[torch.Tensor], b: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask_blocked: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v on [b, h, l, d]; mask_blocked bool [lq, lk], True = blocked."""
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if mask_blocked is not None:
        s = s.masked_fill(mask_blocked, float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v)


def _heads(x: torch.Tensor, h: int) -> torch.Tensor:
    b, l, w = x.shape
    return x.view(b, l, h, w // h).transpose(1, 2)


def perceiver_attention(sd: SD, x: torch.Tensor, latents: torch.Tensor, heads: int) -> torch.Tensor:
    """encoders/resampler.py:81-105: x=LN1(x), latents=LN2(latents), q from latents, k/v from
    cat(x, latents) through one to_kv (chunk(2): K rows first), SDPA, to_out; all bias-free."""
    x = layer_norm(x, sd["norm1.weight"], sd["norm1.bias"])
    latents = layer_norm(latents, sd["norm2.weight"], sd["norm2.bias"])
    q = F.linear(latents, sd["to_q.weight"])
    kv = F.linear(torch.cat((x, latents), dim=-2), sd["to_kv.weight"])
    k, v = kv.chunk(2, dim=-1)
    out = sdpa(_heads(q, heads), _heads(k, heads), _heads(v, heads))
    out = out.transpose(1, 2).reshape(latents.shape[0], latents.shape[1], -1)
    return F.linear(out, sd["to_out.weight"])


def feed_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """encoders/resampler.py:45-52: LN -> Linear(no bias) -> GELU(erf) -> Linear(no bias)."""
    h = layer_norm(x, sd["0.weight"], sd["0.bias"])
    return F.linear(F.gelu(F.linear(h, sd["1.weight"])), sd["3.weight"])


def resampler(sd: SD, x: torch.Tensor, heads: int, depth: int) -> torch.Tensor:
    """encoders/resampler.py:157-174 (no cls token): latents.repeat(N); proj_in; depth x
    {latents += attn(x, latents); latents += ff(latents)}; norm_out(proj_out(latents))."""
    latents = sd["latents"].repeat(x.size(0), 1, 1)
    x = F.linear(x, sd["proj_in.weight"], sd["proj_in.bias"])
    for i in range(depth):
        latents = perceiver_attention(sub(sd, f"layers.{i}.0"), x, latents, heads) + latents
        latents = feed_forward(sub(sd, f"layers.{i}.1"), latents) + latents
    latents = F.linear(latents, sd["proj_out.weight"], sd["proj_out.bias"])
    return layer_norm(latents, sd["norm_out.weight"], sd["norm_out.bias"])


# ----------------------------------------------------------------------------------------------
# torch.nn.TransformerEncoder (post-norm, GELU-erf, batch_first) -- third-party torch; restated so
# that the oracle does not depend on torch's fused fast path
# ----------------------------------------------------------------------------------------------
def encoder_layer(sd: SD, x: torch.Tensor, mask_blocked: torch.Tensor, nhead: int, eps: float = 1e-5) -> torch.Tensor:
    b, l, d = x.shape
    qkv = F.linear(x, sd["self_attn.in_proj_weight"], sd["self_attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    a = sdpa(_heads(q, nhead), _heads(k, nhead), _heads(v, nhead), mask_blocked)
    a = a.transpose(1, 2).reshape(b, l, d)
    a = F.linear(a, sd["self_attn.out_proj.weight"], sd["self_attn.out_proj.bias"])
    x = layer_norm(x + a, sd["norm1.weight"], sd["norm1.bias"], eps)
    f = F.linear(F.gelu(F.linear(x, sd["linear1.weight"], sd["linear1.bias"])), sd["linear2.weight"], sd["linear2.bias"])
    return layer_norm(x + f, sd["norm2.weight"], sd["norm2.bias"], eps)


def transformer_encoder(sd: SD, x: torch.Tensor, mask_blocked: torch.Tensor, nhead: int, num_layers: int) -> torch.Tensor:
    for i in range(num_layers):
        x = encoder_layer(sub(sd, f"layers.{i}"), x, mask_blocked, nhead)
    return x


# ----------------------------------------------------------------------------------------------
# ActionTransformer (CAMA).  The frozen VideoMAE / DINOv2 encoders are third-party and out of scope
# (SURVEY 2.1 #6): the functions take their OUTPUT features.
# ----------------------------------------------------------------------------------------------
class CamaSpec:
    def __init__(self, heads=12, depth=4, nhead=16, num_layers=4, tokens=25, dim=1024, vision_pe_len=256, cond_pe_len=2560):
        self.heads, self.depth, self.nhead, self.num_layers = heads, depth, nhead, num_layers
        self.tokens, self.dim = tokens, dim
        self.vision_pe = sinusoid_table(vision_pe_len, dim)
        self.cond_pe = sinusoid_table(cond_pe_len, dim)


def encode_vision(sd: SD, spec: CamaSpec, feats: torch.Tensor, b: int) -> torch.Tensor:
    """module.py:264-268: Resampler over VideoMAE features [(b k), n, 768] -> [b, k, l, c]."""
    e = resampler(sub(sd, "vision_proj"), feats, spec.heads, spec.depth)
    return e.view(b, -1, e.shape[-2], e.shape[-1])


def encode_condition(sd: SD, spec: CamaSpec, feats: torch.Tensor, b: int) -> torch.Tensor:
    """module.py:270-276 -> :137-143: Resampler over DINOv2 features [(b k), 257, 1024], sinusoid PE
    applied PER IMAGE over its 25 positions, then '(b k) l c -> b (k l) c'."""
    e = resampler(sub(sd, "condition_proj"), feats, spec.heads, spec.depth)
    e = sinusoid_pe(e, spec.cond_pe)
    return e.reshape(b, -1, e.shape[-1])


def cama_forward(sd: SD, spec: CamaSpec, vision_feats: torch.Tensor, cond_feats: torch.Tensor, b: int) -> torch.Tensor:
    """module.py:292-315 (return_loss=False): x = cat(sos, vision_emb[:, :-1]) + vision_pe + condition_emb;
    block-causal TransformerEncoder; -> [b, K, l, c]."""
    vision_emb = encode_vision(sd, spec, vision_feats, b)
    cond_emb = encode_condition(sd, spec, cond_feats, b)
    _, K, l, d = vision_emb.shape
    x = torch.cat([sd["sos_token"].repeat(b, 1, 1), vision_emb[:, :-1].reshape(b, (K - 1) * l, d)], dim=1)
    x = sinusoid_pe(x, spec.vision_pe)
    x = x + cond_emb
    mask = block_causal_mask(K, l)
    y = transformer_encoder(sub(sd, "transformer"), x, mask, spec.nhead, spec.num_layers)
    return y.view(b, K, l, d)


def cama_predict(sd: SD, spec: CamaSpec, vision_feats: torch.Tensor, cond_feats: torch.Tensor, uncond_feats: Optional[torch.Tensor],
                 b: int) -> torch.Tensor:
    """module.py:325-331: answer = forward(...)[:, -1]; with CFG, uncond = encode_vision(zeros)[:, 0] and
    the result is cat([uncond, answer]) (uncond FIRST).  vision_feats already follow batch_forward's
    order (refs flipped, target last: module.py:317-323); uncond_feats are the encoder features of the
    all-zero clip."""
    ans = cama_forward(sd, spec, vision_feats, cond_feats, b)[:, -1]
    if uncond_feats is None:
        return ans
    un = encode_vision(sd, spec, uncond_feats, b)[:, 0]
    return torch.cat([un, ans], dim=0)


# ----------------------------------------------------------------------------------------------
# random-init weights with the checkpoint's key layout
# ----------------------------------------------------------------------------------------------
def random_resampler_sd(g: torch.Generator, embedding_dim: int, dim=1024, heads=12, dim_head=64, depth=4, tokens=25, out_dim=1024,
                        ff_mult=4, std=0.02) -> SD:
    inner = heads * dim_head
    r = lambda *s: torch.randn(*s, generator=g) * std
    sd = {"latents": torch.randn(1, tokens, dim, generator=g) / dim ** 0.5,
          "proj_in.weight": r(dim, embedding_dim), "proj_in.bias": r(dim),
          "proj_out.weight": r(out_dim, dim), "proj_out.bias": r(out_dim),
          "norm_out.weight": 1 + r(out_dim), "norm_out.bias": r(out_dim)}
    for i in range(depth):
        p = f"layers.{i}."
        sd.update({p + "0.norm1.weight": 1 + r(dim), p + "0.norm1.bias": r(dim), p + "0.norm2.weight": 1 + r(dim),
                   p + "0.norm2.bias": r(dim), p + "0.to_q.weight": r(inner, dim), p + "0.to_kv.weight": r(2 * inner, dim),
                   p + "0.to_out.weight": r(dim, inner), p + "1.0.weight": 1 + r(dim), p + "1.0.bias": r(dim),
                   p + "1.1.weight": r(dim * ff_mult, dim), p + "1.3.weight": r(dim, dim * ff_mult)})
    return sd


def random_encoder_sd(g: torch.Generator, d=1024, ff=4096, layers=4, std=0.02) -> SD:
    r = lambda *s: torch.randn(*s, generator=g) * std
    sd = {}
    for i in range(layers):
        p = f"layers.{i}."
        sd.update({p + "self_attn.in_proj_weight": r(3 * d, d), p + "self_attn.in_proj_bias": r(3 * d),
                   p + "self_attn.out_proj.weight": r(d, d), p + "self_attn.out_proj.bias": r(d),
                   p + "linear1.weight": r(ff, d), p + "linear1.bias": r(ff), p + "linear2.weight": r(d, ff),
                   p + "linear2.bias": r(d), p + "norm1.weight": 1 + r(d), p + "norm1.bias": r(d),
                   p + "norm2.weight": 1 + r(d), p + "norm2.bias": r(d)})
    return sd


def random_cama_sd(seed: int = 0, dim=1024, tokens=25, vision_dim=768, cond_dim=1024, heads=12, depth=4, ff=4096, layers=4) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd = {"sos_token": torch.randn(1, tokens, dim, generator=g) / dim ** 0.5}
    for k, v in random_resampler_sd(g, vision_dim, dim, heads, 64, depth, tokens, dim).items():
        sd["vision_proj." + k] = v
    for k, v in random_resampler_sd(g, cond_dim, dim, heads, 64, depth, tokens, dim).items():
        sd["condition_proj." + k] = v
    for k, v in random_encoder_sd(g, dim, ff, layers).items():
        sd["transformer." + k] = v
    return sd
