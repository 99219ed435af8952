"""TEST INFRASTRUCTURE -- CPU restatement (torch fp32, functional, over a plain state dict) of the retrieval side's text embedder: the
`sentence-transformers` model `Alibaba-NLP/gte-base-en-v1.5` that the reference registers as the table's embedding function
(`tools/build_rag_database.py:16-33`, `src/data/rag.py:13-15`; queries are embedded by it inside `table.search(text)`).

PARITY UNPINNED, twice over: the model class is remote code (`Alibaba-NLP/new-impl`, `modeling.py: NewModel`, loaded with trust_remote_code) and the
wrapper is lancedb's `sentence-transformers` registry entry; neither is installed or vendored here.  This file restates the PUBLISHED architecture as
configured by the model's `config.json` -- verify against the real model before relying on it:

  * embeddings: word embedding (vocab 30 528, no token-type table: type_vocab_size 0, no absolute positions) -> LayerNorm(eps 1e-12);
  * 12 post-norm layers of width 768, 12 heads of 64:  qkv = qkv_proj(x) (one Linear with bias, split q | k | v); rotary position embedding on q and k
    (`rotate_half`: dimension i pairs with i + 32); softmax(q k^T / 8 + padding bias) v; o_proj (bias); x = attn_ln(x + attn);
    gated MLP: up_gate_proj (no bias) -> [up | gate], gelu_erf(gate) * up, down_proj (bias); x = mlp_ln(x + mlp);
  * rotary table: NTK scaling with factor 2 over max_position_embeddings 8192 and rope_theta 500 000 -- the class builds its cos / sin cache for
    factor * 8192 positions at construction, which is past max_position_embeddings, so EVERY position uses the scaled frequencies
    inv_freq_i = (theta * factor)^(-2i/64) / factor^(2/64);
  * sentence embedding: the [CLS] (first) token's last hidden state, L2-normalised (lancedb's wrapper: `normalize=True`).

Only tests/ may import this module."""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
CONFIG_BASE = dict(vocab_size=30528, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12,
                   max_position_embeddings=8192, rope_theta=500000.0, rope_scaling_factor=2.0)


def state_shapes(cfg: dict) -> Dict[str, tuple]:
    d, ff = cfg["hidden_size"], cfg["intermediate_size"]
    out = {"embeddings.word_embeddings.weight": (cfg["vocab_size"], d), "embeddings.LayerNorm.weight": (d,), "embeddings.LayerNorm.bias": (d,)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{i}"
        out.update({f"{p}.attention.qkv_proj.weight": (3 * d, d), f"{p}.attention.qkv_proj.bias": (3 * d,), f"{p}.attention.o_proj.weight": (d, d),
                    f"{p}.attention.o_proj.bias": (d,), f"{p}.attn_ln.weight": (d,), f"{p}.attn_ln.bias": (d,), f"{p}.mlp.up_gate_proj.weight": (2 * ff, d),
                    f"{p}.mlp.down_proj.weight": (d, ff), f"{p}.mlp.down_proj.bias": (d,), f"{p}.mlp_ln.weight": (d,), f"{p}.mlp_ln.bias": (d,)})
    return out


def seeded_state(cfg: dict, seed: int) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in state_shapes(cfg).items():
        if k.endswith("ln.weight") or k.endswith("LayerNorm.weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif "word_embeddings" in k:
            sd[k] = torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) * (1.0 / shp[1] ** 0.5)
    return sd


def rope_inv_freq(cfg: dict, head_dim: int = 64) -> torch.Tensor:
    factor = cfg["rope_scaling_factor"]
    base = cfg["rope_theta"] * factor
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    return inv / factor ** (2.0 / head_dim)


def rope_tables(cfg: dict, seq_len: int, head_dim: int = 64):
    freqs = torch.einsum("i,j->ij", torch.arange(seq_len, dtype=torch.float32), rope_inv_freq(cfg, head_dim))
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()                                   # [S, head_dim]


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def encoder(sd: SD, cfg: dict, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NewModel(...).last_hidden_state: input_ids [B, S] -> [B, S, hidden]"""
    d, H, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    hd = d // H
    B, S = input_ids.shape
    x = F.layer_norm(sd["embeddings.word_embeddings.weight"][input_ids], (d,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    cos, sin = rope_tables(cfg, S, hd)
    bias = None
    if attention_mask is not None:
        bias = (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{i}"
        q, k, v = F.linear(x, sd[f"{p}.attention.qkv_proj.weight"], sd[f"{p}.attention.qkv_proj.bias"]).split(d, dim=-1)
        q, k, v = (t.view(B, S, H, hd).transpose(1, 2) for t in (q, k, v))
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        sc = q @ k.transpose(-1, -2) / hd ** 0.5
        if bias is not None:
            sc = sc + bias
        a = (sc.softmax(-1) @ v).transpose(1, 2).reshape(B, S, d)
        x = F.layer_norm(x + F.linear(a, sd[f"{p}.attention.o_proj.weight"], sd[f"{p}.attention.o_proj.bias"]), (d,), sd[f"{p}.attn_ln.weight"], sd[f"{p}.attn_ln.bias"], eps)
        up, gate = F.linear(x, sd[f"{p}.mlp.up_gate_proj.weight"]).split(cfg["intermediate_size"], dim=-1)
        m = F.linear(F.gelu(gate) * up, sd[f"{p}.mlp.down_proj.weight"], sd[f"{p}.mlp.down_proj.bias"])
        x = F.layer_norm(x + m, (d,), sd[f"{p}.mlp_ln.weight"], sd[f"{p}.mlp_ln.bias"], eps)
    return x


def sentence_embedding(sd: SD, cfg: dict, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, normalize: bool = True) -> torch.Tensor:
    """CLS pooling + L2 normalisation: [B, hidden]"""
    e = encoder(sd, cfg, input_ids, attention_mask)[:, 0]
    return F.normalize(e, p=2, dim=1) if normalize else e
