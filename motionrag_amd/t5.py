"""T5 v1.1 encoder on the HIP kernels: the prompt encoder of the CogVideoX path.

The reference loads `T5EncoderModel.from_pretrained(..., subfolder="text_encoder")` (src/projects/cogvideox/module.py:86-90) and diffusers' pipeline calls
`text_encoder(text_input_ids)[0]` on 226 max-length-padded tokens WITHOUT an attention mask.  `T5EncoderModel` here keeps the third-party class's state-dict keys
(`shared.weight`, `encoder.block.N.layer.0.SelfAttention.{q,k,v,o}.weight`, `...relative_attention_bias.weight` in block 0, `...layer.1.DenseReluDense.{wi_0,wi_1,wo}`,
`...layer_norm.weight`, `encoder.final_layer_norm.weight`), so `load_state_dict(hf_model.state_dict())` is the hand-over, and the call returns an object with
`.last_hidden_state` that also indexes as `[0]`.

Per layer, all on `libmrag_hip.so`:
  T5LayerNorm                       -> `mrag_layernorm_bf16` in RMS mode (no mean subtraction, no bias)
  q | k | v                         -> one fused GEMM (no bias)
  softmax(q k^T + position_bias) v  -> `mrag_attn_fwd_bf16` with scale 1 and the additive fp32 bias [H, S, S] (T5 does not scale by 1 / sqrt(d));
                                       the bias table is built once per sequence length on the host from block 0's bucket embedding
  o + residual                      -> GEMM epilogue
  wi_0 | wi_1 -> gelu_new(.) * (.)  -> one GEMM with the gated-GELU (tanh) epilogue, rows interleaved as the GEGLU epilogue wants them
  wo + residual                     -> GEMM epilogue
GPU only.  head_dim (d_kv) must be 64 -- true for every T5 v1.1 size.
"""
import math
from typing import Optional

import torch
from torch import nn

from . import ops
from .dynamicrafter import _CACHE


def _b(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


class T5LayerNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps


class _Attention(nn.Module):
    def __init__(self, d_model: int, inner: int, heads: int, has_bias_table: bool, num_buckets: int):
        super().__init__()
        self.q, self.k, self.v = (nn.Linear(d_model, inner, bias=False) for _ in range(3))
        self.o = nn.Linear(inner, d_model, bias=False)
        if has_bias_table:
            self.relative_attention_bias = nn.Embedding(num_buckets, heads)


class _Holder(nn.Module):
    pass


class _FF(nn.Module):
    def __init__(self, d_model: int, d_ff: int):
        super().__init__()
        self.wi_0 = nn.Linear(d_model, d_ff, bias=False)
        self.wi_1 = nn.Linear(d_model, d_ff, bias=False)
        self.wo = nn.Linear(d_ff, d_model, bias=False)


def _sublayer(inner: nn.Module, name: str, d_model: int, eps: float) -> nn.Module:
    m = _Holder()
    setattr(m, name, inner)
    m.layer_norm = T5LayerNorm(d_model, eps)
    return m


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """T5Attention._relative_position_bucket, bidirectional (the encoder's)"""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rp = torch.abs(relative_position)
    max_exact = num_buckets // 2
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(rp < max_exact, rp, large)


class T5Output:
    def __init__(self, last_hidden_state: torch.Tensor):
        self.last_hidden_state = last_hidden_state

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


class T5EncoderModel(nn.Module):
    """transformers `T5EncoderModel` (v1.1: gated-GELU, no biases), state-dict compatible.  Defaults = the T5-v1.1-XXL encoder CogVideoX ships."""

    def __init__(self, vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64, relative_attention_num_buckets=32,
                 relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu", **_unused):
        super().__init__()
        if d_kv != 64:
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 64")
        if feed_forward_proj != "gated-gelu":
            raise NotImplementedError("T5 v1.1 (gated-gelu) only")
        self.d_model, self.heads, self.eps = d_model, num_heads, layer_norm_epsilon
        self.num_buckets, self.max_distance = relative_attention_num_buckets, relative_attention_max_distance
        inner = num_heads * d_kv
        self.shared = nn.Embedding(vocab_size, d_model)
        self.encoder = _Holder()
        self.encoder.embed_tokens = self.shared                                          # tied, as in transformers (both keys appear in the state dict)
        blocks = []
        for i in range(num_layers):
            blk = _Holder()
            blk.layer = nn.ModuleList([_sublayer(_Attention(d_model, inner, num_heads, i == 0, relative_attention_num_buckets), "SelfAttention", d_model, layer_norm_epsilon),
                                       _sublayer(_FF(d_model, d_ff), "DenseReluDense", d_model, layer_norm_epsilon)])
            blocks.append(blk)
        self.encoder.block = nn.ModuleList(blocks)
        self.encoder.final_layer_norm = T5LayerNorm(d_model, layer_norm_epsilon)

    def _position_bias(self, S: int, device) -> torch.Tensor:
        """T5Attention.compute_bias: fp32 [H, S, S] from block 0's [num_buckets, H] table; built once per S (host arithmetic through torch, then resident)"""
        tab = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight

        def build():
            ctx = torch.arange(S, device=device)[:, None]
            mem = torch.arange(S, device=device)[None, :]
            b = relative_position_bucket(mem - ctx, self.num_buckets, self.max_distance)
            return tab.detach().float()[b].permute(2, 0, 1).contiguous()
        return _CACHE.get(("t5_bias", id(self), S), tab, build)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, **_unused) -> T5Output:
        if not input_ids.is_cuda:
            raise ops.HipOnly("T5EncoderModel: GPU tensors only")
        B, S = input_ids.shape
        H = self.heads
        x = _b(self.shared.weight)[input_ids].contiguous()                               # embedding row gather (plumbing)
        bias = self._position_bias(S, input_ids.device)
        biases = None
        if attention_mask is not None:                                                   # key-padding mask: folded into a per-sample copy of the bias
            neg = torch.finfo(torch.float32).min
            biases = [(bias + (1.0 - attention_mask[b, None, None, :].float()) * neg).contiguous() for b in range(B)]
        for blk in self.encoder.block:
            sa, ff = blk.layer[0], blk.layer[1]
            att, dn = sa.SelfAttention, ff.DenseReluDense
            h = ops.layernorm(x, _b(sa.layer_norm.weight), None, sa.layer_norm.variance_epsilon, rms=True)
            wqkv = _CACHE.get(("t5_qkv", id(att)), (att.q.weight, att.k.weight, att.v.weight),
                              lambda: torch.cat([_b(att.q.weight), _b(att.k.weight), _b(att.v.weight)], 0).contiguous())
            qkv = ops.linear(h, wqkv).view(B, S, 3, H, 64)
            if biases is None:
                a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], scale=1.0, bias=bias)
            else:
                a = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device=x.device)
                for b in range(B):
                    ops.attention(qkv[b:b + 1, :, 0], qkv[b:b + 1, :, 1], qkv[b:b + 1, :, 2], scale=1.0, bias=biases[b], out=a[b:b + 1])
            x = ops.linear(a, _b(att.o.weight), epilogue=ops.EPI_RESID, resid=x)
            h = ops.layernorm(x, _b(ff.layer_norm.weight), None, ff.layer_norm.variance_epsilon, rms=True)
            # hidden_gelu = gelu_new(wi_0 h), hidden_linear = wi_1 h, product: the GEGLU epilogue with value = wi_1, gate = wi_0
            wg = _CACHE.get(("t5_geglu", id(dn)), (dn.wi_0.weight, dn.wi_1.weight),
                            lambda: ops.geglu_interleave(torch.cat([_b(dn.wi_1.weight), _b(dn.wi_0.weight)], 0), None)[0])
            g = ops.linear(h, wg, epilogue=ops.EPI_GEGLU, geglu_tanh=True)
            x = ops.linear(g, _b(dn.wo.weight), epilogue=ops.EPI_RESID, resid=x)
        fl = self.encoder.final_layer_norm
        return T5Output(ops.layernorm(x, _b(fl.weight), None, fl.variance_epsilon, rms=True))
