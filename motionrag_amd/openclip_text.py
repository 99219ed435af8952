"""OpenCLIP text tower on the HIP kernels: DynamiCrafter's prompt encoder.

Mirrors `FrozenOpenCLIPEmbedder` (src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/encoders/condition.py:177-240): `forward(text)` tokenizes (the tokenizer is the
caller's: `open_clip.tokenize` is third-party), `encode_with_transformer(tokens)` = token + positional embedding -> all but the last `layer_idx` residual attention blocks
under the causal mask -> `ln_final`; `return_cls_tokens` picks the end-of-text position (`tokens.argmax(-1)`).  `OpenCLIPTextModel` keeps open_clip's parameter names
(`token_embedding.weight`, `positional_embedding`, `transformer.resblocks.N.{ln_1, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj, ln_2, mlp.c_fc, mlp.c_proj}`,
`ln_final`, `text_projection`, `logit_scale`), so the text half of an open_clip state dict loads as it is (`strict=False` skips `visual.*`).

Per block: LayerNorm -> fused in_proj GEMM -> head_dim-64 attention with the [77, 77] causal byte mask -> out_proj with the residual in its epilogue -> LayerNorm ->
c_fc + GELU(erf) epilogue -> c_proj + residual epilogue.  ViT-H-14's text tower: 24 x 1024, 16 heads of 64.  GPU only."""
from typing import Callable, Optional

import torch
from torch import nn

from . import ops


def _b(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


class _MHA(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * d ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)


class _MLP(nn.Module):
    def __init__(self, d: int, ff: int):
        super().__init__()
        self.c_fc = nn.Linear(d, ff)
        self.c_proj = nn.Linear(ff, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d: int, heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.heads = heads
        self.ln_1 = nn.LayerNorm(d)
        self.attn = _MHA(d)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = _MLP(d, int(d * mlp_ratio))

    def forward(self, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        B, S, D = x.shape
        h = ops.layernorm(x, _b(self.ln_1.weight), _b(self.ln_1.bias), self.ln_1.eps)
        qkv = ops.linear(h, _b(self.attn.in_proj_weight), _b(self.attn.in_proj_bias)).view(B, S, 3, self.heads, 64)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], mask=mask)
        x = ops.linear(a, _b(self.attn.out_proj.weight), _b(self.attn.out_proj.bias), epilogue=ops.EPI_RESID, resid=x)
        h = ops.layernorm(x, _b(self.ln_2.weight), _b(self.ln_2.bias), self.ln_2.eps)
        h = ops.linear(h, _b(self.mlp.c_fc.weight), _b(self.mlp.c_fc.bias), epilogue=ops.EPI_GELU_ERF)
        return ops.linear(h, _b(self.mlp.c_proj.weight), _b(self.mlp.c_proj.bias), epilogue=ops.EPI_RESID, resid=x)


class _Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.resblocks = nn.ModuleList(ResidualAttentionBlock(width, heads) for _ in range(layers))
        self.grad_checkpointing = False


class OpenCLIPTextModel(nn.Module):
    """the text half of an open_clip CLIP model (defaults: ViT-H-14 / laion2b_s32b_b79k)"""

    def __init__(self, vocab_size: int = 49408, width: int = 1024, heads: int = 16, layers: int = 24, context_length: int = 77, embed_dim: int = 1024):
        super().__init__()
        if width != 64 * heads:
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 64")
        self.context_length = context_length
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.randn(context_length, width) * 0.01)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.randn(width, embed_dim) * width ** -0.5)      # unused by the embedder (no pooled output), kept for the key layout
        self.logit_scale = nn.Parameter(torch.ones([]))
        self._mask = {}

    def causal_mask(self, n: int, device) -> torch.Tensor:
        """open_clip `build_attention_mask` as the kernel's byte mask: nonzero = blocked (key after the query)"""
        key = (n, device)
        if key not in self._mask:
            self._mask[key] = torch.ones(n, n, dtype=torch.bool).triu_(1).to(device).contiguous()
        return self._mask[key]


class FrozenOpenCLIPEmbedder(nn.Module):
    """lvdm/modules/encoders/condition.py:177-240.  `model`: an `OpenCLIPTextModel` (or keyword config for one); `tokenizer`: text list -> LongTensor [B, 77]."""

    LAYERS = ["last", "penultimate"]

    def __init__(self, model=None, tokenizer: Optional[Callable] = None, device="cuda", max_length: int = 77, freeze: bool = True, layer: str = "last", **config):
        super().__init__()
        assert layer in self.LAYERS
        self.model = model if isinstance(model, nn.Module) else OpenCLIPTextModel(**config)
        self.tokenizer, self.device, self.max_length = tokenizer, device, max_length
        self.layer, self.layer_idx = layer, {"last": 0, "penultimate": 1}[layer]
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def forward(self, text, return_cls_tokens: bool = False):
        if isinstance(text, torch.Tensor):
            tokens = text
        else:
            if self.tokenizer is None:
                raise ValueError("FrozenOpenCLIPEmbedder: pass token ids or construct with tokenizer= (open_clip.tokenize is third-party)")
            tokens = self.tokenizer(text)
        tokens = tokens.to(self.device)
        z = self.encode_with_transformer(tokens)
        if return_cls_tokens:
            return z[torch.arange(tokens.shape[0], device=z.device), tokens.argmax(-1)], z
        return z

    @torch.no_grad()
    def encode_with_transformer(self, text: torch.Tensor) -> torch.Tensor:
        if not text.is_cuda:
            raise ops.HipOnly("FrozenOpenCLIPEmbedder: token ids on the GPU expected")
        m = self.model
        x = ops.add_rows(_b(m.token_embedding.weight)[text].contiguous(), _b(m.positional_embedding)[:text.shape[1]].contiguous())     # :221-222
        mask = m.causal_mask(text.shape[1], text.device)
        blocks = m.transformer.resblocks
        for i, r in enumerate(blocks):                                                                                                      # :229-237
            if i == len(blocks) - self.layer_idx:
                break
            x = r(x, mask)
        return ops.layernorm(x, _b(m.ln_final.weight), _b(m.ln_final.bias), m.ln_final.eps)                                               # :226

    def encode(self, text):
        return self(text)
