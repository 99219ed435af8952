"""Query / caption text embedder of the retrieval row on the HIP kernels: the encoder the reference registers as the LanceDB table's embedding
function -- sentence-transformers `Alibaba-NLP/gte-base-en-v1.5` in bf16 (`tools/build_rag_database.py:16-33`; `src/data/rag.py:13-15` moves it to the query
device; `table.search(text)` embeds the query with it).

The model class is remote code (`NewModel`, trust_remote_code) behind two third-party wrappers, none of them installed here: `NewModel` below carries the
published parameter names (`embeddings.word_embeddings`, `encoder.layer.N.attention.{qkv_proj, o_proj}`, `attn_ln`, `mlp.{up_gate_proj, down_proj}`, `mlp_ln`), so
the checkpoint's `model.safetensors` loads unchanged; the arithmetic follows `oracle/gte_ref.py` -- PARITY UNPINNED, verify against the real model before use.

Data path per layer (bf16 rows, fp32 accumulation; sequences of one length batched together, so there is no padding and no mask):
  qkv_proj + rotary embedding   one GEMM with the RoPE epilogue (`mrag_gemm_bf16`, EPI_QKNORM_ROPE without the LayerNorm): the checkpoint's `rotate_half` pairing
                                (i, i + 32) becomes the kernel's interleaved pairing (2i, 2i + 1) by permuting the q / k output rows once at load time -- a
                                permutation of head dimensions applied to q and k alike leaves q . k unchanged
  attention                     `mrag_attn_fwd_bf16` (12 heads of 64)
  o_proj + residual, post-LN    GEMM with the residual epilogue, `mrag_layernorm_bf16`
  gated MLP                     up_gate_proj as one GEMM with the `up * gelu_erf(gate)` epilogue, down_proj + residual, post-LN
  sentence embedding            first ([CLS]) token, L2-normalised in fp32 (`normalize=True` of the LanceDB wrapper)
GPU only; no CPU fallback."""
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch import nn

from . import ops
from .dynamicrafter import _CACHE
from .dynamicrafter_vae import _b


class _Holder(nn.Module):
    pass


class NewModel(nn.Module):
    """`Alibaba-NLP/new-impl` NewModel as gte-base-en-v1.5 configures it (rope + NTK factor 2, packed qkv, gated GELU MLP, post-norm, no token types)"""

    def __init__(self, vocab_size: int = 30528, hidden_size: int = 768, num_hidden_layers: int = 12, num_attention_heads: int = 12, intermediate_size: int = 3072,
                 layer_norm_eps: float = 1e-12, max_position_embeddings: int = 8192, rope_theta: float = 500000.0, rope_scaling_factor: float = 2.0, **_unused):
        super().__init__()
        if hidden_size != 64 * num_attention_heads:
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 64")
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size, num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      intermediate_size=intermediate_size, layer_norm_eps=layer_norm_eps, max_position_embeddings=max_position_embeddings,
                                      rope_theta=rope_theta, rope_scaling_factor=rope_scaling_factor)
        self.embeddings = _Holder()
        self.embeddings.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=0)
        self.embeddings.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        layers = []
        for _ in range(num_hidden_layers):
            lyr = _Holder()
            lyr.attention = _Holder()
            lyr.attention.qkv_proj = nn.Linear(hidden_size, 3 * hidden_size)
            lyr.attention.o_proj = nn.Linear(hidden_size, hidden_size)
            lyr.attn_ln = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
            lyr.mlp = _Holder()
            lyr.mlp.up_gate_proj = nn.Linear(hidden_size, 2 * intermediate_size, bias=False)
            lyr.mlp.down_proj = nn.Linear(intermediate_size, hidden_size)
            lyr.mlp_ln = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
            layers.append(lyr)
        self.encoder = _Holder()
        self.encoder.layer = nn.ModuleList(layers)
        self._rope_cache: Dict = {}

    def _rope(self, S: int, device):
        """fp32 [S, 64] tables in the kernel's interleaved layout: columns 2i and 2i + 1 carry frequency i.  NTK scaling: the published class builds its cache for
        factor * max_position_embeddings positions, so every position uses inv_freq_i = (theta * factor)^(-2i/64) / factor^(2/64)."""
        key = (S, str(device))
        if key not in self._rope_cache:
            c = self.config
            inv = 1.0 / ((c.rope_theta * c.rope_scaling_factor) ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64)) / c.rope_scaling_factor ** (2.0 / 64)
            ang = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
            ang = ang.repeat_interleave(2, dim=1)
            if len(self._rope_cache) >= 64:                                     # one table per sequence length the builder meets; bounded
                self._rope_cache.clear()
            self._rope_cache[key] = (ang.cos().contiguous().to(device), ang.sin().contiguous().to(device))
        return self._rope_cache[key]

    @staticmethod
    def _qkv_interleaved(lin: nn.Linear, heads: int):
        """q and k output rows of every head re-ordered (i, i + 32) -> (2i, 2i + 1): `rotate_half` RoPE on the checkpoint's layout == interleaved RoPE on this one"""
        def build():
            d = lin.weight.shape[1]
            per_head = torch.stack([torch.arange(32), torch.arange(32) + 32], dim=1).reshape(-1)                 # [0, 32, 1, 33, ...]
            qk = (torch.arange(heads)[:, None] * 64 + per_head[None, :]).reshape(-1)
            rows = torch.cat([qk, d + qk, 2 * d + torch.arange(d)]).to(lin.weight.device)
            return _b(lin.weight)[rows].contiguous(), _b(lin.bias)[rows].contiguous()
        return _CACHE.get(("gte_qkv", id(lin)), (lin.weight, lin.bias), build)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, **_unused):
        """input_ids [B, S] without padding (all ones `attention_mask`) -> object with `.last_hidden_state` [B, S, hidden] bf16 (`[0]` works too)"""
        if not input_ids.is_cuda:
            raise ops.HipOnly("NewModel: GPU tensors only")
        if attention_mask is not None and not bool(attention_mask.all()):
            raise NotImplementedError("padded batches: group the texts by token count (`SentenceEmbedder.encode` does)")
        c = self.config
        B, S = input_ids.shape
        H, d = c.num_attention_heads, c.hidden_size
        emb = self.embeddings
        x = ops.layernorm(_b(emb.word_embeddings.weight)[input_ids].contiguous(), _b(emb.LayerNorm.weight), _b(emb.LayerNorm.bias), c.layer_norm_eps)
        cos, sin = self._rope(S, input_ids.device)
        for lyr in self.encoder.layer:
            att, mlp = lyr.attention, lyr.mlp
            wqkv, bqkv = self._qkv_interleaved(att.qkv_proj, H)
            qkv = ops.qkv_linear_qknorm_rope(x, wqkv, bqkv, H, None, None, None, None, cos, sin, 0).view(B, S, 3, H, 64)
            a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
            x = ops.linear(a, _b(att.o_proj.weight), _b(att.o_proj.bias), epilogue=ops.EPI_RESID, resid=x)
            x = ops.layernorm(x, _b(lyr.attn_ln.weight), _b(lyr.attn_ln.bias), c.layer_norm_eps)
            wg = _CACHE.get(("gte_geglu", id(mlp.up_gate_proj)), mlp.up_gate_proj.weight, lambda: ops.geglu_interleave(_b(mlp.up_gate_proj.weight), None)[0])
            g = ops.linear(x, wg, epilogue=ops.EPI_GEGLU)                                   # [up | gate] -> up * gelu_erf(gate)
            x = ops.linear(g, _b(mlp.down_proj.weight), _b(mlp.down_proj.bias), epilogue=ops.EPI_RESID, resid=x)
            x = ops.layernorm(x, _b(lyr.mlp_ln.weight), _b(lyr.mlp_ln.bias), c.layer_norm_eps)
        return _Output(x)


class _Output(tuple):
    def __new__(cls, hidden):
        o = super().__new__(cls, (hidden,))
        o.last_hidden_state = hidden
        return o


class SentenceEmbedder:
    """text -> fp32 [768] unit vector: what `RAGDatabase(..., embedder=)` / `add_to_db(..., embedder=)` take in place of the reference's LanceDB embedding function.
    `tokenizer(texts) -> list of token-id lists` ([CLS] ... [SEP] included, no padding) is the model's WordPiece tokenizer -- third-party data (vocab.txt), supplied
    by the caller.  Texts are grouped by token count so that every forward pass is an unpadded batch."""

    def __init__(self, model: NewModel, tokenizer: Callable[[Sequence[str]], List[List[int]]], normalize: bool = True, max_length: int = 8192):
        self.model, self.tokenizer, self.normalize, self.max_length = model, tokenizer, normalize, max_length

    def ndims(self) -> int:
        return self.model.config.hidden_size

    @torch.no_grad()
    def encode(self, texts: Sequence[str]) -> torch.Tensor:
        dev = next(self.model.parameters()).device
        ids = [t[:self.max_length] for t in self.tokenizer(list(texts))]
        out = torch.empty(len(ids), self.ndims(), dtype=torch.float32, device=dev)
        by_len: Dict[int, List[int]] = {}
        for i, t in enumerate(ids):
            by_len.setdefault(len(t), []).append(i)
        for n, rows in by_len.items():
            batch = torch.tensor([ids[i] for i in rows], dtype=torch.long, device=dev)
            cls = self.model(batch).last_hidden_state[:, 0].float()
            if self.normalize:
                cls = cls / cls.norm(dim=1, keepdim=True).clamp_min(1e-12)
            out[torch.tensor(rows, device=dev)] = cls
        return out

    def __call__(self, text: str):
        return self.encode([text])[0].cpu().numpy()
