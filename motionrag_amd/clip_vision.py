"""CLIP vision tower with projection on the HIP kernels: SVD's `image_encoder`.

diffusers' StableVideoDiffusionPipeline loads `transformers.CLIPVisionModelWithProjection` (CLIP-ViT-H/14: 32 layers x 1280, 16 heads of **80**, 257 tokens at 224 x 224,
projection to 1024) and `_encode_image` (src/projects/svd/pipelines/pipeline.py:113-119 through the diffusers base) reads `.image_embeds`.  This module keeps the
class's state-dict keys (`vision_model.embeddings.{class_embedding, patch_embedding.weight, position_embedding.weight}`, `vision_model.pre_layrnorm` [sic],
`vision_model.encoder.layers.N.{self_attn.{q,k,v,out}_proj, layer_norm1, mlp.fc1, mlp.fc2, layer_norm2}`, `vision_model.post_layernorm`, `visual_projection.weight`).

Head dim 80 is outside the flash kernels (built for 64): the attention of this once-per-clip encoder runs on `mrag_attn_small_bf16` (K / V of a head resident in LDS,
fp32 FMAs); a tower with head_dim 64 (CLIP-L) takes the flash kernel.  Everything else is the ViT path of motionrag_amd/encoders.py: pixel rows -> patch GEMM ->
token assembly -> LayerNorm / fused QKV GEMM / attention / output and MLP GEMMs with the residuals in their epilogues."""
import torch
from torch import nn

from . import ops
from .dynamicrafter import _CACHE
from .encoders import assemble_tokens, pixels_to_patch_rows


def _b(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


class _H(nn.Module):
    pass


class _Layer(nn.Module):
    def __init__(self, d: int, ff: int, eps: float):
        super().__init__()
        self.self_attn = _H()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, nn.Linear(d, d))
        self.layer_norm1 = nn.LayerNorm(d, eps=eps)
        self.mlp = _H()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, ff), nn.Linear(ff, d)
        self.layer_norm2 = nn.LayerNorm(d, eps=eps)


class CLIPVisionOutput:
    def __init__(self, image_embeds, last_hidden_state):
        self.image_embeds, self.last_hidden_state = image_embeds, last_hidden_state


class CLIPVisionModelWithProjection(nn.Module):
    """defaults = the CLIP-ViT-H/14 image encoder of stable-video-diffusion-img2vid(-xt)"""

    def __init__(self, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224, patch_size=14, projection_dim=1024,
                 num_channels=3, hidden_act="gelu", layer_norm_eps=1e-5, **_unused):
        super().__init__()
        if hidden_act != "gelu":
            raise NotImplementedError("hidden_act 'gelu' (CLIP-ViT-H) only; quick_gelu towers are not on the reference's path")
        self.head_dim = hidden_size // num_attention_heads
        if self.head_dim not in (32, 64, 80, 96, 128) or self.head_dim * num_attention_heads != hidden_size:
            raise NotImplementedError(f"head_dim {self.head_dim}: no attention kernel")
        self.hidden_size, self.heads, self.patch_size, self.eps = hidden_size, num_attention_heads, patch_size, layer_norm_eps
        n_pos = (image_size // patch_size) ** 2 + 1
        vm = self.vision_model = _H()
        vm.embeddings = _H()
        vm.embeddings.class_embedding = nn.Parameter(torch.randn(hidden_size))
        vm.embeddings.patch_embedding = nn.Conv2d(num_channels, hidden_size, patch_size, stride=patch_size, bias=False)
        vm.embeddings.position_embedding = nn.Embedding(n_pos, hidden_size)
        vm.pre_layrnorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        vm.encoder = _H()
        vm.encoder.layers = nn.ModuleList(_Layer(hidden_size, intermediate_size, layer_norm_eps) for _ in range(num_hidden_layers))
        vm.post_layernorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.visual_projection = nn.Linear(hidden_size, projection_dim, bias=False)

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, **_unused) -> CLIPVisionOutput:
        """already-normalised `pixel_values` [B, 3, H, W] (the feature extractor's output) -> `.image_embeds` [B, projection_dim], `.last_hidden_state`"""
        if not pixel_values.is_cuda:
            raise ops.HipOnly("CLIPVisionModelWithProjection: GPU tensors only")
        B, C, H, W = pixel_values.shape
        ps, D, vm = self.patch_size, self.hidden_size, self.vision_model
        if H != W or H % ps:
            raise ValueError("square inputs with a whole number of patches expected")
        rows = pixels_to_patch_rows(pixel_values[:, None], resize=H, crop=H, mode="bilinear", patch=(1, ps, ps), mean=(0.5,) * C, std=(0.5,) * C)   # identity geometry
        conv = vm.embeddings.patch_embedding

        def build():
            w = _b(conv.weight).reshape(D, -1)
            return torch.nn.functional.pad(w, (0, rows.shape[1] - w.shape[1])).contiguous()
        x = ops.linear(rows, _CACHE.get(("clip_patch", id(conv), rows.shape[1]), conv.weight, build)).view(B, (H // ps) ** 2, D)
        cls = _CACHE.get(("clip_cls", id(self)), vm.embeddings.class_embedding, lambda: _b(vm.embeddings.class_embedding).reshape(1, D).contiguous())
        x = assemble_tokens(x, cls, _b(vm.embeddings.position_embedding.weight).contiguous())
        x = ops.layernorm(x, _b(vm.pre_layrnorm.weight), _b(vm.pre_layrnorm.bias), self.eps)
        S = x.shape[1]
        for L in vm.encoder.layers:
            sa = L.self_attn
            h = ops.layernorm(x, _b(L.layer_norm1.weight), _b(L.layer_norm1.bias), self.eps)
            w, b = _CACHE.get(("clip_qkv", id(sa)), (sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight, sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias),
                              lambda: (torch.cat([_b(sa.q_proj.weight), _b(sa.k_proj.weight), _b(sa.v_proj.weight)], 0).contiguous(),
                                       torch.cat([_b(sa.q_proj.bias), _b(sa.k_proj.bias), _b(sa.v_proj.bias)], 0).contiguous()))
            qkv = ops.linear(h, w, b).view(B, S, 3, self.heads, self.head_dim)
            if self.head_dim == 64:
                a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
            else:
                a = ops.attention_small(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
            x = ops.linear(a, _b(sa.out_proj.weight), _b(sa.out_proj.bias), epilogue=ops.EPI_RESID, resid=x)
            h = ops.layernorm(x, _b(L.layer_norm2.weight), _b(L.layer_norm2.bias), self.eps)
            h = ops.linear(h, _b(L.mlp.fc1.weight), _b(L.mlp.fc1.bias), epilogue=ops.EPI_GELU_ERF)
            x = ops.linear(h, _b(L.mlp.fc2.weight), _b(L.mlp.fc2.bias), epilogue=ops.EPI_RESID, resid=x)
        pooled = ops.layernorm(x[:, 0].contiguous(), _b(vm.post_layernorm.weight), _b(vm.post_layernorm.bias), self.eps)
        return CLIPVisionOutput(ops.linear(pooled, _b(self.visual_projection.weight)), x)
