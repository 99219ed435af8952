"""CLIP vision tower with projection on the HIP kernels: SVD's `image_encoder`.

diffusers' StableVideoDiffusionPipeline loads `transformers.CLIPVisionModelWithProjection` (CLIP-ViT-H/14: 32 layers x 1280, 16 heads of **80**, 257 tokens at 224 x 224,
projection to 1024) and `_encode_image` (src/projects/svd/pipelines/pipeline.py:113-119 through the diffusers base) reads `.image_embeds`.  This module keeps the
class's state-dict keys (`vision_model.embeddings.{class_embedding, patch_embedding.weight, position_embedding.weight}`, `vision_model.pre_layrnorm` [sic],
`vision_model.encoder.layers.N.{self_attn.{q,k,v,out}_proj, layer_norm1, mlp.fc1, mlp.fc2, layer_norm2}`, `vision_model.post_layernorm`, `visual_projection.weight`).

Head dim 80 is outside the flash kernels (built for 64): the attention of this once-per-clip encoder runs on `mrag_attn_small_bf16` (K / V of a head resident in LDS,
fp32 FMAs); a tower with head_dim 64 (CLIP-L) takes the flash kernel.  Everything else is the ViT path of motionrag_amd/encoders.py: pixel rows -> patch GEMM ->
token assembly -> LayerNorm / fused QKV GEMM / attention / output and MLP GEMMs with the residuals in their epilogues."""
from typing import Optional

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
        self.hidden_size, self.heads, self.patch_size, self.eps, self.image_size = hidden_size, num_attention_heads, patch_size, layer_norm_eps, image_size
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
        vm = self.vision_model
        layers = self._layers()
        x = _tower(pixel_values, vm.embeddings.patch_embedding, vm.embeddings.class_embedding, vm.embeddings.position_embedding.weight, vm.pre_layrnorm, layers,
                   self.heads, self.head_dim, self.patch_size, self.eps, id(self))
        pooled = ops.layernorm(x[:, 0].contiguous(), _b(vm.post_layernorm.weight), _b(vm.post_layernorm.bias), self.eps)
        return CLIPVisionOutput(ops.linear(pooled, _b(self.visual_projection.weight)), x)


    MEAN, STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

    def _layers(self):
        out = []
        for L in self.vision_model.encoder.layers:
            sa = L.self_attn
            w, b = _CACHE.get(("clip_qkv", id(sa)), (sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight, sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias),
                              lambda: (torch.cat([_b(sa.q_proj.weight), _b(sa.k_proj.weight), _b(sa.v_proj.weight)], 0).contiguous(),
                                       torch.cat([_b(sa.q_proj.bias), _b(sa.k_proj.bias), _b(sa.v_proj.bias)], 0).contiguous()))
            out.append((L.layer_norm1, w, b, sa.out_proj, L.layer_norm2, L.mlp.fc1, L.mlp.fc2))
        return out

    @torch.no_grad()
    def encode_image(self, image_pm1: torch.Tensor, size: Optional[int] = None) -> CLIPVisionOutput:
        """raw image [B, 3, H, W] in [-1, 1] -> `.image_embeds`: the front half of diffusers' `StableVideoDiffusionPipeline._encode_image` in one pass --
        `_resize_with_antialiasing(image, (224, 224))` (the kornia recipe: Gaussian blur with skimage's sigma rule and a reflect border, then align_corners bicubic),
        `(x + 1) / 2`, the feature extractor's CLIP mean / std -- as the fused pixel kernel's `kornia-bicubic` mode (encoders.kornia_resize_taps folds blur and
        interpolation into one tap table per axis; oracle/kornia_resize_ref.py, parity unpinned), written straight as patch-GEMM rows, then the tower."""
        if not image_pm1.is_cuda:
            raise ops.HipOnly("CLIP vision tower: GPU tensors only")
        vm = self.vision_model
        size = self.image_size if size is None else size
        B = image_pm1.shape[0]
        rows = pixels_to_patch_rows(image_pm1[:, None].contiguous(), resize=size, crop=size, mode="kornia-bicubic", patch=(1, self.patch_size, self.patch_size),
                                    mean=self.MEAN, std=self.STD)
        x = _tower_from_rows(rows, B, size // self.patch_size, vm.embeddings.patch_embedding, vm.embeddings.class_embedding, vm.embeddings.position_embedding.weight,
                             vm.pre_layrnorm, self._layers(), self.heads, self.head_dim, self.eps, id(self))
        pooled = ops.layernorm(x[:, 0].contiguous(), _b(vm.post_layernorm.weight), _b(vm.post_layernorm.bias), self.eps)
        return CLIPVisionOutput(ops.linear(pooled, _b(self.visual_projection.weight)), x)


def _tower(pixel_values, conv, class_embedding, positional, ln_pre, layers, heads, head_dim, ps, eps, owner_id) -> torch.Tensor:
    """patch embedding -> [cls] + positions -> ln_pre -> pre-LN residual attention blocks; returns every token BEFORE the post LayerNorm (transformers'
    `last_hidden_state`, open_clip's `visual.transformer` output).  `layers`: (ln_1, fused qkv weight, fused qkv bias, out_proj, ln_2, fc1, fc2) per block."""
    if not pixel_values.is_cuda:
        raise ops.HipOnly("CLIP vision tower: GPU tensors only")
    B, C, H, W = pixel_values.shape
    D = conv.weight.shape[0]
    if H != W or H % ps:
        raise ValueError("square inputs with a whole number of patches expected")
    rows = pixels_to_patch_rows(pixel_values[:, None], resize=H, crop=H, mode="bilinear", patch=(1, ps, ps), mean=(0.5,) * C, std=(0.5,) * C)   # identity geometry
    return _tower_from_rows(rows, B, H // ps, conv, class_embedding, positional, ln_pre, layers, heads, head_dim, eps, owner_id)


def _tower_from_rows(rows, B, grid, conv, class_embedding, positional, ln_pre, layers, heads, head_dim, eps, owner_id) -> torch.Tensor:
    D = conv.weight.shape[0]

    def build():
        w = _b(conv.weight).reshape(D, -1)
        return torch.nn.functional.pad(w, (0, rows.shape[1] - w.shape[1])).contiguous()
    x = ops.linear(rows, _CACHE.get(("clip_patch", id(conv), rows.shape[1]), conv.weight, build), _b(conv.bias) if conv.bias is not None else None).view(B, grid * grid, D)
    cls = _CACHE.get(("clip_cls", owner_id), class_embedding, lambda: _b(class_embedding).reshape(1, D).contiguous())
    x = assemble_tokens(x, cls, _b(positional).contiguous())
    x = ops.layernorm(x, _b(ln_pre.weight), _b(ln_pre.bias), eps)
    S = x.shape[1]
    for ln1, w, b, out_proj, ln2, fc1, fc2 in layers:
        h = ops.layernorm(x, _b(ln1.weight), _b(ln1.bias), eps)
        qkv = ops.linear(h, w, b).view(B, S, 3, heads, head_dim)
        if head_dim == 64:
            a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
        else:
            a = ops.attention_small(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
        x = ops.linear(a, _b(out_proj.weight), _b(out_proj.bias), epilogue=ops.EPI_RESID, resid=x)
        h = ops.layernorm(x, _b(ln2.weight), _b(ln2.bias), eps)
        h = ops.linear(h, _b(fc1.weight), _b(fc1.bias), epilogue=ops.EPI_GELU_ERF)
        x = ops.linear(h, _b(fc2.weight), _b(fc2.bias), epilogue=ops.EPI_RESID, resid=x)
    return x


# ---------------------------------------------------------------------------------------------------------------- open_clip naming (DynamiCrafter)
class _OCBlock(nn.Module):
    def __init__(self, d: int, ff: int):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = _H()
        self.attn.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * d ** -0.5)
        self.attn.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.attn.out_proj = nn.Linear(d, d)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = _H()
        self.mlp.c_fc, self.mlp.c_proj = nn.Linear(d, ff), nn.Linear(ff, d)


class OpenCLIPVisual(nn.Module):
    """open_clip `VisionTransformer` (the `visual` half of ViT-H-14 / laion2b_s32b_b79k), parameter names as open_clip's: `conv1.weight`, `class_embedding`,
    `positional_embedding`, `ln_pre`, `transformer.resblocks.N.{ln_1, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj, ln_2, mlp.c_fc, mlp.c_proj}`, `ln_post`, `proj`"""

    def __init__(self, width=1280, layers=32, heads=16, mlp_ratio=4.0, image_size=224, patch_size=14, output_dim=1024):
        super().__init__()
        self.head_dim = width // heads
        if self.head_dim not in (32, 64, 80, 96, 128) or self.head_dim * heads != width:
            raise NotImplementedError(f"head_dim {self.head_dim}: no attention kernel")
        self.heads, self.patch_size_ = heads, patch_size
        self.input_patchnorm = False
        self.conv1 = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(width) * width ** -0.5)
        self.positional_embedding = nn.Parameter(torch.randn((image_size // patch_size) ** 2 + 1, width) * width ** -0.5)
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _H()
        self.transformer.resblocks = nn.ModuleList(_OCBlock(width, int(width * mlp_ratio)) for _ in range(layers))
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(torch.randn(width, output_dim) * width ** -0.5)

    @torch.no_grad()
    def tokens(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """every token after the last block, before `ln_post` -- what `encode_with_vision_transformer` returns (condition.py:348-380)"""
        layers = [(r.ln_1, _b(r.attn.in_proj_weight), _b(r.attn.in_proj_bias), r.attn.out_proj, r.ln_2, r.mlp.c_fc, r.mlp.c_proj) for r in self.transformer.resblocks]
        return _tower(pixel_values, self.conv1, self.class_embedding, self.positional_embedding, self.ln_pre, layers, self.heads, self.head_dim, self.patch_size_,
                      self.ln_pre.eps, id(self))


class FrozenOpenCLIPImageEmbedderV2(nn.Module):
    """lvdm/modules/encoders/condition.py:302-380: `forward(image [b, c, h, w] in [-1, 1])` -> tokens [b, 257, 1280] for the image Resampler.  The reference's
    `preprocess` (:328-336) is `kornia.geometry.resize(x, (224, 224), 'bicubic', align_corners=True, antialias)` + `(x + 1) / 2` + CLIP normalisation.  kornia is third-party and
    absent here; its published algorithm (Gaussian blur with skimage's sigma rule and a reflect border, then ATen's align_corners bicubic) is linear and separable, so
    `encoders.kornia_resize_taps` folds blur and interpolation into one tap table per axis and the fused pixel kernel writes the normalised 224 x 224 image straight as
    patch-GEMM rows (oracle/kornia_resize_ref.py, parity unpinned).  `preprocess=` overrides it with any callable [b, 3, h, w] -> normalised [b, 3, 224, 224]."""

    MEAN, STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, model=None, preprocess=None, freeze: bool = True, layer: str = "pooled", antialias: bool = True, **config):
        super().__init__()
        self.model = _H()
        self.model.visual = model if isinstance(model, nn.Module) else OpenCLIPVisual(**config)
        self.preprocess_fn, self.layer, self.antialias = preprocess, layer, antialias
        if freeze:
            self.eval()
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, image: torch.Tensor, no_dropout: bool = False) -> torch.Tensor:
        return self.encode_with_vision_transformer(image)

    @torch.no_grad()
    def encode_with_vision_transformer(self, x: torch.Tensor) -> torch.Tensor:
        if self.preprocess_fn is not None:
            return self.model.visual.tokens(self.preprocess_fn(x))
        if not x.is_cuda:
            raise ops.HipOnly("FrozenOpenCLIPImageEmbedderV2: GPU tensors only")
        v = self.model.visual
        B = x.shape[0]
        rows = pixels_to_patch_rows(x[:, None], resize=224, crop=224, mode="kornia-bicubic" if self.antialias else "kornia-bicubic-noaa",
                                    patch=(1, v.patch_size_, v.patch_size_), mean=self.MEAN, std=self.STD)
        # rows are already the patch-GEMM operand: run the tower from them (same code path as `tokens`, minus the identity pixel pass)
        layers = [(r.ln_1, _b(r.attn.in_proj_weight), _b(r.attn.in_proj_bias), r.attn.out_proj, r.ln_2, r.mlp.c_fc, r.mlp.c_proj) for r in v.transformer.resblocks]
        return _tower_from_rows(rows, B, 224 // v.patch_size_, v.conv1, v.class_embedding, v.positional_embedding, v.ln_pre, layers, v.heads, v.head_dim, v.ln_pre.eps, id(v))
