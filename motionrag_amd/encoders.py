"""Frozen feature encoders in front of CAMA on the HIP kernels.

Mirrors the reference's wrappers  src/projects/condition/encoders/condition.py:360-400 (`VideoMAEEmbedder`) and :561-604 (`DINOImageEmbedder`,
whose `preprocess` is `CLIPImageEmbedder.preprocess`, :503-507): same `forward(video [B, T, C, H, W])` / `forward(images [B, C, H, W])` on inputs
in [-1, 1], same `last_hidden_state` result ([B, 1568, 768] for VideoMAE-B at 16 x 224 x 224, [B, 257, 1024] for DINOv2-L at 224 x 224), same `.dim`.
The models behind them are the third-party `transformers` `VideoMAEModel` / `Dinov2Model` (the reference calls `from_pretrained`); `VideoMAEModel` and
`Dinov2Model` here keep their state-dict keys -- both the 4.44.2 dialect the reference pins (`q_bias` / `v_bias`) and the 5.x one (`query.bias` ...) --
so `load_state_dict(hf_model.state_dict())` is the hand-over; there is no network in this build, so no `from_pretrained`.

Data path, all on `libmrag_hip.so`:
  pixels  -> `mrag_resize_patchify_bf16`: frame gather (uniform sampling) + antialiased Resize + CenterCrop + (x + 1) / 2 + normalisation written
             straight as patch-embedding GEMM rows (no resized image, no im2col buffer)
  tokens  -> patch GEMM (+bias) -> `mrag_assemble_tokens_bf16` ([cls] + position table)
  layers  -> LayerNorm, fused QKV GEMM, head_dim-64 flash attention (attn16 at S = 1568, the 32x32 kernel at S = 257), output / MLP GEMMs with the
             residual (VideoMAE) or LayerScale * x + residual (DINOv2, the AdaLN-gate epilogue with a constant gate) fused in their epilogues.
GPU only; a missing library raises (`motionrag_amd._lib.HipLibraryMissing`).
"""
import math
from typing import Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import ops
from .dynamicrafter import _CACHE

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# ---------------------------------------------------------------------------------------------------------------- host geometry
def resize_output_size(h: int, w: int, size: int):
    """torchvision Resize(int): shorter edge -> `size`, the longer one int(size * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_offsets(h: int, w: int, ch: int, cw: int):
    """torchvision center_crop offsets (Python round: half to even)."""
    if ch > h or cw > w:
        raise ValueError(f"crop {ch}x{cw} larger than the resized image {h}x{w} (torchvision would pad; the reference never does)")
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))


def _filter(x: np.ndarray, mode: str) -> np.ndarray:
    x = np.abs(x).astype(np.float32)
    if mode == "bilinear":
        return np.where(x < 1, np.float32(1) - x, np.float32(0)).astype(np.float32)
    a = np.float32(-0.5)                                   # bicubic, ATen's antialias kernel uses a = -0.5
    w1 = ((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + np.float32(1)
    w2 = (((x - np.float32(5)) * x + np.float32(8)) * x - np.float32(4)) * a
    return np.where(x < 1, w1, np.where(x < 2, w2, np.float32(0))).astype(np.float32)


def resize_taps(in_size: int, out_size: int, mode: str, first: int = 0, count: Optional[int] = None):
    """Source span and weights of output coordinates first .. first + count of torch's `interpolate(mode, align_corners=False, antialias=True)`
    from `in_size` to `out_size` pixels, in fp32 with ATen's operation order (aten/src/ATen/native/cuda/UpSample.cuh, `_compute_weights_span` /
    `_compute_weights`): returns (weights [count, taps] fp32 zero-padded, start [count] int32, n [count] int32)."""
    if mode not in ("bilinear", "bicubic"):
        raise ValueError(mode)
    count = out_size - first if count is None else count
    interp = np.float32(2.0 if mode == "bilinear" else 4.0)
    scale = np.float32(in_size) / np.float32(out_size)
    half = np.float32(0.5)
    support = (interp * half) * scale if scale >= 1 else interp * half
    invscale = np.float32(1.0) / scale if scale >= 1 else np.float32(1.0)
    spans = []
    for i in range(first, first + count):
        center = scale * (np.float32(i) + half)
        xmin = max(int(center - support + half), 0)
        xsize = min(int(center + support + half), in_size) - xmin
        j = np.arange(xsize, dtype=np.float32)
        w = _filter((j + np.float32(xmin) - center + half) * invscale, mode)
        total = np.float32(0)
        for v in w:                                        # sequential fp32 sum, as the device loop
            total = np.float32(total + v)
        if total != 0:
            w = (w / total).astype(np.float32)
        spans.append((xmin, w))
    taps = max(len(w) for _, w in spans)
    W = np.zeros((count, taps), dtype=np.float32)
    start = np.zeros(count, dtype=np.int32)
    n = np.zeros(count, dtype=np.int32)
    for r, (xmin, w) in enumerate(spans):
        W[r, :len(w)] = w
        start[r], n[r] = xmin, len(w)
    return W, start, n


def kornia_resize_taps(in_size: int, out_size: int, sigma: float, ksize: int):
    """One axis of `kornia.geometry.resize(..., interpolation='bicubic', align_corners=True, antialias=True)` (the `preprocess` of the reference's OpenCLIP image
    embedders, lvdm/modules/encoders/condition.py:328-336) as ONE tap table: the library blurs with a normalised Gaussian window (reflect border) and then runs
    ATen's bicubic interpolation (A = -0.75, align_corners=True, clamped indices); both are linear along the axis, so row i of their product
    bicubic[out, in] @ blur[in, in] is the span of source pixels output i reads.  `ksize` = 1 means no blur (not downscaling).  Built in float64, returned as
    (weights [out, taps] fp32, start [out] int32, n [out] int32) like `resize_taps`."""
    blur = np.zeros((in_size, in_size))
    if ksize > 1:
        x = np.arange(ksize, dtype=np.float64) - ksize // 2
        g = np.exp(-x ** 2 / (2.0 * sigma ** 2))
        g /= g.sum()
        for r in range(in_size):
            for m in range(ksize):
                j = r + m - ksize // 2
                j = -j if j < 0 else (2 * (in_size - 1) - j if j >= in_size else j)          # F.pad(mode='reflect'): the edge pixel is not repeated
                blur[r, j] += g[m]
    else:
        blur[np.arange(in_size), np.arange(in_size)] = 1.0
    A = -0.75
    c1 = lambda v: ((A + 2.0) * v - (A + 3.0)) * v * v + 1.0
    c2 = lambda v: ((A * v - 5.0 * A) * v + 8.0 * A) * v - 4.0 * A
    scale = (in_size - 1) / (out_size - 1) if out_size > 1 else 0.0
    full = np.zeros((out_size, in_size))
    for i in range(out_size):
        real = scale * i
        ix = int(np.floor(real))
        t = real - ix
        for k, c in enumerate((c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t))):
            full[i] += c * blur[min(max(ix - 1 + k, 0), in_size - 1)]
    spans = []
    for i in range(out_size):
        nz = np.nonzero(full[i])[0]
        spans.append((int(nz[0]), full[i, nz[0]:nz[-1] + 1]))
    taps = max(len(w) for _, w in spans)
    W = np.zeros((out_size, taps), dtype=np.float32)
    start = np.zeros(out_size, dtype=np.int32)
    n = np.zeros(out_size, dtype=np.int32)
    for r, (x0, w) in enumerate(spans):
        W[r, :len(w)] = w.astype(np.float32)
        start[r], n[r] = x0, len(w)
    return W, start, n


def kornia_blur_geometry(H: int, W: int, oh: int, ow: int):
    """sigma and window size per axis of kornia's antialiasing blur (skimage's rule): only when downscaling, then on both axes"""
    fy, fx = H / oh, W / ow
    if max(fy, fx) <= 1:
        return (0.0, 1), (0.0, 1)
    out = []
    for f in (fy, fx):
        sigma = max((f - 1.0) / 2.0, 0.001)
        k = int(max(2.0 * 2 * sigma, 3))
        out.append((sigma, k + 1 if k % 2 == 0 else k))
    return tuple(out)


class _PixelPlan:
    """tap tables of one (source H x W) -> Resize(resize) -> CenterCrop(crop) geometry, resident on the device; mode 'kornia-bicubic' / 'kornia-bicubic-noaa':
    the squashing resize to crop x crop of `kornia.geometry.resize` (no crop)"""

    def __init__(self, H: int, W: int, resize: int, crop: int, mode: str, device):
        if mode.startswith("kornia-bicubic"):
            (sy, ky), (sx, kx) = kornia_blur_geometry(H, W, crop, crop) if mode == "kornia-bicubic" else ((0.0, 1), (0.0, 1))
            wy, y0, ny = kornia_resize_taps(H, crop, sy, ky) if H != crop or W != crop else kornia_resize_taps(H, H, 0.0, 1)
            wx, x0, nx = kornia_resize_taps(W, crop, sx, kx) if H != crop or W != crop else kornia_resize_taps(W, W, 0.0, 1)
        else:
            nh, nw = resize_output_size(H, W, resize)
            top, left = center_crop_offsets(nh, nw, crop, crop)
            wy, y0, ny = resize_taps(H, nh, mode, top, crop)
            wx, x0, nx = resize_taps(W, nw, mode, left, crop)
        up = lambda a: torch.from_numpy(a).to(device)
        self.wy, self.y0, self.ny, self.wx, self.x0, self.nx = up(wy), up(y0), up(ny), up(wx), up(x0), up(nx)
        self.taps_y, self.taps_x, self.crop = wy.shape[1], wx.shape[1], crop


_PLANS = {}


def pixels_to_patch_rows(src: torch.Tensor, *, resize: int, crop: int, mode: str, patch: Sequence[int], frame_idx: Optional[torch.Tensor] = None,
                         mean=IMAGENET_MEAN, std=IMAGENET_STD, tiled: bool = True) -> torch.Tensor:
    """src [N, T, C, H, W] (bf16 or fp32, values in [-1, 1]) -> patch-embedding GEMM rows [N * T'/pt * crop/ph * crop/pw, Kpad] bf16, where
    T' = len(frame_idx) (default: all T frames) and the columns are (c, dt, dy, dx), zero-padded to a multiple of 64.
    = condition.py:378-382 / :503-507 + the im2col of the patch-embedding convolution, one launch."""
    if src.dim() != 5 or not src.is_cuda:
        raise ops.HipOnly("pixels_to_patch_rows: [N, T, C, H, W] tensor on the GPU expected")
    if src.dtype not in (torch.bfloat16, torch.float32):
        src = src.to(torch.bfloat16)
    if src.stride(-1) != 1 or src.stride(-2) != src.shape[-1]:
        src = src.contiguous()
    N, T, C, H, W = src.shape
    pt, ph, pw = patch
    key = (H, W, resize, crop, mode, src.device)
    plan = _PLANS.get(key)
    if plan is None:
        plan = _PLANS[key] = _PixelPlan(H, W, resize, crop, mode, src.device)
    To = T if frame_idx is None else frame_idx.numel()
    if To % pt or crop % ph or crop % pw or C > 4:
        raise ValueError("frames / crop must be multiples of the patch size; at most 4 channels")
    rows = N * (To // pt) * (crop // ph) * (crop // pw)
    K = C * pt * ph * pw
    out = torch.empty(rows, ops._kpad(K), dtype=torch.bfloat16, device=src.device)
    a = ops._lib.ResizePatchArgs()
    a.src, a.frame_idx = ops._p(src), ops._p(frame_idx)
    a.wy, a.y0, a.ny, a.wx, a.x0, a.nx = ops._p(plan.wy), ops._p(plan.y0), ops._p(plan.ny), ops._p(plan.wx), ops._p(plan.x0), ops._p(plan.nx)
    a.out, a.ldo = ops._p(out), out.stride(0)
    a.s_n, a.s_t, a.s_c = src.stride(0), src.stride(1), src.stride(2)
    a.N, a.T, a.C, a.H, a.W, a.OH, a.OW = N, To, C, H, W, crop, crop
    a.taps_y, a.taps_x, a.pt, a.ph, a.pw = plan.taps_y, plan.taps_x, pt, ph, pw
    a.src_fp32 = 1 if src.dtype == torch.float32 else 0
    a.no_tiling = 0 if tiled else 1
    for c in range(C):                      # ((r + 1) / 2 - mean) / std = r * (0.5 / std) + (0.5 - mean) / std
        a.scale[c], a.shift[c] = 0.5 / std[c], (0.5 - mean[c]) / std[c]
    ops.check(ops._lib.lib().mrag_resize_patchify_bf16(ops._stream(), ops.ctypes.byref(a)), "mrag_resize_patchify_bf16")
    return out


def assemble_tokens(x: torch.Tensor, prefix: Optional[torch.Tensor], pos: Optional[torch.Tensor]) -> torch.Tensor:
    """out[n, j] = (j < P ? prefix[j] : x[n, j - P]) + pos[j]   (x [N, L, D], prefix [P, D] or None, pos [L + P, D] or None)"""
    N, L, D = x.shape
    P = 0 if prefix is None else prefix.shape[0]
    for t in (x, prefix, pos):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or not t.is_cuda):
            raise ops.HipOnly("assemble_tokens: contiguous bf16 GPU tensors expected")
    out = torch.empty(N, L + P, D, dtype=torch.bfloat16, device=x.device)
    ops.check(ops._lib.lib().mrag_assemble_tokens_bf16(ops._stream(), ops._p(x), ops._p(prefix), ops._p(pos), ops._p(out), N, L, P, D), "mrag_assemble_tokens_bf16")
    return out


def uniform_frame_indices(t: int, n: int = 16) -> torch.Tensor:
    """condition.py:396"""
    return torch.linspace(0, t - 1, n).round().long()


def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """VideoMAE's fixed position table: angle[p, j] = p / 10000^(2 (j // 2) / d); sin on even j, cos on odd j (float64 on the host)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid, dtype=np.float64)[None, :]
    ang = pos / np.power(10000.0, 2 * (j // 2) / d_hid)
    tab = np.where((np.arange(d_hid) % 2 == 0)[None, :], np.sin(ang), np.cos(ang))
    return torch.from_numpy(tab.astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------- ViT bodies
def _b(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


class _SelfAttention(nn.Module):
    """`…attention.attention`: query / key / value weights + biases.  Internal layout = transformers 4.44.2's VideoMAE one (weights without bias,
    `q_bias` / `v_bias` parameters) plus a `k_bias`; the 5.x / DINOv2 layout (`query.bias`, `key.bias`, `value.bias`) is mapped on load."""

    def __init__(self, dim: int, bias: bool = True):
        super().__init__()
        self.query, self.key, self.value = (nn.Linear(dim, dim, bias=False) for _ in range(3))
        z = lambda: nn.Parameter(torch.zeros(dim)) if bias else None
        self.q_bias, self.k_bias, self.v_bias = z(), z(), z()
        self._register_load_state_dict_pre_hook(self._remap)

    @staticmethod
    def _remap(sd, prefix, *_):
        for new, old in (("q_bias", "query.bias"), ("k_bias", "key.bias"), ("v_bias", "value.bias")):
            if prefix + old in sd:
                sd[prefix + new] = sd.pop(prefix + old)
        if prefix + "q_bias" in sd and prefix + "k_bias" not in sd:          # 4.44.2 VideoMAE: the key bias is a fixed zero
            sd[prefix + "k_bias"] = torch.zeros_like(sd[prefix + "q_bias"])

    def fused(self):
        """[3D, D] weight and [3D] bias (or None), rebuilt when any constituent changes"""
        parts = (self.query.weight, self.key.weight, self.value.weight, self.q_bias, self.k_bias, self.v_bias)

        def build():
            w = torch.cat([_b(self.query.weight), _b(self.key.weight), _b(self.value.weight)], 0).contiguous()
            b = None if self.q_bias is None else torch.cat([_b(self.q_bias), _b(self.k_bias), _b(self.v_bias)], 0).contiguous()
            return w, b
        return _CACHE.get(("vit_qkv", id(self)), parts, build)


class _Holder(nn.Module):
    pass


def _attn_block(dim: int, bias: bool = True) -> nn.Module:
    m = _Holder()
    m.attention = _SelfAttention(dim, bias)
    m.output = _Holder()
    m.output.dense = nn.Linear(dim, dim)
    return m


class _LayerScale(nn.Module):
    def __init__(self, dim: int, init: float = 1.0):
        super().__init__()
        self.lambda1 = nn.Parameter(init * torch.ones(dim))


def _layer_forward(x: torch.Tensor, heads: int, eps: float, n1: nn.LayerNorm, att: nn.Module, n2: nn.LayerNorm, fc1: nn.Linear, fc2: nn.Linear,
                   ls1: Optional[_LayerScale], ls2: Optional[_LayerScale]) -> torch.Tensor:
    """pre-norm ViT block: x + [ls1 *] proj(attn(norm1(x))), then x + [ls2 *] fc2(gelu(fc1(norm2(x)))); every residual rides in a GEMM epilogue"""
    N, S, D = x.shape
    h = ops.layernorm(x, _b(n1.weight), _b(n1.bias), eps)
    w, b = att.attention.fused()
    qkv = ops.linear(h, w, b).view(N, S, 3, heads, 64)
    a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
    od = att.output.dense
    if ls1 is None:
        x = ops.linear(a, _b(od.weight), _b(od.bias), epilogue=ops.EPI_RESID, resid=x)
    else:
        g = _b(ls1.lambda1)
        x = ops.linear(a, _b(od.weight), _b(od.bias), epilogue=ops.EPI_GATE_RESID, resid=x, gate0=g, gate1=g, rows_per_batch=N * S, split=0, gate_stride=0)
    h = ops.layernorm(x, _b(n2.weight), _b(n2.bias), eps)
    h = ops.linear(h, _b(fc1.weight), _b(fc1.bias), epilogue=ops.EPI_GELU_ERF)
    if ls2 is None:
        return ops.linear(h, _b(fc2.weight), _b(fc2.bias), epilogue=ops.EPI_RESID, resid=x)
    g = _b(ls2.lambda1)
    return ops.linear(h, _b(fc2.weight), _b(fc2.bias), epilogue=ops.EPI_GATE_RESID, resid=x, gate0=g, gate1=g, rows_per_batch=N * S, split=0, gate_stride=0)


class _VideoMAELayer(nn.Module):
    def __init__(self, dim: int, inter: int, eps: float, qkv_bias: bool):
        super().__init__()
        self.attention = _attn_block(dim, qkv_bias)
        self.layernorm_before = nn.LayerNorm(dim, eps=eps)
        self.layernorm_after = nn.LayerNorm(dim, eps=eps)
        self.intermediate = _Holder(); self.intermediate.dense = nn.Linear(dim, inter)
        self.output = _Holder(); self.output.dense = nn.Linear(inter, dim)


class VideoMAEModel(nn.Module):
    """transformers `VideoMAEModel` (encoder only, `bool_masked_pos=None`), state-dict compatible.  Defaults = MCG-NJU/videomae-base-finetuned-ssv2
    (the checkpoint condition.py:365 names): 12 layers x 768, 12 heads, tubelets 2 x 16 x 16 over 16 x 224 x 224, `use_mean_pooling=True` (so the model
    itself ends WITHOUT a final LayerNorm; with `use_mean_pooling=False` a `layernorm` is applied)."""

    def __init__(self, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, image_size=224, patch_size=16,
                 num_channels=3, num_frames=16, tubelet_size=2, layer_norm_eps=1e-12, qkv_bias=True, use_mean_pooling=True, **_unused):
        super().__init__()
        if hidden_size != 64 * num_attention_heads:
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 64")
        self.heads, self.eps = num_attention_heads, layer_norm_eps
        self.patch = (tubelet_size, patch_size, patch_size)
        self.num_frames, self.image_size, self.hidden_size = num_frames, image_size, hidden_size
        self.embeddings = _Holder()
        self.embeddings.patch_embeddings = _Holder()
        self.embeddings.patch_embeddings.projection = nn.Conv3d(num_channels, hidden_size, self.patch, stride=self.patch)
        self.encoder = _Holder()
        self.encoder.layer = nn.ModuleList(_VideoMAELayer(hidden_size, intermediate_size, layer_norm_eps, qkv_bias) for _ in range(num_hidden_layers))
        self.layernorm = None if use_mean_pooling else nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self._pos = {}

    def _position_table(self, tokens: int, device) -> torch.Tensor:
        key = (tokens, device)
        if key not in self._pos:
            self._pos[key] = sinusoid_table(tokens, self.hidden_size).to(device=device, dtype=torch.bfloat16).contiguous()
        return self._pos[key]

    def forward_rows(self, rows: torch.Tensor, batch: int) -> torch.Tensor:
        """patch rows [batch * tokens, Kpad] (from `pixels_to_patch_rows`) -> last_hidden_state [batch, tokens, hidden]"""
        conv = self.embeddings.patch_embeddings.projection

        def build():
            w = _b(conv.weight).reshape(conv.weight.shape[0], -1)
            return torch.nn.functional.pad(w, (0, rows.shape[1] - w.shape[1])).contiguous()
        w = _CACHE.get(("vit_patch", id(conv), rows.shape[1]), conv.weight, build)
        x = ops.linear(rows, w, _b(conv.bias)).view(batch, -1, self.hidden_size)
        x = assemble_tokens(x, None, self._position_table(x.shape[1], x.device))
        for L in self.encoder.layer:
            x = _layer_forward(x, self.heads, self.eps, L.layernorm_before, L.attention, L.layernorm_after, L.intermediate.dense, L.output.dense, None, None)
        if self.layernorm is not None:
            x = ops.layernorm(x, _b(self.layernorm.weight), _b(self.layernorm.bias), self.eps)
        return x

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """already-normalised `pixel_values` [B, T, C, H, W] (the transformers signature; the embedder below skips this entry and feeds rows)"""
        B, T, C, H, W = pixel_values.shape
        if H != W:
            raise ValueError("square frames expected")
        one = (1.0,) * C
        # identity geometry (Resize(H) + CenterCrop(H) are no-ops) and the inverse of the embedder's value map: scale 0.5 / std, shift (0.5 - mean) / std
        # with std = 0.5, mean = 0.5 -> scale 1, shift 0
        rows = pixels_to_patch_rows(pixel_values, resize=H, crop=H, mode="bilinear", patch=self.patch, mean=(0.5,) * C, std=tuple(0.5 * o for o in one))
        return self.forward_rows(rows, B)


class _Dinov2Layer(nn.Module):
    def __init__(self, dim: int, inter: int, eps: float, ls_init: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attention = _attn_block(dim, True)
        self.layer_scale1 = _LayerScale(dim, ls_init)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Holder(); self.mlp.fc1 = nn.Linear(dim, inter); self.mlp.fc2 = nn.Linear(inter, dim)
        self.layer_scale2 = _LayerScale(dim, ls_init)


class Dinov2Model(nn.Module):
    """transformers `Dinov2Model` (MLP variant), state-dict compatible.  Defaults = facebook/dinov2-large (condition.py:568): 24 layers x 1024, 16 heads,
    patch 14, position table of a 37 x 37 grid (518 px) resampled to the input's grid on first use.
    `pos_dialect`: "scale_factor" = transformers 4.44.2 (the reference's pin: bicubic `scale_factor` with the +0.1 offset), "size" = >= 4.45 / 5.x."""

    def __init__(self, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, mlp_ratio=4, image_size=518, patch_size=14, num_channels=3,
                 layer_norm_eps=1e-6, layerscale_value=1.0, use_swiglu_ffn=False, pos_dialect="scale_factor", **_unused):
        super().__init__()
        if hidden_size != 64 * num_attention_heads:
            raise NotImplementedError("the gfx950 attention kernels are built for head_dim 64")
        if use_swiglu_ffn:
            raise NotImplementedError("SwiGLU variant (dinov2-giant) is not on the reference's path")
        self.heads, self.eps, self.hidden_size, self.patch_size, self.pos_dialect = num_attention_heads, layer_norm_eps, hidden_size, patch_size, pos_dialect
        g = image_size // patch_size
        self.embeddings = _Holder()
        self.embeddings.cls_token = nn.Parameter(torch.randn(1, 1, hidden_size))
        self.embeddings.mask_token = nn.Parameter(torch.zeros(1, hidden_size))
        self.embeddings.position_embeddings = nn.Parameter(torch.randn(1, g * g + 1, hidden_size))
        self.embeddings.patch_embeddings = _Holder()
        self.embeddings.patch_embeddings.projection = nn.Conv2d(num_channels, hidden_size, patch_size, stride=patch_size)
        self.encoder = _Holder()
        self.encoder.layer = nn.ModuleList(_Dinov2Layer(hidden_size, int(hidden_size * mlp_ratio), layer_norm_eps, layerscale_value) for _ in range(num_hidden_layers))
        self.layernorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)

    def _position_table(self, gh: int, gw: int) -> torch.Tensor:
        """Dinov2Embeddings.interpolate_pos_encoding, once per grid (load-time constant for a fixed resolution; bicubic in fp32 through torch on the
        table's device -- 1 370 x 1024 numbers, not part of the per-call path)"""
        pe = self.embeddings.position_embeddings

        def build():
            pos = pe.detach().float()
            n = pos.shape[1] - 1
            g0 = int(round(math.sqrt(n)))
            if not (g0 * g0 == n and g0 == gh and g0 == gw):
                patch = pos[:, 1:].reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
                if self.pos_dialect == "size":
                    patch = torch.nn.functional.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
                else:
                    patch = torch.nn.functional.interpolate(patch, scale_factor=(float((gh + 0.1) / g0), float((gw + 0.1) / g0)), mode="bicubic", align_corners=False)
                    if tuple(patch.shape[-2:]) != (gh, gw):
                        raise ValueError("position-table interpolation produced an unexpected grid")
                pos = torch.cat([pos[:, :1], patch.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)
            return pos[0].to(torch.bfloat16).contiguous()
        return _CACHE.get(("dino_pos", id(self), gh, gw, self.pos_dialect), pe, build)

    def forward_rows(self, rows: torch.Tensor, batch: int, gh: int, gw: int) -> torch.Tensor:
        conv = self.embeddings.patch_embeddings.projection

        def build():
            w = _b(conv.weight).reshape(conv.weight.shape[0], -1)
            return torch.nn.functional.pad(w, (0, rows.shape[1] - w.shape[1])).contiguous()
        w = _CACHE.get(("vit_patch", id(conv), rows.shape[1]), conv.weight, build)
        x = ops.linear(rows, w, _b(conv.bias)).view(batch, gh * gw, self.hidden_size)
        cls = _CACHE.get(("dino_cls", id(self)), self.embeddings.cls_token, lambda: _b(self.embeddings.cls_token).reshape(1, -1).contiguous())
        x = assemble_tokens(x, cls, self._position_table(gh, gw))
        for L in self.encoder.layer:
            x = _layer_forward(x, self.heads, self.eps, L.norm1, L.attention, L.norm2, L.mlp.fc1, L.mlp.fc2, L.layer_scale1, L.layer_scale2)
        return ops.layernorm(x, _b(self.layernorm.weight), _b(self.layernorm.bias), self.eps)

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """already-normalised `pixel_values` [B, C, H, W]"""
        B, C, H, W = pixel_values.shape
        if H != W or H % self.patch_size:
            raise ValueError("square inputs with a whole number of patches expected")
        rows = pixels_to_patch_rows(pixel_values[:, None], resize=H, crop=H, mode="bilinear", patch=(1, self.patch_size, self.patch_size),
                                    mean=(0.5,) * C, std=(0.5,) * C)
        return self.forward_rows(rows, B, H // self.patch_size, W // self.patch_size)


# ---------------------------------------------------------------------------------------------------------------- the reference's wrappers
class VideoMAEEmbedder(nn.Module):
    """condition.py:360-400.  `model`: a `VideoMAEModel` (this module's) or keyword config for one; frozen, eval."""

    def __init__(self, model=None, freeze: bool = True, compile: bool = False, resize: int = 224, crop: int = 224, **config):
        super().__init__()
        self.model = model if isinstance(model, nn.Module) else VideoMAEModel(**config)
        self.dim = self.model.hidden_size
        self.resize, self.crop = resize, crop
        self._idx = {}
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def forward(self, video: torch.Tensor) -> torch.Tensor:
        if video.dim() != 5:
            raise ValueError(f"VideoMAEEmbedder expects clips [B, T, C, H, W], got {tuple(video.shape)}")
        T = video.shape[1]
        key = (T, video.device)
        if key not in self._idx:                       # condition.py:396 (the sampled frames are gathered inside the resize kernel)
            self._idx[key] = uniform_frame_indices(T, self.model.num_frames).to(device=video.device, dtype=torch.int32)
        rows = pixels_to_patch_rows(video, resize=self.resize, crop=self.crop, mode="bilinear", patch=self.model.patch, frame_idx=self._idx[key])
        return self.model.forward_rows(rows, video.shape[0])


class DINOImageEmbedder(nn.Module):
    """condition.py:561-604 (preprocess: :503-507).  `resize` / `crop` default to facebook/dinov2-large's processor (shortest_edge 256, crop 224)."""

    def __init__(self, model=None, freeze: bool = True, dtype: torch.dtype = torch.bfloat16, compile: bool = False, model_kwargs=None,
                 resize: int = 256, crop: int = 224, **config):
        super().__init__()
        self.model = model if isinstance(model, nn.Module) else Dinov2Model(**config)
        self.dim = self.model.hidden_size
        self.resize, self.crop = resize, crop
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        if images.dim() != 4:
            raise ValueError(f"DINOImageEmbedder expects images [B, C, H, W], got {tuple(images.shape)}")
        ps = self.model.patch_size
        rows = pixels_to_patch_rows(images[:, None], resize=self.resize, crop=self.crop, mode="bicubic", patch=(1, ps, ps))
        g = self.crop // ps
        return self.model.forward_rows(rows, images.shape[0], g, g)
