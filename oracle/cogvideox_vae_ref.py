"""TEST INFRASTRUCTURE -- CPU restatement (torch fp32, functional, over a plain state dict) of the 3-D causal VAE the CogVideoX pipelines decode
with: diffusers==0.32.2 (`/root/reference/requirements.txt:10`) `AutoencoderKLCogVideoX`, reached from the reference at
`src/projects/cogvideox/module.py:39-40` (`pipe.vae.enable_tiling(); pipe.vae.enable_slicing()`) and through the pipeline's `prepare_latents` /
`decode_latents`.

PARITY UNPINNED: diffusers is not installed in this image and is not vendored under /root/reference, so nothing here can be run against the
real class; this file restates the published algorithm of that version (`models/autoencoders/autoencoder_kl_cogvideox.py`):

  * `CogVideoXCausalConv3d` (pad_mode "first"): zero padding in space; in time the input is preceded by the conv cache (the last
    kernel_t - 1 frames the same convolution saw in the previous frame batch) or, on the first batch, by its own first frame repeated;
  * `CogVideoXSpatialNorm3D`: GroupNorm(32, eps 1e-6) over (C/G, T, H, W) times conv_y(zq') plus conv_b(zq'), zq' = nearest-neighbour
    `F.interpolate` of the latent to the feature's (T, H, W) -- with an odd number T > 1 of frames the first frame and the rest separately;
  * `CogVideoXResnetBlock3D`: norm1 - SiLU - conv1 - norm2 - SiLU - conv2 + shortcut (1x1x1 convolution where the widths differ);
  * `CogVideoXUpsample3D`: nearest x2 in space (and in time for compress_time, first frame apart when T is odd) then a per-frame 3x3 Conv2d;
    `CogVideoXDownsample3D`: avg_pool1d(2, 2) over time (first frame apart when T is odd), F.pad(0,1,0,1), 3x3 stride-2 Conv2d;
  * `_decode` / `_encode`: frame batches of 2 latent / 8 sample frames, the remainder joining the FIRST batch, conv caches carried between;
  * `tiled_decode` / `tiled_encode`: tiles of (sample_height / 2, sample_width / 2) pixels with overlap factors 1/6 (height) and 1/5 (width), each
    tile with its own frame batches and caches, blended linearly over the overlaps (in place, so a tile is blended with its ALREADY blended
    upper / left neighbour) and cropped to the row limits.

Only tests/, `__graft_entry__.smoke()` and bench.py's cpu_baseline leg may import this module."""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

CONFIG_5B = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), layers_per_block=3, latent_channels=16, norm_eps=1e-6,
                 norm_num_groups=32, temporal_compression_ratio=4, sample_height=480, sample_width=720, scaling_factor=0.7)


def state_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """parameter names and shapes of AutoencoderKLCogVideoX(**cfg) (use_quant_conv = use_post_quant_conv = False), in module order"""
    out: Dict[str, Tuple[int, ...]] = {}
    boc, L, zc = tuple(cfg["block_out_channels"]), cfg["layers_per_block"], cfg["latent_channels"]

    def causal(name, cin, cout, k=3):
        out[f"{name}.conv.weight"] = (cout, cin, k, k, k); out[f"{name}.conv.bias"] = (cout,)

    def resnet(name, cin, cout, zq):
        for i, c in ((1, cin), (2, cout)):
            if zq is None:
                out[f"{name}.norm{i}.weight"] = (c,); out[f"{name}.norm{i}.bias"] = (c,)
            else:
                out[f"{name}.norm{i}.norm_layer.weight"] = (c,); out[f"{name}.norm{i}.norm_layer.bias"] = (c,)
                causal(f"{name}.norm{i}.conv_y", zq, c, 1); causal(f"{name}.norm{i}.conv_b", zq, c, 1)
            causal(f"{name}.conv{i}", c if i == 2 else cin, cout)
        if cin != cout:
            out[f"{name}.conv_shortcut.weight"] = (cout, cin, 1, 1, 1); out[f"{name}.conv_shortcut.bias"] = (cout,)

    causal("encoder.conv_in", cfg["in_channels"], boc[0])
    prev = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c, None)
        if i < len(boc) - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3); out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
        prev = c
    for j in range(2):
        resnet(f"encoder.mid_block.resnets.{j}", boc[-1], boc[-1], None)
    out["encoder.norm_out.weight"] = (boc[-1],); out["encoder.norm_out.bias"] = (boc[-1],)
    causal("encoder.conv_out", boc[-1], 2 * zc)

    rev = boc[::-1]
    causal("decoder.conv_in", zc, rev[0])
    for j in range(2):
        resnet(f"decoder.mid_block.resnets.{j}", rev[0], rev[0], zc)
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c, zc)
        if i < len(rev) - 1:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
        prev = c
    out["decoder.norm_out.norm_layer.weight"] = (rev[-1],); out["decoder.norm_out.norm_layer.bias"] = (rev[-1],)
    causal("decoder.norm_out.conv_y", zc, rev[-1], 1); causal("decoder.norm_out.conv_b", zc, rev[-1], 1)
    causal("decoder.conv_out", rev[-1], cfg["out_channels"])
    return out


def seeded_state(cfg: dict, seed: int, std: float = 0.04) -> SD:
    """deterministic weights from (names, shapes, seed): what the tests load into BOTH sides"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in state_shapes(cfg).items():
        if k.endswith("norm_layer.weight") or (".norm" in k and k.endswith(".weight") and len(shp) == 1) or k == "encoder.norm_out.weight":
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            sd[k] = torch.randn(shp, generator=g) * (1.0 / fan_in ** 0.5)
    return sd


# ------------------------------------------------------------------------------------------------------------------ layers
def causal_conv3d(sd: SD, name: str, x: torch.Tensor, cache: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """CogVideoXCausalConv3d.forward, pad_mode 'first'; x [B, C, T, H, W]"""
    w, b = sd[f"{name}.conv.weight"], sd[f"{name}.conv.bias"]
    kt, kh, kw = w.shape[2:]
    if kt > 1:
        front = [cache] if cache is not None else [x[:, :, :1]] * (kt - 1)
        x = torch.cat(front + [x], dim=2)
    new_cache = x[:, :, x.shape[2] - kt + 1:].clone()
    return F.conv3d(x, w, b, padding=(0, kh // 2, kw // 2)), new_cache


def _nearest_to(zq: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    return F.interpolate(zq, size=tuple(size))


def spatial_norm3d(sd: SD, name: str, f: torch.Tensor, zq: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    T = f.shape[2]
    if T > 1 and T % 2 == 1:
        zq = torch.cat([_nearest_to(zq[:, :, :1], (1,) + tuple(f.shape[3:])), _nearest_to(zq[:, :, 1:], (T - 1,) + tuple(f.shape[3:]))], dim=2)
    else:
        zq = _nearest_to(zq, f.shape[2:])
    y, _ = causal_conv3d(sd, f"{name}.conv_y", zq, None)
    b, _ = causal_conv3d(sd, f"{name}.conv_b", zq, None)
    return F.group_norm(f, groups, sd[f"{name}.norm_layer.weight"], sd[f"{name}.norm_layer.bias"], eps) * y + b


def resnet3d(sd: SD, name: str, x: torch.Tensor, zq: Optional[torch.Tensor], cache: dict, groups: int, eps: float) -> torch.Tensor:
    def norm(i, h):
        if zq is None:
            return F.group_norm(h, groups, sd[f"{name}.norm{i}.weight"], sd[f"{name}.norm{i}.bias"], eps)
        return spatial_norm3d(sd, f"{name}.norm{i}", h, zq, groups, eps)
    h = F.silu(norm(1, x))
    h, cache[f"{name}.conv1"] = causal_conv3d(sd, f"{name}.conv1", h, cache.get(f"{name}.conv1"))
    h = F.silu(norm(2, h))
    h, cache[f"{name}.conv2"] = causal_conv3d(sd, f"{name}.conv2", h, cache.get(f"{name}.conv2"))
    if f"{name}.conv_shortcut.weight" in sd:
        x = F.conv3d(x, sd[f"{name}.conv_shortcut.weight"], sd[f"{name}.conv_shortcut.bias"])
    return h + x


def upsample3d(sd: SD, name: str, x: torch.Tensor, compress_time: bool) -> torch.Tensor:
    B, C, T, H, W = x.shape
    if compress_time:
        if T > 1 and T % 2 == 1:
            first = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
            x = torch.cat([first, F.interpolate(x[:, :, 1:], scale_factor=2.0)], dim=2)
        elif T > 1:
            x = F.interpolate(x, scale_factor=2.0)
        else:
            x = F.interpolate(x[:, :, 0], scale_factor=2.0)[:, :, None]
    else:
        x = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), scale_factor=2.0)
        x = x.reshape(B, T, C, 2 * H, 2 * W).permute(0, 2, 1, 3, 4)
    B, C, T, H, W = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), sd[f"{name}.conv.weight"], sd[f"{name}.conv.bias"], padding=1)
    return y.reshape(B, T, -1, H, W).permute(0, 2, 1, 3, 4)


def downsample3d(sd: SD, name: str, x: torch.Tensor, compress_time: bool) -> torch.Tensor:
    if compress_time:
        B, C, T, H, W = x.shape
        v = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T)
        if T % 2 == 1:
            first, rest = v[..., 0], v[..., 1:]
            if rest.shape[-1] > 0:
                rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
            v = torch.cat([first[..., None], rest], dim=-1)
        else:
            v = F.avg_pool1d(v, kernel_size=2, stride=2)
        x = v.reshape(B, H, W, C, v.shape[-1]).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1))
    B, C, T, H, W = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), sd[f"{name}.conv.weight"], sd[f"{name}.conv.bias"], stride=2)
    return y.reshape(B, T, -1, y.shape[-2], y.shape[-1]).permute(0, 2, 1, 3, 4)


# ------------------------------------------------------------------------------------------------------------------ encoder / decoder
def decoder3d(sd: SD, cfg: dict, z: torch.Tensor, cache: dict) -> torch.Tensor:
    """CogVideoXDecoder3D.forward on one frame batch; z [B, zc, T, h, w]"""
    g, eps, L = cfg["norm_num_groups"], cfg["norm_eps"], cfg["layers_per_block"]
    n_blocks = len(cfg["block_out_channels"])
    levels = int(round(torch.log2(torch.tensor(float(cfg["temporal_compression_ratio"]))).item()))
    h, cache["decoder.conv_in"] = causal_conv3d(sd, "decoder.conv_in", z, cache.get("decoder.conv_in"))
    for j in range(2):
        h = resnet3d(sd, f"decoder.mid_block.resnets.{j}", h, z, cache, g, eps)
    for i in range(n_blocks):
        for j in range(L + 1):
            h = resnet3d(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, z, cache, g, eps)
        if i < n_blocks - 1:
            h = upsample3d(sd, f"decoder.up_blocks.{i}.upsamplers.0", h, i < levels)
    h = F.silu(spatial_norm3d(sd, "decoder.norm_out", h, z, g, eps))
    h, cache["decoder.conv_out"] = causal_conv3d(sd, "decoder.conv_out", h, cache.get("decoder.conv_out"))
    return h


def encoder3d(sd: SD, cfg: dict, x: torch.Tensor, cache: dict) -> torch.Tensor:
    """CogVideoXEncoder3D.forward on one frame batch; x [B, 3, T, H, W] -> moments [B, 2 zc, T', H / 8, W / 8]"""
    g, eps, L = cfg["norm_num_groups"], cfg["norm_eps"], cfg["layers_per_block"]
    n_blocks = len(cfg["block_out_channels"])
    levels = int(round(torch.log2(torch.tensor(float(cfg["temporal_compression_ratio"]))).item()))
    h, cache["encoder.conv_in"] = causal_conv3d(sd, "encoder.conv_in", x, cache.get("encoder.conv_in"))
    for i in range(n_blocks):
        for j in range(L):
            h = resnet3d(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, cache, g, eps)
        if i < n_blocks - 1:
            h = downsample3d(sd, f"encoder.down_blocks.{i}.downsamplers.0", h, i < levels)
    for j in range(2):
        h = resnet3d(sd, f"encoder.mid_block.resnets.{j}", h, None, cache, g, eps)
    h = F.silu(F.group_norm(h, g, sd["encoder.norm_out.weight"], sd["encoder.norm_out.bias"], 1e-6))
    h, cache["encoder.conv_out"] = causal_conv3d(sd, "encoder.conv_out", h, cache.get("encoder.conv_out"))
    return h


def frame_batches(num_frames: int, batch: int) -> List[Tuple[int, int]]:
    """the [start, end) frame ranges of `_decode` / `_encode`: the remainder joins the first batch"""
    n, rem = max(num_frames // batch, 1), num_frames % batch
    return [(batch * k + (0 if k == 0 else rem), batch * (k + 1) + rem) for k in range(n)]


def _run_batched(fn, sd, cfg, x, batch):
    cache: dict = {}
    return torch.cat([fn(sd, cfg, x[:, :, a:b], cache) for a, b in frame_batches(x.shape[2], batch)], dim=2)


def tile_geometry(cfg: dict) -> dict:
    """the tiling constants of AutoencoderKLCogVideoX.__init__ (tile_overlap_factor_height 1/6, _width 1/5)"""
    sh, sw = cfg["sample_height"] // 2, cfg["sample_width"] // 2
    f = 2 ** (len(cfg["block_out_channels"]) - 1)
    lh, lw = int(sh / f), int(sw / f)
    oh, ow = 1 / 6, 1 / 5
    return dict(sample=(sh, sw), latent=(lh, lw),
                dec_overlap=(int(lh * (1 - oh)), int(lw * (1 - ow))), dec_blend=(int(sh * oh), int(sw * ow)), dec_limit=(sh - int(sh * oh), sw - int(sw * ow)),
                enc_overlap=(int(sh * (1 - oh)), int(sw * (1 - ow))), enc_blend=(int(lh * oh), int(lw * ow)), enc_limit=(lh - int(lh * oh), lw - int(lw * ow)))


def _blend_v(a, b, extent):
    extent = min(a.shape[3], b.shape[3], extent)
    for y in range(extent):
        b[:, :, :, y, :] = a[:, :, :, -extent + y, :] * (1 - y / extent) + b[:, :, :, y, :] * (y / extent)
    return b


def _blend_h(a, b, extent):
    extent = min(a.shape[4], b.shape[4], extent)
    for x in range(extent):
        b[:, :, :, :, x] = a[:, :, :, :, -extent + x] * (1 - x / extent) + b[:, :, :, :, x] * (x / extent)
    return b


def _tiled(fn, sd, cfg, x, batch, tile, overlap, blend, limit):
    H, W = x.shape[3:]
    rows = [[_run_batched(fn, sd, cfg, x[:, :, :, i:i + tile[0], j:j + tile[1]], batch) for j in range(0, W, overlap[1])] for i in range(0, H, overlap[0])]
    out_rows = []
    for i, row in enumerate(rows):
        out = []
        for j, t in enumerate(row):
            if i > 0:
                t = _blend_v(rows[i - 1][j], t, blend[0])
            if j > 0:
                t = _blend_h(row[j - 1], t, blend[1])
            out.append(t[:, :, :, :limit[0], :limit[1]])
        out_rows.append(torch.cat(out, dim=4))
    return torch.cat(out_rows, dim=3)


def decode(sd: SD, cfg: dict, z: torch.Tensor, tiling: bool = True) -> torch.Tensor:
    """AutoencoderKLCogVideoX.decode(z).sample with slicing (per-sample) and, if `tiling`, tiled decode past the latent tile size; z [B, zc, T, h, w]"""
    g = tile_geometry(cfg)
    outs = []
    for zs in z.split(1):
        if tiling and (zs.shape[4] > g["latent"][1] or zs.shape[3] > g["latent"][0]):
            outs.append(_tiled(decoder3d, sd, cfg, zs, 2, g["latent"], g["dec_overlap"], g["dec_blend"], g["dec_limit"]))
        else:
            outs.append(_run_batched(decoder3d, sd, cfg, zs, 2))
    return torch.cat(outs)


def encode_moments(sd: SD, cfg: dict, x: torch.Tensor, tiling: bool = True) -> torch.Tensor:
    """the `h` of AutoencoderKLCogVideoX.encode (mean | logvar along dim 1); x [B, 3, T, H, W]"""
    g = tile_geometry(cfg)
    outs = []
    for xs in x.split(1):
        if tiling and (xs.shape[4] > g["sample"][1] or xs.shape[3] > g["sample"][0]):
            outs.append(_tiled(encoder3d, sd, cfg, xs, 8, g["sample"], g["enc_overlap"], g["enc_blend"], g["enc_limit"]))
        else:
            outs.append(_run_batched(encoder3d, sd, cfg, xs, 8))
    return torch.cat(outs)


def encode_sample(sd: SD, cfg: dict, x: torch.Tensor, noise: torch.Tensor, tiling: bool = True) -> torch.Tensor:
    """DiagonalGaussianDistribution(h).sample(): mean + exp(0.5 clamp(logvar, -30, 20)) * noise"""
    mean, logvar = encode_moments(sd, cfg, x, tiling).chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
