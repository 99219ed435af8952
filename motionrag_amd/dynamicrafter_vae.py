"""KL-VAE decode after the DynamiCrafter denoising loop (the per-frame KL-VAE) on the HIP kernels.

Mirrors, with the reference's parameter names (so `first_stage_model.*` checkpoint keys load unchanged):
  * `Decoder` / `ResnetBlock` / `AttnBlock` / `Upsample`  src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/networks/ae_modules.py:27-79,116-132,156-215,472-584
  * `AutoencoderKL.decode`                                lvdm/models/autoencoder.py:104-107  (`post_quant_conv` then the decoder)
  * `LatentDiffusion.decode_core` / `decode_first_stage`  lvdm/models/ddpm3d.py:668-690      (`1 / scale_factor * z`, frames folded into the batch)
  * `Encoder` / `Downsample`, `AutoencoderKL.encode`, `DiagonalGaussianDistribution`, `get_first_stage_encoding` / `encode_first_stage`
        ae_modules.py:93-113,370-470; autoencoder.py:97-102; lvdm/distributions.py:24-43; ddpm3d.py:633-666   (the conditioning image's latent)

Data path (channels-last rows, all frames of a clip in one batch -- 16 x 576 x 1024 x 128 bf16 is 2.4 GB of the 288 GB):
  GroupNorm(32, eps 1e-6) + swish      -> `mrag_groupnorm_bf16` (statistics + fused SiLU)
  3x3 convolutions (+ nearest x2)      -> `mrag_conv_bf16` implicit GEMM, the ResnetBlock's `x + h` in the epilogue; conv_in (3 / 4 channels) through the row gather
  Downsample (pad (0,1,0,1), stride 2) -> the same implicit GEMM with `asym_pad` (taps start at the pixel itself, zero row / column at the bottom / right)
  1x1 convolutions / nin_shortcut      -> `mrag_gemm_bf16`
  AttnBlock (1 head, head_dim 512)     -> q / k projections, S = q k^T as a GEMM per frame, `mrag_softmax_rows_bf16`, V^T produced directly by a GEMM with swapped
                                          operands (no transpose pass), O = P V^T^T as a GEMM (+ the value bias: rows of P sum to 1), proj_out with the residual.
GPU only; no CPU fallback.
"""
from typing import Optional, Sequence

import torch
from torch import nn

from . import ops
from .dynamicrafter import _CACHE, _lin_w, conv3x3


def _b(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = t.detach()
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


def Normalize(in_channels: int, num_groups: int = 32) -> nn.GroupNorm:
    """ae_modules.py:16-17"""
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _gn_swish(x: torch.Tensor, gn: nn.GroupNorm, silu: bool = True) -> torch.Tensor:
    """x [N, H, W, C] -> same; GroupNorm over (H W, C / G) per (n, group) [+ x * sigmoid(x)]"""
    N, H, W, C = x.shape
    return ops.groupnorm(x.view(N, H * W, C), _b(gn.weight), _b(gn.bias), gn.num_groups, gn.eps, silu=silu).view(N, H, W, C)


class ResnetBlock(nn.Module):
    """ae_modules.py:156-215 with temb_channels = 0 (the autoencoder has no timestep embedding)"""

    def __init__(self, *, in_channels: int, out_channels: Optional[int] = None, conv_shortcut: bool = False, dropout: float = 0.0, temb_channels: int = 0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if temb_channels:
            raise NotImplementedError("the KL-VAE decoder runs with temb_channels = 0")
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x: torch.Tensor, temb=None) -> torch.Tensor:
        h = conv3x3(_gn_swish(x, self.norm1), self.conv1)
        h = _gn_swish(h, self.norm2)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                x = conv3x3(x, self.conv_shortcut)
            else:
                x = ops.linear(x, _b(_lin_w(self.nin_shortcut)), _b(self.nin_shortcut.bias))
        return conv3x3(h, self.conv2, resid=x)                               # x + h in the convolution's epilogue


class AttnBlock(nn.Module):
    """ae_modules.py:27-79: one head over all pixels of a frame, head_dim = channels (512)"""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        N, H, W, C = x.shape
        S = H * W
        h = _gn_swish(x, self.norm, silu=False).view(N, S, C)
        q = ops.linear(h, _b(_lin_w(self.q)), _b(self.q.bias))                                   # [N, S, C]
        k = ops.linear(h, _b(_lin_w(self.k)), _b(self.k.bias))
        wv = _b(_lin_w(self.v))
        a = torch.empty(N, S, C, dtype=torch.bfloat16, device=x.device)
        scores = torch.empty(S, S, dtype=torch.bfloat16, device=x.device)                         # one frame at a time: 170 MB at 72 x 128
        for n in range(N):
            ops.linear(q[n], k[n], out=scores)                                                    # w_[i, j] = sum_c q[i, c] k[j, c]        (:67)
            ops.softmax_rows(scores, scale=float(C) ** -0.5, out=scores)                          # * c^-0.5, softmax over j                (:68-69)
            vt = ops.linear(wv, h[n])                                                             # V^T [C, S] = Wv h^T: no transpose pass
            ops.linear(scores, vt, _b(self.v.bias), out=a[n])                                     # sum_j w_[i, j] v[j, c] (+ b_v: rows sum to 1)  (:72-75)
        return ops.linear(a, _b(_lin_w(self.proj_out)), _b(self.proj_out.bias), epilogue=ops.EPI_RESID, resid=x.view(N, S, C)).view(N, H, W, C)


class Upsample(nn.Module):
    """ae_modules.py:116-132: nearest x2, then a 3x3 convolution -- one launch (the convolution's gather reads pixel (y >> 1, x >> 1))"""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv = False is not on the reference's path")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return conv3x3(x, self.conv, upsample=True)


class _Level(nn.Module):
    pass


def _make_level(widths, attn_here: bool) -> "_Level":
    """one resolution level: ResnetBlocks over consecutive (in, out) widths [+ an AttnBlock after each when the YAML's attn_resolutions lists this resolution]"""
    lv = _Level()
    lv.block = nn.ModuleList(ResnetBlock(in_channels=a, out_channels=b) for a, b in zip(widths[:-1], widths[1:]))
    lv.attn = nn.ModuleList(AttnBlock(b) for b in widths[1:]) if attn_here else nn.ModuleList()
    return lv


def _make_mid(width: int) -> "_Level":
    mid = _Level()
    mid.block_1, mid.attn_1, mid.block_2 = ResnetBlock(in_channels=width, out_channels=width), AttnBlock(width), ResnetBlock(in_channels=width, out_channels=width)
    return mid


def _run_level(lv: "_Level", h: torch.Tensor) -> torch.Tensor:
    for j, blk in enumerate(lv.block):
        h = blk(h)
        if len(lv.attn):
            h = lv.attn[j](h)
    return h


class Decoder(nn.Module):
    """ae_modules.py:472-584 (attn_type 'vanilla').  Module tree = the reference's (`conv_in`, `mid.{block_1, attn_1, block_2}`, `up.{level}.{block.N, attn.N, upsample.conv}`
    with level 0 the finest, `norm_out`, `conv_out`), built from a per-level width plan instead of the reference's running counters."""

    def __init__(self, *, ch: int, out_ch: int, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int, attn_resolutions: Sequence[int] = (), dropout: float = 0.0,
                 resamp_with_conv: bool = True, in_channels: int = 3, resolution: int = 256, z_channels: int = 4, give_pre_end: bool = False, tanh_out: bool = False,
                 use_linear_attn: bool = False, attn_type: str = "vanilla", **ignorekwargs):
        super().__init__()
        if use_linear_attn or attn_type != "vanilla":
            raise NotImplementedError("only the vanilla AttnBlock is on the reference's path")
        if tanh_out:
            raise NotImplementedError("tanh_out is not used by the shipped configs")
        levels = len(ch_mult)
        self.num_resolutions, self.num_res_blocks, self.give_pre_end, self.tanh_out, self.out_ch = levels, num_res_blocks, give_pre_end, tanh_out, out_ch
        widths = [ch * m for m in ch_mult]                                     # level i runs at width ch * ch_mult[i]; the decoder walks them coarse -> fine
        self.conv_in = nn.Conv2d(z_channels, widths[-1], 3, 1, 1)
        self.mid = _make_mid(widths[-1])
        ups, entering = [None] * levels, widths[-1]
        for i in range(levels - 1, -1, -1):
            lv = _make_level([entering] + [widths[i]] * (num_res_blocks + 1), (resolution >> i) in attn_resolutions)
            if i > 0:
                lv.upsample = Upsample(widths[i], resamp_with_conv)
            ups[i], entering = lv, widths[i]
        self.up = nn.ModuleList(ups)
        self.norm_out = Normalize(widths[0])
        self.conv_out = nn.Conv2d(widths[0], out_ch, 3, 1, 1)
        self.peak_channels = ch * ch_mult[min(1, levels - 1)]                  # channels of the finest level's upsampled input: the largest activation

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z [N, H, W, z_channels] channels-last -> [N, H * 2^(levels-1), W * 2^(levels-1), out_ch]"""
        h = _conv_small_cin(z, self.conv_in, "vae_dec_in")
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i in range(self.num_resolutions - 1, -1, -1):
            h = _run_level(self.up[i], h)
            if i > 0:
                h = self.up[i].upsample(h)
        if self.give_pre_end:
            return h
        h = _gn_swish(h, self.norm_out)
        conv = self.conv_out                                                  # out_ch = 3: the GEMM wants N % 4 == 0 -> one zero output channel, sliced off

        def build():
            cout, cin = conv.weight.shape[:2]
            pad = (-cout) % 4
            w = _b(conv.weight).permute(0, 2, 3, 1).reshape(cout, 9 * cin)
            b = _b(conv.bias)
            if pad:
                w = torch.cat([w, torch.zeros(pad, 9 * cin, dtype=w.dtype, device=w.device)], 0)
                b = torch.cat([b, torch.zeros(pad, dtype=b.dtype, device=b.device)], 0)
            return w.contiguous(), b.contiguous()
        wk, bk = _CACHE.get(("vae_out", id(conv)), (conv.weight, conv.bias), build)
        y = ops.conv_implicit(h.contiguous(), wk, bk, ops.CONV_3X3)
        return y[..., :self.out_ch]


class Downsample(nn.Module):
    """ae_modules.py:93-113: F.pad(x, (0, 1, 0, 1)) + Conv2d(3, stride 2, padding 0) -- one implicit-GEMM launch"""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv = False is not on the reference's path")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        conv = self.conv
        C = x.shape[-1]
        if C % 64:
            raise NotImplementedError("Downsample: channel counts are multiples of 64 on the reference's path")
        wk = _CACHE.get(("c3", id(conv)), conv.weight, lambda: _b(conv.weight).permute(0, 2, 3, 1).reshape(conv.weight.shape[0], 9 * C).contiguous())
        return ops.conv_implicit(x.contiguous(), wk, _b(conv.bias), ops.CONV_3X3, stride=2, asym_pad=True)


def _conv_small_cin(z: torch.Tensor, conv: nn.Conv2d, tag: str) -> torch.Tensor:
    """3x3 convolution from 3 (RGB) or 4 (latent) channels: the row gather moves 16-byte (8-channel) granules, so the input and the kernel get zero channels"""
    N, H, W, cz = z.shape
    cp = (cz + 7) // 8 * 8
    if cp != cz:
        z = torch.nn.functional.pad(z, (0, cp - cz))
    kp = ops._kpad(9 * cp)

    def build():
        w = torch.nn.functional.pad(_b(conv.weight), (0, 0, 0, 0, 0, cp - cz)).permute(0, 2, 3, 1).reshape(conv.weight.shape[0], 9 * cp)
        return torch.nn.functional.pad(w, (0, kp - 9 * cp)).contiguous()
    wk = _CACHE.get((tag, id(conv), cp), conv.weight, build)
    return ops.linear(ops.im2col3x3(z.contiguous()), wk, _b(conv.bias)).view(N, H, W, -1)


class Encoder(nn.Module):
    """ae_modules.py:370-470; module tree `conv_in`, `down.{level}.{block.N, attn.N, downsample.conv}` (level 0 the finest), `mid`, `norm_out`, `conv_out`"""

    def __init__(self, *, ch: int, out_ch: int = 3, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int, attn_resolutions: Sequence[int] = (), dropout: float = 0.0,
                 resamp_with_conv: bool = True, in_channels: int = 3, resolution: int = 256, z_channels: int = 4, double_z: bool = True, use_linear_attn: bool = False,
                 attn_type: str = "vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn or attn_type != "vanilla":
            raise NotImplementedError("only the vanilla AttnBlock is on the reference's path")
        levels = len(ch_mult)
        self.num_resolutions, self.num_res_blocks = levels, num_res_blocks
        widths = [ch * m for m in ch_mult]
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        downs, entering = [], ch
        for i in range(levels):                                               # fine -> coarse
            lv = _make_level([entering] + [widths[i]] * num_res_blocks, (resolution >> i) in attn_resolutions)
            if i < levels - 1:
                lv.downsample = Downsample(widths[i], resamp_with_conv)
            downs.append(lv)
            entering = widths[i]
        self.down = nn.ModuleList(downs)
        self.mid = _make_mid(widths[-1])
        self.norm_out = Normalize(widths[-1])
        self.conv_out = nn.Conv2d(widths[-1], (2 if double_z else 1) * z_channels, 3, 1, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, H, W, 3] channels-last -> moments [N, H / 2^(levels-1), W / 2^(levels-1), 2 z_channels]"""
        h = _conv_small_cin(x, self.conv_in, "vae_enc_in")
        for i, lv in enumerate(self.down):
            h = _run_level(lv, h)
            if i < self.num_resolutions - 1:
                h = lv.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return conv3x3(_gn_swish(h, self.norm_out), self.conv_out)


class DiagonalGaussianDistribution:
    """The posterior `AutoencoderKL.encode` returns (lvdm/distributions.py:24-43): `parameters` [N, 2 z, h, w] = mean | log-variance (clamped to [-30, 20]).
    The inference path reads `sample(noise)`, `mode()`, `mean`, `std`, `logvar`, `var`; a few KB per frame, kept in fp32 on the device."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters, self.deterministic = parameters, deterministic
        moments = parameters.float()
        z = moments.shape[1] // 2
        self.mean, self.logvar = moments[:, :z], moments[:, z:].clamp(-30.0, 20.0)
        spread = torch.zeros_like(self.mean) if deterministic else None
        self.std = spread if deterministic else (0.5 * self.logvar).exp()
        self.var = spread if deterministic else self.logvar.exp()

    def sample(self, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        eps = torch.randn(self.mean.shape) if noise is None else noise          # like the reference, the default draw is on the host generator
        return torch.addcmul(self.mean, self.std, eps.to(device=self.mean.device, dtype=torch.float32))

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKL(nn.Module):
    """lvdm/models/autoencoder.py:13-107.  Constructor keywords as the reference's YAML (`ddconfig`, `embed_dim`; the rest is accepted and ignored);
    state-dict keys `encoder.*`, `decoder.*`, `quant_conv.*`, `post_quant_conv.*` as the reference's (`loss.*` of a training checkpoint: strict=False)."""

    def __init__(self, ddconfig: dict, embed_dim: int, lossconfig=None, **_ignored):
        super().__init__()
        assert ddconfig["double_z"]
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim

    @torch.no_grad()
    def encode(self, x: torch.Tensor, **kwargs) -> DiagonalGaussianDistribution:
        """x [N, 3, H, W] in [-1, 1] -> posterior over z [N, embed_dim, H / 8, W / 8]  (autoencoder.py:97-102)"""
        if not x.is_cuda:
            raise ops.HipOnly("AutoencoderKL.encode: GPU tensors only")
        xc = x.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
        h = self.encoder(xc)
        moments = ops.linear(h, _b(_lin_w(self.quant_conv)), _b(self.quant_conv.bias))
        return DiagonalGaussianDistribution(moments.permute(0, 3, 1, 2))

    @torch.no_grad()
    def decode(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        """z [N, C, h, w] -> images [N, out_ch, 8 h, 8 w] (bf16), as autoencoder.py:104-107.  Frames are independent (the reference's `perframe_ae` decodes them one
        by one); here as many go through one batch as keep the largest activation -- the finest level's upsampled input -- below the convolution's 2^31-element
        limit (14 frames at 576 x 1024)."""
        if not z.is_cuda:
            raise ops.HipOnly("AutoencoderKL.decode: GPU tensors only")
        N, _, h, w = z.shape
        up = 2 ** (self.decoder.num_resolutions - 1)
        per_frame = (up * h) * (up * w) * self.decoder.peak_channels
        chunk = max(1, min(N, (2 ** 31 - 1) // per_frame))
        if chunk < N:
            chunk = max(1, N // -(-N // chunk))                                # even chunks (16 frames -> 2 x 8)
        outs = []
        for i in range(0, N, chunk):
            zc = z[i:i + chunk].to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()       # channels-last rows (a few KB per frame)
            zc = ops.linear(zc, _b(_lin_w(self.post_quant_conv)), _b(self.post_quant_conv.bias))
            outs.append(self.decoder(zc).permute(0, 3, 1, 2))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def encode_first_stage(first_stage_model: AutoencoderKL, x: torch.Tensor, scale_factor: float = 0.18215, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LatentDiffusion.encode_first_stage + get_first_stage_encoding (ddpm3d.py:633-666): x [b, 3, t, H, W] or [n, 3, H, W] in [-1, 1] ->
    scale_factor * posterior.sample(noise) (fp32).  `noise` = None draws like the reference (CPU generator); pass zeros for the posterior mean."""
    reshape_back = x.dim() == 5
    if reshape_back:
        b, c, t, h, w = x.shape
        x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    z = scale_factor * first_stage_model.encode(x).sample(noise=noise)
    if reshape_back:
        z = z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)
    return z


def decode_first_stage(first_stage_model: AutoencoderKL, z: torch.Tensor, scale_factor: float = 0.18215, perframe_ae: bool = True) -> torch.Tensor:
    """LatentDiffusion.decode_core (ddpm3d.py:668-686): z [b, c, t, h, w] -> video [b, 3, t, 8 h, 8 w].  `perframe_ae` only bounds memory in the reference (one frame
    per call, same arithmetic per frame); here all frames go through one batch."""
    reshape_back = z.dim() == 5
    if reshape_back:
        b, c, t, h, w = z.shape
        z = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    out = first_stage_model.decode((1.0 / scale_factor) * z.float())
    if reshape_back:
        out = out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4)
    return out


class FirstStage:
    """The first-stage methods `image_guided_synthesis` (dynamicrafter_pipeline.py) reads from the LatentVisualDiffusion-shaped model object, ddpm3d.py:633-690:
    mix into / attach to that object (`model.encode_first_stage = fs.encode_first_stage`, ...) or subclass."""

    def __init__(self, first_stage_model: AutoencoderKL, scale_factor: float = 0.18215, perframe_ae: bool = True):
        self.first_stage_model, self.scale_factor, self.perframe_ae = first_stage_model, scale_factor, perframe_ae

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        """ddpm3d.py:633-640: a posterior is sampled, a tensor passes through; both are scaled"""
        if hasattr(encoder_posterior, "sample") and hasattr(encoder_posterior, "mode"):
            return self.scale_factor * encoder_posterior.sample(noise=noise)
        if torch.is_tensor(encoder_posterior):
            return self.scale_factor * encoder_posterior
        raise NotImplementedError(f"no first-stage encoding for {type(encoder_posterior).__name__}")

    @torch.no_grad()
    def encode_first_stage(self, x: torch.Tensor) -> torch.Tensor:
        return encode_first_stage(self.first_stage_model, x, self.scale_factor)

    @torch.no_grad()
    def decode_first_stage(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        return decode_first_stage(self.first_stage_model, z, self.scale_factor, self.perframe_ae)
