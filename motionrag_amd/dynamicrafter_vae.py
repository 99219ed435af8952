"""KL-VAE decode after the DynamiCrafter denoising loop (the per-frame KL-VAE) on the HIP kernels.

Mirrors, with the reference's parameter names (so `first_stage_model.*` checkpoint keys load unchanged):
  * `Decoder` / `ResnetBlock` / `AttnBlock` / `Upsample`  src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/networks/ae_modules.py:27-79,116-132,156-215,472-584
  * `AutoencoderKL.decode`                                lvdm/models/autoencoder.py:104-107  (`post_quant_conv` then the decoder)
  * `LatentDiffusion.decode_core` / `decode_first_stage`  lvdm/models/ddpm3d.py:668-690      (`1 / scale_factor * z`, frames folded into the batch)
Encoding (the conditioning image's latent) is not built: its stride-2 convolutions pad asymmetrically ((0, 1, 0, 1)), a gather the implicit-GEMM
convolution does not have yet -- `encode` raises.

Data path (channels-last rows, all frames of a clip in one batch -- 16 x 576 x 1024 x 128 bf16 is 2.4 GB of the 288 GB):
  GroupNorm(32, eps 1e-6) + swish      -> `mrag_groupnorm_bf16` (statistics + fused SiLU)
  3x3 convolutions (+ nearest x2)      -> `mrag_conv_bf16` implicit GEMM, the ResnetBlock's `x + h` in the epilogue; conv_in (4 channels) through the row gather
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


class Decoder(nn.Module):
    """ae_modules.py:472-584 (attn_type 'vanilla', no attention at the up levels unless `attn_resolutions` asks for it)"""

    def __init__(self, *, ch: int, out_ch: int, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int, attn_resolutions: Sequence[int] = (), dropout: float = 0.0,
                 resamp_with_conv: bool = True, in_channels: int = 3, resolution: int = 256, z_channels: int = 4, give_pre_end: bool = False, tanh_out: bool = False,
                 use_linear_attn: bool = False, attn_type: str = "vanilla", **ignorekwargs):
        super().__init__()
        if use_linear_attn or attn_type != "vanilla":
            raise NotImplementedError("only the vanilla AttnBlock is on the reference's path")
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.give_pre_end, self.tanh_out, self.out_ch = give_pre_end, tanh_out, out_ch
        if tanh_out:
            raise NotImplementedError("tanh_out is not used by the shipped configs")
        block_in = ch * ch_mult[-1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = _Level()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)
        self.peak_channels = ch * ch_mult[min(1, len(ch_mult) - 1)]            # channels of the finest level's upsampled input: the largest activation

    def _conv_in(self, z: torch.Tensor) -> torch.Tensor:
        """3x3 convolution from z_channels = 4: the row gather moves 16-byte (8-channel) granules, so the latent and the kernel get 4 zero channels"""
        N, H, W, cz = z.shape
        conv = self.conv_in
        cp = (cz + 7) // 8 * 8
        if cp != cz:
            z = torch.nn.functional.pad(z, (0, cp - cz))
        kp = ops._kpad(9 * cp)

        def build():
            w = torch.nn.functional.pad(_b(conv.weight), (0, 0, 0, 0, 0, cp - cz)).permute(0, 2, 3, 1).reshape(conv.weight.shape[0], 9 * cp)
            return torch.nn.functional.pad(w, (0, kp - 9 * cp)).contiguous()
        wk = _CACHE.get(("vae_in", id(conv), cp), conv.weight, build)
        return ops.linear(ops.im2col3x3(z.contiguous()), wk, _b(conv.bias)).view(N, H, W, -1)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z [N, H, W, z_channels] channels-last -> [N, H * 2^(levels-1), W * 2^(levels-1), out_ch]"""
        h = self._conv_in(z)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        for i_level in reversed(range(self.num_resolutions)):
            lv = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = lv.block[i_block](h)
                if len(lv.attn) > 0:
                    h = lv.attn[i_block](h)
            if i_level != 0:
                h = lv.upsample(h)
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


class AutoencoderKL(nn.Module):
    """lvdm/models/autoencoder.py:13-107, decode side.  Constructor keywords as the reference's YAML (`ddconfig`, `embed_dim`; the rest is accepted and ignored);
    state-dict keys `decoder.*`, `post_quant_conv.*` (an `encoder.*` / `quant_conv.*` / `loss.*` checkpoint loads with strict=False)."""

    def __init__(self, ddconfig: dict, embed_dim: int, lossconfig=None, **_ignored):
        super().__init__()
        self.decoder = Decoder(**ddconfig)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim

    def encode(self, x, **kwargs):
        raise NotImplementedError("KL-VAE encode is not built (asymmetric-padding stride-2 convolutions); decode is")

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
