"""TEST INFRASTRUCTURE (oracle) -- fp32 restatement of SVD's VAE, THIRD-PARTY `diffusers==0.32.2` `AutoencoderKLTemporalDecoder` (requirements.txt:10; not vendored, not
installed): `Encoder` (DownEncoderBlock2D x 4, UNetMidBlock2D with one single-head attention), `TemporalDecoder` (MidBlockTemporalDecoder, UpBlockTemporalDecoder of
SpatioTemporalResBlock = ResnetBlock2D + TemporalResnetBlock blended by a learned AlphaBlender with switch_spatial_to_temporal_mix, `time_conv_out`), `quant_conv`.
Anchored on the reference's call sites: src/projects/svd/pipelines/pipeline.py (`self.vae.encode(image).latent_dist.mode()`, `decode_latents` -> `vae.decode(latents,
num_frames=...)`), src/projects/svd/module.py:38-47.  **PARITY UNPINNED**: restated from the published architecture; the only checks are self-consistency (the spatial
half equals the pinned LVDM KL-VAE arithmetic of oracle/dynamicrafter_vae_ref.py: same ResNet / attention / asymmetric-padding blocks under other names)."""
import torch
import torch.nn.functional as F


def _gn(x, sd, p, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=eps)


def resnet2d(x, sd, p, eps=1e-6):
    h = F.conv2d(F.silu(_gn(x, sd, p + ".norm1", eps)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(h, sd, p + ".norm2", eps)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def temporal_resnet(x, sd, p, eps=1e-5):
    """x [b, c, t, h, w]: GroupNorm over (c/g, t, h, w), Conv3d (3, 1, 1)"""
    h = F.conv3d(F.silu(_gn(x, sd, p + ".norm1", eps)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=(1, 0, 0))
    h = F.conv3d(F.silu(_gn(h, sd, p + ".norm2", eps)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=(1, 0, 0))
    return x + h


def spatio_temporal_res(x, sd, p, num_frames):
    """SpatioTemporalResBlock(temb None, merge_strategy 'learned', switch_spatial_to_temporal_mix=True), image_only_indicator = 0"""
    s = resnet2d(x, sd, p + ".spatial_res_block", 1e-6)
    bf, c, h, w = s.shape
    s5 = s.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    t5 = temporal_resnet(s5, sd, p + ".temporal_res_block", 1e-5)
    alpha = 1.0 - torch.sigmoid(sd[p + ".time_mixer.mix_factor"].float())
    out = alpha * s5 + (1.0 - alpha) * t5
    return out.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


def attention(x, sd, p):
    """diffusers Attention(heads=1, dim_head=C, norm_num_groups=32, eps=1e-6, bias=True, residual_connection=True) on [B, C, H, W]"""
    b, c, h, w = x.shape
    t = _gn(x.reshape(b, c, h * w), sd, p + ".group_norm", 1e-6).transpose(1, 2)
    q, k, v = (F.linear(t, sd[p + f".to_{n}.weight"], sd[p + f".to_{n}.bias"]) for n in "qkv")
    a = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1) @ v
    a = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + a.transpose(1, 2).reshape(b, c, h, w)


def encoder(x, sd, n_blocks, layers_per_block):
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(n_blocks):
        for j in range(layers_per_block):
            h = resnet2d(h, sd, f"encoder.down_blocks.{i}.resnets.{j}")
        q = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        if q + ".weight" in sd:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[q + ".weight"], sd[q + ".bias"], stride=2)
    h = resnet2d(h, sd, "encoder.mid_block.resnets.0")
    h = attention(h, sd, "encoder.mid_block.attentions.0")
    h = resnet2d(h, sd, "encoder.mid_block.resnets.1")
    h = F.silu(_gn(h, sd, "encoder.conv_norm_out", 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])               # moments (mean | logvar)


def decoder(z, sd, n_blocks, layers_per_block, num_frames):
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = spatio_temporal_res(h, sd, "decoder.mid_block.resnets.0", num_frames)
    for j in range(1, layers_per_block):
        h = attention(h, sd, f"decoder.mid_block.attentions.{j - 1}")
        h = spatio_temporal_res(h, sd, f"decoder.mid_block.resnets.{j}", num_frames)
    for i in range(n_blocks):
        for j in range(layers_per_block + 1):
            h = spatio_temporal_res(h, sd, f"decoder.up_blocks.{i}.resnets.{j}", num_frames)
        q = f"decoder.up_blocks.{i}.upsamplers.0.conv"
        if q + ".weight" in sd:
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[q + ".weight"], sd[q + ".bias"], padding=1)
    h = F.silu(_gn(h, sd, "decoder.conv_norm_out", 1e-6))
    h = F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
    bf, c, hh, ww = h.shape
    h5 = h.reshape(bf // num_frames, num_frames, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = F.conv3d(h5, sd["decoder.time_conv_out.weight"], sd["decoder.time_conv_out.bias"], padding=(1, 0, 0))
    return h5.permute(0, 2, 1, 3, 4).reshape(bf, c, hh, ww)
