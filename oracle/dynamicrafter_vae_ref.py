"""TEST INFRASTRUCTURE (oracle) -- fp32 restatement of the DynamiCrafter KL-VAE decode path.  Only tests/ may import this.

Restates, function by function:
  Decoder.forward / ResnetBlock.forward / AttnBlock.forward / Upsample.forward
      src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/networks/ae_modules.py:545-584, 195-215, 54-79, 128-132
  AutoencoderKL.decode                        lvdm/models/autoencoder.py:104-107
  LatentDiffusion.decode_core                 lvdm/models/ddpm3d.py:668-686
Pinned by tests/golden/dc_vae.npz: outputs of the reference's own `AutoencoderKL` class (imported from /root/reference by oracle/gen_golden_vae.py through the
stub-import harness of oracle/gen_golden.py) on seeded weights and inputs."""
import torch
import torch.nn.functional as F


def _swish(x):
    return x * torch.sigmoid(x)


def _gn(x, sd, p):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _conv(x, sd, p, padding):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def resnet_block(x, sd, p):
    h = _conv(_swish(_gn(x, sd, p + ".norm1")), sd, p + ".conv1", 1)
    h = _conv(_swish(_gn(h, sd, p + ".norm2")), sd, p + ".conv2", 1)
    if p + ".nin_shortcut.weight" in sd:
        x = _conv(x, sd, p + ".nin_shortcut", 0)
    elif p + ".conv_shortcut.weight" in sd:
        x = _conv(x, sd, p + ".conv_shortcut", 1)
    return x + h


def attn_block(x, sd, p):
    h = _gn(x, sd, p + ".norm")
    q, k, v = _conv(h, sd, p + ".q", 0), _conv(h, sd, p + ".k", 0), _conv(h, sd, p + ".v", 0)
    b, c, hh, ww = q.shape
    w_ = torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), k.reshape(b, c, hh * ww)) * (int(c) ** -0.5)
    w_ = torch.softmax(w_, dim=2)
    h = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(h, sd, p + ".proj_out", 0)


def decoder(z, sd, num_resolutions, num_res_blocks, p="decoder"):
    h = _conv(z, sd, p + ".conv_in", 1)
    h = resnet_block(h, sd, p + ".mid.block_1")
    h = attn_block(h, sd, p + ".mid.attn_1")
    h = resnet_block(h, sd, p + ".mid.block_2")
    for i_level in reversed(range(num_resolutions)):
        for i_block in range(num_res_blocks + 1):
            h = resnet_block(h, sd, f"{p}.up.{i_level}.block.{i_block}")
            if f"{p}.up.{i_level}.attn.{i_block}.norm.weight" in sd:
                h = attn_block(h, sd, f"{p}.up.{i_level}.attn.{i_block}")
        if i_level != 0:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"{p}.up.{i_level}.upsample.conv", 1)
    return _conv(_swish(_gn(h, sd, p + ".norm_out")), sd, p + ".conv_out", 1)


def autoencoder_decode(z, sd, num_resolutions, num_res_blocks):
    return decoder(_conv(z, sd, "post_quant_conv", 0), sd, num_resolutions, num_res_blocks)


def decode_core(z, sd, num_resolutions, num_res_blocks, scale_factor=0.18215):
    """z [b, c, t, h, w] (encoder_type '2d'): frames folded into the batch, 1 / scale_factor, decode, unfolded"""
    b, c, t, h, w = z.shape
    z = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    out = autoencoder_decode(1.0 / scale_factor * z, sd, num_resolutions, num_res_blocks)
    return out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4)


def encoder(x, sd, num_resolutions, num_res_blocks, p="encoder"):
    """Encoder.forward, ae_modules.py:436-470 (Downsample: F.pad (0, 1, 0, 1) + stride-2 convolution without padding, :106-110)"""
    h = _conv(x, sd, p + ".conv_in", 1)
    for i_level in range(num_resolutions):
        for i_block in range(num_res_blocks):
            h = resnet_block(h, sd, f"{p}.down.{i_level}.block.{i_block}")
            if f"{p}.down.{i_level}.attn.{i_block}.norm.weight" in sd:
                h = attn_block(h, sd, f"{p}.down.{i_level}.attn.{i_block}")
        if i_level != num_resolutions - 1:
            q = f"{p}.down.{i_level}.downsample.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[q + ".weight"], sd[q + ".bias"], stride=2)
    h = resnet_block(h, sd, p + ".mid.block_1")
    h = attn_block(h, sd, p + ".mid.attn_1")
    h = resnet_block(h, sd, p + ".mid.block_2")
    return _conv(_swish(_gn(h, sd, p + ".norm_out")), sd, p + ".conv_out", 1)


def autoencoder_encode_moments(x, sd, num_resolutions, num_res_blocks):
    """AutoencoderKL.encode up to the posterior's parameters (autoencoder.py:97-100)"""
    return _conv(encoder(x, sd, num_resolutions, num_res_blocks), sd, "quant_conv", 0)


def first_stage_encoding(moments, noise, scale_factor=0.18215):
    """DiagonalGaussianDistribution(moments).sample(noise) * scale_factor (distributions.py:24-40, ddpm3d.py:633-640)"""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return scale_factor * (mean + std * noise)
