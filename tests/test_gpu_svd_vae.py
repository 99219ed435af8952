"""SVD's temporal-decoder VAE (diffusers AutoencoderKLTemporalDecoder, third-party: parity UNPINNED) on the GPU against the fp32 restatement oracle/svd_vae_ref.py:
encode moments / mode and chunked decode at a reduced width and at the shipped configuration."""
import pytest
import torch

from oracle import svd_vae_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


def _make(block_out, layers, seed):
    from motionrag_amd import svd_vae as V
    torch.manual_seed(seed)
    m = V.AutoencoderKLTemporalDecoder(block_out_channels=block_out, layers_per_block=layers)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("mix_factor"):
                p.copy_(torch.randn(1, generator=g))                       # a real blend (the default 0.0 gives alpha = 0.5)
            elif p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m, sd


@pytest.mark.parametrize("block_out,layers,hw", [((64, 128), 1, (16, 24)), ((128, 256, 512, 512), 2, (8, 12))])
def test_svd_vae_encode_decode_vs_oracle(hip, block_out, layers, hw):
    m, sd = _make(block_out, layers, 31 + layers)
    m = m.to(DEV, torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    up = 2 ** (len(block_out) - 1)
    img = torch.rand(2, 3, hw[0] * up, hw[1] * up, generator=g) * 2 - 1
    post = m.encode(img.to(DEV)).latent_dist
    want_m = R.encoder(img.to(torch.bfloat16).float(), sd, len(block_out), layers)
    assert post.parameters.shape == want_m.shape == (2, 8, hw[0], hw[1])
    assert rel(post.parameters, want_m) <= 3e-2
    assert torch.equal(post.mode(), post.parameters.float()[:, :4])
    # decode two clips of 3 frames, as decode_latents drives it (one chunk = whole clips)
    z = torch.randn(6, 4, hw[0], hw[1], generator=g)
    out = m.decode(z.to(DEV), num_frames=3)
    want = R.decoder(z.to(torch.bfloat16).float(), sd, len(block_out), layers, num_frames=3)
    assert out.sample.shape == want.shape == (6, 3, hw[0] * up, hw[1] * up) and out[0] is out.sample
    assert rel(out.sample, want) <= 3e-2
    # frames of a clip are coupled (temporal blocks + time_conv_out): decoding with another clip length must change the result
    other = m.decode(z.to(DEV), num_frames=2).sample
    assert (other.float() - out.sample.float()).abs().max().item() > 0
    with pytest.raises(ValueError):
        m.decode(z.to(DEV), num_frames=4)


def test_svd_vae_shipped_key_layout():
    from motionrag_amd import svd_vae as V
    m = V.AutoencoderKLTemporalDecoder()
    keys = set(m.state_dict().keys())
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.resnets.1.conv2.bias", "encoder.down_blocks.2.downsamplers.0.conv.weight",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "encoder.mid_block.attentions.0.to_out.0.bias", "encoder.mid_block.attentions.0.group_norm.weight",
              "quant_conv.weight", "decoder.mid_block.resnets.1.temporal_res_block.conv1.weight", "decoder.mid_block.resnets.0.time_mixer.mix_factor",
              "decoder.up_blocks.2.resnets.0.spatial_res_block.conv_shortcut.weight", "decoder.up_blocks.0.upsamplers.0.conv.weight", "decoder.time_conv_out.weight",
              "decoder.conv_norm_out.weight"):
        assert k in keys, k
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in keys and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in keys
    assert tuple(m.state_dict()["decoder.time_conv_out.weight"].shape) == (3, 3, 3, 1, 1)
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 97.74) < 0.01      # the published size of SVD's VAE
