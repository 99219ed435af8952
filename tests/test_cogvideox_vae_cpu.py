"""CPU side of the CogVideoX VAE row: the product's module tree carries exactly the parameter names / shapes of diffusers' AutoencoderKLCogVideoX as the
oracle lists them, frame batching and tile geometry agree between product and oracle, and the oracle is self-consistent (tiling a latent no larger than a
tile is the identity; frame batches with caches equal one causal pass where GroupNorm statistics do not couple the frames)."""
import torch

from oracle import cogvideox_vae_ref as R

TOY = dict(in_channels=3, out_channels=3, block_out_channels=(32, 64, 64, 64), layers_per_block=1, latent_channels=16, norm_eps=1e-6, norm_num_groups=32,
           temporal_compression_ratio=4, sample_height=96, sample_width=160, scaling_factor=0.7)


def test_module_tree_matches_diffusers_names():
    from motionrag_amd.cogvideox_vae import AutoencoderKLCogVideoX
    for cfg in (TOY, R.CONFIG_5B):
        with torch.device("meta"):
            m = AutoencoderKLCogVideoX(**cfg)
        have = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert have == R.state_shapes(cfg)
    assert sum(torch.Size(s).numel() for s in R.state_shapes(R.CONFIG_5B).values()) == 215_583_907      # the published size of the CogVideoX VAE


def test_frame_batches_and_tile_geometry():
    from motionrag_amd.cogvideox_vae import AutoencoderKLCogVideoX, frame_batches
    for n in (1, 2, 3, 4, 5, 13, 49):
        for b in (2, 8):
            assert frame_batches(n, b) == R.frame_batches(n, b)
    assert R.frame_batches(13, 2) == [(0, 3), (3, 5), (5, 7), (7, 9), (9, 11), (11, 13)]
    with torch.device("meta"):
        m = AutoencoderKLCogVideoX(**R.CONFIG_5B)
    g = R.tile_geometry(R.CONFIG_5B)
    assert (m.tile_sample_min_height, m.tile_sample_min_width) == g["sample"] == (240, 360)
    assert (m.tile_latent_min_height, m.tile_latent_min_width) == g["latent"] == (30, 45)
    assert g["dec_overlap"] == (25, 36) and g["dec_blend"] == (40, 72) and g["dec_limit"] == (200, 288)


def test_oracle_self_consistency():
    sd = R.seeded_state(TOY, 5)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 16, 3, 6, 10, generator=g)                      # exactly one tile: tiled == untiled
    assert torch.equal(R.decode(sd, TOY, z, tiling=True), R.decode(sd, TOY, z, tiling=False))
    # the conv cache makes two frame batches equal one pass of the same causal convolution
    x = torch.randn(1, 32, 6, 5, 7, generator=g)
    name = "decoder.up_blocks.3.resnets.0.conv2"
    whole, _ = R.causal_conv3d(sd, name, x, None)
    a, c = R.causal_conv3d(sd, name, x[:, :, :2], None)
    b, _ = R.causal_conv3d(sd, name, x[:, :, 2:], c)
    assert torch.allclose(torch.cat([a, b], 2), whole, atol=1e-5)
    y = R.decode(sd, TOY, torch.randn(1, 16, 5, 6, 10, generator=g), tiling=False)
    assert y.shape == (1, 3, 17, 48, 80)
    m = R.encode_moments(sd, TOY, torch.randn(1, 3, 9, 48, 80, generator=g), tiling=False)
    assert m.shape == (1, 32, 3, 6, 10)
