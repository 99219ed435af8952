"""The CogVideoX 3-D causal VAE (diffusers 0.32.2 AutoencoderKLCogVideoX, third-party: parity UNPINNED) on the GPU against the fp32 restatement
oracle/cogvideox_vae_ref.py: the causal 3x3x3 implicit GEMM with its conv cache, the spatially conditioned GroupNorm, the seam blend, and the
decoder / encoder with frame batches and tiling at a reduced width; the shipped configuration through size-independent properties."""
import pytest
import torch
import torch.nn.functional as F

from oracle import cogvideox_vae_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOY = dict(in_channels=3, out_channels=3, block_out_channels=(64, 128, 128, 128), layers_per_block=1, latent_channels=16, norm_eps=1e-6, norm_num_groups=32,
           temporal_compression_ratio=4, sample_height=96, sample_width=160, scaling_factor=0.7)


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


def bf(t):
    return t.to(torch.bfloat16).float()


def _model(cfg, seed):
    from motionrag_amd.cogvideox_vae import AutoencoderKLCogVideoX
    sd = {k: bf(v) for k, v in R.seeded_state(cfg, seed).items()}
    m = AutoencoderKLCogVideoX(**cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV, torch.bfloat16), sd


@pytest.mark.parametrize("T,H,W,cin,cout", [(1, 6, 10, 64, 64), (3, 9, 7, 128, 64), (2, 16, 24, 64, 192), (5, 20, 33, 64, 128)])
def test_causal_conv3d_with_cache_matches_conv3d(hip, T, H, W, cin, cout):
    """two consecutive frame batches through the same convolution: first-frame context, then the carried cache"""
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(T * 100 + H)
    w = bf(torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5)
    b = bf(torch.randn(cout, generator=g) * 0.1)
    sd = {"c.conv.weight": w, "c.conv.bias": b}
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, -1).contiguous().to(DEV, torch.bfloat16)
    cache_ref, cache = None, None
    for step in range(2):
        x = bf(torch.randn(1, cin, T, H, W, generator=g))
        want, cache_ref = R.causal_conv3d(sd, "c", x, cache_ref)
        stack = torch.empty(T + 2, H, W, cin, dtype=torch.bfloat16, device=DEV)
        stack[2:] = x[0].permute(1, 2, 3, 0).to(DEV)
        stack[:2] = cache if cache is not None else stack[2:3].expand(2, -1, -1, -1)
        resid = bf(torch.randn(T, H, W, cout, generator=g)).to(DEV, torch.bfloat16) if step else None
        got = ops.conv_implicit(stack, wk, b.to(DEV, torch.bfloat16), ops.CONV_3X3, t_frames=T, resid=resid)
        cache = stack[T:].clone()
        want_cl = want[0].permute(1, 2, 3, 0) + (resid.float().cpu() if step else 0)
        assert rel(got, want_cl) < 6e-3                                        # bf16 output rounding (2^-9 per element) over fp32 accumulation
        assert torch.equal(cache.float().cpu(), cache_ref[0].permute(1, 2, 3, 0))


@pytest.mark.parametrize("T,Tz,H,W,shift,C", [(1, 1, 8, 12, 0, 64), (3, 3, 8, 12, 0, 128), (5, 3, 16, 24, 1, 128), (9, 3, 32, 16, 2, 64), (4, 2, 16, 8, 1, 256), (8, 2, 32, 48, 3, 512)])
def test_spatial_norm_matches_oracle(hip, T, Tz, H, W, shift, C):
    from motionrag_amd import cogvideox_vae as V
    g = torch.Generator().manual_seed(T + 7 * C)
    norm = V.CogVideoXSpatialNorm3D(C, 16)
    sd = {}
    with torch.no_grad():
        for n, p in norm.named_parameters():
            p.copy_(bf(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.2) + (1.0 if n == "norm_layer.weight" else 0.0)))
            sd[f"n.{n}"] = p.detach().clone()
    norm = norm.to(DEV, torch.bfloat16)
    f = bf(torch.randn(1, C, T, H, W, generator=g) * 1.5 + 0.3)
    zq = bf(torch.randn(1, 16, Tz, H >> shift, W >> shift, generator=g))
    want = F.silu(R.spatial_norm3d(sd, "n", f, zq, 32, 1e-6))[0].permute(1, 2, 3, 0)
    zq64 = V._pad_channels(zq[0].permute(1, 2, 3, 0).to(DEV, torch.bfloat16)).contiguous()
    stack = V._norm_into_stack(f[0].permute(1, 2, 3, 0).contiguous().to(DEV, torch.bfloat16), norm, zq64, 32, 1e-6)
    assert rel(stack[2:], want) < 8e-3                                          # the bf16 rounding of the latent-resolution conv_y / conv_b maps, then of the output


def test_blend_tile_matches_oracle_loops(hip):
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(3)
    T, th, tw, C = 3, 12, 20, 4
    tiles = [[bf(torch.randn(1, C, T, th if i < 1 else 7, tw if j < 2 else 9, generator=g)) for j in range(3)] for i in range(2)]
    dev = [[t[0].permute(1, 2, 3, 0).contiguous().to(DEV, torch.bfloat16) for t in row] for row in tiles]
    for i in range(2):
        for j in range(3):
            if i:
                tiles[i][j] = R._blend_v(tiles[i - 1][j], tiles[i][j], 5)
            if j:
                tiles[i][j] = R._blend_h(tiles[i][j - 1], tiles[i][j], 12)
            ops.blend_tile(dev[i][j], dev[i - 1][j] if i else None, dev[i][j - 1] if j else None, 5, 12)
            assert rel(dev[i][j], tiles[i][j][0].permute(1, 2, 3, 0)) < 4e-3


@pytest.mark.parametrize("frames,tiling", [(1, False), (5, False), (5, True), (4, True)])
def test_decode_matches_oracle(hip, frames, tiling):
    m, sd = _model(TOY, 11)
    if tiling:
        m.enable_tiling()
    m.enable_slicing()
    g = torch.Generator().manual_seed(frames)
    z = bf(torch.randn(2 if frames == 5 and not tiling else 1, 16, frames, 12, 20, generator=g))
    want = R.decode(sd, TOY, z, tiling=tiling)
    got = m.decode(z.to(DEV)).sample
    assert got.dtype == torch.bfloat16
    assert rel(got, want) < 3e-2                                                # ~20 bf16 convolutions deep against fp32


@pytest.mark.parametrize("frames,hw,tiling", [(1, (96, 160), True), (9, (48, 80), False), (17, (32, 48), False)])
def test_encode_matches_oracle(hip, frames, hw, tiling):
    m, sd = _model(TOY, 12)
    if tiling:
        m.enable_tiling()
    g = torch.Generator().manual_seed(frames)
    x = bf(torch.rand(1, 3, frames, *hw, generator=g) * 2 - 1)
    want = R.encode_moments(sd, TOY, x, tiling=tiling)
    post = m.encode(x.to(DEV)).latent_dist
    assert rel(post.parameters, want) < 3e-2
    noise = torch.randn(post.mean.shape, generator=g)
    want_s = R.encode_sample(sd, TOY, x, noise, tiling=tiling)
    assert rel(post.sample(noise=noise.to(DEV)), want_s) < 3e-2


def test_tiles_on_side_streams_equal_serial_tiles(hip):
    m, _ = _model(TOY, 15)
    m.enable_tiling()
    g = torch.Generator().manual_seed(9)
    z = bf(torch.randn(1, 16, 5, 12, 20, generator=g)).to(DEV)
    m.tile_streams = 1
    serial = m.decode(z).sample
    m.tile_streams = 3
    for _ in range(3):
        assert torch.equal(m.decode(z).sample, serial)


def test_pipeline_decode_latents_through_the_vae(hip):
    """CogVideoXImageToVideoCTPipeline.decode_latents (1 / scaling_factor, [b, F, 16, h, w] -> [b, 3, f, H, W]) on this class"""
    from motionrag_amd.cogvideox import CogVideoXImageToVideoCTPipeline
    m, sd = _model(TOY, 13)
    pipe = CogVideoXImageToVideoCTPipeline(vae=m)
    g = torch.Generator().manual_seed(1)
    lat = bf(torch.randn(1, 3, 16, 6, 10, generator=g))
    got = pipe.decode_latents(lat.to(DEV))
    want = R.decode(sd, TOY, lat.permute(0, 2, 1, 3, 4) / 0.7, tiling=False)
    assert rel(got, want) < 3e-2


def test_shipped_configuration_one_tile_properties(hip):
    """CogVideoX-5B's VAE width (215.6 M parameters) on one 30 x 45 latent tile of 5 frames: shape, finiteness, run-to-run determinism, and causality --
    the first frame batch's output does not depend on later latent frames"""
    m, _ = _model(R.CONFIG_5B, 14)
    assert sum(p.numel() for p in m.parameters()) == 215_583_907
    g = torch.Generator().manual_seed(2)
    z = bf(torch.randn(1, 16, 5, 30, 45, generator=g)).to(DEV)
    a = m.decode(z).sample
    assert a.shape == (1, 3, 17, 240, 360) and torch.isfinite(a.float()).all()
    assert torch.equal(a, m.decode(z).sample)
    z2 = z.clone(); z2[:, :, 3:] = 0
    assert torch.equal(m.decode(z2).sample[:, :, :9], a[:, :, :9])


def test_shipped_width_decode_and_encode_match_oracle(hip):
    """the full 128 / 256 / 256 / 512 widths with four resnets per up block on a small latent (two frame batches: 3 + 2 latent frames -> 17 frames), and the
    shipped encoder on a single image -- against the fp32 restatement on the host cores"""
    m, sd = _model(R.CONFIG_5B, 21)
    g = torch.Generator().manual_seed(4)
    z = bf(torch.randn(1, 16, 5, 6, 8, generator=g))
    want = R.decode(sd, R.CONFIG_5B, z, tiling=False)
    assert rel(m.decode(z.to(DEV)).sample, want) < 3e-2
    x = bf(torch.rand(1, 3, 1, 64, 96, generator=g) * 2 - 1)
    assert rel(m.encode(x.to(DEV)).latent_dist.parameters, R.encode_moments(sd, R.CONFIG_5B, x, tiling=False)) < 3e-2


def test_shipped_configuration_full_tiled_decode(hip):
    """the headline clip's latents [1, 16, 13, 60, 90] through the tiled decode the reference configures (nine tiles x six frame batches): 49 x 480 x 720 out, finite,
    and the three-stream fan-out of the tiles is bit-equal to issuing them on one stream"""
    m, _ = _model(R.CONFIG_5B, 22)
    m.enable_tiling()
    m.enable_slicing()
    z = bf(torch.randn(1, 16, 13, 60, 90, generator=torch.Generator().manual_seed(6))).to(DEV)
    m.tile_streams = 3
    a = m.decode(z).sample
    assert a.shape == (1, 3, 49, 480, 720) and torch.isfinite(a.float()).all() and a.float().abs().mean().item() > 1e-3
    m.tile_streams = 1
    assert torch.equal(m.decode(z).sample, a)
    img = bf(torch.rand(1, 3, 1, 480, 720, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(DEV)
    post = m.encode(img).latent_dist
    assert post.mean.shape == (1, 16, 1, 60, 90) and torch.isfinite(post.mean).all()
