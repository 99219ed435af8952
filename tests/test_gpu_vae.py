"""KL-VAE decode on the GPU through the C ABI against the reference's own AutoencoderKL outputs (tests/golden/dc_vae.npz) and, at the shipped decoder
configuration, against the fp32 oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import dynamicrafter_vae_ref as R
from oracle.seeded import seeded_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


def test_softmax_rows(hip):
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(5)
    for rows, cols, ld in ((7, 96, 96), (33, 9216, 9216), (5, 100, 104), (4, 520, 520)):
        buf = (torch.randn(rows, ld, generator=g) * 6).to(torch.bfloat16)
        x = buf[:, :cols]
        want = torch.softmax(x.float() * 0.37, dim=1)
        got = ops.softmax_rows(x.to(DEV) if ld == cols else buf.to(DEV)[:, :cols], scale=0.37)
        assert got.shape == (rows, cols)
        err = (got.float().cpu() - want).abs()
        assert bool((err <= want * 2.0 ** -7 + 1e-6).all()), float(err.max())        # one bf16 rounding of a probability
        assert float((got.float().sum(1).cpu() - 1).abs().max()) < 2e-2


def test_vae_decode_equals_reference_autoencoder(hip, golden_dir):
    from motionrag_amd import dynamicrafter_vae as V
    G = np.load(os.path.join(golden_dir, "dc_vae.npz"))
    shapes = [tuple(int(v) for v in s.split(",")) if s else () for s in G["shapes"].tolist()]
    sd = seeded_sd(G["keys"].tolist(), shapes, int(G["weight_seed"]), float(G["weight_std"]))
    m = V.AutoencoderKL(dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0),
                        embed_dim=4)
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV, torch.bfloat16)
    y = m.decode(torch.from_numpy(G["z"]).to(DEV))
    assert y.shape == G["y"].shape
    assert rel(y, torch.from_numpy(G["y"])) <= 2e-2                              # vs the REFERENCE's AutoencoderKL.decode
    x = torch.from_numpy(G["x_attn"]).to(DEV, torch.bfloat16).permute(0, 2, 3, 1).contiguous()
    ya = m.decoder.mid.attn_1(x).permute(0, 3, 1, 2)
    assert rel(ya, torch.from_numpy(G["y_attn"])) <= 1e-2                        # vs the REFERENCE's AttnBlock.forward
    y5 = V.decode_first_stage(m, torch.from_numpy(G["z5"]).to(DEV), scale_factor=0.18215)
    assert y5.shape == G["y5"].shape and rel(y5, torch.from_numpy(G["y5"])) <= 2e-2
    # encode: posterior parameters and a sample with the fixture's noise, vs the REFERENCE's AutoencoderKL.encode / DiagonalGaussianDistribution.sample
    post = m.encode(torch.from_numpy(G["x_img"]).to(DEV))
    assert post.parameters.shape == G["moments"].shape
    assert rel(post.parameters, torch.from_numpy(G["moments"])) <= 2e-2
    z = V.encode_first_stage(m, torch.from_numpy(G["x_img"]).to(DEV), noise=torch.from_numpy(G["noise"]))
    assert rel(z, torch.from_numpy(G["z_enc"])) <= 2e-2
    assert torch.equal(V.encode_first_stage(m, torch.from_numpy(G["x_img"]).to(DEV), noise=torch.zeros_like(post.mean)), 0.18215 * post.mode())


@pytest.mark.parametrize("H,W,C,Cout", [(10, 14, 64, 64), (9, 7, 128, 192), (32, 48, 64, 64)])
def test_asymmetric_pad_stride2_conv(hip, H, W, C, Cout):
    """Downsample (ae_modules.py:106-110): F.pad(x, (0, 1, 0, 1)) + Conv2d(3, stride 2, padding 0) as one implicit-GEMM launch; odd sizes included"""
    import torch.nn.functional as F
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(2, C, H, W, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.05).to(torch.bfloat16)
    b = (torch.randn(Cout, generator=g) * 0.1).to(torch.bfloat16)
    want = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C).contiguous()
    got = ops.conv_implicit(x.permute(0, 2, 3, 1).contiguous().to(DEV), wk.to(DEV), b.to(DEV), ops.CONV_3X3, stride=2, asym_pad=True)
    assert got.shape == (2, want.shape[2], want.shape[3], Cout)
    assert rel(got.permute(0, 3, 1, 2), want) <= 1e-2


def test_vae_decode_shipped_config_vs_oracle(hip):
    """the shipped decoder (ch 128, mult (1, 2, 4, 4), 2 + 1 blocks per level, 512-channel single-head mid attention) on 2 frames of a 24 x 40 latent -> 192 x 320"""
    from motionrag_amd import dynamicrafter_vae as V
    torch.manual_seed(9)
    m = V.AutoencoderKL(dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[],
                             dropout=0.0), embed_dim=4)
    g = torch.Generator().manual_seed(10)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    z = torch.randn(1, 4, 2, 24, 40, generator=g) * 0.18215
    want = R.decode_core(z.to(torch.bfloat16).float(), sd, 4, 2)
    got = V.decode_first_stage(m.to(DEV, torch.bfloat16), z.to(DEV, torch.bfloat16))
    assert got.shape == want.shape == (1, 3, 2, 192, 320)
    assert rel(got, want) <= 3e-2
