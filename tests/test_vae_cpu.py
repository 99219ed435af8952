"""KL-VAE decode (DynamiCrafter): the oracle restatement against outputs of the reference's OWN AutoencoderKL class
(tests/golden/dc_vae.npz, oracle/gen_golden_vae.py), and the state-dict layout of the product module.  No GPU compute."""
import os

import numpy as np
import torch

from oracle import dynamicrafter_vae_ref as R
from oracle.seeded import seeded_sd

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dc_vae.npz"))


def golden_sd():
    shapes = [tuple(int(v) for v in s.split(",")) if s else () for s in G["shapes"].tolist()]
    return seeded_sd(G["keys"].tolist(), shapes, int(G["weight_seed"]), float(G["weight_std"]))


def test_oracle_decode_equals_reference_autoencoder():
    sd = golden_sd()
    y = R.autoencoder_decode(torch.from_numpy(G["z"]), sd, num_resolutions=2, num_res_blocks=1)
    np.testing.assert_allclose(y.numpy(), G["y"], rtol=2e-4, atol=2e-5)
    ya = R.attn_block(torch.from_numpy(G["x_attn"]), sd, "decoder.mid.attn_1")
    np.testing.assert_allclose(ya.numpy(), G["y_attn"], rtol=2e-4, atol=2e-5)
    y5 = R.decode_core(torch.from_numpy(G["z5"]), sd, 2, 1, scale_factor=0.18215)
    np.testing.assert_allclose(y5.numpy(), G["y5"], rtol=2e-4, atol=2e-5)


def test_oracle_encode_equals_reference_autoencoder():
    sd = golden_sd()
    mom = R.autoencoder_encode_moments(torch.from_numpy(G["x_img"]), sd, 2, 1)
    np.testing.assert_allclose(mom.numpy(), G["moments"], rtol=2e-4, atol=2e-5)
    z = R.first_stage_encoding(mom, torch.from_numpy(G["noise"]))
    np.testing.assert_allclose(z.numpy(), G["z_enc"], rtol=2e-4, atol=2e-5)


def test_product_module_has_the_reference_key_layout():
    from motionrag_amd import dynamicrafter_vae as V
    m = V.AutoencoderKL(dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0),
                        embed_dim=4, lossconfig={"target": "torch.nn.Identity"})
    sd = golden_sd()
    assert set(m.state_dict().keys()) == set(G["keys"].tolist())      # encoder.*, decoder.*, quant_conv.*, post_quant_conv.* exactly as the reference names them
    m.load_state_dict(sd, strict=True)
    # the shipped configuration (configs/dynamicrafter/MotionRAG_open.yml:245-259): 83.65 M parameters (SD's KL-f8 autoencoder)
    full = V.AutoencoderKL(dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[],
                                dropout=0.0), embed_dim=4)
    assert abs(sum(p.numel() for p in full.parameters()) / 1e6 - 83.65) < 0.01
    import pytest
    from motionrag_amd import ops
    with pytest.raises(ops.HipOnly):
        full.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(ops.HipOnly):
        full.encode(torch.zeros(1, 3, 8, 8))
