"""CLIP vision tower (SVD's image_encoder): oracle restatement vs the REAL transformers.CLIPVisionModelWithProjection (tests/golden/clip_vision.npz); product key layout."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_vision_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_vision.npz"))


def golden_sd():
    return {k[3:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith("sd.")}


def test_oracle_equals_transformers():
    last, emb = R.clip_vision(golden_sd(), 2, torch.from_numpy(G["pixel_values"]))
    np.testing.assert_allclose(last.numpy(), G["last_hidden_state"], atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(emb.numpy(), G["image_embeds"], atol=3e-5, rtol=1e-5)


def test_product_key_layout():
    from motionrag_amd import clip_vision as C, ops
    d, heads, layers, ff, img, patch, proj = (int(v) for v in G["cfg"])
    m = C.CLIPVisionModelWithProjection(hidden_size=d, intermediate_size=ff, num_hidden_layers=layers, num_attention_heads=heads, image_size=img, patch_size=patch, projection_dim=proj)
    sd = golden_sd()
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    with pytest.raises(ops.HipOnly):
        m(torch.zeros(1, 3, img, img))
    with pytest.raises(NotImplementedError):
        C.CLIPVisionModelWithProjection(hidden_act="quick_gelu", num_hidden_layers=1)
    full = C.CLIPVisionModelWithProjection(num_hidden_layers=1)
    assert full.head_dim == 80 and tuple(full.vision_model.embeddings.position_embedding.weight.shape) == (257, 1280)


def test_kornia_tap_tables_equal_two_stage_resize():
    """blur and align_corners bicubic folded into one tap table per axis (`encoders.kornia_resize_taps`) reproduce the two-stage restatement of
    kornia.geometry.resize (oracle/kornia_resize_ref.py); rows sum to 1, so the reference's later (x + 1) / 2 and normalisation commute with the resize"""
    import numpy as np
    from motionrag_amd import encoders as E
    from oracle import kornia_resize_ref as K
    for H, W in ((576, 1024), (320, 512), (100, 300), (150, 180), (480, 720)):
        x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(H)) * 2 - 1
        want = K.resize(x, (224, 224), True).numpy()
        (sy, ky), (sx, kx) = E.kornia_blur_geometry(H, W, 224, 224)
        assert K.blur_geometry((H, W), (224, 224))[2] == ((ky, kx) if max(H, W) > 224 else (1, 1))
        wy, y0, ny = E.kornia_resize_taps(H, 224, sy, ky)
        wx, x0, nx = E.kornia_resize_taps(W, 224, sx, kx)
        My, Mx = np.zeros((224, H)), np.zeros((224, W))
        for i in range(224):
            My[i, y0[i]:y0[i] + ny[i]] = wy[i, :ny[i]]
            Mx[i, x0[i]:x0[i] + nx[i]] = wx[i, :nx[i]]
        assert np.allclose(My.sum(1), 1.0, atol=1e-6) and np.allclose(Mx.sum(1), 1.0, atol=1e-6)
        got = My @ x.double().numpy() @ Mx.T                                  # [224, H] @ [1, 3, H, W] @ [W, 224]
        assert np.abs(got - want).max() < 2e-4
