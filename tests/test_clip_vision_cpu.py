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
