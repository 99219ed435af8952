"""OpenCLIP text tower (DynamiCrafter's prompt encoder): the oracle restatement against the REAL transformers.CLIPTextModel outputs stored under
open_clip's parameter names (tests/golden/openclip_text.npz), and the product module's key layout.  No GPU compute."""
import os

import numpy as np
import pytest
import torch

from oracle import openclip_text_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "openclip_text.npz"))


def golden_sd():
    return {k[3:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith("sd.")}


def test_oracle_equals_transformers_clip_text_model():
    sd = golden_sd()
    tokens = torch.from_numpy(G["tokens"])
    np.testing.assert_allclose(R.encode_with_transformer(sd, tokens, heads=2, layer_idx=0).numpy(), G["last"], atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(R.encode_with_transformer(sd, tokens, heads=2, layer_idx=1).numpy(), G["penultimate"], atol=3e-5, rtol=1e-5)
    assert float(R.causal_mask(4)[0, 1]) == float("-inf") and float(R.causal_mask(4)[1, 0]) == 0.0


def test_product_key_layout_and_guards():
    from motionrag_amd import openclip_text as T, ops
    d, heads, layers, vocab = (int(v) for v in G["cfg"])
    m = T.OpenCLIPTextModel(vocab_size=vocab, width=d, heads=heads, layers=layers, embed_dim=64)
    missing, unexpected = m.load_state_dict(golden_sd(), strict=False)
    assert not unexpected and set(missing) == {"text_projection", "logit_scale"}          # the tower itself is complete
    emb = T.FrozenOpenCLIPEmbedder(m, layer="penultimate")
    assert emb.layer_idx == 1 and T.FrozenOpenCLIPEmbedder(m, layer="last").layer_idx == 0
    with pytest.raises(ValueError):
        emb(["a prompt"])                                                                  # no tokenizer supplied
    with pytest.raises(ops.HipOnly):
        emb.encode_with_transformer(torch.zeros(1, 77, dtype=torch.long))
    assert m.causal_mask(5, "cpu").tolist()[0] == [False, True, True, True, True]
    with pytest.raises(NotImplementedError):
        T.OpenCLIPTextModel(width=1280, heads=16)                                          # head_dim 80 (the image tower's): no kernel
