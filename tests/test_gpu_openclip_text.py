"""OpenCLIP text tower on the GPU through the C ABI against the REAL transformers.CLIPTextModel outputs (tests/golden/openclip_text.npz), and the ViT-H-14 text width
(1024, 16 heads) against the fp32 oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import openclip_text_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


def test_openclip_text_equals_transformers_golden(hip, golden_dir):
    from motionrag_amd import openclip_text as T
    G = np.load(os.path.join(golden_dir, "openclip_text.npz"))
    d, heads, layers, vocab = (int(v) for v in G["cfg"])
    m = T.OpenCLIPTextModel(vocab_size=vocab, width=d, heads=heads, layers=layers, embed_dim=64)
    m.load_state_dict({k[3:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith("sd.")}, strict=False)
    m = m.to(DEV, torch.bfloat16)
    tokens = torch.from_numpy(G["tokens"])
    for layer in ("last", "penultimate"):
        emb = T.FrozenOpenCLIPEmbedder(m, tokenizer=lambda text: tokens, layer=layer)
        z = emb(["a", "b", "c"])
        assert z.shape == (3, 77, d) and rel(z, torch.from_numpy(G[layer])) <= 2e-2      # vs the REAL CLIPTextModel (hidden_states[-1 - layer_idx] -> final LN)
        cls, z2 = emb(tokens.to(DEV), return_cls_tokens=True)
        assert torch.equal(z2, z) and torch.equal(cls, z[torch.arange(3), tokens.argmax(-1)])


def test_openclip_text_vit_h_width_vs_oracle(hip):
    from motionrag_amd import openclip_text as T
    torch.manual_seed(13)
    m = T.OpenCLIPTextModel(vocab_size=1000, width=1024, heads=16, layers=3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    tokens = torch.randint(1, 999, (2, 77))
    want = R.encode_with_transformer(sd, tokens, heads=16, layer_idx=1)
    got = T.FrozenOpenCLIPEmbedder(m.to(DEV, torch.bfloat16), layer="penultimate").encode_with_transformer(tokens.to(DEV))
    assert rel(got, want) <= 2e-2
