"""T5 v1.1 prompt encoder: the oracle restatement against outputs of the REAL transformers.T5EncoderModel (tests/golden/t5.npz), and the
product module's state-dict layout.  No GPU compute."""
import os

import numpy as np
import pytest
import torch

from oracle import t5_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "t5.npz"))


def golden_sd():
    return {k[3:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith("sd.")}


def cfg():
    d, h, dk, dff, layers, vocab = (int(v) for v in G["cfg"])
    return dict(d_model=d, num_heads=h, d_kv=dk, d_ff=dff, num_layers=layers, vocab_size=vocab, eps=1e-6)


def test_oracle_t5_equals_transformers():
    sd, c = golden_sd(), cfg()
    ids = torch.from_numpy(G["ids"])
    np.testing.assert_allclose(R.t5_encoder(sd, c, ids).numpy(), G["y"], atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(R.t5_encoder(sd, c, ids, torch.from_numpy(G["mask"])).numpy(), G["y_masked"], atol=3e-5, rtol=1e-5)
    # bucket function spot values (bidirectional, 32 buckets, max distance 128): 0 -> 0, +1 -> 17, -1 -> 1, +-8 and beyond are logarithmic, far = 15 / 31
    rp = torch.tensor([0, 1, -1, 7, -7, 8, -8, 127, -127, 500, -500])
    assert R.relative_position_bucket(rp).tolist() == [0, 17, 1, 23, 7, 24, 8, 31, 15, 31, 15]


def test_product_t5_key_layout_and_guards():
    from motionrag_amd import ops, t5
    c = cfg()
    m = t5.T5EncoderModel(vocab_size=c["vocab_size"], d_model=c["d_model"], d_kv=64, d_ff=c["d_ff"], num_layers=c["num_layers"], num_heads=c["num_heads"])
    sd = golden_sd()
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    assert torch.equal(t5.relative_position_bucket(torch.arange(-300, 300)), R.relative_position_bucket(torch.arange(-300, 300)))
    with pytest.raises(ops.HipOnly):
        m(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        t5.T5EncoderModel(d_kv=32, num_layers=1)
    # the shipped size: T5-v1.1-XXL encoder, 4.76 B parameters
    assert abs(sum(p.numel() for p in t5.T5EncoderModel(num_layers=1).parameters()) / 1e6 - (32128 * 4096 + 4 * 4096 * 4096 + 3 * 4096 * 10240 + 64 * 32 + 3 * 4096) / 1e6) < 1e-3
