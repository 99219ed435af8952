"""T5 v1.1 encoder on the GPU through the C ABI: the new kernel modes on their own (RMSNorm, attention with an additive score bias, gated-GELU-tanh epilogue),
the reduced model against the REAL transformers.T5EncoderModel outputs (tests/golden/t5.npz), and two layers at the full T5-XXL width against the fp32 oracle
and the live transformers class."""
import os

import numpy as np
import pytest
import torch

from oracle import t5_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


def test_rmsnorm_bias_attention_and_gated_gelu_tanh(hip):
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(37, 4096, generator=g) * 3 + 0.5).to(torch.bfloat16)
    w = (1 + 0.2 * torch.randn(4096, generator=g)).to(torch.bfloat16)
    got = ops.layernorm(x.to(DEV), w.to(DEV), None, 1e-6, rms=True)
    assert rel(got, R.rms_norm(x.float(), w.float(), 1e-6)) <= 4e-3
    # attention: scale 1 (T5), additive bias per head; ragged S (226) and more than one 64-key tile
    B, S, H = 2, 226, 3
    q, k, v = ((torch.randn(B, S, H, 64, generator=g) * 0.4).to(torch.bfloat16) for _ in range(3))
    bias = torch.randn(H, S, S, generator=g) * 2
    want = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) + bias[None], dim=-1)
    want = torch.einsum("bhqk,bkhd->bqhd", want, v.float()).reshape(B, S, H * 64)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), scale=1.0, bias=bias.to(DEV))
    assert rel(got, want) <= 1e-2
    got_m = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), scale=1.0, bias=bias.to(DEV), mask=(torch.rand(S, S, generator=g) < 0.3).to(DEV))   # bias AND mask
    assert torch.isfinite(got_m.float()).all()
    # gated GELU (tanh): [value | gate] = [wi_1 | wi_0]
    M, K, F_ = 300, 128, 256
    h = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w0, w1 = ((torch.randn(F_, K, generator=g) * 0.1).to(torch.bfloat16) for _ in range(2))
    wg = ops.geglu_interleave(torch.cat([w1, w0], 0).to(DEV), None)[0]
    got = ops.linear(h.to(DEV), wg, epilogue=ops.EPI_GEGLU, geglu_tanh=True)
    want = R.gelu_new(h.float() @ w0.float().T) * (h.float() @ w1.float().T)
    assert rel(got, want) <= 1e-2
    got_erf = ops.linear(h.to(DEV), wg, epilogue=ops.EPI_GEGLU)                  # the default (erf) is a different function
    assert rel(got_erf, torch.nn.functional.gelu(h.float() @ w0.float().T) * (h.float() @ w1.float().T)) <= 1e-2


def test_t5_reduced_equals_transformers_golden(hip, golden_dir):
    from motionrag_amd import t5
    G = np.load(os.path.join(golden_dir, "t5.npz"))
    d, h, dk, dff, layers, vocab = (int(v) for v in G["cfg"])
    m = t5.T5EncoderModel(vocab_size=vocab, d_model=d, d_kv=dk, d_ff=dff, num_layers=layers, num_heads=h)
    m.load_state_dict({k[3:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith("sd.")}, strict=True)
    m = m.to(DEV, torch.bfloat16)
    ids = torch.from_numpy(G["ids"]).to(DEV)
    out = m(ids)
    assert out[0] is out.last_hidden_state and out[0].shape == (2, 226, d)
    assert rel(out[0], torch.from_numpy(G["y"])) <= 2e-2                         # vs the REAL transformers.T5EncoderModel, as the pipeline calls it (no mask)
    assert rel(m(ids, attention_mask=torch.from_numpy(G["mask"]).to(DEV))[0], torch.from_numpy(G["y_masked"])) <= 2e-2


def test_t5_xxl_width_two_layers_vs_oracle_and_live_transformers(hip):
    """d_model 4096, 64 heads x 64, d_ff 10240 (the T5-v1.1-XXL encoder CogVideoX ships) at 2 of its 24 layers, 226 tokens, batch 2 (prompt + negative prompt)"""
    from motionrag_amd import t5
    torch.manual_seed(8)
    m = t5.T5EncoderModel(vocab_size=512, num_layers=2)
    with torch.no_grad():                                                         # T5's own initialisation (T5PreTrainedModel._init_weights, factor 1): scores of O(1) --
        for n, p in m.named_parameters():                                         # T5 does not scale q k^T, so N(0, 0.02) weights at d_model 4096 would give logits of +-13
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
            elif "relative_attention_bias" in n or n.startswith("shared"):
                p.normal_(0.0, 1.0)
            elif n.endswith("SelfAttention.q.weight"):
                p.normal_(0.0, (4096 * 64) ** -0.5)
            elif n.endswith(("SelfAttention.k.weight", "SelfAttention.v.weight", "wi_0.weight", "wi_1.weight")):
                p.normal_(0.0, 4096 ** -0.5)
            elif n.endswith("SelfAttention.o.weight"):
                p.normal_(0.0, (64 * 64) ** -0.5)
            elif n.endswith("wo.weight"):
                p.normal_(0.0, 10240 ** -0.5)
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ids = torch.randint(0, 512, (2, 226))
    cfg = dict(d_model=4096, num_heads=64, d_kv=64, num_layers=2, eps=1e-6)
    want = R.t5_encoder(sd, cfg, ids)
    got = m.to(DEV, torch.bfloat16)(ids.to(DEV))[0]
    assert got.shape == (2, 226, 4096) and rel(got, want) <= 2e-2
    try:
        from transformers import T5Config, T5EncoderModel
    except Exception:
        return
    hf = T5EncoderModel(T5Config(vocab_size=512, d_model=4096, d_kv=64, d_ff=10240, num_layers=2, num_heads=64, feed_forward_proj="gated-gelu", dropout_rate=0.0,
                                 tie_word_embeddings=False)).eval()
    hf.load_state_dict(sd, strict=True)
    with torch.no_grad():
        live = hf(input_ids=ids).last_hidden_state
    assert rel(want, live) <= 1e-4 and rel(got, live) <= 2e-2
