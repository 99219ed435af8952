"""TEST INFRASTRUCTURE (oracle) -- fp32 restatement of the T5 v1.1 encoder that encodes CogVideoX's prompt.

Where the reference uses it: src/projects/cogvideox/module.py:86-90 (`T5EncoderModel.from_pretrained(..., subfolder="text_encoder")`), consumed by diffusers'
`CogVideoXImageToVideoPipeline._get_t5_prompt_embeds` (226 tokens, padding="max_length", NO attention mask: `text_encoder(text_input_ids)[0]`).
THIRD-PARTY `transformers` (pinned 4.44.2, requirements.txt:9): `T5EncoderModel` / `T5Stack` / `T5Block` / `T5Attention` / `T5LayerNorm` /
`T5DenseGatedActDense`, restated from the published architecture.  Pinned by tests/golden/t5.npz (outputs of the REAL class in this image, random init, reduced
config, oracle/gen_golden_t5.py) and, on the GPU image, by a live comparison at the full T5-XXL width."""
import math

import torch
import torch.nn.functional as F


def rms_norm(x, w, eps):
    """T5LayerNorm: no mean subtraction, no bias"""
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """T5Attention._relative_position_bucket, bidirectional=True (encoder)"""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rp = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, large)


def position_bias(table, S, num_buckets=32, max_distance=128):
    """T5Attention.compute_bias: [H, S, S] from the [num_buckets, H] embedding of block 0 (shared by every layer)"""
    ctx = torch.arange(S)[:, None]
    mem = torch.arange(S)[None, :]
    return table[relative_position_bucket(mem - ctx, num_buckets, max_distance)].permute(2, 0, 1)


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_encoder(sd, cfg, input_ids, attention_mask=None):
    """T5EncoderModel(input_ids[, attention_mask]).last_hidden_state; cfg: d_model, num_heads, d_kv, num_layers, eps, num_buckets, max_distance"""
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg["eps"]
    x = sd["shared.weight"].float()[input_ids]
    B, S, _ = x.shape
    bias = position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].float(), S, cfg.get("num_buckets", 32),
                         cfg.get("max_distance", 128))[None]
    if attention_mask is not None:
        bias = bias + (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer."
        h = rms_norm(x, sd[p + "0.layer_norm.weight"].float(), eps)
        sp = lambda t: t.view(B, S, H, dk).transpose(1, 2)
        q, k, v = (sp(F.linear(h, sd[p + f"0.SelfAttention.{n}.weight"].float())) for n in "qkv")
        a = torch.softmax(q @ k.transpose(-1, -2) + bias, dim=-1) @ v                     # NO 1/sqrt(d) scaling in T5
        x = x + F.linear(a.transpose(1, 2).reshape(B, S, H * dk), sd[p + "0.SelfAttention.o.weight"].float())
        h = rms_norm(x, sd[p + "1.layer_norm.weight"].float(), eps)
        g = gelu_new(F.linear(h, sd[p + "1.DenseReluDense.wi_0.weight"].float())) * F.linear(h, sd[p + "1.DenseReluDense.wi_1.weight"].float())
        x = x + F.linear(g, sd[p + "1.DenseReluDense.wo.weight"].float())
    return rms_norm(x, sd["encoder.final_layer_norm.weight"].float(), eps)
