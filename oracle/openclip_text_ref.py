"""TEST INFRASTRUCTURE (oracle) -- fp32 restatement of the OpenCLIP text tower as DynamiCrafter's prompt encoder drives it.

Restates the in-tree wrapper  src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/encoders/condition.py:177-240 (`FrozenOpenCLIPEmbedder`:
`encode_with_transformer` = token_embedding + positional_embedding -> the first `len(resblocks) - layer_idx` residual attention blocks under the causal `attn_mask`
-> `ln_final`; layer "last" -> layer_idx 0, "penultimate" -> 1) over THIRD-PARTY `open_clip` (not installed): `ResidualAttentionBlock` = x + attn(ln_1(x)), x + mlp(ln_2(x))
with nn.MultiheadAttention (in_proj q | k | v) and `mlp = c_fc -> GELU -> c_proj`.
Pinned by tests/golden/openclip_text.npz: the REAL `transformers.CLIPTextModel` (the same architecture under other parameter names; hidden_act "gelu" as the laion ViT-H
text tower) with its weights renamed to open_clip's keys by oracle/gen_golden_openclip_text.py -- `hidden_states[-1 - layer_idx]` through `final_layer_norm`."""
import torch
import torch.nn.functional as F


def causal_mask(n: int) -> torch.Tensor:
    """open_clip `build_attention_mask`: additive, -inf above the diagonal"""
    return torch.full((n, n), float("-inf")).triu_(1)


def encode_with_transformer(sd: dict, tokens: torch.Tensor, heads: int, layer_idx: int = 1, eps: float = 1e-5) -> torch.Tensor:
    x = sd["token_embedding.weight"].float()[tokens] + sd["positional_embedding"].float()
    B, S, D = x.shape
    mask = causal_mask(S)
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.resblocks."))
    for i in range(n_layers - layer_idx):
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (D,), sd[p + "ln_1.weight"].float(), sd[p + "ln_1.bias"].float(), eps)
        q, k, v = F.linear(h, sd[p + "attn.in_proj_weight"].float(), sd[p + "attn.in_proj_bias"].float()).chunk(3, dim=-1)
        sp = lambda t: t.view(B, S, heads, D // heads).transpose(1, 2)
        a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * (D // heads) ** -0.5 + mask, dim=-1) @ sp(v)
        x = x + F.linear(a.transpose(1, 2).reshape(B, S, D), sd[p + "attn.out_proj.weight"].float(), sd[p + "attn.out_proj.bias"].float())
        h = F.layer_norm(x, (D,), sd[p + "ln_2.weight"].float(), sd[p + "ln_2.bias"].float(), eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.c_fc.weight"].float(), sd[p + "mlp.c_fc.bias"].float()))
        x = x + F.linear(h, sd[p + "mlp.c_proj.weight"].float(), sd[p + "mlp.c_proj.bias"].float())
    return F.layer_norm(x, (D,), sd["ln_final.weight"].float(), sd["ln_final.bias"].float(), eps)
