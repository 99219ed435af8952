"""TEST INFRASTRUCTURE (oracle) -- fp32 restatement of the CLIP vision tower with projection: SVD's `image_encoder` (THIRD-PARTY `transformers`
`CLIPVisionModelWithProjection`, loaded by diffusers' StableVideoDiffusionPipeline; called at src/projects/svd/pipelines/pipeline.py:113-119 through `_encode_image`:
`self.image_encoder(image).image_embeds`).  Pinned by tests/golden/clip_vision.npz (the REAL class, random init, reduced config with head_dim 80 like ViT-H)."""
import torch
import torch.nn.functional as F


def clip_vision(sd, heads, pixel_values, eps=1e-5):
    """returns (last_hidden_state, image_embeds)"""
    p = "vision_model."
    w = sd[p + "embeddings.patch_embedding.weight"].float()
    x = F.conv2d(pixel_values.float(), w, None, stride=w.shape[2:]).flatten(2).transpose(1, 2)
    x = torch.cat([sd[p + "embeddings.class_embedding"].float().expand(x.shape[0], 1, -1), x], dim=1) + sd[p + "embeddings.position_embedding.weight"].float()
    B, S, D = x.shape
    x = F.layer_norm(x, (D,), sd[p + "pre_layrnorm.weight"].float(), sd[p + "pre_layrnorm.bias"].float(), eps)
    n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(p + "encoder.layers."))
    for i in range(n_layers):
        q = f"{p}encoder.layers.{i}."
        h = F.layer_norm(x, (D,), sd[q + "layer_norm1.weight"].float(), sd[q + "layer_norm1.bias"].float(), eps)
        sp = lambda t: t.view(B, S, heads, D // heads).transpose(1, 2)
        qq, kk, vv = (sp(F.linear(h, sd[q + f"self_attn.{n}_proj.weight"].float(), sd[q + f"self_attn.{n}_proj.bias"].float())) for n in "qkv")
        a = torch.softmax(qq @ kk.transpose(-1, -2) * (D // heads) ** -0.5, dim=-1) @ vv
        x = x + F.linear(a.transpose(1, 2).reshape(B, S, D), sd[q + "self_attn.out_proj.weight"].float(), sd[q + "self_attn.out_proj.bias"].float())
        h = F.layer_norm(x, (D,), sd[q + "layer_norm2.weight"].float(), sd[q + "layer_norm2.bias"].float(), eps)
        h = F.gelu(F.linear(h, sd[q + "mlp.fc1.weight"].float(), sd[q + "mlp.fc1.bias"].float()))
        x = x + F.linear(h, sd[q + "mlp.fc2.weight"].float(), sd[q + "mlp.fc2.bias"].float())
    pooled = F.layer_norm(x[:, 0], (D,), sd[p + "post_layernorm.weight"].float(), sd[p + "post_layernorm.bias"].float(), eps)
    return x, F.linear(pooled, sd["visual_projection.weight"].float())
