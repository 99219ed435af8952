"""The short-sequence attention kernel for head dims other than 64 (mrag_attn_small_bf16) and the CLIP vision tower on it, through the C ABI: against fp32 attention, the
REAL transformers.CLIPVisionModelWithProjection outputs (tests/golden/clip_vision.npz), and the CLIP-ViT-H width against the fp32 oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_vision_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


@pytest.mark.parametrize("B,S,H,D", [(2, 257, 16, 80), (3, 17, 2, 80), (1, 250, 3, 128), (2, 50, 4, 32), (1, 257, 5, 96)])
def test_attention_small(hip, B, S, H, D):
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(S + D)
    qkv = (torch.randn(B, S, 3, H, D, generator=g) * 0.7).to(torch.bfloat16)
    q, k, v = (qkv[:, :, i].float() for i in range(3))
    want = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q, k) * D ** -0.5, dim=-1)
    want = torch.einsum("bhqk,bkhd->bqhd", want, v).reshape(B, S, H * D)
    dq = qkv.to(DEV)
    got = ops.attention_small(dq[:, :, 0], dq[:, :, 1], dq[:, :, 2])               # strided views of the fused buffer
    assert rel(got, want) <= 6e-3
    from motionrag_amd._lib import HipError
    if D == 80:
        with pytest.raises(HipError):                                              # K and V of one head must fit the LDS image
            big = torch.zeros(1, 900, 3, 1, 80, dtype=torch.bfloat16, device=DEV)
            ops.attention_small(big[:, :, 0], big[:, :, 1], big[:, :, 2])


def test_clip_vision_equals_transformers_golden(hip, golden_dir):
    from motionrag_amd import clip_vision as C
    G = np.load(os.path.join(golden_dir, "clip_vision.npz"))
    d, heads, layers, ff, img, patch, proj = (int(v) for v in G["cfg"])
    m = C.CLIPVisionModelWithProjection(hidden_size=d, intermediate_size=ff, num_hidden_layers=layers, num_attention_heads=heads, image_size=img, patch_size=patch, projection_dim=proj)
    m.load_state_dict({k[3:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith("sd.")}, strict=True)
    out = m.to(DEV, torch.bfloat16)(torch.from_numpy(G["pixel_values"]).to(DEV))
    assert out.image_embeds.shape == (3, proj)
    assert rel(out.last_hidden_state, torch.from_numpy(G["last_hidden_state"])) <= 2e-2      # vs the REAL transformers class
    assert rel(out.image_embeds, torch.from_numpy(G["image_embeds"])) <= 2e-2


def test_clip_vit_h_width_vs_oracle(hip):
    """1280 wide, 16 heads of 80, 257 tokens at 224 x 224 (CLIP-ViT-H/14), 3 of its 32 layers"""
    from motionrag_amd import clip_vision as C
    torch.manual_seed(17)
    m = C.CLIPVisionModelWithProjection(num_hidden_layers=3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p)) if "class_embedding" not in n else p.normal_(0.0, 0.02)
            elif "position_embedding" in n:
                p.normal_(0.0, 0.02)
            p.copy_(p.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    pix = torch.randn(2, 3, 224, 224)
    want_last, want_emb = R.clip_vision(sd, 16, pix.to(torch.bfloat16).float())
    out = m.to(DEV, torch.bfloat16)(pix.to(DEV, torch.bfloat16))
    assert out.last_hidden_state.shape == (2, 257, 1280) and out.image_embeds.shape == (2, 1024)
    assert rel(out.last_hidden_state, want_last) <= 2e-2 and rel(out.image_embeds, want_emb) <= 2e-2


def test_clip_encode_image_is_resize_normalise_then_forward(hip):
    """`encode_image` = diffusers' SVD `_encode_image` front half -- antialiased 224 x 224 resize of the [-1, 1] image (the kornia recipe the package copied),
    `(x + 1) / 2`, CLIP mean / std -- fused into the tower's pixel kernel: equal to the tower's `forward` on the oracle's preprocessed pixels, at the SVD
    resolution; and the SVD pipeline takes this path when it is given no feature extractor"""
    from motionrag_amd import clip_vision as C
    from motionrag_amd import svd
    from oracle import kornia_resize_ref as K
    torch.manual_seed(19)
    m = C.CLIPVisionModelWithProjection(num_hidden_layers=2)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "position_embedding" in n or "class_embedding" in n:
                p.normal_(0.0, 0.02)
    m = m.to(DEV, torch.bfloat16)
    img = (torch.rand(2, 3, 576, 1024) * 2 - 1).to(torch.bfloat16)
    got = m.encode_image(img.to(DEV))
    want = m(K.preprocess(img.float(), True).to(DEV, torch.bfloat16))                  # oracle pixels -> the same tower
    assert got.image_embeds.shape == (2, 1024) and rel(got.image_embeds, want.image_embeds.float().cpu()) <= 2e-2
    assert rel(got.last_hidden_state, want.last_hidden_state.float().cpu()) <= 2e-2
    pipe = svd.StableVideoDiffusionPipeline(vae=None, image_encoder=m, unet=torch.nn.Linear(1, 1).to(DEV), scheduler=None, feature_extractor=None)
    emb = pipe._encode_image_base(img.float(), DEV, True)
    assert emb.shape == (4, 1, 1024) and torch.all(emb[:2] == 0) and torch.equal(emb[2:, 0], got.image_embeds)


def _hf_to_openclip_visual(sd, layers):
    out = {"conv1.weight": sd["vision_model.embeddings.patch_embedding.weight"], "class_embedding": sd["vision_model.embeddings.class_embedding"],
           "positional_embedding": sd["vision_model.embeddings.position_embedding.weight"], "ln_pre.weight": sd["vision_model.pre_layrnorm.weight"],
           "ln_pre.bias": sd["vision_model.pre_layrnorm.bias"], "ln_post.weight": sd["vision_model.post_layernorm.weight"], "ln_post.bias": sd["vision_model.post_layernorm.bias"],
           "proj": sd["visual_projection.weight"].t().contiguous()}
    for i in range(layers):
        h, o = f"vision_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[o + "attn.in_proj_weight"] = torch.cat([sd[h + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        out[o + "attn.in_proj_bias"] = torch.cat([sd[h + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        for a, b in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            out[o + a + ".weight"], out[o + a + ".bias"] = sd[h + b + ".weight"], sd[h + b + ".bias"]
    return out


def test_openclip_visual_tower_under_open_clip_names(hip, golden_dir):
    """DynamiCrafter's FrozenOpenCLIPImageEmbedderV2 (lvdm/modules/encoders/condition.py:302-380) returns every token after the last block: with the REAL
    transformers weights renamed to open_clip's keys, that is the golden's `last_hidden_state`"""
    from motionrag_amd import clip_vision as C
    G = np.load(os.path.join(golden_dir, "clip_vision.npz"))
    d, heads, layers, ff, img, patch, proj = (int(v) for v in G["cfg"])
    sd = {k[3:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith("sd.")}
    v = C.OpenCLIPVisual(width=d, layers=layers, heads=heads, mlp_ratio=ff / d, image_size=img, patch_size=patch, output_dim=proj)
    v.load_state_dict(_hf_to_openclip_visual(sd, layers), strict=True)
    emb = C.FrozenOpenCLIPImageEmbedderV2(v.to(DEV, torch.bfloat16), preprocess=lambda x: x)          # the fixture's pixel_values are already normalised
    z = emb(torch.from_numpy(G["pixel_values"]).to(DEV))
    assert z.shape == (3, (img // patch) ** 2 + 1, d) and rel(z, torch.from_numpy(G["last_hidden_state"])) <= 2e-2
    # built-in preprocess (kornia's antialiased bicubic squash to 224 x 224 + CLIP normalisation, fused into the pixel kernel) at the shipped geometry:
    # equal to the tower run on the oracle's preprocessed pixels
    from oracle import kornia_resize_ref as K
    big = C.FrozenOpenCLIPImageEmbedderV2(width=160, layers=1, heads=2, mlp_ratio=2.0).to(DEV, torch.bfloat16)
    x = (torch.rand(2, 3, 320, 512) * 2 - 1).to(torch.bfloat16)
    t = big(x.to(DEV))
    by_hand = C.FrozenOpenCLIPImageEmbedderV2(big.model.visual, preprocess=lambda im: K.preprocess(im.float().cpu()).to(DEV, torch.bfloat16))(x.to(DEV))
    assert t.shape == (2, 257, 160) and rel(t, by_hand) <= 1e-2


@pytest.mark.parametrize("H,W,antialias", [(576, 1024, True), (320, 512, True), (100, 300, True), (150, 180, True), (224, 224, True), (576, 1024, False)])
def test_kornia_preprocess_pixels_match_oracle(hip, H, W, antialias):
    """the fused pixel kernel with the folded blur x bicubic tap tables against the two-stage restatement of kornia.geometry.resize + normalize (parity unpinned)"""
    from motionrag_amd.encoders import pixels_to_patch_rows
    from oracle import kornia_resize_ref as K
    g = torch.Generator().manual_seed(H + W)
    x = (torch.rand(2, 3, H, W, generator=g) * 2 - 1).to(torch.bfloat16)
    want = K.preprocess(x.float(), antialias)                                                  # [2, 3, 224, 224]
    rows = pixels_to_patch_rows(x.to(DEV)[:, None], resize=224, crop=224, mode="kornia-bicubic" if antialias else "kornia-bicubic-noaa", patch=(1, 14, 14),
                                mean=K.CLIP_MEAN, std=K.CLIP_STD)
    got = rows[:, :3 * 196].float().cpu().view(2, 16, 16, 3, 14, 14).permute(0, 3, 1, 4, 2, 5).reshape(2, 3, 224, 224)
    assert (got - want).abs().max().item() <= 2.5e-2 and rel(got, want) <= 4e-3             # bf16 output rounding of values up to ~2.7
