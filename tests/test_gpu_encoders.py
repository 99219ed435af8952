"""Frozen feature encoders on the GPU, through the C ABI: the fused pixel kernel against ATen's antialiased resize, the two ViT
bodies against outputs of the REAL `transformers` classes (tests/golden/encoders.npz) and -- at the full VideoMAE-B / DINOv2-L configurations -- against
the fp32 oracle (and the live `transformers` class when it is importable on the box)."""
import os

import numpy as np
import pytest
import torch

from oracle import encoders_ref as E

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    return ((g - w).norm() / w.norm()).item()


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "encoders.npz"))


def _sd(G, tag):
    return {k[len(tag) + 4:]: torch.from_numpy(G[k].view(np.int16).copy()).view(torch.bfloat16).float() for k in G.files if k.startswith(tag + ".sd.")}


def _rows_from_pixels(pix, patch):
    """oracle-side patch rows: pix [N, T, C, H, W] -> [N * T/pt * H/ph * W/pw, C * pt * ph * pw] in the (c, dt, dy, dx) column order"""
    N, T, C, H, W = pix.shape
    pt, ph, pw = patch
    x = pix.reshape(N, T // pt, pt, C, H // ph, ph, W // pw, pw).permute(0, 1, 4, 6, 3, 2, 5, 7)
    return x.reshape(N * (T // pt) * (H // ph) * (W // pw), C * pt * ph * pw)


@pytest.mark.parametrize("src_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("geom", [
    dict(N=2, T=7, H=40, W=52, resize=32, crop=32, mode="bilinear", patch=(2, 16, 16), frames=4),       # the golden fixture's geometry
    dict(N=2, T=20, H=240, W=426, resize=224, crop=224, mode="bilinear", patch=(2, 16, 16), frames=16),  # VideoMAE: 16 of 20 frames, 240p -> 224
    dict(N=3, T=1, H=480, W=720, resize=256, crop=224, mode="bicubic", patch=(1, 14, 14), frames=None),  # DINOv2: 480 x 720 -> 256 -> 224, K 588 -> 640
    dict(N=2, T=1, H=100, W=80, resize=256, crop=224, mode="bicubic", patch=(1, 14, 14), frames=None),   # up-scaling, portrait
])
def test_pixels_to_patch_rows_equals_aten_resize(hip, src_dtype, geom):
    from motionrag_amd import encoders as P
    g = torch.Generator().manual_seed(7)
    src = (torch.rand(geom["N"], geom["T"], 3, geom["H"], geom["W"], generator=g) * 2 - 1).to(src_dtype)
    idx = None if geom["frames"] is None else E.uniform_frame_indices(geom["T"], geom["frames"])
    ref_in = src.float() if idx is None else src.float()[:, idx]
    want = _rows_from_pixels(E.preprocess(ref_in, geom["resize"], geom["crop"], geom["mode"]), geom["patch"])
    got = P.pixels_to_patch_rows(src.to(DEV), resize=geom["resize"], crop=geom["crop"], mode=geom["mode"], patch=geom["patch"],
                                 frame_idx=None if idx is None else idx.to(DEV, torch.int32))
    K = want.shape[1]
    assert got.shape == (want.shape[0], (K + 63) // 64 * 64) and got.dtype == torch.bfloat16
    assert float(got[:, K:].float().abs().max()) == 0.0 if got.shape[1] > K else True          # GEMM K padding is zero
    err = (got[:, :K].float().cpu() - want).abs()
    # one bf16 rounding of values up to ~2.7: |err| <= 2^-8 |ref| + 1e-3 (fp32 tap arithmetic vs ATen's: ~1e-6)
    assert bool((err <= want.abs() * 2.0 ** -8 + 1e-3).all()), float(err.max())
    # the LDS-tiled kernel (default) and the per-pixel kernel sum in the same order: bit-identical rows
    got_px = P.pixels_to_patch_rows(src.to(DEV), resize=geom["resize"], crop=geom["crop"], mode=geom["mode"], patch=geom["patch"],
                                    frame_idx=None if idx is None else idx.to(DEV, torch.int32), tiled=False)
    assert torch.equal(got, got_px)


def test_videomae_reduced_equals_transformers_golden(hip, golden_dir):
    from motionrag_amd import encoders as P
    G = _golden(golden_dir)
    h, heads, layers, tub, patch, frames = (int(v) for v in G["vmae.cfg"])
    kw = dict(hidden_size=h, num_attention_heads=heads, num_hidden_layers=layers, intermediate_size=256, image_size=32, patch_size=patch, num_frames=frames,
              tubelet_size=tub, layer_norm_eps=float(G["vmae.eps"]))
    sd = _sd(G, "vmae")
    video = torch.from_numpy(G["vmae.video"]).to(DEV)
    for tag, mean_pool in (("vmae", True), ("vmae_ln", False)):
        m = P.VideoMAEModel(use_mean_pooling=mean_pool, **kw)
        missing, unexpected = m.load_state_dict({k: v for k, v in sd.items() if not (mean_pool and k.startswith("layernorm."))}, strict=False)
        assert not missing and not unexpected
        emb = P.VideoMAEEmbedder(m.to(DEV, torch.bfloat16), resize=32, crop=32)
        y = emb(video)                                                        # fp32 pixels in, as the dataset delivers them
        assert y.shape == (2, 8, 128) and emb.dim == 128
        assert rel(y, torch.from_numpy(G[f"{tag}.last_hidden_state"])) <= 2e-2          # vs the REAL transformers VideoMAEModel
        y2 = m(torch.from_numpy(G["vmae.pixel_values"]).to(DEV))              # the transformers entry: normalised pixel_values
        assert rel(y2, torch.from_numpy(G[f"{tag}.last_hidden_state"])) <= 2e-2


def test_dinov2_reduced_equals_transformers_golden(hip, golden_dir):
    from motionrag_amd import encoders as P
    G = _golden(golden_dir)
    h, heads, layers, patch = (int(v) for v in G["dino.cfg"])
    m = P.Dinov2Model(hidden_size=h, num_attention_heads=heads, num_hidden_layers=layers, mlp_ratio=2, image_size=70, patch_size=patch,
                      layer_norm_eps=float(G["dino.eps"]), pos_dialect="size")
    missing, unexpected = m.load_state_dict(_sd(G, "dino"), strict=False)
    assert not missing and not unexpected
    m = m.to(DEV, torch.bfloat16)
    images = torch.from_numpy(G["dino.images"]).to(DEV)
    for tag in ("dino", "dino_native"):                                       # resampled 5 x 5 -> 4 x 4 position table, and the native grid
        rs, cr = (int(v) for v in G[f"{tag}.resize_crop"])
        y = P.DINOImageEmbedder(m, resize=rs, crop=cr)(images)
        assert y.shape == (3, (cr // patch) ** 2 + 1, h)
        assert rel(y, torch.from_numpy(G[f"{tag}.last_hidden_state"])) <= 2e-2           # vs the REAL transformers Dinov2Model
        y2 = m(torch.from_numpy(G[f"{tag}.pixel_values"]).to(DEV))
        assert rel(y2, torch.from_numpy(G[f"{tag}.last_hidden_state"])) <= 2e-2


def _randomise_(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            p.copy_(p.to(torch.bfloat16).float())                             # bf16-exact, shared bit for bit with the fp32 side
    return m


def test_videomae_base_full_size_vs_oracle(hip):
    """VideoMAE-B exactly as condition.py:365 configures it (12 x 768, 1 568 tokens, no final LayerNorm), one clip of 24 frames at 240 x 320"""
    from motionrag_amd import encoders as P
    torch.manual_seed(11)
    m = _randomise_(P.VideoMAEModel(), 12)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    video = torch.rand(1, 24, 3, 240, 320) * 2 - 1
    cfg = dict(hidden_size=768, num_attention_heads=12, num_hidden_layers=12, layer_norm_eps=1e-12, num_frames=16)
    # the oracle speaks the transformers dialects: q_bias / k_bias / v_bias -> 5.x names
    osd = {k.replace("attention.attention.q_bias", "attention.attention.query.bias").replace("attention.attention.k_bias", "attention.attention.key.bias")
            .replace("attention.attention.v_bias", "attention.attention.value.bias"): v for k, v in sd.items()}
    want = E.videomae_embedder(osd, cfg, video)
    emb = P.VideoMAEEmbedder(m.to(DEV, torch.bfloat16))
    got = emb(video.to(DEV, torch.bfloat16))
    assert got.shape == (1, 1568, 768)
    # the bf16 input copy is part of the product path (`precision: bf16-true`); the oracle sees the fp32 pixels: stated bound 3 % over 12 layers
    assert rel(got, want) <= 3e-2


def test_dinov2_large_full_size_vs_oracle_and_live_transformers(hip):
    """DINOv2-L as condition.py:568 configures it (24 x 1024, 257 tokens at 224, 37 x 37 position table resampled to 16 x 16)"""
    from motionrag_amd import encoders as P
    torch.manual_seed(21)
    m = _randomise_(P.Dinov2Model(pos_dialect="size"), 22)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    images = torch.rand(2, 3, 480, 720) * 2 - 1
    cfg = dict(hidden_size=1024, num_attention_heads=16, num_hidden_layers=24, patch_size=14, layer_norm_eps=1e-6)
    osd = {k.replace("attention.attention.q_bias", "attention.attention.query.bias").replace("attention.attention.k_bias", "attention.attention.key.bias")
            .replace("attention.attention.v_bias", "attention.attention.value.bias"): v for k, v in sd.items()}
    want = E.dino_embedder(osd, cfg, images, pos_dialect="size")
    got = P.DINOImageEmbedder(m.to(DEV, torch.bfloat16))(images.to(DEV, torch.bfloat16))
    assert got.shape == (2, 257, 1024)
    assert rel(got, want) <= 3e-2
    try:
        from transformers import Dinov2Config, Dinov2Model
    except Exception:                                                         # the oracle comparison above stands on its own
        return
    hf = Dinov2Model(Dinov2Config(hidden_size=1024, num_attention_heads=16, num_hidden_layers=24, mlp_ratio=4, image_size=518, patch_size=14,
                                  attn_implementation="eager")).eval()
    missing, unexpected = hf.load_state_dict(osd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)              # the state-dict hand-over works in the other direction too
    with torch.no_grad():
        live = hf(E.preprocess(images, 256, 224, "bicubic")).last_hidden_state
    assert rel(want, live) <= 1e-4                                            # oracle == the real class at the real architecture
    assert rel(got, live) <= 3e-2


def test_cama_predict_from_raw_pixels(hip):
    """ActionTransformer.predict with BOTH frozen encoders on the HIP path (module.py:264-276, 311-331): raw videos in, motion tokens out, against the
    fp32 oracle fed with the oracle encoders' features"""
    from motionrag_amd import cama, encoders as P
    from oracle import cama_ref
    torch.manual_seed(3)
    vm = _randomise_(P.VideoMAEModel(hidden_size=128, num_attention_heads=2, num_hidden_layers=2, intermediate_size=256, image_size=32, num_frames=4), 4)
    dm = _randomise_(P.Dinov2Model(hidden_size=128, num_attention_heads=2, num_hidden_layers=2, mlp_ratio=2, image_size=28, pos_dialect="size"), 5)
    ren = lambda sd: {k.replace("attention.attention.q_bias", "attention.attention.query.bias").replace("attention.attention.k_bias", "attention.attention.key.bias")
                       .replace("attention.attention.v_bias", "attention.attention.value.bias"): v.detach().clone() for k, v in sd.items()}
    vsd, dsd = ren(vm.state_dict()), ren(dm.state_dict())
    vcfg = dict(hidden_size=128, num_attention_heads=2, num_hidden_layers=2, layer_norm_eps=1e-12, num_frames=4)
    dcfg = dict(hidden_size=128, num_attention_heads=2, num_hidden_layers=2, patch_size=14, layer_norm_eps=1e-6)
    model = cama.build_cama(P.VideoMAEEmbedder(vm, resize=32, crop=32), P.DINOImageEmbedder(dm, resize=32, crop=28), vision_dim=128, cond_dim=128,
                            dim=128, tokens=5, heads=2, depth=1, nhead=2, ff=256, layers=1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith(("vision_model.", "condition_model."))}
    model = model.to(DEV, torch.bfloat16)
    b, k = 1, 3
    batch = {"ref_videos": torch.rand(b, k, 6, 3, 36, 44) * 2 - 1, "video": torch.rand(b, 6, 3, 36, 44) * 2 - 1}
    out = model.predict({n: v.to(DEV, torch.bfloat16) for n, v in batch.items()}, do_classifier_free_guidance=True)
    vis = lambda x: E.videomae_embedder(vsd, vcfg, x.float(), resize=32, crop=32)
    con = lambda x: E.dino_embedder(dsd, dcfg, x.float(), resize=32, crop=28, pos_dialect="size")
    bf = {n: v.to(torch.bfloat16).float() for n, v in batch.items()}
    videos = torch.cat([bf["ref_videos"].flip(1), bf["video"][:, None]], dim=1)           # module.py:319-321
    K = videos.shape[1]
    vfeat = vis(videos.reshape(b * K, *videos.shape[2:]))
    cfeat = con(videos[:, :, 0].reshape(b * K, *videos.shape[3:]))
    ufeat = vis(torch.zeros(b, *videos.shape[2:]))
    spec = cama_ref.CamaSpec(heads=2, depth=1, nhead=2, num_layers=1, tokens=5, dim=128)
    want = cama_ref.cama_predict({k_: v.to(torch.bfloat16).float() for k_, v in sd.items()}, spec, vfeat, cfeat, ufeat, b)
    assert out.shape == want.shape == (2 * b, 5, 128)
    assert rel(out, want) <= 3e-2
