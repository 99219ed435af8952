"""GPU parity of the HIP attention processors and the adapter pipelines' glue against golden vectors produced by the REFERENCE'S OWN classes
(tests/golden/{cog,svd}_attn_processor.npz, adapter_pipelines.npz; generator oracle/gen_golden_attn_processor.py).  No restatement sits
between the product and the reference here: weights and inputs are the fixture's bf16-representable values, the expected outputs are what
`APAdapterCogVideoXAttnProcessor2_0.__call__` / `APAdapterAttnProcessor2_0.__call__` returned in fp32 on the CPU.

Tolerance (bf16 activations between the kernels vs an fp32 reference): relative Frobenius error <= 1 %, every element within
3 % + 4 % of the mean magnitude (four bf16-rounded GEMM results in a chain: a bf16 ulp at |x| ~ 2 is 0.016; measured worst element 0.044 at mean
magnitude 0.99)."""
import json
import os

import numpy as np
import pytest
import torch

from test_attn_processor_golden_cpu import action_embedder, cog_case_inputs, load
from test_gpu_kernels import close as close_elem

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, want, rel_l2=1e-2):
    g, w = got.float().cpu(), torch.as_tensor(want).float()
    assert g.shape == w.shape and torch.isfinite(g).all()
    l2 = ((g - w).norm() / w.norm()).item()
    assert l2 <= rel_l2, f"relative L2 error {l2:.4f} > {rel_l2}"
    close_elem(got, w, rtol=3e-2, atol_frac=4e-2)


def _load_weights(attn, proc, g):
    sd = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("attn.")}
    sd.update({"processor." + k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("proc.")})
    attn.set_processor(proc)
    missing, unexpected = attn.load_state_dict(sd, strict=True)
    return attn.to(DEV, torch.bfloat16)


def dev(a):
    return torch.from_numpy(np.asarray(a)).to(DEV, torch.bfloat16)


@pytest.mark.parametrize("fold", [True, False])
def test_cogvideox_hip_processor_against_the_reference_class(hip, golden_dir, fold, monkeypatch):
    from motionrag_amd import attn_processor as ap
    monkeypatch.setattr(ap, "FOLD_IP_QUERY", fold)
    g, meta = load(golden_dir, "cog_attn_processor.npz")
    attn = ap.Attention(meta["D"], heads=meta["H"], dim_head=64, bias=True, out_bias=True, qk_norm="layer_norm", eps=1e-6)
    proc = ap.APAdapterCogVideoXAttnProcessor2_0(meta["D"], meta["ip_dim"])
    attn = _load_weights(attn, proc, g)
    hidden, enc = dev(g["hidden"]), dev(g["enc"])
    for name, c in meta["cases"].items():
        rope, ip = cog_case_inputs(g, name)
        proc.scale = [c["scale"]]
        ip = ip.to(DEV, torch.bfloat16)
        if rope is not None:
            rope = (rope[0].to(DEV), rope[1].to(DEV))
        if name == "rope_list_kwarg":
            got_h, got_e = attn(hidden, enc, image_rotary_emb=list(rope), action_hidden_states=ip)
        elif name == "norope_kwarg":
            got_h, got_e = attn(hidden, enc, image_rotary_emb=None, action_hidden_states=ip)
        else:
            got_h, got_e = attn(hidden, enc, image_rotary_emb=(rope, ip))
        close(got_h, g[f"{name}.h"])
        close(got_e, g[f"{name}.e"])


def test_svd_hip_processor_against_the_reference_class(hip, golden_dir):
    from motionrag_amd import attn_processor as ap
    from motionrag_amd.svd import TupleTensor
    g, meta = load(golden_dir, "svd_attn_processor.npz")
    C, H, cd, F = meta["C"], meta["H"], meta["cross_dim"], meta["F"]
    attn = ap.Attention(C, cross_attention_dim=cd, heads=H, dim_head=64, bias=False, out_bias=True)
    proc = ap.APAdapterAttnProcessor2_0(C, cd)
    attn = _load_weights(attn, proc, g)
    hidden, hidden4, img, img2, img3, act = (dev(g[k]) for k in ("hidden", "hidden4", "img", "img2", "img3", "act"))
    close(attn(hidden, (img, act)), g["out.tuple"])
    close(attn(hidden, (img3, act)), g["out.tuple_img3"])
    close(attn(hidden, TupleTensor([img2, act]).to(DEV, torch.bfloat16).repeat_interleave(F, dim=0)), g["out.tupletensor"])
    close(attn(hidden, img, action_hidden_states=act), g["out.kwarg"])
    close(attn(hidden4, (img3, act)), g["out.hidden4"])
    attn.residual_connection = True
    close(attn(hidden, (img, act)), g["out.resid"])
    close(attn(hidden4, (img3, act)), g["out.hidden4_resid"])
    attn.residual_connection = False
    proc.scale = [0.0]
    close(attn(hidden, (img, act)), g["out.scale_zero"])
    proc.scale = [0.6]
    close(attn(hidden, (img, act)), g["out.scale_06"])
    proc.scale = [1.0]
    attn.rescale_output_factor = 2.0
    close(attn(hidden, (img, act)), g["out.rescale2"])
    # `block_residual` (the caller's x of `x = attn2(norm2(x)) + x`, added in the output projection's epilogue): the reference divides only the processor's
    # output (:139) and the caller adds x afterwards -> x + out / f, NOT (x + out) / f
    xres = dev(g["hidden4"].reshape(2 * F, C, 36).transpose(0, 2, 1).copy())           # any [2F, 36, C] tensor
    want = torch.from_numpy(g["out.rescale2"]) + xres.float().cpu()
    close(attn(hidden, (img, act), block_residual=xres), want)


class _CT:
    """condition_transformer stand-in of the generator (oracle.gen_golden_attn_processor.CTRecorder) on the GPU"""

    def __init__(self, seed):
        from oracle.stubs import ConditionTransformerStub
        self.inner = ConditionTransformerStub(dim=16, seed=seed).to(DEV)

    def predict(self, batch, do_classifier_free_guidance=False):
        self.batch, self.cfg = batch, do_classifier_free_guidance
        y = self.inner.predict(batch)
        return torch.cat([torch.zeros_like(y), y]) if do_classifier_free_guidance else y


@torch.no_grad()
def test_adapter_pipelines_glue_against_the_reference_classes(hip, golden_dir):
    """`prepare_action_embeddings` / `_prepare_rotary_positional_embeddings` (cogvideox/pipeline.py:46-78,117-130) and the SVD pipelines'
    `prepare_action_embeddings` / `_encode_image` / stage-2 image hop (svd/pipelines/pipeline.py:99-119,154-158) of the product classes against the
    reference classes' outputs.  fp32 glue around `mrag_weighted_sum_bf16` (bf16 in / out): 1 % bound as above."""
    from motionrag_amd import cogvideox, svd
    g, meta = load(golden_dir, "adapter_pipelines.npz")
    emb_fp32 = action_embedder(meta).to(DEV)
    emb = lambda v: emb_fp32(v).to(torch.bfloat16)      # noqa: E731 -- the frozen embedder's tokens arrive in the model dtype (precision: bf16-true)
    proj = torch.nn.Linear(16, 24)
    proj.load_state_dict({"weight": torch.from_numpy(g["proj.weight"]), "bias": torch.from_numpy(g["proj.bias"])})
    proj = proj.to(DEV, torch.bfloat16)               # the fused tokens leave `mrag_weighted_sum_bf16` in bf16 (reference precision: bf16-true)
    ref_videos = torch.from_numpy(g["ref_videos"]).to(DEV)
    metadata = [{"ref_video_distance": d} for d in g["dist"].tolist()]
    model = torch.nn.Linear(1, 1).to(DEV)                                     # `_execution_device` reads the transformer's / unet's device
    for fusion in ("mean", "weight", "top1", "concat"):
        p = cogvideox.CogVideoXImageToVideoActionPipeline(transformer=model, action_embedder=emb, action_proj_model=proj, ref_fusion_type=fusion)
        close(p.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=False), g[f"cog.action.{fusion}.nocfg"])
        if meta["cog"][f"{fusion}.cfg"] == "ok":
            close(p.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=True), g[f"cog.action.{fusion}.cfg"])
        else:
            with pytest.raises((RuntimeError, ValueError)):
                p.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=True)
    p.action_emb = torch.zeros(2, 5, 24, device=DEV)
    (cos, sin), tok = p._prepare_rotary_positional_embeddings(2, 3, 5, DEV)
    assert tok is p.action_emb and cos.shape == (30, 64)                    # the hook returns ((cos, sin), action_emb) (:57)
    # stage 2: the batch handed to the condition transformer
    ct = _CT(meta["ct_seed"])
    p = cogvideox.CogVideoXImageToVideoCTPipeline(transformer=model, condition_transformer=ct)
    got = p.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=True, image=torch.from_numpy(g["image01"]).to(DEV))
    assert ct.cfg is True and ct.batch["ref_videos"] is ref_videos
    np.testing.assert_array_equal(ct.batch["video"].cpu().numpy(), g["cog.ct.video"])
    np.testing.assert_allclose(got.cpu().numpy(), g["cog.ct.action_emb"], rtol=1e-4, atol=1e-5)
    # SVD stage 1
    for fusion in ("mean", "weight", "top1"):
        p = svd.SVDActionPipeline(unet=model, action_embedder=emb, action_proj_model=proj, ref_fusion_type=fusion)
        close(p.prepare_action_embeddings(ref_videos, metadata), g[f"svd.action.{fusion}"])
    img_enc = torch.nn.Linear(3 * 8 * 8, 12)
    img_enc.load_state_dict({"weight": torch.from_numpy(g["img_enc.weight"]), "bias": torch.from_numpy(g["img_enc.bias"])})
    img_enc = img_enc.to(DEV)
    p = svd.SVDActionPipeline(unet=model, image_encoder=lambda x: img_enc(x.reshape(x.shape[0], -1).float())[:, None], feature_extractor=None,
                              action_embedder=emb, action_proj_model=proj, ref_fusion_type="mean")
    p.action_emb = p.prepare_action_embeddings(ref_videos, metadata)
    tt = p._encode_image(torch.from_numpy(g["image01"]).to(DEV), DEV, True)
    assert isinstance(tt, svd.TupleTensor)
    t0, t1 = tt.to_tuple()
    np.testing.assert_allclose(t0.float().cpu().numpy(), g["svd.action.tt0"], rtol=1e-4, atol=1e-5)
    assert t1 is p.action_emb
    # SVD stage 2: uint8 image -> / 127.5 - 1 -> repeated over the clip length
    ct = _CT(meta["ct_seed"])
    p = svd.SVDCTPipeline(unet=model, condition_transformer=ct)
    p._run = lambda *a, **k: "ran"                                             # the diffusers body is not under test here
    assert p(ref_videos=ref_videos, metadata=metadata, image=[torch.from_numpy(im).permute(2, 0, 1) for im in g["image_u8"]]) == "ran"
    assert ct.cfg is True and ct.batch["ref_videos"] is ref_videos
    np.testing.assert_array_equal(ct.batch["video"].cpu().numpy(), g["svd.ct.video"])
    np.testing.assert_allclose(p.action_emb.cpu().numpy(), g["svd.ct.action_emb"], rtol=1e-4, atol=1e-5)
