"""The reference's PIPELINE protocol driven end to end on the GPU with stub third-party modules (text encoder, VAEs, CLIP image
encoder, frozen action embedder): `pipe(prompt=, image=, negative_prompt=, output_type='pt', ref_videos=, metadata=, ...)`, `.frames` /
`frames[0]`, `eval_pipeline` post-processing (first-16 / uniform sampling, `* 2 - 1`), the stage-1 Action pipelines with `condition_fusion`,
and the retrieval consumer contract.  References: src/projects/cogvideox/pipeline.py:13-130, cogvideox/module.py:163-223,
src/projects/svd/pipelines/pipeline.py:25-160, svd/module.py:145-191, src/data/dataset.py:285-312, src/data/datamodule.py:225-265."""
import numpy as np
import pytest
import torch

from test_gpu_models import _small_dit, close
from test_oracle_golden import svd_tiny

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close_chained_cfg(got, want, rel_l2=4e-2, p999_frac=0.2, max_frac=0.35):
    """Two chained CFG steps of a bf16 UNet against the fp32 oracle.  Guidance multiplies the two branches' independent bf16 errors, so single
    elements scatter widely while the aggregate error stays put; an absolute slack wide enough for the worst element on EVERY element would hide
    real regressions.  Three instruments instead: relative Frobenius error <= 4 % (measured 3.0 %); the 99.9th percentile of the element error
    beyond 5 % of the element, in units of the mean magnitude, <= 0.2 (the level no element but one outlier of 4 096 has reached); the
    maximum <= 0.35 (the bound of rounds 3-4: the one measured outlier reaches 0.32; ADVICE r5 asked for it back)."""
    g, w = got.float().cpu(), want.float().cpu()
    assert g.shape == w.shape and torch.isfinite(g).all()
    l2 = ((g - w).norm() / w.norm()).item()
    assert l2 <= rel_l2, f"relative L2 error {l2:.4f} > {rel_l2}"
    s = w.abs().mean().item()
    e = (((g - w).abs() - 5e-2 * w.abs()).clamp_min(0) / s).flatten()
    p999 = torch.quantile(e, 0.999).item()
    assert p999 <= p999_frac, f"99.9th percentile of the element error {p999:.3f} of the mean magnitude > {p999_frac}"
    assert e.max().item() <= max_frac, f"largest element error {e.max().item():.3f} of the mean magnitude > {max_frac}"


class StubText:
    """T5 stand-in: deterministic [b, 10, 64] embeddings from the strings"""

    def __call__(self, texts):
        out = []
        for t in texts:
            g = torch.Generator().manual_seed(sum(map(ord, t)) + 7)
            out.append(torch.randn(10, 64, generator=g))
        return torch.stack(out).to(DEV, torch.bfloat16)


class StubVAE:
    """8x spatial / 4x temporal toy VAE: encode = 8x8 average pool of the 3 channels tiled to C latent channels; decode = nearest upsample"""

    class Cfg:
        scaling_factor = 0.7

    config = Cfg()

    def __init__(self, c_lat):
        self.c = c_lat

    def encode(self, x):                      # [b, 3, 1, H, W] in [-1, 1]
        z = torch.nn.functional.avg_pool2d(x[:, :, 0].float(), 8)                     # [b, 3, h, w]
        return z.repeat(1, (self.c + 2) // 3, 1, 1)[:, : self.c].unsqueeze(2)

    def decode(self, z):                      # [b, C, F, h, w] -> [b, 3, 4 (F - 1) + 1, 8 h, 8 w]
        v = torch.tanh(z[:, :3].float())
        v = torch.nn.functional.interpolate(v, scale_factor=(1, 8, 8), mode="nearest")
        idx = torch.arange(4 * (v.shape[2] - 1) + 1, device=v.device) // 4
        return v[:, :, idx]


class StubCAMA:
    """condition_transformer stand-in with the CAMA protocol: predict(batch, do_classifier_free_guidance) -> [2b, 25, 64], uncond first"""

    def __init__(self):
        self.calls = []

    def predict(self, batch, do_classifier_free_guidance=False):
        self.calls.append({k: tuple(v.shape) for k, v in batch.items()})
        b = batch["ref_videos"].shape[0]
        g = torch.Generator().manual_seed(5)
        cond = torch.randn(b, 25, 64, generator=g) + batch["video"].float().mean().cpu()
        un = torch.randn(b, 25, 64, generator=g)
        return torch.cat([un, cond] if do_classifier_free_guidance else [cond]).to(DEV, torch.bfloat16)


def _pipe_inputs(b=1):
    g = torch.Generator().manual_seed(21)
    image = torch.rand(b, 3, 64, 96, generator=g) * 2 - 1                             # eval_pipeline takes [-1, 1]
    ref_videos = torch.randn(b, 9, 8, 3, 16, 16, generator=g).to(DEV, torch.bfloat16)
    metadata = [{"ref_video_distance": (0.1 + 0.08 * torch.arange(9)).tolist()} for _ in range(b)]
    return image.to(DEV), ref_videos, metadata


def test_cogvideox_ct_pipeline_reference_call_surface(hip):
    from motionrag_amd import cogvideox as cvx
    cfg, sd, dit = _small_dit()
    cama = StubCAMA()
    pipe = cvx.CogVideoXImageToVideoCTPipeline(tokenizer=None, text_encoder=StubText(), vae=StubVAE(8), transformer=dit,
                                               scheduler=cvx.CogVideoXDDIMScheduler(), condition_transformer=cama)
    pipe.set_progress_bar_config(disable=True)
    image, ref_videos, metadata = _pipe_inputs()
    kw = dict(num_frames=9, num_inference_steps=2, guidance_scale=6.0, height=64, width=96)
    out = pipe(prompt=["a dog runs"], image=image / 2 + 0.5, negative_prompt=["blurry"], output_type="pt", ref_videos=ref_videos, metadata=metadata,
               generator=torch.Generator().manual_seed(3), **kw)
    assert out.frames.shape == (1, 9, 3, 64, 96) and out[0] is out.frames
    assert 0.0 <= out.frames.min().item() and out.frames.max().item() <= 1.0
    assert cama.calls[-1] == {"ref_videos": (1, 9, 8, 3, 16, 16), "video": (1, 8, 3, 64, 96)}       # image repeated over the reference clip length (:127-128)
    # the same call, latent output, equals the hot loop driven by hand with the same noise, embeddings and image latents
    lat = pipe(prompt=["a dog runs"], image=image / 2 + 0.5, negative_prompt=["blurry"], output_type="latent", ref_videos=ref_videos,
               metadata=metadata, generator=torch.Generator().manual_seed(3), **kw).frames
    noise = torch.randn(1, 3, 8, 8, 12, generator=torch.Generator().manual_seed(3), dtype=torch.bfloat16).to(DEV)   # randn_tensor draws IN bf16
    te = StubText()
    pe = torch.cat([te(["blurry"]), te(["a dog runs"])])
    il = pipe.encode_image_latents(image / 2 + 0.5, 3).to(DEV, torch.bfloat16)
    assert il.shape == (1, 3, 8, 8, 12) and il[:, 1:].abs().max().item() == 0               # first frame's latents, zero-padded
    ae = cama.predict({"ref_videos": ref_videos, "video": (image / 2 + 0.5)[:, None].expand(-1, 8, -1, -1, -1).to(torch.bfloat16)}, True)
    want = pipe.denoise(noise.clone(), il.contiguous(), pe.contiguous(), ae, num_inference_steps=2, guidance_scale=6.0)
    assert torch.equal(lat, want)
    # eval_pipeline (module.py:197-223): denormalise the image, first-16 / uniform sampling, back to [-1, 1]
    vid = cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, ref_videos, metadata, generator=torch.Generator().manual_seed(3), **kw)
    assert vid.shape == (1, 9, 3, 64, 96) and torch.allclose(vid, out.frames * 2 - 1)
    # uniform sampling of 16 of the decoded frames (module.py:214-216); a clip longer than the model's sample length runs on a regenerated positional
    # table (diffusers' patch embedding), another resolution is refused
    class LongVAE(StubVAE):
        def decode(self, z):
            return super().decode(z).repeat_interleave(3, dim=2)                                      # 27 frames out of 3 latent frames
    pipe.vae = LongVAE(8)
    uni = cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, ref_videos, metadata, sample_method="uniform",
                            generator=torch.Generator().manual_seed(3), **kw)
    assert uni.shape == (1, 16, 3, 64, 96)
    longer = pipe(prompt=["x"], image=image / 2 + 0.5, negative_prompt=["y"], output_type="latent", ref_videos=ref_videos, metadata=metadata,
                  **dict(kw, num_frames=33)).frames
    assert longer.shape == (1, 9, 8, 8, 12) and torch.isfinite(longer.float()).all()
    with pytest.raises(ValueError):
        pipe(prompt=["x"], image=(image / 2 + 0.5)[..., :80], negative_prompt=["y"], output_type="latent", ref_videos=ref_videos, metadata=metadata,
             **dict(kw, width=80))
    with pytest.raises(ValueError):
        cvx.eval_pipeline(pipe, image, ["x"], ["y"], torch.bfloat16, ref_videos, metadata, sample_method="bogus", **kw)
    # the shipped eval_pipeline_call_kwargs (MotionRAG_open.yml:189-194): scheduler 'dpm', sample_method null, guidance 3 -- every decoded frame comes back
    # the motion-injected pipelines are BUILT with the DPM scheduler (module.py:44, 183, 255); the YAML's `scheduler` entry is validated and dropped for them
    pipe.vae = StubVAE(8)
    ddim = pipe.scheduler
    shipped = dict(num_inference_steps=3, num_frames=9, guidance_scale=3, sample_method=None, scheduler="dpm", height=64, width=96)
    cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, ref_videos, metadata, generator=torch.Generator().manual_seed(3), **shipped)
    assert pipe.scheduler is ddim                                                                  # not swapped behind the caller's back
    pipe = cvx.CogVideoXImageToVideoCTPipeline(tokenizer=None, text_encoder=StubText(), vae=StubVAE(8), transformer=dit, condition_transformer=cama)
    assert isinstance(pipe.scheduler, cvx.CogVideoXDPMScheduler)                                   # the constructor's default is the reference's choice
    dpm = cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, ref_videos, metadata, generator=torch.Generator().manual_seed(3),
                            **dict(shipped, scheduler="ddim"))
    assert isinstance(pipe.scheduler, cvx.CogVideoXDPMScheduler) and dpm.shape == (1, 9, 3, 64, 96) and torch.isfinite(dpm.float()).all()
    dpm2 = cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, ref_videos, metadata, generator=torch.Generator().manual_seed(3), **shipped)
    assert torch.equal(dpm, dpm2)                                                                  # the sampler's noise comes from the generator
    with pytest.raises(ValueError):
        cvx.eval_pipeline(pipe, image, ["x"], ["y"], torch.bfloat16, ref_videos, metadata, **dict(shipped, scheduler="euler"))


def test_cogvideox_set_attention_processors_by_name(hip):
    """cogvideox/module.py:163-175: adapters only on the listed processor names; state-dict keys as in Motion-Adapter.ckpt"""
    from motionrag_amd import cogvideox as cvx
    from motionrag_amd.attn_processor import APAdapterCogVideoXAttnProcessor2_0
    dit = cvx.CogVideoXTransformer3DModel(num_layers=3, num_attention_heads=2, in_channels=16, out_channels=8, time_embed_dim=64, text_embed_dim=64,
                                          max_text_seq_length=10, sample_frames=3, sample_height=8, sample_width=12)
    cvx.set_attention_processors(dit, ["transformer_blocks.0.attn1.processor", "transformer_blocks.2.attn1.processor"], 64)
    kinds = [isinstance(p, APAdapterCogVideoXAttnProcessor2_0) for p in dit.attn_processors.values()]
    assert kinds == [True, False, True]
    keys = set(dit.state_dict().keys())
    assert "transformer_blocks.2.attn1.processor.to_q_ip.0.weight" in keys and "transformer_blocks.1.attn1.processor.to_q_ip.0.weight" not in keys


def test_cogvideox_action_pipeline_condition_fusion(hip):
    """stage 1 (pipeline.py:59-78): action_embedder over the (b k) clips -> condition_fusion over k (HIP kernel) -> uncond first -> action_proj_model"""
    from motionrag_amd import cogvideox as cvx
    from oracle import cama_ref
    cfg, sd, dit = _small_dit()

    class Embedder(torch.nn.Module):
        def forward(self, clips):                     # [(b k), f, c, h, w] -> [(b k), 25, 64]
            m = clips.float().mean(dim=(1, 2, 3, 4))
            base = torch.linspace(-1, 1, 25 * 64, device=clips.device).view(1, 25, 64)
            return (base * (1 + m.view(-1, 1, 1)) + m.view(-1, 1, 1)).to(torch.bfloat16)

    proj = torch.nn.Identity()
    image, ref_videos, metadata = _pipe_inputs(b=2)
    for mode in ("mean", "weight", "top1"):
        pipe = cvx.CogVideoXImageToVideoActionPipeline(text_encoder=StubText(), vae=StubVAE(8), transformer=dit, scheduler=cvx.CogVideoXDDIMScheduler(),
                                                       action_embedder=Embedder(), action_proj_model=proj, ref_fusion_type=mode)
        got = pipe.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=True)
        emb = Embedder()(ref_videos.reshape(18, 8, 3, 16, 16)).view(2, 9, 25, 64)
        want = torch.cat([Embedder()(torch.zeros_like(ref_videos[:, 0])).float().cpu(),
                          cama_ref.condition_fusion(emb.float().cpu(), mode, [m["ref_video_distance"] for m in metadata] if mode == "weight" else None)])
        assert got.shape == (4, 25, 64)
        assert ((got.float().cpu() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6).all(), mode
    out = pipe(prompt=["a", "b"], image=image / 2 + 0.5, negative_prompt=["", ""], output_type="latent", ref_videos=ref_videos, metadata=metadata,
               num_frames=9, num_inference_steps=1, guidance_scale=3.0, height=64, width=96, generator=torch.Generator().manual_seed(1))
    assert out.frames.shape == (2, 3, 8, 8, 12) and torch.isfinite(out.frames.float()).all()


def test_svd_ct_pipeline_matches_oracle_loop(hip):
    """SVDCTPipeline.__call__ (pipeline.py:147-160) + the restated diffusers body: 2 Euler steps with per-frame guidance on the reduced-width
    UNet against the fp32 oracle loop (svd_ref.unet_forward + euler_cfg_step) fed the same noise / embeddings; then eval_pipeline's
    `.frames[:, :16] * 2 - 1`"""
    from motionrag_amd import svd, svd_unet
    from oracle import svd_ref
    unet, cfg, inp = svd_tiny()
    sdict = {k: v.float() for k, v in unet.state_dict().items()}
    unet = unet.to(DEV)
    b, Fr, h, w = 1, 4, 16, 16

    class ImgEnc:
        def __call__(self, x):                        # [b, 3, H, W] -> [b, 64]
            return torch.tanh(x.float().mean(dim=(2, 3)).repeat(1, 22)[:, :64]).to(torch.bfloat16)

    class VAE:
        class Cfg:
            scaling_factor = 0.18215
        config = Cfg()

        def encode(self, x):
            return torch.nn.functional.avg_pool2d(x.float(), 8).repeat(1, 2, 1, 1)[:, :4]

        def decode(self, z, num_frames=None):
            return torch.tanh(torch.nn.functional.interpolate(z[:, :3].float(), scale_factor=8, mode="nearest"))

    cama = StubCAMA()
    pipe = svd.SVDCTPipeline(vae=VAE(), image_encoder=ImgEnc(), unet=unet, scheduler=svd_unet.EulerDiscreteScheduler(), feature_extractor=None,
                             condition_transformer=cama)
    g = torch.Generator().manual_seed(8)
    img255 = torch.rand(b, 3, 8 * h, 8 * w, generator=g) * 255.0
    ref_videos = torch.randn(b, 9, 8, 3, 16, 16, generator=g).to(DEV, torch.bfloat16)
    kw = dict(height=8 * h, width=8 * w, num_frames=Fr, num_inference_steps=2, min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=127,
              noise_aug_strength=0.02)
    got = pipe(image=img255, ref_videos=ref_videos, metadata=None, output_type="latent", generator=torch.Generator().manual_seed(9), **kw).frames
    # ---- oracle loop on the same random stream
    gen = torch.Generator().manual_seed(9)
    img = img255 / 127.5 - 1.0
    noise = torch.randn(img.shape, generator=gen, dtype=torch.bfloat16)                    # randn_tensor draws IN the pipeline dtype
    sig = svd_ref.karras_sigmas(2)
    lat = (torch.randn(b, Fr, 4, h, w, generator=gen, dtype=torch.bfloat16) * float((sig[0] ** 2 + 1) ** 0.5)).float()
    emb = ImgEnc()(img).float().unsqueeze(1)
    emb2 = torch.cat([torch.zeros_like(emb), emb])
    act = cama.predict({"ref_videos": ref_videos, "video": img[:, None].expand(-1, 8, -1, -1, -1).to(DEV, torch.bfloat16)}, True).float().cpu()
    z = VAE().encode((img.to(torch.bfloat16) + 0.02 * noise)).to(torch.bfloat16).float()   # the noise-augmented image is formed in bf16
    il = torch.cat([torch.zeros_like(z), z])[:, None].expand(-1, Fr, -1, -1, -1)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    gs = torch.linspace(1.0, 3.0, Fr)
    for i in range(2):
        s, sn = float(sig[i]), float(sig[i + 1])
        scaled = (lat / (s * s + 1) ** 0.5).to(torch.bfloat16).float()
        x = torch.cat([torch.cat([scaled, scaled]), il], dim=2)
        v = svd_ref.unet_forward(sdict, cfg, x, torch.tensor(0.25 * np.log(s)), emb2, ids, act)
        lat = svd_ref.euler_cfg_step(v[:b].double(), v[b:].double(), lat.double(), s, sn, gs.double()).float().to(torch.bfloat16).float()
    close_chained_cfg(got, lat)                              # Frobenius + 99.9th percentile + loose maximum (see the helper)
    vid = svd.eval_pipeline(pipe, img255 / 127.5 - 1.0, ref_videos=ref_videos, metadata=None, generator=torch.Generator().manual_seed(9), **kw)
    assert vid.shape == (b, Fr, 3, 8 * h, 8 * w) and -1.0 <= vid.min().item() and vid.max().item() <= 1.0
    # the uint8 hop of svd/module.py:181 (tensor2PIL -> denormalize -> uint8) + pipeline.py:154-155 (pil_to_tensor / 127.5 - 1): eval_pipeline's clip is
    # the clip of the pipeline fed the reference's own quantised image, NOT of the float image it was handed
    pm1 = img255 / 127.5 - 1.0
    u8 = (torch.clip((pm1 + 1.0) / 2.0, 0.0, 1.0) * 255).to(torch.uint8)            # src/utils/pipeline.py:178-184, verbatim arithmetic on the host
    assert (u8.float() != img255).any()                                                # the hop is not the identity on this image
    want = pipe(image=u8, ref_videos=ref_videos, metadata=None, output_type="pt", generator=torch.Generator().manual_seed(9), **kw).frames[:, :16] * 2 - 1
    assert torch.equal(vid, want)
    unq = pipe(image=img255, ref_videos=ref_videos, metadata=None, output_type="pt", generator=torch.Generator().manual_seed(9), **kw).frames[:, :16] * 2 - 1
    assert not torch.equal(vid, unq)


def test_retrieval_consumer_contract_and_fan_out(hip):
    """datamodule.py:225-265 fan-out (`top_k = K + 3`, self-exclusion, select) as one batched launch, then dataset.py:285-312: first K rows,
    `_distance` list, zero video + 1.0 for a failed or dropped reference"""
    from motionrag_amd import rag
    annos = rag.synthetic_captions(600)
    embed = rag.hash_embedder(768)
    emb = np.stack([embed(a["motion_caption"]) for a in annos])
    db = rag.RAGDatabase.from_arrays(emb, rag.prepare_annotations(annos, "motion_caption", "openvid"), prefilter=True)
    for a, e in zip(annos, emb):
        a["text_embedding"] = e
    rag.attach_ref_videos(annos, db, ref_video_num=9, chunk=256)
    a = annos[17]
    assert len(a["ref_videos"]) == 12 and set(a["ref_videos"][0]) == {"video", "start_sec", "end_sec", "_distance"}
    assert all(r["video"] != a["video"] for r in a["ref_videos"])
    d = [r["_distance"] for r in a["ref_videos"]]
    assert d == sorted(d)
    video = torch.randn(1, 8, 3, 4, 4)

    def load_clip(row):
        if row["video"] == a["ref_videos"][2]["video"]:
            raise IOError("missing file")
        return torch.full((1, 8, 3, 4, 4), float(int(row["video"][5:11])))

    refs, dist = rag.get_ref_videos(a, video, load_clip, ref_video_num=9)
    assert refs.shape == (9, 8, 3, 4, 4) and len(dist) == 9
    assert dist[2] == 1.0 and refs[2].abs().max().item() == 0 and dist[0] == a["ref_videos"][0]["_distance"]
    assert refs[0, 0, 0, 0, 0].item() == float(int(a["ref_videos"][0]["video"][5:11]))
    refs, dist = rag.get_ref_videos(a, video, load_clip, ref_video_num=9, uncond_video_ratio=1.0)        # all references dropped
    assert dist == [1.0] * 9 and refs.abs().max().item() == 0


def test_retrieval_filter_order_on_multi_clip_videos(hip):
    """src/data/rag.py:54-58 on a table with 6 clips per video (the multi-clip datasets MotionRAG retrieves from): lancedb 0.14.0 applies
    `video != "<self>"` AFTER taking the K + 3 = 12 nearest rows, so a query whose own video fills 5 of them gets 7 references, not 9;
    `get_ref_videos` (dataset.py:285-312) then leaves zero videos in the unused slots and a distance list as long as the rows it was handed.
    Both filter orders bit-exact against the C oracle in the matching mode."""
    from motionrag_amd import rag
    from oracle import topk_ref
    from test_oracle_golden import multi_clip_db
    rng = np.random.default_rng(5)
    emb, group = multi_clip_db(rng, n_videos=120, clips=6, dim=768, spread=0.02)
    annos = [{"motion_caption": f"clip {i}", "id": i, "video": f"video_{i // 6:04d}.mp4", "start_sec": float(4 * (i % 6)), "end_sec": float(4 * (i % 6) + 4)}
             for i in range(len(emb))]
    rows = rag.prepare_annotations(annos, "motion_caption", "openvid")
    queries = [dict(a, text_embedding=emb[a["id"]]) for a in annos[3::97]]          # the query clips are themselves rows of the table
    own = np.array([a["id"] // 6 for a in queries], np.int32)
    q = np.stack([a["text_embedding"] for a in queries])
    got = {}
    for pre in (False, True):
        db = rag.RAGDatabase.from_arrays(emb, rows, prefilter=pre)
        qa = [dict(a) for a in queries]
        rag.attach_ref_videos(qa, db, ref_video_num=9)
        want_r, want_d = topk_ref.topk(emb, q, 12, "l2", group, own, mode="f32chain", postfilter=not pre)
        for a, wr, wd in zip(qa, want_r, want_d):
            keep = wr >= 0
            assert [(r["video"], r["start_sec"]) for r in a["ref_videos"]] == [(annos[i]["video"], annos[i]["start_sec"]) for i in wr[keep]]
            np.testing.assert_array_equal(np.array([r["_distance"] for r in a["ref_videos"]], np.float32), wd[keep].astype(np.float32))
            assert all(r["video"] != a["video"] for r in a["ref_videos"])
        got[pre] = qa
    assert all(len(a["ref_videos"]) == 12 for a in got[True])                      # pre-filter: always K + 3 rows
    assert all(len(a["ref_videos"]) == 6 for a in got[False])                      # lancedb's order: the query's 6 clips were in the 12 nearest
    for a_post, a_pre in zip(got[False], got[True]):
        assert a_post["ref_videos"] == a_pre["ref_videos"][:6]                     # the survivors are the head of the pre-filtered list
    a = got[False][0]
    video = torch.randn(1, 8, 3, 4, 4)
    refs, dist = rag.get_ref_videos(a, video, lambda row: torch.full((1, 8, 3, 4, 4), row["start_sec"] + 1.0), ref_video_num=9)
    assert refs.shape == (9, 8, 3, 4, 4) and len(dist) == 6                        # dataset.py:296: only the rows that exist are visited
    assert refs[6:].abs().max().item() == 0 and all(refs[i].abs().min().item() > 0 for i in range(6))
    assert dist == [r["_distance"] for r in a["ref_videos"]]


def test_svd_ct_pipeline_all_native_components(hip):
    """SVDCTPipeline with EVERY model on the HIP path: the UNet, the temporal-decoder VAE (`svd_vae`), the CLIP image encoder with head_dim 80 (`clip_vision`) and CAMA.
    Composition check of the duck-typed hand-offs (`latent_dist.mode()`, `.sample`, `.image_embeds`, `config.scaling_factor`): shapes, finiteness, determinism, and the
    decoded frames equal decoding the pipeline's own latents by hand."""
    from motionrag_amd import clip_vision, svd, svd_unet, svd_vae
    unet, cfg, inp = svd_tiny()
    unet = unet.to(DEV)
    torch.manual_seed(41)
    vae = svd_vae.AutoencoderKLTemporalDecoder(block_out_channels=(64, 64, 128, 128), layers_per_block=1).to(DEV, torch.bfloat16)
    enc = clip_vision.CLIPVisionModelWithProjection(hidden_size=160, intermediate_size=320, num_hidden_layers=1, num_attention_heads=2, image_size=28, patch_size=14,
                                                    projection_dim=64).to(DEV, torch.bfloat16)

    def feature_extractor(x):                      # stands for the 224 x 224 resize + CLIP normalisation of the third-party processor
        return torch.nn.functional.interpolate(x.float(), size=(28, 28), mode="bilinear").to(DEV, torch.bfloat16)

    pipe = svd.SVDCTPipeline(vae=vae, image_encoder=enc, unet=unet, scheduler=svd_unet.EulerDiscreteScheduler(), feature_extractor=feature_extractor,
                             condition_transformer=StubCAMA())
    b, Fr, h, w = 1, 4, 16, 16
    g = torch.Generator().manual_seed(8)
    img255 = torch.rand(b, 3, 8 * h, 8 * w, generator=g) * 255.0
    ref_videos = torch.randn(b, 9, 8, 3, 16, 16, generator=g).to(DEV, torch.bfloat16)
    kw = dict(height=8 * h, width=8 * w, num_frames=Fr, num_inference_steps=2, min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=127,
              noise_aug_strength=0.02)
    lat = pipe(image=img255, ref_videos=ref_videos, metadata=None, output_type="latent", generator=torch.Generator().manual_seed(9), **kw).frames
    frames = pipe(image=img255, ref_videos=ref_videos, metadata=None, output_type="pt", generator=torch.Generator().manual_seed(9), **kw).frames
    frames2 = pipe(image=img255, ref_videos=ref_videos, metadata=None, output_type="pt", generator=torch.Generator().manual_seed(9), **kw).frames
    assert lat.shape == (b, Fr, 4, h, w) and frames.shape[0] == b and frames.shape[-2:] == (8 * h, 8 * w)
    assert torch.isfinite(frames.float()).all() and torch.equal(frames, frames2)
    by_hand = vae.decode((lat.float().to(DEV) / vae.config.scaling_factor).flatten(0, 1), num_frames=Fr).sample
    assert by_hand.numel() == frames.numel()


def test_dynamicrafter_pipeline_all_native_components(hip, golden_dir):
    """DynamiCrafterPipelineRef / image_guided_synthesis with EVERY model on the HIP path: the UNet + DDIM sampler, the KL-VAE (`dynamicrafter_vae.FirstStage`),
    the OpenCLIP text tower (`openclip_text`) as `get_learned_conditioning`, the OpenCLIP image tower with head_dim 80 (`clip_vision`) + the image Resampler, and CAMA's
    condition transformer stub.  Composition check of the duck-typed hand-offs: shapes, finiteness, determinism under fixed noise."""
    import json
    import types
    from motionrag_amd import cama, clip_vision, dynamicrafter as dc, dynamicrafter_vae as V, openclip_text as T
    from motionrag_amd.dynamicrafter_pipeline import DynamiCrafterPipelineRef
    from oracle import stubs
    from test_oracle_golden import dc_unet_fixture
    _, sd, _, _, _ = dc_unet_fixture(golden_dir)
    unet = dc.UNetModel(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=(1, 2), num_res_blocks=1, channel_mult=(1, 2),
                        num_head_channels=64, transformer_depth=1, context_dim=64, use_linear=True, temporal_conv=True, temporal_attention=True,
                        temporal_self_att_only=True, use_relative_position=False, temporal_length=4, addition_attention=True,
                        image_cross_attention=True, action_cross_attention=True, default_fs=10, fs_condition=True)
    unet.load_state_dict(sd, strict=True)
    unet = unet.to(DEV, torch.bfloat16)
    torch.manual_seed(77)
    vae = V.AutoencoderKL(dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 1, 2, 2], num_res_blocks=1, attn_resolutions=[],
                               dropout=0.0), embed_dim=4).to(DEV, torch.bfloat16)
    fs = V.FirstStage(vae, scale_factor=0.18215)
    text_model = T.OpenCLIPTextModel(vocab_size=100, width=64, heads=1, layers=2, context_length=7, embed_dim=64).to(DEV, torch.bfloat16)
    tokens = {"": torch.zeros(1, 7, dtype=torch.long), "a prompt": torch.tensor([[98, 5, 6, 7, 99, 0, 0]])}
    text = T.FrozenOpenCLIPEmbedder(text_model, tokenizer=lambda ts: torch.cat([tokens[t] for t in ts]), layer="penultimate")
    visual = clip_vision.OpenCLIPVisual(width=160, layers=1, heads=2, mlp_ratio=2.0, image_size=28, patch_size=14, output_dim=32)
    embedder = clip_vision.FrozenOpenCLIPImageEmbedderV2(visual.to(DEV, torch.bfloat16),
                                                         preprocess=lambda x: torch.nn.functional.interpolate(x.float(), size=(28, 28), mode="bilinear").to(torch.bfloat16))
    proj = cama.Resampler(dim=64, depth=1, dim_head=64, heads=2, num_queries=3, embedding_dim=160, output_dim=64, video_length=4).to(DEV, torch.bfloat16)
    model = types.SimpleNamespace(
        model=types.SimpleNamespace(conditioning_key="hybrid", diffusion_model=unet), uncond_type="empty_seq", action_embedder=None,
        embedder=embedder, image_proj_model=proj, condition_transformer=stubs.ConditionTransformerStub(dim=64).to(DEV), get_learned_conditioning=text,
        encode_first_stage=fs.encode_first_stage, decode_first_stage=fs.decode_first_stage)
    pipe = DynamiCrafterPipelineRef(model)
    g = torch.Generator().manual_seed(3)
    image = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    ref_videos = torch.randn(1, 2, 4, 3, 16, 16, generator=g)
    x_T = torch.randn(1, 4, 4, 8, 8, generator=g)
    noises = [torch.randn(1, 4, 4, 8, 8, generator=g) for _ in range(5)]

    def run():
        torch.manual_seed(5)                          # the posterior sample of the conditioning frame draws from the global CPU generator, like the reference
        return pipe(image=image.to(DEV), positive_prompt=["a prompt"], negative_prompt=None, height=64, width=64, num_frames=4, num_inference_steps=5, eta=1.0,
                    unconditional_guidance_scale=2.0, frame_stride=15, ref_videos=ref_videos.to(DEV), x_T=x_T, noises=noises)
    a, b = run(), run()
    assert a.shape == (1, 4, 3, 64, 64) and torch.isfinite(a.float()).all() and torch.equal(a, b)


def test_cogvideox_ct_pipeline_all_native_components(hip):
    """CogVideoXImageToVideoCTPipeline with the DiT, the T5 prompt encoder (`t5`) and the 3-D causal VAE (`cogvideox_vae`, image encode + decode) on the HIP path:
    the call equals the same components driven by hand (prompt ids -> T5, image -> posterior sample -> 0.7 z, hot loop, 1 / 0.7 -> decode -> [0, 1])."""
    from motionrag_amd import cogvideox as cvx, t5
    from motionrag_amd.cogvideox_vae import AutoencoderKLCogVideoX
    cfg, sd, dit = _small_dit()
    torch.manual_seed(2)
    vae = AutoencoderKLCogVideoX(block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=8, sample_height=64, sample_width=96).to(DEV, torch.bfloat16)
    text = t5.T5EncoderModel(vocab_size=128, d_model=64, d_ff=128, num_layers=2, num_heads=2).to(DEV, torch.bfloat16)

    class Tok:
        def __call__(self, texts, max_length=226, **_):
            ids = torch.zeros(len(texts), 10, dtype=torch.long)
            for r, t in enumerate(texts):
                w = [1 + ord(c) % 120 for c in t][:9]
                ids[r, :len(w)] = torch.tensor(w)
            return type("Enc", (), {"input_ids": ids})()
    cama = StubCAMA()
    pipe = cvx.CogVideoXImageToVideoCTPipeline(tokenizer=Tok(), text_encoder=text, vae=vae, transformer=dit, scheduler=cvx.CogVideoXDDIMScheduler(),
                                               condition_transformer=cama)
    image, ref_videos, metadata = _pipe_inputs()
    kw = dict(num_frames=9, num_inference_steps=2, guidance_scale=6.0, height=64, width=96)
    out = pipe(prompt=["a dog runs"], image=image / 2 + 0.5, negative_prompt=["blurry"], output_type="pt", ref_videos=ref_videos, metadata=metadata,
               generator=torch.Generator().manual_seed(3), **kw).frames
    assert out.shape == (1, 9, 3, 64, 96) and torch.isfinite(out.float()).all() and 0.0 <= out.min().item() and out.max().item() <= 1.0
    lat = pipe(prompt=["a dog runs"], image=image / 2 + 0.5, negative_prompt=["blurry"], output_type="latent", ref_videos=ref_videos, metadata=metadata,
               generator=torch.Generator().manual_seed(3), **kw).frames
    by_hand = vae.decode(lat.permute(0, 2, 1, 3, 4).float() / 0.7).sample                         # [1, 3, 9, 64, 96]
    want = (by_hand.permute(0, 2, 1, 3, 4).float() / 2 + 0.5).clamp(0, 1)
    assert ((out.float() - want.float().to(out.device)).norm() / want.float().norm()).item() < 1e-2
    # the image latents the loop was conditioned on: posterior sample with the generator's draw, times the scaling factor, zero-padded in time
    g = torch.Generator().manual_seed(3)
    il = pipe.encode_image_latents(image / 2 + 0.5, 3, generator=g)
    post = vae.encode((image / 2 + 0.5).float().mul(2).sub(1).unsqueeze(2)).latent_dist
    g2 = torch.Generator().manual_seed(3)
    want_il = post.sample(g2) * 0.7
    assert il.shape == (1, 3, 8, 8, 12) and torch.equal(il[:, 0], want_il[:, :, 0]) and il[:, 1:].abs().max().item() == 0


def test_cogvideox_baseline_pipeline_without_motion_injection(hip):
    """diffusers' plain CogVideoXImageToVideoPipeline as the reference's baseline module calls it (cogvideox/module.py:55-79, configs/cogvideox/baseline_open.yml):
    no ref_videos, plain joint attention; its latents equal the oracle DiT + DDIM loop without the adapter branch, and differ from the motion-injected run"""
    from motionrag_amd import cogvideox as cvx
    from oracle import cogvideox_ref
    from test_gpu_models import _bf_round
    cfg, sd, dit = _small_dit()
    pipe = cvx.CogVideoXImageToVideoPipeline(tokenizer=None, text_encoder=StubText(), vae=StubVAE(8), transformer=dit, scheduler=cvx.make_scheduler("ddim"))
    image, ref_videos, metadata = _pipe_inputs()
    kw = dict(num_frames=9, num_inference_steps=2, guidance_scale=6.0, height=64, width=96)
    vid = cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, None, None, generator=torch.Generator().manual_seed(3), **kw)
    assert vid.shape == (1, 9, 3, 64, 96) and torch.isfinite(vid.float()).all()
    # the YAML's `scheduler` entry selects the BASELINE pipe's scheduler (module.py:28-35); absent = 'ddim'
    cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, None, None, generator=torch.Generator().manual_seed(3), scheduler="dpm", **kw)
    assert isinstance(pipe.scheduler, cvx.CogVideoXDPMScheduler)
    cvx.eval_pipeline(pipe, image, ["a dog runs"], ["blurry"], torch.bfloat16, None, None, generator=torch.Generator().manual_seed(3), **kw)
    assert isinstance(pipe.scheduler, cvx.CogVideoXDDIMScheduler)
    lat = pipe(prompt=["a dog runs"], image=image / 2 + 0.5, negative_prompt=["blurry"], output_type="latent", generator=torch.Generator().manual_seed(3), **kw).frames
    # by hand with the oracle: same noise, embeddings, image latents; ip_hidden = None
    noise = torch.randn(1, 3, 8, 8, 12, generator=torch.Generator().manual_seed(3), dtype=torch.bfloat16)
    te = StubText()
    text = torch.cat([te(["blurry"]), te(["a dog runs"])]).float().cpu()
    il = pipe.encode_image_latents(image / 2 + 0.5, 3).to(torch.bfloat16).float().cpu()
    ac = cogvideox_ref.ddim_alphas_cumprod()
    cos, sin = cogvideox_ref.rope_3d(64, 3, 4, 6)
    x = noise.float()
    sdr = _bf_round(sd)
    for t in cogvideox_ref.ddim_timesteps(2):
        inp = torch.cat([torch.cat([x] * 2), torch.cat([il] * 2)], dim=2)
        v = cogvideox_ref.dit_forward(sdr, cfg, inp, text, torch.full((2,), float(t)), (cos, sin), None, ip_scale=0.0)
        x = cogvideox_ref.cfg_ddim_step(v, x, 6.0, cogvideox_ref.ddim_coeffs(ac, int(t), 2)).to(torch.bfloat16).float()
    close(lat, x, rel_l2=4e-2, atol_frac=0.15)          # two chained bf16 CFG steps against fp32; one element of 2 304 reaches 0.137 of the mean magnitude with the bf16-drawn noise
    inj = cvx.CogVideoXImageToVideoCTPipeline(tokenizer=None, text_encoder=StubText(), vae=StubVAE(8), transformer=dit, scheduler=cvx.make_scheduler("ddim"),
                                              condition_transformer=StubCAMA())
    lat_inj = inj(prompt=["a dog runs"], image=image / 2 + 0.5, negative_prompt=["blurry"], output_type="latent", ref_videos=ref_videos, metadata=metadata,
                  generator=torch.Generator().manual_seed(3), **kw).frames
    assert not torch.equal(lat_inj, lat)


def test_svd_baseline_pipeline_without_motion_injection(hip):
    """diffusers' plain StableVideoDiffusionPipeline as the reference's baseline module runs it (configs/svd/baseline_open.yml): image embedding only, plain
    attention processors -- two Euler / CFG steps against the oracle UNet without motion tokens"""
    from motionrag_amd import svd, svd_unet
    from oracle import svd_ref
    cfg = dict(in_channels=8, out_channels=4, block_out_channels=(64, 128), addition_time_embed_dim=64, projection_class_embeddings_input_dim=192,
               layers_per_block=1, cross_attention_dim=64, num_attention_heads=(1, 2))
    unet = svd_unet.UNetSpatioTemporalConditionModel(**cfg)
    g0 = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for k, p in unet.named_parameters():
            if k.endswith("mix_factor"):
                p.copy_(torch.randn(p.shape, generator=g0))
            elif p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g0) if "norm" in k and k.endswith("weight") else 0.05 * torch.randn(p.shape, generator=g0))
            else:
                p.copy_(torch.randn(p.shape, generator=g0) * (0.7 / p[0].numel() ** 0.5))
    unet = unet.to(torch.bfloat16)
    sdict = {k: v.float() for k, v in unet.state_dict().items()}
    unet = unet.to(DEV)
    b, Fr, h, w = 1, 4, 16, 16

    class ImgEnc:
        def __call__(self, x):
            return torch.tanh(x.float().mean(dim=(2, 3)).repeat(1, 22)[:, :64]).to(torch.bfloat16)

    class VAE:
        class Cfg:
            scaling_factor = 0.18215
        config = Cfg()

        def encode(self, x):
            return torch.nn.functional.avg_pool2d(x.float(), 8).repeat(1, 2, 1, 1)[:, :4]

        def decode(self, z, num_frames=None):
            return torch.tanh(torch.nn.functional.interpolate(z[:, :3].float(), scale_factor=8, mode="nearest"))
    pipe = svd.StableVideoDiffusionPipeline(vae=VAE(), image_encoder=ImgEnc(), unet=unet, scheduler=svd_unet.EulerDiscreteScheduler(), feature_extractor=None)
    g = torch.Generator().manual_seed(8)
    img255 = torch.rand(b, 3, 8 * h, 8 * w, generator=g) * 255.0
    kw = dict(height=8 * h, width=8 * w, num_frames=Fr, num_inference_steps=2, min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=127,
              noise_aug_strength=0.02)
    got = pipe(image=img255, output_type="latent", generator=torch.Generator().manual_seed(9), **kw).frames
    gen = torch.Generator().manual_seed(9)
    img = img255 / 127.5 - 1.0
    noise = torch.randn(img.shape, generator=gen, dtype=torch.bfloat16)                    # randn_tensor draws IN the pipeline dtype
    sig = svd_ref.karras_sigmas(2)
    lat = (torch.randn(b, Fr, 4, h, w, generator=gen, dtype=torch.bfloat16) * float((sig[0] ** 2 + 1) ** 0.5)).float()
    emb = ImgEnc()(img).float().unsqueeze(1)
    emb2 = torch.cat([torch.zeros_like(emb), emb])
    z = VAE().encode((img.to(torch.bfloat16) + 0.02 * noise)).to(torch.bfloat16).float()   # the noise-augmented image is formed in bf16
    il = torch.cat([torch.zeros_like(z), z])[:, None].expand(-1, Fr, -1, -1, -1)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    gs = torch.linspace(1.0, 3.0, Fr)
    for i in range(2):
        s, sn = float(sig[i]), float(sig[i + 1])
        scaled = (lat / (s * s + 1) ** 0.5).to(torch.bfloat16).float()
        x = torch.cat([torch.cat([scaled, scaled]), il], dim=2)
        v = svd_ref.unet_forward(sdict, cfg, x, torch.tensor(0.25 * np.log(s)), emb2, ids, None)
        lat = svd_ref.euler_cfg_step(v[:b].double(), v[b:].double(), lat.double(), s, sn, gs.double()).float().to(torch.bfloat16).float()
    close_chained_cfg(got, lat)                              # as the CT test: two chained bf16 CFG steps
    vid = svd.eval_pipeline(pipe, img255 / 127.5 - 1.0, generator=torch.Generator().manual_seed(9), **kw)
    assert vid.shape == (b, Fr, 3, 8 * h, 8 * w) and -1.0 <= vid.min().item() and vid.max().item() <= 1.0
    u8 = (torch.clip((img255 / 127.5 - 1.0 + 1.0) / 2.0, 0.0, 1.0) * 255).to(torch.uint8)   # the reference's tensor2PIL quantisation (utils/pipeline.py:178-184)
    want = pipe(image=u8, output_type="pt", generator=torch.Generator().manual_seed(9), **kw).frames[:, :16] * 2 - 1
    assert torch.equal(vid, want)
