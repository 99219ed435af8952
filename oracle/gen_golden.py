"""ORACLE TOOLING (test infrastructure; runs only in the build container, never on the GPU box).

Generates tests/golden/*.npz by IMPORTING the reference's own Python code from /root/reference and
running it on seeded inputs.  The reference sources never travel: only inputs' seeds and the
reference's OUTPUTS are stored.  Weights are re-generated from seeds by oracle/cama_ref.py's
random_*_sd helpers (torch CPU generator), so fixtures stay small.

    python -m oracle.gen_golden            # rewrites tests/golden/

What is pinned here:
  G1  Resampler.forward                      src/projects/condition/encoders/resampler.py:108-174
  G2  SinusoidPositionalEmbeddings table     src/projects/condition/position_embeddings.py:149-174
  G3  ConditionTransformer.get_mask          src/projects/condition/module.py:131-135
  G4  condition_fusion (4 modes)             src/projects/condition/utils.py:7-36
  G5  ActionTransformer.predict (stub feature encoders; CFG)   src/projects/condition/module.py:255-331
  G7  DynamiCrafter CrossAttention.efficient_forward (self / text+image+action: the `to_q_a(out)`
      adapter arithmetic shared with attn_processor.py)   .../lvdm/modules/attention.py:171-223
Not pinnable (third-party code absent from /root/reference): diffusers (CogVideoX / SVD blocks,
schedulers), lancedb (retrieval).  See oracle/cogvideox_ref.py and oracle/topk_oracle.c headers.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _Anything:
    """permissive stand-in for any class / function of an absent third-party package"""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def _stub(name: str, **attrs):
    import importlib.machinery
    m = _StubModule(name)
    m.__path__ = []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    import transformers  # noqa: F401  (real; must be imported before torchvision is stubbed)

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

    for n in ["cv2", "kornia", "kornia.augmentation", "open_clip", "torchvision", "torchvision.transforms", "torchvision.transforms.v2",
              "torchvision.transforms.v2.functional", "torchvision.transforms.functional", "torchvision.utils", "torchvision.io",
              "lightning", "lightning.pytorch.utilities", "lightning.pytorch.utilities.types", "lightning.pytorch.cli",
              "lightning.pytorch.loggers", "lightning.pytorch.callbacks", "diffusers", "diffusers.models", "diffusers.models.lora",
              "diffusers.models.attention_processor", "diffusers.models.embeddings", "diffusers.utils", "timm", "timm.models",
              "timm.models.layers", "omegaconf", "decord"]:
        _stub(n)
    _stub("lightning.pytorch", LightningModule=LightningModule)
    sys.modules["lightning"].pytorch = sys.modules["lightning.pytorch"]
    if REF not in sys.path:
        sys.path.insert(0, REF)


def _load_file(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FeatureStub(nn.Module):
    """deterministic stand-in for the frozen VideoMAE / DINOv2 encoders: features depend only on the
    per-sample mean of the input, so that `zeros` (the CFG unconditional clip) maps to a fixed feature."""

    def __init__(self, tokens: int, dim: int, seed: int):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("base", torch.randn(tokens, dim, generator=g))
        self.register_buffer("dirn", torch.randn(tokens, dim, generator=g))
        self.dim = dim

    def forward(self, x):
        m = x.reshape(x.shape[0], -1).float().mean(dim=1)
        return self.base[None] + m[:, None, None] * self.dirn[None]


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.manual_seed(0)
    from oracle import cama_ref

    cond = f"{REF}/src/projects/condition"
    res_mod = _load_file("ref_resampler", f"{cond}/encoders/resampler.py")
    pe_mod = _load_file("ref_pe", f"{cond}/position_embeddings.py")
    utils_mod = _load_file("ref_utils", f"{cond}/utils.py")

    # ---- G1 Resampler (full width, depth 4, 12 heads, 25 queries) ----
    g = torch.Generator().manual_seed(101)
    sd = cama_ref.random_resampler_sd(g, embedding_dim=768)
    ref = res_mod.Resampler(dim=1024, depth=4, dim_head=64, heads=12, num_queries=25, embedding_dim=768, output_dim=1024)
    ref.load_state_dict(sd, strict=True)
    x = torch.randn(2, 40, 768, generator=torch.Generator().manual_seed(102))
    with torch.no_grad():
        y = ref(x)
    np.savez(os.path.join(OUT, "resampler.npz"), weight_seed=101, input_seed=102, input_shape=np.array([2, 40, 768]), out=y.numpy())

    # ---- G2 sinusoid tables ----
    t256 = pe_mod.SinusoidPositionalEmbeddings(1024, 256).pos_table[0]
    t2560 = pe_mod.SinusoidPositionalEmbeddings(1024, 2560).pos_table[0]
    rows = np.array([0, 1, 2, 24, 25, 249, 255])
    np.savez(os.path.join(OUT, "sinusoid.npz"), rows=rows, t256=t256[rows].numpy(), t2560_first25=t2560[:25].numpy()[:, ::16],
             pe_applied=pe_mod.SinusoidPositionalEmbeddings(64, 32)(torch.ones(2, 5, 64)).numpy())

    # ---- G3 block-causal mask (via the reference ActionTransformer class, stub harness) ----
    module = importlib.import_module("src.projects.condition.module")
    mk = module.ConditionTransformer.get_mask
    holder = types.SimpleNamespace(device=torch.device("cpu"))
    np.savez(os.path.join(OUT, "mask.npz"), m4x3=mk(holder, 4, 3).numpy(), m10x25=np.packbits(mk(holder, 10, 25).numpy()))

    # ---- G4 condition_fusion ----
    emb = torch.randn(2, 3, 4, 8, generator=torch.Generator().manual_seed(103))
    dist = [[0.1, 0.3, 0.5], [0.2, 0.25, 0.9]]
    np.savez(os.path.join(OUT, "fusion.npz"), emb=emb.numpy(), dist=np.array(dist, dtype=np.float32),
             mean=utils_mod.condition_fusion(emb, "mean").numpy(), weight=utils_mod.condition_fusion(emb, "weight", dist).numpy(),
             concat=utils_mod.condition_fusion(emb, "concat").numpy(), top1=utils_mod.condition_fusion(emb, "top1").numpy())

    # ---- G5 ActionTransformer.predict (k = 3 refs, b = 2, CFG), reference classes end to end ----
    sdc = cama_ref.random_cama_sd(seed=104)
    vis, con = FeatureStub(48, 768, 105), FeatureStub(33, 1024, 106)
    layer = nn.TransformerEncoderLayer(d_model=1024, nhead=16, dim_feedforward=4096, dropout=0.0, activation="gelu", batch_first=True)
    at = module.ActionTransformer(
        condition_model=con, vision_model=vis,
        condition_proj=res_mod.Resampler(dim=1024, depth=4, dim_head=64, heads=12, num_queries=25, embedding_dim=1024, output_dim=1024),
        vision_proj=res_mod.Resampler(dim=1024, depth=4, dim_head=64, heads=12, num_queries=25, embedding_dim=768, output_dim=1024),
        transformer=nn.TransformerEncoder(layer, num_layers=4),
        condition_pe=pe_mod.SinusoidPositionalEmbeddings(1024, 2560), vision_pe=pe_mod.SinusoidPositionalEmbeddings(1024, 256))
    missing, unexpected = at.load_state_dict(sdc, strict=False)
    assert not unexpected and all(k.startswith(("vision_model", "condition_model")) for k in missing), (missing, unexpected)
    at.eval()
    gi = torch.Generator().manual_seed(107)
    batch = {"ref_videos": torch.randn(2, 3, 4, 3, 8, 8, generator=gi), "video": torch.randn(2, 4, 3, 8, 8, generator=gi)}
    with torch.no_grad():
        out = at.predict(batch, do_classifier_free_guidance=True)
        fwd = at.batch_forward(batch, return_loss=False)
    np.savez(os.path.join(OUT, "cama_predict.npz"), weight_seed=104, vis_seed=105, con_seed=106, input_seed=107,
             ref_videos_shape=np.array([2, 3, 4, 3, 8, 8]), vis_tokens=48, con_tokens=33, predict=out.numpy(), forward_last=fwd[:, -1].numpy(),
             forward_first=fwd[:, 0].numpy())

    # ---- G7 DynamiCrafter CrossAttention.efficient_forward ----
    # lvdm/basics.py:11 imports `..utils.utils` (needs cv2: stubbed); load it under a synthetic package root so the
    # relative imports resolve without executing src/projects/dynamicrafter/__init__.py
    dc = f"{REF}/src/projects/dynamicrafter/DynamiCrafter"
    pkg = types.ModuleType("dcroot"); pkg.__path__ = [dc]; sys.modules["dcroot"] = pkg
    attn_mod = importlib.import_module("dcroot.lvdm.modules.attention")
    torch.manual_seed(108)
    ca = attn_mod.CrossAttention(query_dim=128, context_dim=96, heads=2, dim_head=64, image_cross_attention=True,
                                 image_cross_attention_scale=0.7, action_cross_attention=True, action_cross_attention_scale=1.0)
    sa = attn_mod.CrossAttention(query_dim=128, heads=2, dim_head=64)
    gi = torch.Generator().manual_seed(109)
    xq = torch.randn(3, 50, 128, generator=gi)
    ctx = {"prompt": torch.randn(3, 77, 96, generator=gi), "image": torch.randn(3, 16, 96, generator=gi),
           "action": torch.randn(3, 25, 96, generator=gi)}
    with torch.no_grad():
        y_cross = ca.efficient_forward(xq, ctx)
        y_self = sa.efficient_forward(xq)
    blob = {f"ca.{k}": v.numpy() for k, v in ca.state_dict().items()}
    blob.update({f"sa.{k}": v.numpy() for k, v in sa.state_dict().items()})
    np.savez(os.path.join(OUT, "dc_cross_attention.npz"), x=xq.numpy(), prompt=ctx["prompt"].numpy(), image=ctx["image"].numpy(),
             action=ctx["action"].numpy(), y_cross=y_cross.numpy(), y_self=y_self.numpy(), **blob)
    # ---- G8-G12 DynamiCrafter UNet blocks, reduced-width UNetModel, schedule tables, DDIM steps ----
    unet_ref = gen_dynamicrafter(attn_mod)
    gen_dc_pipeline(unet_ref, res_mod)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.0f} KiB")


def seeded_state(module, seed: int, std: float = 0.08):
    """overwrite EVERY parameter (zero-initialised ones included: SURVEY App. D.6) from a seeded generator, in sorted key order;
    norm scales get 1 + noise.  tests regenerate the same values from (keys, shapes, seed)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    new = {}
    for k in sorted(sd):
        v = torch.randn(sd[k].shape, generator=g) * std
        if k.endswith("weight") and sd[k].dim() == 1:
            v = v + 1.0
        new[k] = v
    module.load_state_dict(new, strict=True)
    SEEDED_META[seed] = {"std": std, "keys": sorted(sd), "shapes": [list(sd[k].shape) for k in sorted(sd)]}
    return new


SEEDED_META = {}


def gen_dynamicrafter(attn_mod):
    import importlib
    net = importlib.import_module("dcroot.lvdm.modules.networks.openaimodel3d")
    ud = importlib.import_module("dcroot.lvdm.models.utils_diffusion")
    ddim_mod = importlib.import_module("dcroot.lvdm.models.samplers.ddim")
    gi = torch.Generator().manual_seed(201)
    r = lambda *s: torch.randn(*s, generator=gi)
    C, cd = 64, 96
    ctx = {"prompt": r(4, 7, cd), "image": r(4, 5, cd), "action": r(4, 25, cd)}
    blob = {}

    # G8 transformers (in_channels 64 -> 1 head x 64)
    st = attn_mod.SpatialTransformer(C, 1, 64, depth=1, context_dim=cd, use_linear=True, use_checkpoint=False, image_cross_attention=True,
                                     action_cross_attention=True).eval()
    seeded_state(st, 202)
    x4 = r(4, C, 6, 5)
    tt = attn_mod.TemporalTransformer(C, 2, 64, depth=1, context_dim=cd, use_linear=True, use_checkpoint=False, only_self_att=True,
                                      relative_position=False, temporal_length=4).eval()      # inner 128 != 64, like init_attn
    seeded_state(tt, 203)
    x5 = r(2, C, 4, 3, 5)
    with torch.no_grad():
        blob.update(st_x=x4.numpy(), st_y=st(x4, ctx).numpy(), tt_x=x5.numpy(), tt_y=tt(x5).numpy())
    blob.update({f"ctx_{k}": v.numpy() for k, v in ctx.items()})

    # G9 ResBlock (+ temporal conv, skip 1x1), Downsample, Upsample
    rb = net.ResBlock(C, 128, 0.0, out_channels=96, dims=2, use_temporal_conv=True).eval()
    seeded_state(rb, 204)
    rb2 = net.ResBlock(C, 128, 0.0, out_channels=C, dims=2, use_temporal_conv=False).eval()
    seeded_state(rb2, 205)
    dn, up = net.Downsample(C, True, dims=2, out_channels=C).eval(), net.Upsample(C, True, dims=2, out_channels=C).eval()
    seeded_state(dn, 206); seeded_state(up, 207)
    xr, emb = r(4, C, 6, 5), r(4, 128)
    with torch.no_grad():
        blob.update(rb_x=xr.numpy(), rb_emb=emb.numpy(), rb_y=rb(xr, emb, batch_size=2).numpy(), rb2_y=rb2(xr, emb, batch_size=2).numpy(),
                    dn_y=dn(xr).numpy(), up_y=up(xr).numpy())
    import json
    blob["meta"] = np.array(json.dumps({name: dict(seed=seed, **SEEDED_META[seed]) for name, seed in
                                        (("st", 202), ("tt", 203), ("rb", 204), ("rb2", 205), ("dn", 206), ("up", 207))}))
    np.savez(os.path.join(OUT, "dc_blocks.npz"), **blob)

    # G10 reduced-width UNetModel (every zero-init layer re-randomised)
    unet = net.UNetModel(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=(1, 2), num_res_blocks=1, channel_mult=(1, 2),
                         num_head_channels=64, transformer_depth=1, context_dim=64, use_linear=True, use_checkpoint=False, temporal_conv=True,
                         temporal_attention=True, temporal_self_att_only=True, use_relative_position=False, temporal_length=4,
                         addition_attention=True, image_cross_attention=True, action_cross_attention=True, default_fs=10, fs_condition=True).eval()
    sd = seeded_state(unet, 208, std=0.05)
    gi = torch.Generator().manual_seed(209)
    x = torch.randn(2, 8, 4, 8, 8, generator=gi)
    ctx = {"image": torch.randn(2, 4 * 3, 64, generator=gi), "prompt": torch.randn(2, 7, 64, generator=gi), "action": torch.randn(2, 25, 64, generator=gi)}
    ts, fs = torch.tensor([481, 34]), torch.tensor([15, 15])
    with torch.no_grad():
        y = unet(x, ts, context=ctx, fs=fs)
    keys = sorted(sd)
    np.savez(os.path.join(OUT, "dc_unet.npz"), seed=208, std=0.05, keys=np.array(keys), shapes=np.array([list(sd[k].shape) + [0] * (5 - sd[k].dim()) for k in keys]),
             ndims=np.array([sd[k].dim() for k in keys]), input_seed=209, timesteps=ts.numpy(), fs=fs.numpy(), y=y.numpy())

    # G11 schedule tables
    betas = ud.rescale_zero_terminal_snr(ud.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012))
    ac = np.cumprod(1.0 - betas)
    t30, t50 = ud.make_ddim_timesteps("uniform", 30, 1000, verbose=False), ud.make_ddim_timesteps("uniform", 50, 1000, verbose=False)
    sig, al, alp = ud.make_ddim_sampling_parameters(torch.tensor(ac, dtype=torch.float32), t30, 1.0, verbose=False)
    temb = ud.timestep_embedding(torch.tensor([0, 1, 481, 999]), 64)

    # G12 three stochastic DDIM steps (eta = 1, CFG 2.0, v-param, dynamic rescale) against a duck-typed model
    class Duck:
        num_timesteps, parameterization, use_dynamic_rescale, device = 1000, "v", True, torch.device("cpu")

        def __init__(self):
            self.alphas_cumprod_np = ac
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.sqrt_alphas_cumprod = torch.tensor(np.sqrt(ac), dtype=torch.float32)
            self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)
            self.scale_arr = torch.tensor(np.concatenate((np.linspace(1.0, 0.3, 400), np.full(1000, 0.3))), dtype=torch.float32)

        def apply_model(self, x, t, c, **kw):
            return 0.5 * x + c["shift"] * torch.cos(t.float() / 100.0).view(-1, 1, 1, 1, 1)

        def predict_start_from_z_and_v(self, x, t, v):
            return self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * x - self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * v

        def predict_eps_from_z_and_v(self, x, t, v):
            return self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * v + self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * x

    ddim_mod.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)      # the reference forces .to("cuda")
    smp = ddim_mod.DDIMSampler(Duck())
    smp.make_schedule(30, ddim_eta=1.0, verbose=False)
    gi = torch.Generator().manual_seed(210)
    xT = torch.randn(2, 4, 4, 8, 8, generator=gi)
    c, uc = {"shift": torch.randn(2, 1, 1, 1, 1, generator=gi)}, {"shift": torch.randn(2, 1, 1, 1, 1, generator=gi)}
    xs, noises, x = [], [], xT
    steps = np.flip(smp.ddim_timesteps)
    for i in range(3):
        index = len(steps) - i - 1
        torch.manual_seed(300 + i)
        noises.append(torch.randn(x.shape).numpy())
        torch.manual_seed(300 + i)
        tsx = torch.full((2,), int(steps[i]), dtype=torch.long)
        x, _ = smp.p_sample_ddim(x, c, tsx, index=index, unconditional_guidance_scale=2.0, unconditional_conditioning=uc)
        xs.append(x.numpy())
    gen_dynamicrafter.tables = (ac, betas)
    np.savez(os.path.join(OUT, "dc_schedule.npz"), alphas_cumprod=ac, t30=t30, t50=t50, sigmas=sig.numpy(), alphas=al.numpy(), alphas_prev=alp.numpy(),
             temb=temb.numpy(), xT=xT.numpy(), c_shift=c["shift"].numpy(), uc_shift=uc["shift"].numpy(), noises=np.stack(noises), xs=np.stack(xs),
             scale_arr=smp.model.scale_arr.numpy())
    return unet


def gen_dc_pipeline(unet, res_mod):
    """G14: the reference's OWN image_guided_synthesis / DynamiCrafterPipelineRef glue (inference.py:174-305, pipelines/pipeline.py:64-115)
    driven end to end on CPU: reduced-width reference UNetModel (the G10 weights), reference Resampler as image_proj_model, reference
    DDIMSampler, deterministic stand-ins (oracle/stubs.py) for the third-party encoders / VAE.  The torch.randn stream the sampler draws
    (x_T, then one eta-noise per step) is recorded so that the GPU test can replay it."""
    import importlib
    from oracle import stubs
    inf = importlib.import_module("dcroot.scripts.evaluation.inference")
    ddim_mod = importlib.import_module("dcroot.lvdm.models.samplers.ddim")
    ac, betas = gen_dynamicrafter.tables

    class Wrapper(nn.Module):                       # DiffusionWrapper, conditioning_key 'hybrid' (ddpm3d.py:1378-1382)
        conditioning_key = "hybrid"

        def __init__(self, dm):
            super().__init__()
            self.diffusion_model = dm

        def forward(self, x, t, c_concat=None, c_crossattn=None, **kwargs):
            return self.diffusion_model(torch.cat([x] + c_concat, dim=1), t, context=c_crossattn, **kwargs)

    class DuckLVD(nn.Module):                       # the attributes image_guided_synthesis and DDIMSampler read from LatentVisualDiffusion
        num_timesteps, parameterization, use_dynamic_rescale, uncond_type = 1000, "v", True, "empty_seq"
        action_embedder = None

        def __init__(self):
            super().__init__()
            self.model = Wrapper(unet)
            self.embedder = stubs.ImageEmbedderStub(tokens=9, dim=48)
            self.image_proj_model = res_mod.Resampler(dim=64, depth=2, dim_head=64, heads=2, num_queries=3, embedding_dim=48, output_dim=64, video_length=4).eval()
            seeded_state(self.image_proj_model, 404, std=0.08)
            self.condition_transformer = stubs.ConditionTransformerStub(dim=64)
            self.first_stage = stubs.FirstStageStub()
            self.text = stubs.TextStub(tokens=7, dim=64)
            self.alphas_cumprod_np = ac
            self.register_buffer("betas", torch.tensor(betas, dtype=torch.float32))
            self.register_buffer("alphas_cumprod", torch.tensor(ac, dtype=torch.float32))
            self.register_buffer("alphas_cumprod_prev", torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32))
            self.register_buffer("sqrt_alphas_cumprod", torch.tensor(np.sqrt(ac), dtype=torch.float32))
            self.register_buffer("sqrt_one_minus_alphas_cumprod", torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32))
            self.register_buffer("scale_arr", torch.tensor(np.concatenate((np.linspace(1.0, 0.3, 400), np.full(1000, 0.3))), dtype=torch.float32))

        device = torch.device("cpu")

        def get_learned_conditioning(self, prompts):
            return self.text(prompts)

        def encode_first_stage(self, x):
            return self.first_stage.encode_first_stage(x)

        def decode_first_stage(self, z):
            return self.first_stage.decode_first_stage(z)

        def apply_model(self, x_noisy, t, cond, **kwargs):       # ddpm3d.py:745-760 (dict branch)
            return self.model(x_noisy, t, **cond, **kwargs)

        def predict_start_from_z_and_v(self, x, t, v):           # ddpm3d.py:251-256
            return self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * x - self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * v

        def predict_eps_from_z_and_v(self, x, t, v):             # ddpm3d.py:258-263
            return self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * v + self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1, 1) * x

    ddim_mod.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)      # the reference forces .to("cuda")
    model = DuckLVD().eval()
    pipe_cls = type("Pipe", (), {})                  # pipelines/pipeline.py cannot be imported (pulls ddpm3d -> lightning/torchvision): call the glue directly
    gi = torch.Generator().manual_seed(405)
    b, T, H, W = 1, 4, 64, 64
    image = torch.rand(b, 3, H, W, generator=gi) * 2 - 1
    ref_videos = torch.rand(b, 3, T, 3, 16, 16, generator=gi) * 2 - 1
    prompts = ["a corgi running on the beach"]
    drawn = []
    real_randn = torch.randn

    def recording_randn(*a, **k):
        k.pop("device", None)
        v = real_randn(*a, **k)
        drawn.append(v.clone())
        return v

    torch.manual_seed(406)
    torch.randn = recording_randn
    try:
        with torch.no_grad():
            videos = image[:, :, None].expand(-1, -1, T, -1, -1)                         # DynamiCrafterPipelineRef.__call__ :95
            out = inf.image_guided_synthesis(model=model, prompts=prompts, videos=videos, noise_shape=[b, 4, T, H // 8, W // 8], n_samples=1, ddim_steps=5,
                                             ddim_eta=1.0, unconditional_guidance_scale=2.0, cfg_img=None, fs=15, text_input=True, multiple_cond_cfg=False,
                                             loop=False, interp=False, timestep_spacing="uniform", guidance_rescale=0.0, ref_videos=ref_videos,
                                             ref_fusion_type=None, metadata=None)
    finally:
        torch.randn = real_randn
    frames = out[:, 0].permute(0, 2, 1, 3, 4)                                            # 'b 1 c t h w -> b t c h w'  :115
    shape5 = [d for d in drawn if tuple(d.shape) == (b, 4, T, H // 8, W // 8)]
    assert len(shape5) == 6, [tuple(d.shape) for d in drawn]                             # x_T + one noise per DDIM step (5 steps)
    import json
    np.savez(os.path.join(OUT, "dc_pipeline.npz"), image=image.numpy(), ref_videos=ref_videos.numpy(), prompt=np.array(prompts[0]), frames=frames.numpy(),
             x_T=shape5[0].numpy(), noises=np.stack([d.numpy() for d in shape5[1:]]),
             proj_meta=np.array(json.dumps(dict(seed=404, **SEEDED_META[404]))))


if __name__ == "__main__":
    main()
