"""ORACLE TOOLING (test infrastructure; runs only in the build container, never on the GPU box).

Pins the motion-injection boundary to the REFERENCE'S OWN CODE: imports

  * /root/reference/src/projects/condition/attn_processor.py   (`APAdapterCogVideoXAttnProcessor2_0.__call__` :176-283,
                                                                 `APAdapterAttnProcessor2_0.__call__` :18-141)
  * /root/reference/src/projects/svd/pipelines/pipeline.py      (`TupleTensor` :25-57, `SVDActionPipeline.__call__` / `_encode_image` :93-119,
                                                                 `SVDCTPipeline.__call__` :147-160)
  * /root/reference/src/projects/cogvideox/pipeline.py          (`_prepare_rotary_positional_embeddings` :46-57,
                                                                 `prepare_action_embeddings` :59-78 and :117-130, `__call__` :80-89)

under a stub of the absent third-party package `diffusers==0.32.2`, runs them on seeded CPU inputs and writes

    tests/golden/cog_attn_processor.npz   tests/golden/svd_attn_processor.npz   tests/golden/adapter_pipelines.npz

    python -m oracle.gen_golden_attn_processor

Only inputs and the reference's OUTPUTS are stored; neither the reference sources nor this stub travel to the GPU box.

What the stub supplies (everything else that executes is the reference's own text):
  * `diffusers.models.attention_processor.Attention` -- a name for a type annotation; `CogVideoXAttnProcessor2_0 = object`;
    `IPAdapterAttnProcessor2_0.__init__` building `to_k_ip` / `to_v_ip` ModuleLists and `self.scale` (what the reference's subclass
    `super().__init__` expects, attn_processor.py:11-16) -- constructor plumbing, no arithmetic;
  * `diffusers.models.embeddings.apply_rotary_emb` -- THE ONE RESTATED PIECE OF ARITHMETIC (diffusers 0.32.2, `use_real=True,
    use_real_unbind_dim=-1`: `x * cos + stack([-x_imag, x_real]) * sin` in fp32); it is cross-checked against a complex rotation
    in tests/test_oracle_golden.py::test_rope_matches_complex_rotation;
  * `StableVideoDiffusionPipeline` / `CogVideoXImageToVideoPipeline` base classes that only RECORD what the reference subclasses hand
    them (`_encode_image`, `_prepare_rotary_positional_embeddings`, `__call__` arguments), `VideoProcessor`, `pil_to_tensor`
    (numpy HWC uint8 -> CHW tensor, torchvision's definition);
  * a duck-typed `attn` module with the attributes the processors read (`to_q/k/v`, `to_out`, `norm_q/k`, `heads`, `spatial_norm`, ...).
"""
from __future__ import annotations

import importlib
import json
import os
import sys

import numpy as np
import torch
from torch import nn

from . import gen_golden as gg

REF = gg.REF
OUT = gg.OUT


# ------------------------------------------------------------------------------------------------ the diffusers stub
class Attention(nn.Module):
    """duck-typed `diffusers.models.attention_processor.Attention`: the attributes the two processors read, nothing else"""

    def __init__(self, query_dim, cross_attention_dim=None, heads=2, dim_head=64, bias=False, out_bias=True, qk_norm=None, eps=1e-6,
                 residual_connection=False, rescale_output_factor=1.0):
        super().__init__()
        inner = heads * dim_head
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.is_cross_attention = heads, cross_attention_dim is not None
        self.spatial_norm = self.group_norm = None
        self.norm_cross = None
        self.residual_connection, self.rescale_output_factor = residual_connection, rescale_output_factor
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.norm_q = nn.LayerNorm(dim_head, eps=eps) if qk_norm == "layer_norm" else None
        self.norm_k = nn.LayerNorm(dim_head, eps=eps) if qk_norm == "layer_norm" else None


class IPAdapterAttnProcessor2_0(nn.Module):
    """constructor plumbing of diffusers' class (the reference's subclass calls it and adds `to_q_ip`)"""

    def __init__(self, hidden_size, cross_attention_dim=None, num_tokens=(4,), scale=1.0):
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        if not isinstance(num_tokens, (tuple, list)):
            num_tokens = [num_tokens]
        self.num_tokens = num_tokens
        if not isinstance(scale, list):
            scale = [scale] * len(num_tokens)
        self.scale = scale
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in num_tokens])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in num_tokens])


def apply_rotary_emb(x, freqs_cis):
    """diffusers 0.32.2 `apply_rotary_emb(x, (cos, sin))`, use_real=True, use_real_unbind_dim=-1 (restated; see the module docstring)"""
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rotated.float() * sin).to(x.dtype)


class _RecordingPipeline:
    """base of the two diffusers pipelines: keeps the registered modules and records what the subclass hands to the hooks"""

    def __init__(self, **modules):
        self.register_modules(**modules)
        self.vae_scale_factor = 8

    def register_modules(self, **modules):
        for k, v in modules.items():
            setattr(self, k, v)


class StableVideoDiffusionPipeline(_RecordingPipeline):
    def _encode_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance):
        emb = self.image_encoder(image)                                   # [b, 1, D]
        return torch.cat([torch.zeros_like(emb), emb]) if do_classifier_free_guidance else emb

    def __call__(self, *args, **kwargs):
        self.base_call = (args, kwargs)
        # what diffusers does next with the hook's result: `.to`, `.repeat_interleave(num_frames, dim=0)`, `[idx]`, `.shape`, `.dtype`
        return self._encode_image(kwargs["image_for_clip"], "cpu", 1, True)


class CogVideoXImageToVideoPipeline(_RecordingPipeline):
    def _prepare_rotary_positional_embeddings(self, height, width, num_frames, device):
        return ("cos-table", "sin-table")                                  # placeholders: the override must pass them through untouched

    def __call__(self, *args, **kwargs):
        self.base_call = (args, kwargs)
        return self._prepare_rotary_positional_embeddings(480, 720, 13, "cpu")


class VideoProcessor:
    def __init__(self, *a, **k):
        pass


def pil_to_tensor(img):
    return torch.from_numpy(np.asarray(img)).permute(2, 0, 1).contiguous()


def install():
    gg.install_stubs()
    ap = sys.modules["diffusers.models.attention_processor"]
    ap.Attention, ap.IPAdapterAttnProcessor2_0, ap.CogVideoXAttnProcessor2_0 = Attention, IPAdapterAttnProcessor2_0, object
    sys.modules["diffusers.models.embeddings"].apply_rotary_emb = apply_rotary_emb
    d = sys.modules["diffusers"]
    d.StableVideoDiffusionPipeline, d.CogVideoXImageToVideoPipeline = StableVideoDiffusionPipeline, CogVideoXImageToVideoPipeline
    gg._stub("diffusers.video_processor", VideoProcessor=VideoProcessor)
    sys.modules["torchvision.transforms.v2.functional"].pil_to_tensor = pil_to_tensor
    sys.modules["torchvision.transforms.functional"].pil_to_tensor = pil_to_tensor
    gg._stub("src.utils.pipeline", tensor2PIL=None)                        # imported by svd/pipelines/pipeline.py:13, unused by the pinned lines


# ------------------------------------------------------------------------------------------------ seeded fixtures
def _randomise(module: nn.Module, seed: int, std: float):
    g = torch.Generator().manual_seed(seed)
    for n, p in module.named_parameters():
        with torch.no_grad():
            p.copy_(torch.randn(p.shape, generator=g) * std)
            if n.endswith(("norm_q.weight", "norm_k.weight")):
                p.add_(1.0)
            p.copy_(p.to(torch.bfloat16).float())                          # bf16-representable: the GPU tests load the very same values


def _bf(*ts):
    """inputs rounded to bf16-representable fp32 values (the reference then runs in fp32 on exactly what the bf16 product path reads)"""
    return [t.to(torch.bfloat16).float() for t in ts]


def _sd(attn, proc):
    sd = {f"attn.{k}": v.detach().numpy() for k, v in attn.state_dict().items()}
    sd.update({f"proc.{k}": v.detach().numpy() for k, v in proc.state_dict().items()})
    return sd


def rope_3d(head_dim, t, h, w):
    from .cogvideox_ref import rope_3d as r                                # restated table builder: an INPUT of these fixtures, stored in the npz
    return r(head_dim, t, h, w)


def gen_cog(apm):
    D, H, ipd, text_len, (t, h, w) = 128, 2, 96, 6, (2, 3, 5)
    attn = Attention(D, heads=H, dim_head=64, bias=True, out_bias=True, qk_norm="layer_norm", eps=1e-6)
    proc = apm.APAdapterCogVideoXAttnProcessor2_0(D, ipd)
    _randomise(attn, 501, 0.15)
    _randomise(proc, 502, 0.15)
    g = torch.Generator().manual_seed(503)
    hidden = torch.randn(2, t * h * w, D, generator=g)
    enc = torch.randn(2, text_len, D, generator=g)
    ip2 = torch.randn(2, 25, ipd, generator=g)
    ip1 = torch.randn(1, 25, ipd, generator=g)
    hidden, enc, ip2, ip1 = _bf(hidden, enc, ip2, ip1)
    cos, sin = rope_3d(64, t, h, w)
    out = {"hidden": hidden, "enc": enc, "ip2": ip2, "ip1": ip1, "cos": cos, "sin": sin}
    cases = {
        # name: (image_rotary_emb, action_hidden_states kwarg, scale)
        "rope_tuple": (((cos, sin), ip2), None, 1.0),                      # the shipped path: ((cos, sin), ip) smuggled through the rope hook :189-190
        "rope_tuple_repeat": (((cos, sin), ip1), None, 1.0),               # B' = 1, B = 2: the '(b r)' repeat :254-256
        "norope_kwarg": (None, ip2, 1.0),                                  # no rotary table, tokens through the keyword :192-193
        "rope_list_kwarg": ([cos, sin], ip2, 1.0),                         # a non-tuple rope + keyword (a plain (cos, sin) TUPLE would be unpacked as (rope, ip) by :189)
        "scale_half": (((cos, sin), ip2), None, 0.5),
        "scale_zero": (((cos, sin), ip2), None, 0.0),                      # skip :243-249
    }
    with torch.no_grad():
        for name, (rope, kw, scale) in cases.items():
            proc.scale = [scale]
            oh, oe = proc(attn, hidden.clone(), enc.clone(), action_hidden_states=kw, image_rotary_emb=rope)
            out[f"{name}.h"], out[f"{name}.e"] = oh, oe
        # the reference's unpack quirk (:189): a plain (cos, sin) tuple IS "a tuple whose [1] is a tensor" -> sin becomes the motion tokens.
        # Recorded as the behaviour the product deliberately does not reproduce.
        try:
            proc(attn, hidden.clone(), enc.clone(), action_hidden_states=ip2, image_rotary_emb=(cos, sin))
            quirk = "ran"
        except Exception as e:                                             # sin [30, 64] through to_k_ip(96 -> 128): shape error
            quirk = type(e).__name__
    meta = {"D": D, "H": H, "ip_dim": ipd, "text_len": text_len, "thw": [t, h, w], "cases": {k: {"scale": v[2]} for k, v in cases.items()},
            "plain_cos_sin_tuple_with_kwarg": quirk}
    np.savez(os.path.join(OUT, "cog_attn_processor.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()}, **_sd(attn, proc))
    return meta


def gen_svd(apm, svd_pipe):
    C, H, cd, F = 192, 3, 96, 3
    attn = Attention(C, cross_attention_dim=cd, heads=H, dim_head=64, bias=False, out_bias=True)
    proc = apm.APAdapterAttnProcessor2_0(C, cd)
    _randomise(attn, 511, 0.05)
    _randomise(proc, 512, 0.05)
    g = torch.Generator().manual_seed(513)
    hidden = torch.randn(2 * F, 36, C, generator=g)                        # [(b f), hw, c]
    hidden4 = torch.randn(2 * F, C, 4, 9, generator=g)                     # the 4-D form :48-50
    img = torch.randn(2 * F, 1, cd, generator=g)                           # CLIP image embedding per frame
    img2 = torch.randn(2, 1, cd, generator=g)                              # before diffusers' repeat_interleave
    act = torch.randn(2, 25, cd, generator=g)                              # motion tokens [2b, 25, cd]: r = F :109-111
    img3 = torch.randn(2 * F, 3, cd, generator=g)
    hidden, hidden4, img, img2, act, img3 = _bf(hidden, hidden4, img, img2, act, img3)
    out = {"hidden": hidden, "hidden4": hidden4, "img": img, "img2": img2, "img3": img3, "act": act}
    TT = svd_pipe.TupleTensor
    with torch.no_grad():
        def run(name, h, ehs, **kw):
            out["out." + name] = proc(attn, h.clone(), ehs, **kw)
        run("tuple", hidden, (img, act))
        run("tuple_img3", hidden, (img3, act))                             # 3 image tokens: the result depends on the queries (1 token: softmax == 1)
        tt = TT([img2, act]).to(torch.float32).repeat_interleave(F, dim=0)  # what diffusers' UNet does to encoder_hidden_states
        assert isinstance(tt, TT) and tt.shape == (2 * F, 1, cd) and tuple.__getitem__(tt, 1).shape == (2 * F, 25, cd)
        run("tupletensor", hidden, tt)                                     # tokens already repeated per frame: r = 1
        run("kwarg", hidden, img, action_hidden_states=act)
        run("hidden4", hidden4, (img3, act))
        attn.residual_connection = True
        run("resid", hidden, (img, act))
        run("hidden4_resid", hidden4, (img3, act))
        attn.residual_connection = False
        proc.scale = [0.0]
        run("scale_zero", hidden, (img, act))
        proc.scale = [0.6]
        run("scale_06", hidden, (img, act))
        proc.scale = [1.0]
        attn.rescale_output_factor = 2.0
        run("rescale2", hidden, (img, act))
        attn.rescale_output_factor = 1.0
    # TupleTensor's own contract (pipeline.py:25-57)
    t0 = TT([img2, act])
    tt_contract = {"getitem_is_first": bool(torch.equal(t0[1], img2[1])), "shape": list(t0.shape), "size0": int(t0.size(0)), "dtype": str(t0.dtype),
                   "to_tuple_len": len(t0.to_tuple()), "to_keeps_type": isinstance(t0.to(torch.float64), TT),
                   "repeat_shapes": [list(x.shape) for x in t0.repeat_interleave(F, dim=0).to_tuple()], "is_tuple": isinstance(t0, tuple)}
    meta = {"C": C, "H": H, "cross_dim": cd, "F": F, "tuple_tensor": tt_contract}
    np.savez(os.path.join(OUT, "svd_attn_processor.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()}, **_sd(attn, proc))
    return meta


class ActionEmbedderStub(nn.Module):
    """frozen action embedder of the stage-1 pipelines: videos [n, f, c, h, w] -> tokens [n, t, c]; depends on the clip's mean and its mean
    absolute value so that `zeros` maps to a fixed embedding"""

    def __init__(self, tokens=5, dim=16, seed=521):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        for n in ("base", "d1", "d2"):
            self.register_buffer(n, torch.randn(tokens, dim, generator=g))

    def forward(self, v):
        f = v.reshape(v.shape[0], -1).float()
        return (self.base[None] + f.mean(1)[:, None, None] * self.d1[None] + f.abs().mean(1)[:, None, None] * self.d2[None]).to(v.dtype)


class CTRecorder(nn.Module):
    """`condition_transformer` stand-in: records the batch the pipeline builds and returns tokens that depend on it"""

    def __init__(self):
        super().__init__()
        from .stubs import ConditionTransformerStub
        self.inner = ConditionTransformerStub(dim=16, seed=523)

    def predict(self, batch, do_classifier_free_guidance=False):
        self.batch, self.cfg = batch, do_classifier_free_guidance
        y = self.inner.predict(batch)
        return torch.cat([torch.zeros_like(y), y]) if do_classifier_free_guidance else y


def gen_pipelines(cog_pipe, svd_pipe):
    g = torch.Generator().manual_seed(531)
    ref_videos = torch.randn(2, 3, 4, 3, 8, 8, generator=g)                # [b, k, f, c, h, w]
    dist = [[0.1, 0.3, 0.5], [0.2, 0.25, 0.9]]
    metadata = [{"ref_video_distance": d} for d in dist]
    image01 = torch.rand(2, 3, 8, 8, generator=g)                          # CogVideoX eval_pipeline hands image / 2 + 0.5
    image_u8 = (torch.rand(2, 8, 8, 3, generator=g) * 255).to(torch.uint8)  # the SVD pipeline gets PIL images: HWC uint8
    emb, proj = ActionEmbedderStub(), nn.Linear(16, 24)
    _randomise(proj, 522, 0.3)
    out = {"ref_videos": ref_videos, "dist": torch.tensor(dist), "image01": image01, "image_u8": image_u8,
           "proj.weight": proj.weight.detach(), "proj.bias": proj.bias.detach()}
    meta = {"embedder": {"tokens": 5, "dim": 16, "seed": 521}, "ct_seed": 523, "cog": {}, "svd": {}}
    with torch.no_grad():
        # ---- CogVideoX stage 1 (pipeline.py:59-78) ----
        for fusion in ("mean", "weight", "top1", "concat"):
            p = cog_pipe.CogVideoXImageToVideoActionPipeline(tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None,
                                                             action_embedder=emb, action_proj_model=proj, ref_fusion_type=fusion)
            out[f"cog.action.{fusion}.nocfg"] = p.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=False)
            try:
                out[f"cog.action.{fusion}.cfg"] = p.prepare_action_embeddings(ref_videos, metadata, do_classifier_free_guidance=True)
                meta["cog"][f"{fusion}.cfg"] = "ok"
            except RuntimeError:                                           # 'concat': [b, t, c] cannot be concatenated with [b, k t, c] :75
                meta["cog"][f"{fusion}.cfg"] = "RuntimeError"
        # __call__ (:80-89) + the rope hook (:46-57): action_emb set with CFG, hook returns (base tables, action_emb)
        p = cog_pipe.CogVideoXImageToVideoActionPipeline(tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None,
                                                         action_embedder=emb, action_proj_model=proj, ref_fusion_type="mean")
        hook = p(ref_videos=ref_videos, metadata=metadata, image=image01, prompt=["a"])
        assert hook[0] == ("cos-table", "sin-table") and hook[1] is p.action_emb and torch.equal(p.action_emb, out["cog.action.mean.cfg"])
        meta["cog"]["base_call_kwargs"] = sorted(p.base_call[1])           # ref_videos / metadata are NOT forwarded to the diffusers __call__
        # ---- CogVideoX stage 2 (pipeline.py:117-130) ----
        ct = CTRecorder()
        p = cog_pipe.CogVideoXImageToVideoCTPipeline(tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None, condition_transformer=ct)
        p(ref_videos=ref_videos, metadata=metadata, image=image01, prompt=["a"])
        out["cog.ct.video"], out["cog.ct.action_emb"] = ct.batch["video"], p.action_emb
        assert ct.cfg is True and ct.batch["ref_videos"] is ref_videos
        # ---- SVD stage 1 (pipeline.py:93-119) ----
        img_enc = nn.Linear(3 * 8 * 8, 12)
        _randomise(img_enc, 524, 0.1)
        out["img_enc.weight"], out["img_enc.bias"] = img_enc.weight.detach(), img_enc.bias.detach()
        clip = lambda x: img_enc(x.reshape(x.shape[0], -1))[:, None]       # noqa: E731
        for fusion in ("mean", "weight", "top1"):
            p = svd_pipe.SVDActionPipeline(vae=_Cfg(), image_encoder=clip, unet=None, scheduler=None, feature_extractor=None,
                                           action_embedder=emb, action_proj_model=proj, ref_fusion_type=fusion)
            tt = p(ref_videos=ref_videos, metadata=metadata, image_for_clip=image01)
            assert isinstance(tt, svd_pipe.TupleTensor)
            out[f"svd.action.{fusion}"] = p.action_emb
            if fusion == "mean":
                out["svd.action.tt0"], out["svd.action.tt1"] = tt.to_tuple()
                meta["svd"]["base_call_kwargs"] = sorted(p.base_call[1])
        # ---- SVD stage 2 (pipeline.py:147-160): PIL -> uint8 CHW -> / 127.5 - 1 -> repeated over the reference clips' frame count ----
        from types import SimpleNamespace
        pil = [SimpleNamespace(__array_interface__=im.numpy().__array_interface__, _keep=im) for im in image_u8]   # np.asarray(img) as for a PIL image
        ct = CTRecorder()
        p = svd_pipe.SVDCTPipeline(vae=_Cfg(), image_encoder=clip, unet=None, scheduler=None, feature_extractor=None, condition_transformer=ct)
        p(ref_videos=ref_videos, metadata=metadata, image=pil, image_for_clip=image01)
        out["svd.ct.video"], out["svd.ct.action_emb"] = ct.batch["video"], p.action_emb
        assert ct.cfg is True
    np.savez(os.path.join(OUT, "adapter_pipelines.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()})
    return meta


class _Cfg:
    dtype = torch.float32


def main():
    install()
    apm = gg._load_file("ref_attn_processor", f"{REF}/src/projects/condition/attn_processor.py")
    svd_pipe = importlib.import_module("src.projects.svd.pipelines.pipeline")
    cog_pipe = importlib.import_module("src.projects.cogvideox.pipeline")
    print(json.dumps({"cog": gen_cog(apm), "svd": gen_svd(apm, svd_pipe), "pipelines": gen_pipelines(cog_pipe, svd_pipe)}, indent=1))


if __name__ == "__main__":
    main()
