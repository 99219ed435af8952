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
              "timm.models.layers"]:
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
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
