"""ORACLE TOOLING (test infrastructure; runs only in the build container, never on the GPU box).

Golden vectors from the REFERENCE'S OWN classes at the widths and row counts at which the product library dispatches to its PRODUCTION kernels --
the persistent four-wave GEMM (`gemm_w4_kernel`, incl. the fused qk-LayerNorm + RoPE and GEGLU epilogues), `attn16_kernel`,
`ip_attn_folded_kernel`, the 256x320 tile and the implicit-GEMM convolutions on 256-row tiles -- not the 128x128 fallback tile, the separate
`qknorm_rope` pass and `attn_flash` that the reduced-width fixtures of gen_golden.py / gen_golden_attn_processor.py reach
(mrag_gemm_bf16 takes the big tiles from 192 tiles of 256x256 up, csrc/gemm_bf16.hip; attn16 from Sq > 128, Skv >= 256, csrc/attn16.hip).

    python -m oracle.gen_golden_fullwidth            # writes tests/golden/fullwidth_{cog,svd,dc}.npz

  cog  `APAdapterCogVideoXAttnProcessor2_0.__call__`   /root/reference/src/projects/condition/attn_processor.py:176-283
       D = 3072, 48 heads x 64, text 226 + video (3, 30, 45) = 4 276 rows, B = 2, rope on, motion tokens [1, 25, 1024] (the `(b r)` repeat).  (Three latent
       frames, not two: at 2 x 2 926 rows the to_out projection is 276 tiles of 256x256 -- two rounds on 256 CUs, the second 8 % full -- and the library then
       takes the 320-wide 8-wave tile; at 2 x 4 276 rows it runs on the persistent four-wave kernel like the 35 552-row benchmark shape.)
  svd  `APAdapterAttnProcessor2_0.__call__`            .../attn_processor.py:18-141
       C = 320, 5 heads, cross 1024, hidden [2 F = 4, 72 x 128 = 9 216, 320] (the real level-0 resolution), motion tokens [2, 25, 1024]
  dc   `SpatialTransformer` C = 320 / 5 heads / context 1024 (text 77 + image 16 + action 25 tokens), `TemporalTransformer`, `ResBlock` 320 -> 640
       with the temporal convolution, on [b t = 2 x 5, 320, 48, 64]
       .../dynamicrafter/DynamiCrafter/lvdm/modules/attention.py:316-445, .../networks/openaimodel3d.py:211-281

The diffusers stub is gen_golden_attn_processor.py's (constructor plumbing + the one restated `apply_rotary_emb`).  Weights and inputs are NOT stored:
they are regenerated from (keys, shapes, seed, std) / input seeds by oracle.seeded (torch CPU generator walk), rounded to bf16-representable
values -- the reference then runs in fp32 on exactly what the bf16 product path reads.  Outputs are stored for a fixed sample of rows / pixels
(the `rows_*` index arrays in each file), fp32, so every file stays below 8 MB.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import time
import types

import numpy as np
import torch

from . import gen_golden as gg
from . import gen_golden_attn_processor as gap
from .seeded import seeded_sd

OUT = gg.OUT
REF = gg.REF


def bf(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def seeded_inputs(seed: int, shapes: dict) -> dict:
    """named N(0, 1) inputs from one generator walk (dict order), bf16-representable; tests/fullwidth.py regenerates them from the same call"""
    g = torch.Generator().manual_seed(int(seed))
    return {k: bf(torch.randn(*shp, generator=g)) for k, shp in shapes.items()}


def load_seeded(module: torch.nn.Module, seed: int, std: float) -> dict:
    """overwrite every parameter from (sorted keys, shapes, seed, std), bf16-representable; returns the fixture's meta entry"""
    sd0 = module.state_dict()
    keys = sorted(sd0)
    shapes = [list(sd0[k].shape) for k in keys]
    sd = {k: bf(v) for k, v in seeded_sd(keys, shapes, seed, std).items()}
    module.load_state_dict(sd, strict=True)
    return {"seed": seed, "std": std, "keys": keys, "shapes": shapes}


def sample_rows(n: int, count: int, edges=(0, 1, 127, 128, 191, 192, 255, 256, 257, 383, 384, 511, 512)) -> np.ndarray:
    """row sample: tile-boundary rows of the 128 / 192 / 256-row kernels, the last rows (ragged tiles), and an even spread"""
    base = {e for e in edges if e < n} | {n - 1, n - 2, max(n - 65, 0), max(n - 129, 0)}
    spread = np.linspace(0, n - 1, max(count - len(base), 2)).round().astype(np.int64)
    return np.array(sorted(base | set(spread.tolist())), dtype=np.int64)


COG = dict(D=3072, H=48, ip_dim=1024, text_len=226, thw=(3, 30, 45), B=2, attn_seed=701, proc_seed=702, input_seed=703, std=0.02)
SVD = dict(C=320, H=5, cross_dim=1024, F=2, hw=(72, 128), attn_seed=711, proc_seed=712, input_seed=713, std=0.03)
DC = dict(C=320, heads=5, ctx_dim=1024, B=2, T=5, hw=(48, 64), st_seed=721, tt_seed=722, rb_seed=723, input_seed=724, emb_dim=1280, out_ch=640)


def cog_inputs():
    c = COG
    t, h, w = c["thw"]
    return seeded_inputs(c["input_seed"], {"hidden": (c["B"], t * h * w, c["D"]), "enc": (c["B"], c["text_len"], c["D"]), "ip1": (1, 25, c["ip_dim"])})


def gen_cog(apm):
    c = COG
    attn = gap.Attention(c["D"], heads=c["H"], dim_head=64, bias=True, out_bias=True, qk_norm="layer_norm", eps=1e-6)
    proc = apm.APAdapterCogVideoXAttnProcessor2_0(c["D"], c["ip_dim"])
    meta = dict(c, attn=load_seeded(attn, c["attn_seed"], c["std"]), proc=load_seeded(proc, c["proc_seed"], c["std"]))
    x = cog_inputs()
    cos, sin = gap.rope_3d(64, *c["thw"])
    t0 = time.time()
    with torch.no_grad():
        oh, oe = proc(attn, x["hidden"].clone(), x["enc"].clone(), image_rotary_emb=((cos, sin), x["ip1"]))
        proc.scale = [0.0]
        oh0, _ = proc(attn, x["hidden"].clone(), x["enc"].clone(), image_rotary_emb=((cos, sin), x["ip1"]))
    rows_v, rows_t = sample_rows(oh.shape[1], 120), sample_rows(oe.shape[1], 24, edges=(0, 1, 127, 128, 191, 192))
    meta["seconds"] = round(time.time() - t0, 1)
    meta["motion_branch_rel"] = float((oh - oh0).norm() / oh.norm())         # the adapter branch's share of the result (a test that ignored it would pass below this)
    np.savez(os.path.join(OUT, "fullwidth_cog.npz"), meta=json.dumps(meta), rows_v=rows_v, rows_t=rows_t,
             h=oh[:, rows_v].numpy(), e=oe[:, rows_t].numpy(), h_scale0=oh0[:, rows_v[::4]].numpy())
    return {k: meta[k] for k in ("seconds", "motion_branch_rel")}


def svd_inputs():
    c = SVD
    hw = c["hw"][0] * c["hw"][1]
    return seeded_inputs(c["input_seed"], {"hidden": (2 * c["F"], hw, c["C"]), "img": (2 * c["F"], 1, c["cross_dim"]), "act": (2, 25, c["cross_dim"])})


def gen_svd(apm):
    c = SVD
    attn = gap.Attention(c["C"], cross_attention_dim=c["cross_dim"], heads=c["H"], dim_head=64, bias=False, out_bias=True)
    proc = apm.APAdapterAttnProcessor2_0(c["C"], c["cross_dim"])
    meta = dict(c, attn=load_seeded(attn, c["attn_seed"], c["std"]), proc=load_seeded(proc, c["proc_seed"], c["std"]))
    x = svd_inputs()
    with torch.no_grad():
        out = proc(attn, x["hidden"].clone(), (x["img"], x["act"]))
        attn.residual_connection = True
        out_r = proc(attn, x["hidden"].clone(), (x["img"], x["act"]))
    rows = sample_rows(out.shape[1], 250)
    np.savez(os.path.join(OUT, "fullwidth_svd.npz"), meta=json.dumps(meta), rows=rows, out=out[:, rows].numpy(), out_resid=out_r[:, rows[::4]].numpy())
    return {"rows": int(rows.size)}


def dc_inputs():
    c = DC
    n, (h, w) = c["B"] * c["T"], c["hw"]
    return seeded_inputs(c["input_seed"], {"x": (n, c["C"], h, w), "prompt": (n, 77, c["ctx_dim"]), "image": (n, 16, c["ctx_dim"]), "action": (n, 25, c["ctx_dim"]),
                                          "emb": (n, c["emb_dim"])})


def gen_dc():
    c = DC
    dc = f"{REF}/src/projects/dynamicrafter/DynamiCrafter"
    pkg = types.ModuleType("dcroot"); pkg.__path__ = [dc]; sys.modules["dcroot"] = pkg
    attn_mod = importlib.import_module("dcroot.lvdm.modules.attention")
    net = importlib.import_module("dcroot.lvdm.modules.networks.openaimodel3d")
    x = dc_inputs()
    ctx = {k: x[k] for k in ("prompt", "image", "action")}
    n, (h, w) = c["B"] * c["T"], c["hw"]
    st = attn_mod.SpatialTransformer(c["C"], c["heads"], 64, depth=1, context_dim=c["ctx_dim"], use_linear=True, use_checkpoint=False, image_cross_attention=True,
                                     action_cross_attention=True).eval()
    tt = attn_mod.TemporalTransformer(c["C"], c["heads"], 64, depth=1, context_dim=c["ctx_dim"], use_linear=True, use_checkpoint=False, only_self_att=True,
                                      relative_position=False, temporal_length=c["T"]).eval()
    rb = net.ResBlock(c["C"], c["emb_dim"], 0.0, out_channels=c["out_ch"], dims=2, use_temporal_conv=True).eval()
    meta = dict(c, st=load_seeded(st, c["st_seed"], 0.03), tt=load_seeded(tt, c["tt_seed"], 0.03), rb=load_seeded(rb, c["rb_seed"], 0.02))
    x5 = x["x"].view(c["B"], c["T"], c["C"], h, w).permute(0, 2, 1, 3, 4).contiguous()          # [b, c, t, h, w]
    t0 = time.time()
    with torch.no_grad():
        st_y = st(x["x"], ctx)                                                                    # [n, C, h, w]
        tt_y = tt(x5)                                                                             # [b, C, t, h, w]
        rb_y = rb(x["x"], x["emb"], batch_size=c["B"])                                            # [n, out_ch, h, w]
    meta["seconds"] = round(time.time() - t0, 1)
    pix = sample_rows(h * w, 60, edges=(0, 1, 63, 64, 255, 256, 257))                             # pixel sample, every frame
    take = lambda y: y.flatten(2)[:, :, pix].permute(0, 2, 1).contiguous().numpy()              # noqa: E731  [n, pix, C]
    tt_rows = tt_y.permute(0, 2, 1, 3, 4).reshape(n, c["C"], h, w)                                # back to (b t) frames
    np.savez(os.path.join(OUT, "fullwidth_dc.npz"), meta=json.dumps(meta), pix=pix, st_y=take(st_y), tt_y=take(tt_rows), rb_y=take(rb_y))
    return {"seconds": meta["seconds"], "pixels": int(pix.size)}


def main():
    os.makedirs(OUT, exist_ok=True)
    gap.install()
    apm = gg._load_file("ref_attn_processor", f"{REF}/src/projects/condition/attn_processor.py")
    report = {"cog": gen_cog(apm), "svd": gen_svd(apm), "dc": gen_dc()}
    for f in ("fullwidth_cog.npz", "fullwidth_svd.npz", "fullwidth_dc.npz"):
        report[f] = f"{os.path.getsize(os.path.join(OUT, f)) / 2 ** 20:.2f} MiB"
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
