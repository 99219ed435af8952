"""CPU suite: the oracle restatements of the motion-injection boundary against golden vectors produced by the REFERENCE'S OWN classes
(oracle/gen_golden_attn_processor.py imports /root/reference/src/projects/condition/attn_processor.py, svd/pipelines/pipeline.py and
cogvideox/pipeline.py under a diffusers stub).  Bound: 1e-5 (fp32 CPU on both sides; only the summation order of SDPA differs)."""
import json
import os

import numpy as np
import torch

from oracle import cama_ref, cogvideox_ref, svd_ref


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    return g, json.loads(str(g["meta"]))


def split_sd(g):
    attn = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("attn.")}
    proc = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("proc.")}
    return attn, proc


def cog_case_inputs(g, name):
    """(rope, ip) of a CogVideoX fixture case"""
    cos, sin = torch.from_numpy(g["cos"]), torch.from_numpy(g["sin"])
    rope = None if name.startswith("norope") else (cos, sin)
    ip = torch.from_numpy(g["ip1"] if name.endswith("repeat") else g["ip2"])
    return rope, ip


def test_cogvideox_processor_oracle_matches_the_reference_class(golden_dir):
    g, meta = load(golden_dir, "cog_attn_processor.npz")
    attn, proc = split_sd(g)
    hidden, enc = torch.from_numpy(g["hidden"]), torch.from_numpy(g["enc"])
    assert set(meta["cases"]) == {"rope_tuple", "rope_tuple_repeat", "norope_kwarg", "rope_list_kwarg", "scale_half", "scale_zero"}
    for name, c in meta["cases"].items():
        rope, ip = cog_case_inputs(g, name)
        h, e = cogvideox_ref.adapter_attn_processor(attn, proc, hidden, enc, rope, ip, meta["H"], scale=c["scale"])
        np.testing.assert_allclose(h.numpy(), g[f"{name}.h"], rtol=1e-5, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(e.numpy(), g[f"{name}.e"], rtol=1e-5, atol=1e-5, err_msg=name)
    # the cases are not degenerate: rope, the repeat, and the scale each change the result
    assert np.abs(g["rope_tuple.h"] - g["norope_kwarg.h"]).max() > 1e-2
    assert np.abs(g["rope_tuple.h"] - g["rope_tuple_repeat.h"]).max() > 1e-2
    assert np.abs(g["rope_tuple.h"] - g["scale_zero.h"]).max() > 1e-2
    np.testing.assert_array_equal(g["rope_tuple.h"], g["rope_list_kwarg.h"])     # tuple-smuggled and keyword tokens: same arithmetic
    # the reference unpacks a plain (cos, sin) TUPLE as (rope, ip) (:189) and fails inside to_k_ip; the product does not reproduce that
    assert meta["plain_cos_sin_tuple_with_kwarg"] != "ran"


def test_svd_processor_oracle_matches_the_reference_class(golden_dir):
    g, meta = load(golden_dir, "svd_attn_processor.npz")
    sd = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("attn.")}
    sd.update({"processor." + k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("proc.")})
    w = svd_ref.SD(sd)
    hidden, hidden4, img, img2, img3, act = (torch.from_numpy(g[k]) for k in ("hidden", "hidden4", "img", "img2", "img3", "act"))
    H, F = meta["H"], meta["F"]
    cases = {
        "tuple": dict(hidden=hidden), "kwarg": dict(hidden=hidden), "tuple_img3": dict(hidden=hidden, img=img3), "hidden4": dict(hidden=hidden4, img=img3),
        # TupleTensor([image_emb [2, 1, cd], tokens [2, 25, cd]]).repeat_interleave(F): diffusers repeated BOTH members per frame, r = 1 inside
        "tupletensor": dict(hidden=hidden, img=img2.repeat_interleave(F, dim=0), ip=act.repeat_interleave(F, dim=0)),
        "resid": dict(hidden=hidden, residual_connection=True), "hidden4_resid": dict(hidden=hidden4, img=img3, residual_connection=True),
        "scale_zero": dict(hidden=hidden, scale=0.0), "scale_06": dict(hidden=hidden, scale=0.6), "rescale2": dict(hidden=hidden, rescale_output_factor=2.0),
    }
    for name, c in cases.items():
        c = dict(c)
        got = svd_ref.adapter_processor_call(w, c.pop("hidden"), c.pop("img", img), c.pop("ip", act), H, **c)
        np.testing.assert_allclose(got.numpy(), g["out." + name], rtol=1e-5, atol=1e-5, err_msg=name)
    np.testing.assert_array_equal(g["out.tuple"], g["out.kwarg"])
    assert np.abs(g["out.tuple"] - g["out.scale_zero"]).max() > 1e-3
    assert np.abs(g["out.tuple_img3"][:, 0] - g["out.tuple_img3"][:, 1]).max() > 1e-3        # rows differ: the fixture is not the 1-key degenerate case


def test_tuple_tensor_contract_is_the_reference_classes(golden_dir):
    """motionrag_amd.svd.TupleTensor against what the reference's TupleTensor (svd/pipelines/pipeline.py:25-57) answered in the generator"""
    from motionrag_amd.svd import TupleTensor
    g, meta = load(golden_dir, "svd_attn_processor.npz")
    want = meta["tuple_tensor"]
    img2, act = torch.from_numpy(g["img2"]), torch.from_numpy(g["act"])
    t0 = TupleTensor([img2, act])
    got = {"getitem_is_first": bool(torch.equal(t0[1], img2[1])), "shape": list(t0.shape), "size0": int(t0.size(0)), "dtype": str(t0.dtype),
           "to_tuple_len": len(t0.to_tuple()), "to_keeps_type": isinstance(t0.to(torch.float64), TupleTensor),
           "repeat_shapes": [list(x.shape) for x in t0.repeat_interleave(meta["F"], dim=0).to_tuple()], "is_tuple": isinstance(t0, tuple)}
    assert got == want
    assert all(x.dtype == torch.float64 for x in t0.to(torch.float64).to_tuple())
    assert isinstance(t0.cpu(), TupleTensor)


def action_embedder(meta):
    from oracle.gen_golden_attn_processor import ActionEmbedderStub       # plain torch stand-in (no reference import at module level)
    return ActionEmbedderStub(**meta["embedder"])


def test_adapter_pipeline_glue_restated(golden_dir):
    """`prepare_action_embeddings` (cogvideox/pipeline.py:59-78) and SVDActionPipeline.__call__ (svd/pipelines/pipeline.py:99-110) restated with
    the pinned `condition_fusion` oracle reproduce the reference's outputs; the stage-2 pipelines' `video` batch entry is the repeated image"""
    g, meta = load(golden_dir, "adapter_pipelines.npz")
    emb = action_embedder(meta)
    ref_videos, dist = torch.from_numpy(g["ref_videos"]), g["dist"].tolist()
    W, b_ = torch.from_numpy(g["proj.weight"]), torch.from_numpy(g["proj.bias"])
    bsz, k = ref_videos.shape[:2]
    tok = emb(ref_videos.flatten(0, 1)).unflatten(0, (bsz, k))
    unc = emb(torch.zeros_like(ref_videos[:, 0]))
    for fusion in ("mean", "weight", "top1", "concat"):
        fused = cama_ref.condition_fusion(tok, fusion, dist)
        np.testing.assert_allclose(torch.nn.functional.linear(fused, W, b_).numpy(), g[f"cog.action.{fusion}.nocfg"], rtol=1e-5, atol=1e-5)
        if fusion != "concat":
            want = torch.nn.functional.linear(torch.cat([unc, fused]), W, b_).numpy()       # uncond FIRST (:75)
            np.testing.assert_allclose(want, g[f"cog.action.{fusion}.cfg"], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(want, g[f"svd.action.{fusion}"], rtol=1e-5, atol=1e-5)
    assert meta["cog"]["concat.cfg"] == "RuntimeError"
    assert meta["cog"]["base_call_kwargs"] == ["image", "prompt"] and meta["svd"]["base_call_kwargs"] == ["image_for_clip"]
    # stage 2: the condition transformer sees the conditioning image repeated over the reference clips' frame count
    f = ref_videos.shape[2]
    np.testing.assert_array_equal(g["cog.ct.video"], np.repeat(g["image01"][:, None], f, axis=1))
    u8 = torch.from_numpy(g["image_u8"]).permute(0, 3, 1, 2).float() / 127.5 - 1.0          # pil_to_tensor + / 127.5 - 1 (:154-155)
    np.testing.assert_array_equal(g["svd.ct.video"], np.repeat(u8.numpy()[:, None], f, axis=1))
    # `_encode_image` returns TupleTensor([image_embedding (CFG: zeros first), action_emb]) (:113-119)
    np.testing.assert_array_equal(g["svd.action.tt1"], g["svd.action.mean"])
    assert g["svd.action.tt0"].shape[0] == 2 * bsz and not g["svd.action.tt0"][:bsz].any()


def test_packed_score_layout_of_the_folded_motion_branch():
    """host logic of round 6's packed score blocks (attn_processor.packed_score_layout): the pitch that drops a 256-column tile of the score GEMM, the
    alignment constraint of mrag_ip_attn_folded_bf16, and the fall-back to 32 columns per head"""
    from motionrag_amd.attn_processor import packed_score_layout
    assert packed_score_layout(48, 25) == (26, 1280)            # the DiT: 48 x 26 = 1 248 -> five tiles instead of six
    assert packed_score_layout(2, 25) == (32, 64)               # nothing to gain below one tile
    assert packed_score_layout(48, 31) == (32, 1536)            # no even pitch below 32 holds 31 keys and their shift
    assert packed_score_layout(16, 9) == (10, 256)              # 16 x 32 = 512 (two tiles) -> 15 x 10 + 32 = 182 (one)
    for heads in (5, 8, 10, 16, 24, 40, 48, 64):
        for keys in range(1, 33):
            ks, nw = packed_score_layout(heads, keys)
            assert keys <= ks <= 32 and nw % 4 == 0 and nw >= (((heads - 1) * ks) & ~7) + 32
            assert all(((ks * h) & 7) + keys <= 32 for h in range(heads))
            assert nw <= heads * 32 and (ks == 32 or nw % 256 == 0)
