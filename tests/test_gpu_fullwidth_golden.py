"""GPU parity of the HIP modules against golden vectors produced by the REFERENCE'S OWN classes at the widths and row counts where the library
dispatches to its PRODUCTION kernels, with the dispatch asserted through the library's launch counters (`ops.dispatched`, include/mrag_hip.h:
enum mrag_kernel_id).  Fixtures: tests/golden/fullwidth_{cog,svd,dc}.npz (generator oracle/gen_golden_fullwidth.py); the CPU suite checks the oracle
restatements against the same files (tests/test_fullwidth_golden_cpu.py).  No restatement sits between the product and the reference here.

Tolerance (bf16 activations between the kernels vs the reference's fp32): relative Frobenius error <= 1 % per output; the element bound is the
one of the reduced-width fixture tests (3 % + 4 % of the mean magnitude) on >= 99.9 % of the elements and twice that on every element."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fullwidth as fw

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, want, name, rel_l2=1e-2):
    g, w = got.float().cpu(), torch.as_tensor(np.asarray(want)).float()
    assert g.shape == w.shape, (name, g.shape, w.shape)
    assert torch.isfinite(g).all(), name
    l2 = ((g - w).norm() / w.norm()).item()
    assert l2 <= rel_l2, f"{name}: relative L2 error {l2:.4f} > {rel_l2}"
    err, tol = (g - w).abs(), 3e-2 * w.abs() + 4e-2 * w.abs().mean()
    frac = (err > tol).float().mean().item()
    assert frac <= 1e-3, f"{name}: {frac:.2e} of the elements outside 3 % + 4 % of the mean magnitude"
    assert (err <= 2 * tol).all(), f"{name}: max error {err.max().item():.4g} beyond twice the element bound"
    return l2


def dev(t):
    return t.to(DEV, torch.bfloat16)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def test_cogvideox_processor_full_width_on_the_production_kernels(hip, golden_dir):
    """attn_processor.py:176-283 at D = 3072 / 48 heads / text 226 + video 4 050 rows, B = 2, rope on, B' = 1: the fused QKV projection with qk-LayerNorm + RoPE
    in the persistent four-wave GEMM's epilogue, attn16, the folded motion branch, to_out on the persistent GEMM"""
    from motionrag_amd import attn_processor as ap, ops
    from oracle import cogvideox_ref
    g, meta = fw.load(golden_dir, "fullwidth_cog.npz")
    attn = ap.Attention(meta["D"], heads=meta["H"], dim_head=64, bias=True, out_bias=True, qk_norm="layer_norm", eps=1e-6)
    proc = ap.APAdapterCogVideoXAttnProcessor2_0(meta["D"], meta["ip_dim"])
    attn.set_processor(proc)
    sd = dict(fw.weights(meta["attn"]))
    sd.update({"processor." + k: v for k, v in fw.weights(meta["proc"]).items()})
    attn.load_state_dict(sd, strict=True)
    attn = attn.to(DEV, torch.bfloat16)
    x = fw.cog_inputs()
    cos, sin = cogvideox_ref.rope_3d(64, *meta["thw"])              # the table builder is an INPUT of the fixture (same call as the generator's)
    rope = (cos.to(DEV), sin.to(DEV))
    rv, rt = torch.from_numpy(g["rows_v"]), torch.from_numpy(g["rows_t"])
    with ops.dispatched() as d:
        h, e = attn(dev(x["hidden"]), dev(x["enc"]), image_rotary_emb=(rope, dev(x["ip1"])))
        torch.cuda.synchronize()
    c = d.counts
    assert c.get("GEMM_W4_QKNORM_ROPE", 0) == 1 and "QKNORM_ROPE" not in c, c          # the fused epilogue, not the separate norm + RoPE pass
    assert c.get("ATTN16", 0) + c.get("ATTN16_KSPLIT", 0) == 1 and "ATTN_FLASH" not in c and "ATTN_FLASH_KSPLIT" not in c, c
    assert c.get("IP_ATTN_FOLDED", 0) == 1, c
    assert c.get("GEMM_W4", 0) >= 1, c                                                  # to_out [8 552, 3072] x [3072, 3072]
    close(h[:, rv.to(DEV)], g["h"], "hidden")
    close(e[:, rt.to(DEV)], g["e"], "text")
    # scale 0: the motion branch off (:243-249) -- the joint attention alone, which the branch's larger magnitude would otherwise mask
    proc.scale = [0.0]
    with ops.dispatched() as d:
        h0, _ = attn(dev(x["hidden"]), dev(x["enc"]), image_rotary_emb=(rope, dev(x["ip1"])))
        torch.cuda.synchronize()
    assert "IP_ATTN_FOLDED" not in d.counts and d.counts.get("GEMM_W4_QKNORM_ROPE", 0) == 1, d.counts
    close(h0[:, rv[::4].to(DEV)], g["h_scale0"], "hidden, scale 0")


def test_svd_processor_full_width_on_the_wide_tile(hip, golden_dir):
    """attn_processor.py:18-141 at C = 320 / 5 heads / 4 frames of 72 x 128 pixels: to_q, to_q_ip and to_out on the N = K = 320 kernel (round 5; the 256x320 tile before)"""
    from motionrag_amd import attn_processor as ap, ops
    g, meta = fw.load(golden_dir, "fullwidth_svd.npz")
    attn = ap.Attention(meta["C"], cross_attention_dim=meta["cross_dim"], heads=meta["H"], dim_head=64, bias=False, out_bias=True)
    proc = ap.APAdapterAttnProcessor2_0(meta["C"], meta["cross_dim"])
    attn.set_processor(proc)
    sd = dict(fw.weights(meta["attn"]))
    sd.update({"processor." + k: v for k, v in fw.weights(meta["proc"]).items()})
    attn.load_state_dict(sd, strict=True)
    attn = attn.to(DEV, torch.bfloat16)
    x = fw.svd_inputs()
    rows = torch.from_numpy(g["rows"]).to(DEV)
    with ops.dispatched() as d:
        out = attn(dev(x["hidden"]), (dev(x["img"]), dev(x["act"])))
        torch.cuda.synchronize()
    assert d.counts.get("GEMM_N320K320", 0) == 3, d.counts                              # to_q, to_q_ip, to_out: 36 864 rows x 320 x 320 on the register-resident-weight kernel; the 1- and 25-row K / V projections take the small tile
    close(out[:, rows], g["out"], "out")
    attn.residual_connection = True
    close(attn(dev(x["hidden"]), (dev(x["img"]), dev(x["act"])))[:, rows[::4]], g["out_resid"], "out + residual")


def test_dynamicrafter_blocks_full_width_on_the_production_kernels(hip, golden_dir):
    """lvdm attention.py:316-445, openaimodel3d.py:211-281 at C = 320 / 5 heads / context 1024, 2 clips x 5 frames of 48 x 64: the GEGLU projection on the
    persistent four-wave GEMM, the K = 320 linears with N = 320 / 960 on the register-resident-weight kernel (gemm_k320_kernel, round 5), FF2 on the 256x320 tile, attn16
    for the spatial self-attention, the 3x3 and (3,1,1) implicit-GEMM convolutions on 256-row tiles"""
    from motionrag_amd import dynamicrafter as dc, ops
    g, meta = fw.load(golden_dir, "fullwidth_dc.npz")
    C, B, T, (h, w), cd = meta["C"], meta["B"], meta["T"], meta["hw"], meta["ctx_dim"]
    x = fw.dc_inputs()
    ctx = {k: dev(x[k]) for k in ("prompt", "image", "action")}
    pix = torch.from_numpy(g["pix"]).to(DEV)
    take = lambda y: y.reshape(B * T, h * w, -1)[:, pix]                               # noqa: E731  channels-last rows [n, pix, C]

    def load(m, name):
        m.load_state_dict(fw.weights(meta[name]), strict=True)
        return m.to(DEV, torch.bfloat16)

    st = load(dc.SpatialTransformer(C, meta["heads"], 64, depth=1, context_dim=cd, use_linear=True, image_cross_attention=True, action_cross_attention=True), "st")
    with ops.dispatched() as d:
        y = st(dev(nhwc(x["x"])), ctx)
        torch.cuda.synchronize()
    c = d.counts
    # K = 320 linears on the register-resident-weight kernel (round 5): proj_in, to_q (cross), to_out x 2, to_q_a, proj_out (N = 320), the fused QKV (N = 960); FF1 + GEGLU
    # (N = 2 560) on the persistent four-wave kernel; FF2 (N = 320, K = 1 280) on the 256x320 tile
    assert c.get("GEMM_N320K320", 0) >= 6 and c.get("GEMM_256x320", 0) >= 1 and c.get("GEMM_W4_GEGLU", 0) == 1, c
    assert c.get("ATTN16", 0) + c.get("ATTN16_KSPLIT", 0) == 1, c                       # the 3 072-key spatial self-attention
    close(take(y), g["st_y"], "SpatialTransformer")

    tt = load(dc.TemporalTransformer(C, meta["heads"], 64, depth=1, context_dim=cd, use_linear=True, temporal_length=T), "tt")
    with ops.dispatched() as d:
        y = tt(dev(nhwc(x["x"])), B)
        torch.cuda.synchronize()
    assert d.counts.get("ATTN_TINY", 0) >= 2 and d.counts.get("GEMM_N320K320", 0) >= 3 and d.counts.get("GEMM_W4_GEGLU", 0) == 1, d.counts   # two 5-frame self-attentions per pixel (each in row chunks)
    close(take(y), g["tt_y"], "TemporalTransformer")

    rb = load(dc.ResBlock(C, meta["emb_dim"], 0.0, out_channels=meta["out_ch"], use_temporal_conv=True), "rb")
    with ops.dispatched() as d:
        y = rb(dev(nhwc(x["x"])), dev(F.silu(x["emb"])), batch_size=B)
        torch.cuda.synchronize()
    c = d.counts
    big3 = sum(c.get(k, 0) for k in ("CONV3_W4", "CONV3_256x256", "CONV3_256x320", "CONV3_256x128"))
    bigt = sum(c.get(k, 0) for k in ("CONVT_W4", "CONVT_256x256", "CONVT_256x320"))
    assert big3 == 2 and bigt == 4 and "CONV3_128x128" not in c and "CONVT_128x128" not in c, c      # in / out 3x3 convolutions, four (3,1,1) ones
    close(take(y), g["rb_y"], "ResBlock")
