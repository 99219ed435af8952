"""CPU suite: the oracle restatements at the PRODUCTION widths against golden vectors produced by the reference's own classes
(oracle/gen_golden_fullwidth.py: `APAdapterCogVideoXAttnProcessor2_0` at D = 3072 / 48 heads / 4 276 rows, `APAdapterAttnProcessor2_0` at C = 320 /
9 216 pixels, DynamiCrafter `SpatialTransformer` / `TemporalTransformer` / `ResBlock` 320 -> 640 at 10 frames of 48 x 64).  The GPU tests
(test_gpu_fullwidth_golden.py) compare the HIP modules with the SAME files and assert which kernels ran.
Bound: fp32 CPU on both sides, only summation orders differ -- relative Frobenius error <= 1e-5, every element within 1e-4 of the output's mean magnitude."""
import numpy as np
import torch

import fullwidth as fw
from oracle import cogvideox_ref, dynamicrafter_ref, svd_ref


def close(got, want, name):
    want = np.asarray(want)
    l2 = fw.rel_l2(got, want)
    assert l2 <= 1e-5, f"{name}: relative L2 {l2:.2e}"
    np.testing.assert_allclose(np.asarray(got), want, rtol=1e-4, atol=1e-4 * float(np.abs(want).mean()), err_msg=name)


def test_cogvideox_processor_oracle_at_full_width(golden_dir):
    g, meta = fw.load(golden_dir, "fullwidth_cog.npz")
    assert (meta["D"], meta["H"], meta["text_len"], meta["thw"]) == (3072, 48, 226, [3, 30, 45])
    x = fw.cog_inputs()
    rope = cogvideox_ref.rope_3d(64, *meta["thw"])
    attn, proc = fw.weights(meta["attn"]), fw.weights(meta["proc"])
    rv, rt = g["rows_v"], g["rows_t"]
    with torch.no_grad():
        h, e = cogvideox_ref.adapter_attn_processor(attn, proc, x["hidden"], x["enc"], rope, x["ip1"], meta["H"], scale=1.0)
        h0, _ = cogvideox_ref.adapter_attn_processor(attn, proc, x["hidden"], x["enc"], rope, x["ip1"], meta["H"], scale=0.0)
    close(h[:, rv].numpy(), g["h"], "hidden")
    close(e[:, rt].numpy(), g["e"], "text")
    close(h0[:, rv[::4]].numpy(), g["h_scale0"], "hidden, scale 0")
    assert meta["motion_branch_rel"] > 0.05                      # the adapter branch is a visible part of the fixture


def test_svd_processor_oracle_at_full_width(golden_dir):
    g, meta = fw.load(golden_dir, "fullwidth_svd.npz")
    assert (meta["C"], meta["H"], meta["hw"]) == (320, 5, [72, 128])
    x = fw.svd_inputs()
    sd = {k: v for k, v in fw.weights(meta["attn"]).items()}
    sd.update({"processor." + k: v for k, v in fw.weights(meta["proc"]).items()})
    w = svd_ref.SD(sd)
    rows = g["rows"]
    with torch.no_grad():
        out = svd_ref.adapter_processor_call(w, x["hidden"], x["img"], x["act"], meta["H"])
        out_r = svd_ref.adapter_processor_call(w, x["hidden"], x["img"], x["act"], meta["H"], residual_connection=True)
    close(out[:, rows].numpy(), g["out"], "out")
    close(out_r[:, rows[::4]].numpy(), g["out_resid"], "out + residual")


def test_dynamicrafter_blocks_oracle_at_full_width(golden_dir):
    g, meta = fw.load(golden_dir, "fullwidth_dc.npz")
    C, B, T, (h, w) = meta["C"], meta["B"], meta["T"], meta["hw"]
    assert (C, meta["heads"], meta["ctx_dim"], meta["out_ch"]) == (320, 5, 1024, 640)
    x = fw.dc_inputs()
    ctx = {k: x[k] for k in ("prompt", "image", "action")}
    pix = g["pix"]
    take = lambda y: y.flatten(2)[:, :, pix].permute(0, 2, 1).numpy()          # noqa: E731
    with torch.no_grad():
        st_y = dynamicrafter_ref.spatial_transformer(fw.weights(meta["st"]), x["x"], ctx, heads=meta["heads"])
        x5 = x["x"].view(B, T, C, h, w).permute(0, 2, 1, 3, 4)
        tt_y = dynamicrafter_ref.temporal_transformer(fw.weights(meta["tt"]), x5, heads=meta["heads"])
        rb_y = dynamicrafter_ref.res_block(fw.weights(meta["rb"]), x["x"], x["emb"], batch_size=B)
    close(take(st_y), g["st_y"], "SpatialTransformer")
    close(take(tt_y.permute(0, 2, 1, 3, 4).reshape(B * T, C, h, w)), g["tt_y"], "TemporalTransformer")
    close(take(rb_y), g["rb_y"], "ResBlock")
