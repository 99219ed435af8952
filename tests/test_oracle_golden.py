"""CPU suite: the oracle (oracle/*.py, oracle/topk_oracle.c) against the golden vectors generated from the
reference's own code (oracle/gen_golden.py) and against independent restatements."""
import os

import numpy as np
import pytest
import torch

from oracle import cama_ref, cogvideox_ref, dynamicrafter_ref, topk_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_resampler_matches_reference(golden_dir):
    g = _load(golden_dir, "resampler.npz")
    sd = cama_ref.random_resampler_sd(torch.Generator().manual_seed(int(g["weight_seed"])), embedding_dim=768)
    x = torch.randn(*g["input_shape"].tolist(), generator=torch.Generator().manual_seed(int(g["input_seed"])))
    y = cama_ref.resampler(sd, x, heads=12, depth=4)
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-4, atol=2e-5)


def test_sinusoid_table_matches_reference(golden_dir):
    g = _load(golden_dir, "sinusoid.npz")
    t256 = cama_ref.sinusoid_table(256, 1024)[0]
    np.testing.assert_array_equal(t256[g["rows"]].numpy(), g["t256"])
    t2560 = cama_ref.sinusoid_table(2560, 1024)[0]
    np.testing.assert_array_equal(t2560[:25].numpy()[:, ::16], g["t2560_first25"])
    pe = cama_ref.sinusoid_pe(torch.ones(2, 5, 64), cama_ref.sinusoid_table(32, 64))
    np.testing.assert_array_equal(pe.numpy(), g["pe_applied"])


def test_block_causal_mask_matches_reference(golden_dir):
    g = _load(golden_dir, "mask.npz")
    np.testing.assert_array_equal(cama_ref.block_causal_mask(4, 3).numpy(), g["m4x3"])
    np.testing.assert_array_equal(np.packbits(cama_ref.block_causal_mask(10, 25).numpy()), g["m10x25"])


def test_condition_fusion_matches_reference(golden_dir):
    g = _load(golden_dir, "fusion.npz")
    emb = torch.from_numpy(g["emb"])
    for mode in ("mean", "concat", "top1"):
        np.testing.assert_allclose(cama_ref.condition_fusion(emb, mode).numpy(), g[mode], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(cama_ref.condition_fusion(emb, "weight", g["dist"].tolist()).numpy(), g["weight"], rtol=1e-5, atol=1e-6)


class _Stub:
    """same deterministic feature stub as oracle/gen_golden.py::FeatureStub"""

    def __init__(self, tokens, dim, seed):
        g = torch.Generator().manual_seed(seed)
        self.base = torch.randn(tokens, dim, generator=g)
        self.dirn = torch.randn(tokens, dim, generator=g)

    def __call__(self, x):
        m = x.reshape(x.shape[0], -1).float().mean(dim=1)
        return self.base[None] + m[:, None, None] * self.dirn[None]


def cama_fixture_inputs(g):
    """rebuild the G5 inputs: features in batch_forward order (refs flipped, target last)"""
    gi = torch.Generator().manual_seed(int(g["input_seed"]))
    shp = g["ref_videos_shape"].tolist()
    ref_videos = torch.randn(*shp, generator=gi)
    video = torch.randn(shp[0], *shp[2:], generator=gi)
    vis = _Stub(int(g["vis_tokens"]), 768, int(g["vis_seed"]))
    con = _Stub(int(g["con_tokens"]), 1024, int(g["con_seed"]))
    videos = torch.cat([ref_videos.flip(1), video[:, None]], dim=1)               # module.py:319-320
    b, K = videos.shape[:2]
    vfeat = vis(videos.reshape(b * K, *videos.shape[2:]))
    cfeat = con(videos[:, :, 0].reshape(b * K, *videos.shape[3:]))                # ref_images = videos[:, :, 0]
    ufeat = vis(torch.zeros(b, *videos.shape[2:]))
    return dict(ref_videos=ref_videos, video=video, vfeat=vfeat, cfeat=cfeat, ufeat=ufeat, b=b, vis=vis, con=con)


def test_cama_predict_matches_reference(golden_dir):
    g = _load(golden_dir, "cama_predict.npz")
    sd = cama_ref.random_cama_sd(seed=int(g["weight_seed"]))
    inp = cama_fixture_inputs(g)
    spec = cama_ref.CamaSpec()
    fwd = cama_ref.cama_forward(sd, spec, inp["vfeat"], inp["cfeat"], inp["b"])
    np.testing.assert_allclose(fwd[:, -1].numpy(), g["forward_last"], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(fwd[:, 0].numpy(), g["forward_first"], rtol=2e-4, atol=5e-5)
    out = cama_ref.cama_predict(sd, spec, inp["vfeat"], inp["cfeat"], inp["ufeat"], inp["b"])
    assert out.shape == (2 * inp["b"], 25, 1024)
    np.testing.assert_allclose(out.numpy(), g["predict"], rtol=2e-4, atol=5e-5)


def test_encoder_restatement_matches_torch():
    torch.manual_seed(1)
    d, nhead, ff, L = 128, 2, 256, 2
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=nhead, dim_feedforward=ff, dropout=0.0, activation="gelu", batch_first=True)
    enc = torch.nn.TransformerEncoder(layer, num_layers=L).eval()
    for p in enc.parameters():
        torch.nn.init.normal_(p, std=0.05)
    x = torch.randn(2, 12, d)
    mask = cama_ref.block_causal_mask(4, 3)
    with torch.no_grad():
        want = enc(x, mask)
    got = cama_ref.transformer_encoder(dict(enc.state_dict()), x, mask, nhead, L)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


def test_dc_cross_attention_matches_reference(golden_dir):
    g = _load(golden_dir, "dc_cross_attention.npz")
    ca = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ca.")}
    sa = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sa.")}
    x = torch.from_numpy(g["x"])
    ctx = {k: torch.from_numpy(g[k]) for k in ("prompt", "image", "action")}
    y = dynamicrafter_ref.cross_attention(ca, x, ctx, heads=2, image_scale=0.7, action_scale=1.0)
    np.testing.assert_allclose(y.numpy(), g["y_cross"], rtol=1e-4, atol=1e-5)
    y = dynamicrafter_ref.cross_attention(sa, x, None, heads=2)
    np.testing.assert_allclose(y.numpy(), g["y_self"], rtol=1e-4, atol=1e-5)


def test_adapter_processor_equals_dc_arithmetic():
    """attn_processor.py's `to_q_ip(out)` branch is DynamiCrafter's `to_q_a(out)` branch with no text/image part:
    with identical weights the two oracle restatements must agree (ties the unpinned CogVideoX processor restatement
    to the pinned DynamiCrafter one)."""
    g = torch.Generator().manual_seed(3)
    D, H, ipd = 128, 2, 96
    r = lambda *s: torch.randn(*s, generator=g) * 0.2
    attn = {"to_q.weight": r(D, D), "to_k.weight": r(D, D), "to_v.weight": r(D, D), "to_out.0.weight": r(D, D), "to_out.0.bias": r(D)}
    proc = {"to_q_ip.0.weight": r(D, D), "to_k_ip.0.weight": r(D, ipd), "to_v_ip.0.weight": r(D, ipd)}
    hidden, enc, ip = r(2, 20, D), r(2, 4, D), r(1, 25, ipd)
    h, e = cogvideox_ref.adapter_attn_processor(attn, proc, hidden, enc, None, ip, H, scale=1.0)
    x = torch.cat([enc, hidden], dim=1)
    dc_sd = {"to_q.weight": attn["to_q.weight"], "to_k.weight": attn["to_k.weight"], "to_v.weight": attn["to_v.weight"],
             "to_q_a.weight": proc["to_q_ip.0.weight"], "to_k_a.weight": proc["to_k_ip.0.weight"], "to_v_a.weight": proc["to_v_ip.0.weight"],
             "to_out.0.weight": attn["to_out.0.weight"], "to_out.0.bias": attn["to_out.0.bias"]}
    # DC cross-attends to context['prompt']; feed the joint sequence itself as the prompt to get self-attention + action
    y = dynamicrafter_ref.cross_attention(dc_sd, x, {"prompt": x, "action": ip.repeat(2, 1, 1)}, heads=H)
    np.testing.assert_allclose(torch.cat([e, h], dim=1).numpy(), y.numpy(), rtol=1e-5, atol=1e-6)


def test_rope_matches_complex_rotation():
    cos, sin = cogvideox_ref.rope_3d(64, 2, 3, 4)
    assert cos.shape == (24, 64)
    x = torch.randn(1, 2, 24, 64)
    y = cogvideox_ref.apply_rotary_emb(x, cos, sin)
    xc = torch.view_as_complex(x.reshape(1, 2, 24, 32, 2))
    ang = torch.atan2(sin[:, 0::2], cos[:, 0::2])
    want = torch.view_as_real(xc * torch.polar(torch.ones_like(ang), ang)[None, None]).reshape(1, 2, 24, 64)
    np.testing.assert_allclose(y.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    # positions: t varies slowest, w fastest; the t block is the first 16 dims
    assert torch.allclose(cos[0:12, :16], cos[0, :16].expand(12, 16)) and not torch.allclose(cos[12, :16], cos[0, :16])


def test_ddim_schedule_properties():
    ac = cogvideox_ref.ddim_alphas_cumprod()
    assert ac.shape == (1000,) and abs(ac[-1]) < 1e-12 and np.all(np.diff(ac) < 0)
    ts = cogvideox_ref.ddim_timesteps(50)
    assert ts[0] == 999 and ts[-1] == 19 and len(ts) == 50 and np.all(np.diff(ts) == -20)
    # one exact step: with v = 0 the update keeps x0 = sa * x, x_prev = (a + b sa) x
    sa, sb, a, b = cogvideox_ref.ddim_coeffs(ac, 999, 50)
    assert abs(sa) < 1e-6 and abs(sb - 1.0) < 1e-9
    x = torch.randn(1, 4, 8)
    out = cogvideox_ref.cfg_ddim_step(torch.zeros(2, 4, 8), x, 6.0, (sa, sb, a, b))
    np.testing.assert_allclose(out.numpy(), ((a + b * sa) * x).numpy(), rtol=1e-6)


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_topk_oracle_c_vs_numpy(metric):
    rng = np.random.default_rng(5)
    db = rng.standard_normal((700, 96)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[rng.integers(0, 700, 9)] + 0.05 * rng.standard_normal((9, 96)).astype(np.float32)
    group = (np.arange(700) // 3).astype(np.int32)
    excl = group[rng.integers(0, 700, 9)]
    r64, d64 = topk_ref.topk(db, q, 12, metric, group, excl, mode="f64")
    rnp, dnp = topk_ref.topk_numpy(db, q, 12, metric, group, excl)
    np.testing.assert_array_equal(r64, rnp)
    np.testing.assert_allclose(d64, dnp, rtol=1e-12, atol=1e-12)
    r32, d32 = topk_ref.topk(db, q, 12, metric, group, excl, mode="f32chain")
    np.testing.assert_array_equal(r32, r64)        # gaps here are >> fp32 rounding
    np.testing.assert_allclose(d32, d64, rtol=0, atol=5e-6)
    for qi in range(9):                             # the filter really removed the excluded group
        assert not np.any(group[r32[qi]] == excl[qi])
    # mode 2 -- the fan-out kernel's order (one chain over the features 8c, 8c+4, 8c+1, ...; L2 through |q|^2 + |x|^2 - 2 q.x): same ranks, fp32-close distances,
    # and its chain is a PERMUTATION of the feature order: on features that are exactly representable sums it equals the plain dot product
    rm, dm = topk_ref.topk(db, q, 12, metric, group, excl, mode="f32mfma")
    np.testing.assert_array_equal(rm, r64)
    np.testing.assert_allclose(dm, d64, rtol=0, atol=5e-6)
    ints = rng.integers(-8, 9, (40, 96)).astype(np.float32)                       # small integers: every partial sum is exact in fp32, any order gives the same bits
    ri, di = topk_ref.topk(ints, ints[:3], 5, metric, mode="f32mfma")
    rc, dc = topk_ref.topk(ints, ints[:3], 5, metric, mode="f32chain")
    np.testing.assert_array_equal(ri, rc)
    np.testing.assert_array_equal(di, dc)


def test_topk_fanout_l2_second_scoring_removes_the_cancellation():
    """the fan-out form scores l2 as |q|^2 + |x|^2 - 2 q.x -- on unnormalised 768-d N(0, 1) embeddings a row's distance to itself came
    out as ~1e-3, near-duplicates as noise or negative.  Mode f32mfma_raw keeps that arithmetic; mode f32mfma (= the kernel since round 6) scores the 16
    selected candidates again in the direct form: self-distance exactly 0, nothing negative, the distances of the scan form (f32chain) bit for bit."""
    rng = np.random.default_rng(11)
    db = rng.standard_normal((400, 768)).astype(np.float32)
    db[1] = db[0] + np.float32(9.2e-5) * rng.standard_normal(768).astype(np.float32)     # a near-duplicate: true distance ~6.5e-6
    q = db[:8].copy()
    raw_r, raw_d = topk_ref.topk(db, q, 12, "l2", mode="f32mfma_raw")
    new_r, new_d = topk_ref.topk(db, q, 12, "l2", mode="f32mfma")
    ch_r, ch_d = topk_ref.topk(db, q, 12, "l2", mode="f32chain")
    f64_r, f64_d = topk_ref.topk(db, q, 12, "l2", mode="f64")
    assert np.abs(raw_d[:, 0]).max() > 1e-5                       # the expansion: self-distance is rounding noise of |q|^2 + |x|^2 ...
    assert np.all(new_d[:, 0] == 0.0) and np.all(new_r[:, 0] == np.arange(8))   # ... the second scoring: exactly 0, the row itself first
    assert np.all(new_d >= 0.0)
    np.testing.assert_array_equal(new_r, ch_r)                    # well-separated neighbours: the same rows as the scan form ...
    np.testing.assert_array_equal(new_d, ch_d)                    # ... and the same bits (one definition of a row's distance whatever the call shape)
    np.testing.assert_array_equal(new_r, f64_r)
    assert abs(new_d[0, 1] - f64_d[0, 1]) < 1e-9 and abs(raw_d[0, 1] - f64_d[0, 1]) > 1e-6   # the near-duplicate: 6.5e-6 exact vs noise


def multi_clip_db(rng, n_videos=50, clips=6, dim=64, spread=0.05):
    """a database in which every video contributes `clips` near-duplicate clips (the multi-clip-per-video datasets MotionRAG retrieves from):
    a query built from one clip has its own video's other clips among its nearest rows, so lancedb's post-filter and a pre-filter disagree"""
    centres = rng.standard_normal((n_videos, dim)).astype(np.float32)
    db = np.repeat(centres, clips, axis=0) + spread * rng.standard_normal((n_videos * clips, dim)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    group = (np.arange(n_videos * clips) // clips).astype(np.int32)
    return db, group


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_topk_oracle_filter_order(metric):
    """lancedb `where(..., prefilter=False)` (post-filter, 0.14.0's default) against `prefilter=True`: C oracle == numpy restatement in both
    orders; the orders differ exactly where the query's own video fills the head of the unfiltered list"""
    rng = np.random.default_rng(11)
    db, group = multi_clip_db(rng)
    own = np.array([0, 7, 31, 49], np.int32)
    q = db[own * 6 + 2] + 0.01 * rng.standard_normal((4, 64)).astype(np.float32)
    k = 12
    r_pre, d_pre = topk_ref.topk(db, q, k, metric, group, own, mode="f64")
    r_post, d_post = topk_ref.topk(db, q, k, metric, group, own, mode="f64", postfilter=True)
    for rows, dist, post in ((r_pre, d_pre, False), (r_post, d_post, True)):
        rn, dn = topk_ref.topk_numpy(db, q, k, metric, group, own, postfilter=post)
        np.testing.assert_array_equal(rows, rn)
        np.testing.assert_allclose(dist, dn, rtol=1e-12, atol=1e-12)
    r_none, _ = topk_ref.topk(db, q, k, metric, mode="f64")
    for qi in range(4):
        assert (r_pre[qi] >= 0).all() and not np.any(group[r_pre[qi]] == own[qi])               # pre-filter: always k rows
        n_own = int(np.sum(group[r_none[qi]] == own[qi]))
        assert n_own == 6                                                                     # all six clips of the query's video lead the unfiltered list
        kept = r_post[qi][r_post[qi] >= 0]
        assert len(kept) == k - n_own and (r_post[qi][len(kept):] == -1).all() and np.isinf(d_post[qi][len(kept):]).all()
        np.testing.assert_array_equal(kept, [r for r in r_none[qi] if group[r] != own[qi]])     # survivors keep their order
        np.testing.assert_array_equal(kept, r_pre[qi][:len(kept)])                             # ... and are the head of the pre-filtered list
    # the fp32-chain mode (what the HIP kernel is compared with bit for bit) ranks the same rows in both orders
    np.testing.assert_array_equal(topk_ref.topk(db, q, k, metric, group, own, postfilter=True)[0], r_post)
    # no exclusion given: the flag changes nothing
    np.testing.assert_array_equal(topk_ref.topk(db, q, k, metric, postfilter=True)[0], r_none)


def test_topk_oracle_ties_and_short_results():
    db = np.zeros((5, 32), dtype=np.float32)
    db[3, 0] = 1.0
    q = np.zeros((1, 32), dtype=np.float32)
    rows, dist = topk_ref.topk(db, q, 4, "l2")
    assert rows.tolist() == [[0, 1, 2, 4]] and dist.tolist() == [[0.0, 0.0, 0.0, 0.0]]     # ties by ascending row
    rows, dist = topk_ref.topk(db, q, 8, "l2", group=np.zeros(5, np.int32) + np.array([0, 0, 0, 1, 1], np.int32), exclude=np.array([0], np.int32))
    assert rows.tolist() == [[4, 3, -1, -1, -1, -1, -1, -1]] and np.isinf(dist[0, 2:]).all()
    rows, dist = topk_ref.topk(db, q, 4, "l2", group=np.array([0, 0, 0, 1, 1], np.int32), exclude=np.array([0], np.int32), postfilter=True)
    assert rows.tolist() == [[4, -1, -1, -1]] and dist[0, 0] == 0.0 and np.isinf(dist[0, 1:]).all()     # the 4 nearest are rows 0, 1, 2, 4: one survives


# ---------------------------------------------------------------------------------------------- DynamiCrafter UNet (G8-G12)
def _dc_blocks(golden_dir):
    import json
    from oracle.seeded import seeded_sd
    g = _load(golden_dir, "dc_blocks.npz")
    meta = json.loads(str(g["meta"]))
    sds = {n: seeded_sd(m["keys"], m["shapes"], m["seed"], m["std"]) for n, m in meta.items()}
    return g, sds


def test_dc_transformers_match_reference(golden_dir):
    g, sds = _dc_blocks(golden_dir)
    ctx = {k: torch.from_numpy(g[f"ctx_{k}"]) for k in ("prompt", "image", "action")}
    y = dynamicrafter_ref.spatial_transformer(sds["st"], torch.from_numpy(g["st_x"]), ctx, heads=1)
    np.testing.assert_allclose(y.numpy(), g["st_y"], rtol=1e-4, atol=2e-5)
    y = dynamicrafter_ref.temporal_transformer(sds["tt"], torch.from_numpy(g["tt_x"]), heads=2)
    np.testing.assert_allclose(y.numpy(), g["tt_y"], rtol=1e-4, atol=2e-5)


def test_dc_resblock_and_resample_match_reference(golden_dir):
    g, sds = _dc_blocks(golden_dir)
    x, emb = torch.from_numpy(g["rb_x"]), torch.from_numpy(g["rb_emb"])
    np.testing.assert_allclose(dynamicrafter_ref.res_block(sds["rb"], x, emb, batch_size=2).numpy(), g["rb_y"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dynamicrafter_ref.res_block(sds["rb2"], x, emb, batch_size=2).numpy(), g["rb2_y"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dynamicrafter_ref.downsample(sds["dn"], x).numpy(), g["dn_y"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dynamicrafter_ref.upsample(sds["up"], x).numpy(), g["up_y"], rtol=1e-4, atol=2e-5)


def dc_unet_fixture(golden_dir):
    from oracle.seeded import seeded_sd
    g = _load(golden_dir, "dc_unet.npz")
    shapes = [s[:n].tolist() for s, n in zip(g["shapes"], g["ndims"])]
    sd = seeded_sd([str(k) for k in g["keys"]], shapes, int(g["seed"]), float(g["std"]))
    gi = torch.Generator().manual_seed(int(g["input_seed"]))
    x = torch.randn(2, 8, 4, 8, 8, generator=gi)
    ctx = {"image": torch.randn(2, 12, 64, generator=gi), "prompt": torch.randn(2, 7, 64, generator=gi), "action": torch.randn(2, 25, 64, generator=gi)}
    spec = dynamicrafter_ref.UNetSpec(model_channels=64, attention_resolutions=(1, 2), num_res_blocks=1, channel_mult=(1, 2), context_dim=64,
                                      temporal_length=4)
    return g, sd, spec, x, ctx


def test_dc_unet_matches_reference(golden_dir):
    g, sd, spec, x, ctx = dc_unet_fixture(golden_dir)
    y = dynamicrafter_ref.unet_forward(sd, spec, x, torch.from_numpy(g["timesteps"]), ctx, torch.from_numpy(g["fs"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=2e-4, atol=5e-5)


def test_dc_schedule_and_ddim_steps_match_reference(golden_dir):
    g = _load(golden_dir, "dc_schedule.npz")
    ac = dynamicrafter_ref.dc_schedule()
    np.testing.assert_allclose(ac, g["alphas_cumprod"], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(dynamicrafter_ref.dc_ddim_timesteps(30), g["t30"])      # 31 entries
    np.testing.assert_array_equal(dynamicrafter_ref.dc_ddim_timesteps(50), g["t50"])
    assert len(g["t30"]) == 31 and g["t30"][-1] == 991
    sig, al, alp = dynamicrafter_ref.dc_ddim_params(ac, g["t30"], 1.0)
    np.testing.assert_allclose(sig.numpy(), g["sigmas"], rtol=1e-6); np.testing.assert_allclose(al.numpy(), g["alphas"], rtol=1e-6)
    np.testing.assert_allclose(alp.numpy(), g["alphas_prev"], rtol=1e-6)
    np.testing.assert_allclose(dynamicrafter_ref.timestep_embedding(torch.tensor([0, 1, 481, 999]), 64).numpy(), g["temb"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(dynamicrafter_ref.dc_scale_arr(), g["scale_arr"])
    # three eta = 1 steps with the recorded noise
    ac32 = torch.tensor(ac, dtype=torch.float32)
    scale = torch.from_numpy(g["scale_arr"])
    ts = g["t30"]
    sc_t = scale[ts]
    sc_prev = torch.cat([sc_t[0:1], sc_t[:-1]])
    x = torch.from_numpy(g["xT"])
    c, uc = torch.from_numpy(g["c_shift"]), torch.from_numpy(g["uc_shift"])
    for i in range(3):
        index = len(ts) - 1 - i
        t = int(ts[index])
        f = math_cos(t)
        vc, vu = 0.5 * x + c * f, 0.5 * x + uc * f
        x, _ = dynamicrafter_ref.dc_ddim_step(vc, vu, x, torch.from_numpy(g["noises"][i]), 2.0, ac32, t, sig[index], al[index], alp[index],
                                              sc_t[index], sc_prev[index])
        np.testing.assert_allclose(x.numpy(), g["xs"][i], rtol=1e-4, atol=1e-5)


def math_cos(t):
    import math
    return math.cos(t / 100.0)


# ------------------------------------------------------------------------------------------------ SVD (parity unpinned: self-consistency only)
def svd_tiny(seed=0):
    """reduced-width SVD UNet with the motion adapters installed: (model on CPU, state dict, config, inputs)"""
    import torch
    from motionrag_amd import svd, svd_unet
    cfg = dict(in_channels=8, out_channels=4, block_out_channels=(64, 128), addition_time_embed_dim=64, projection_class_embeddings_input_dim=192,
               layers_per_block=1, cross_attention_dim=64, num_attention_heads=(1, 2))
    unet = svd_unet.UNetSpatioTemporalConditionModel(**cfg)
    names = [n for n in unet.attn_processors if "temporal_transformer_blocks" not in n and n.endswith("attn2.processor")]
    hidden = {n: dict(unet.named_modules())[n[: -len(".processor")]].to_q.in_features for n in names}
    svd.set_attention_processors(unet, names, 64, hidden)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in unet.named_parameters():
            if k.endswith("mix_factor"):
                p.copy_(torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g) if "norm" in k and k.endswith("weight") else 0.05 * torch.randn(p.shape, generator=g))
            else:
                fan = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / fan ** 0.5))
    unet = unet.to(torch.bfloat16)
    B, Fr = 2, 4
    inp = dict(sample=torch.randn(B, Fr, 8, 16, 16, generator=g), timestep=torch.tensor(1.3), image=torch.randn(B, 1, 64, generator=g),
               action=torch.randn(B, 25, 64, generator=g), added_time_ids=torch.tensor([[6.0, 127.0, 0.02]] * B))
    inp = {k: (v.to(torch.bfloat16).float() if k != "added_time_ids" and k != "timestep" else v) for k, v in inp.items()}
    return unet, cfg, inp


def test_svd_oracle_self_consistency():
    import torch
    from oracle import svd_ref
    unet, cfg, inp = svd_tiny()
    sd = unet.state_dict()
    assert any(k.endswith("attn2.processor.to_q_ip.0.weight") for k in sd), "adapter weights must be state-dict keys (Motion-Adapter.ckpt layout)"
    assert "down_blocks.0.resnets.0.temporal_res_block.conv1.weight" in sd and "mid_block.attentions.0.time_mixer.mix_factor" in sd
    y = svd_ref.unet_forward(sd, cfg, inp["sample"], inp["timestep"], inp["image"], inp["added_time_ids"], inp["action"])
    assert y.shape == (2, 4, 4, 16, 16) and torch.isfinite(y).all() and y.std() > 1e-3
    y0 = svd_ref.unet_forward(sd, cfg, inp["sample"], inp["timestep"], inp["image"], inp["added_time_ids"], None)
    assert (y - y0).abs().max() > 1e-4, "the motion tokens must reach the output"
    # the Euler step in its linear form (what the kernel takes) equals the textbook form
    sig = svd_ref.karras_sigmas(25)
    assert abs(float(sig[0]) - 700.0) < 1e-9 and abs(float(sig[24]) - 0.002) < 1e-12 and float(sig[25]) == 0.0
    g = torch.Generator().manual_seed(1)
    x, vu, vc = (torch.randn(1, 4, 4, 8, 8, generator=g, dtype=torch.float64) for _ in range(3))
    gs = torch.linspace(1.0, 3.0, 4, dtype=torch.float64)
    for i in (0, 7, 24):
        want = svd_ref.euler_cfg_step(vu, vc, x, float(sig[i]), float(sig[i + 1]), gs)
        cx, cv = svd_ref.euler_coeffs(float(sig[i]), float(sig[i + 1]))
        got = cx * x + cv * (vu + gs.view(1, -1, 1, 1, 1) * (vc - vu))
        assert torch.allclose(got, want, rtol=1e-10, atol=1e-10)


def test_dc_pipeline_glue_oracle_matches_reference(golden_dir):
    """row a20: the oracle's restatement of image_guided_synthesis / DynamiCrafterPipelineRef equals the reference's own run (G14)"""
    import json
    from oracle import stubs
    from oracle.seeded import seeded_sd
    g = _load(golden_dir, "dc_pipeline.npz")
    _, sd, spec, _, _ = dc_unet_fixture(golden_dir)
    pm = json.loads(str(g["proj_meta"]))
    proj_sd = seeded_sd(pm["keys"], pm["shapes"], pm["seed"], pm["std"])
    frames = dynamicrafter_ref.image_guided_synthesis_ref(
        sd, spec, proj_sd, stubs.ImageEmbedderStub(tokens=9, dim=48), stubs.TextStub(tokens=7, dim=64), stubs.FirstStageStub(),
        stubs.ConditionTransformerStub(dim=64), torch.from_numpy(g["image"]), [str(g["prompt"])], torch.from_numpy(g["ref_videos"]), num_frames=4,
        ddim_steps=5, guidance=2.0, fs=15, x_T=torch.from_numpy(g["x_T"]), noises=[torch.from_numpy(n) for n in g["noises"]])
    np.testing.assert_allclose(frames.numpy(), g["frames"], rtol=2e-3, atol=2e-3)
