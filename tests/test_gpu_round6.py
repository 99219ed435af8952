"""GPU parity tests of the kernels that round 6 added or re-routed, through the C ABI (motionrag_amd.ops -> ctypes), each with its dispatch asserted:
the persistent GEMM's tail rectangle and per-sample weights, the packed score blocks of the folded motion branch, the GroupNorm fold by the last-arriving
statistics workgroup, the persistent AdaLN LayerNorm, the 256x256 implicit-GEMM convolutions on the slim tap cursor, the stream-copy probe and the second
(direct-form) scoring of the retrieval fan-out.  Tolerances: the ones of tests/test_gpu_kernels.py (bf16 outputs: 2 % of |want| + 2 % of the mean magnitude;
layout-only changes: bit-equal)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def close(got, want, scale=None, rtol=2e-2, atol_frac=2e-2):
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all(), "non-finite output"
    s = want.abs().mean().item() if scale is None else scale
    err = (got - want).abs()
    bad = err > rtol * want.abs() + atol_frac * s
    assert not bad.any(), f"{bad.sum().item()} / {bad.numel()} outside tolerance; max err {err.max().item():.4g}, scale {s:.4g}"


# ---------------------------------------------------------------------------------------------- persistent GEMM: tail rectangle
@pytest.mark.parametrize("tm,tn,K,epi", [(33, 16, 512, "none"), (35, 15, 512, "resid"), (35, 15, 1536, "gelu"), (139, 48, 320, "none")])
def test_gemm_w4_tail_rectangle(hip, tm, tn, K, epi):
    """(opt-in) a persistent launch whose last round would hold <= 32 tiles stops in front of a rectangle of the last row group's last tile columns, which runs as its
    own launch of 128x128 tiles (launch_w4 / plan_tail_rect; the DiT's FF1 is 139 x 48 = 6 672 tiles = 26 rounds + 16): bit-equal to the one-launch form
    (same K order, same rounding points), the fp32 reference within the bf16 tolerance, both launches counted"""
    from motionrag_amd import ops
    M, N = tm * 256 - 100, tn * 256
    tiles = tm * tn
    assert 0 < tiles % 256 <= 32
    g = torch.Generator().manual_seed(tm * 100 + tn)
    x, w, b = bf(torch.randn(M, K, generator=g)).to(DEV), bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV), bf(torch.randn(N, generator=g)).to(DEV)
    r = bf(torch.randn(M, N, generator=g)).to(DEV) if epi == "resid" else None
    kw = {"none": {}, "resid": dict(epilogue=ops.EPI_RESID, resid=r), "gelu": dict(epilogue=ops.EPI_GELU_TANH)}[epi]
    ops.TUNING["gemm"] = ops.GEMM_TUNE_TAIL_RECT     # MRAG_GEMM_TUNE_TAIL_RECT (opt-in: measured equal to the partial round on the DiT's FF1)
    try:
        with ops.dispatched() as d:
            got = ops.linear(x, w, b, **kw)
    finally:
        ops.TUNING["gemm"] = 0
    assert d.counts.get("GEMM_W4") == 1 and d.counts.get("GEMM_W4_TAIL_RECT") == 1 and d.counts.get("GEMM_128x128") == 1, d.counts
    with ops.dispatched() as d1:
        one = ops.linear(x, w, b, **kw)             # the shipped dispatch: one launch (the persistent kernel, or the 320-wide tile where that finishes in fewer rounds)
    assert "GEMM_W4_TAIL_RECT" not in d1.counts and sum(d1.counts.values()) == 1, d1.counts
    assert torch.equal(got, one)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 900, M)])        # a sample of rows incl. the rectangle's
    acc = x[rows].float() @ w.float().t() + b.float()
    want = {"none": acc, "resid": (r[rows].float() + acc.to(torch.bfloat16).float()) if r is not None else None,
            "gelu": torch.nn.functional.gelu(acc, approximate="tanh")}[epi]
    close(got[rows], want, scale=acc.abs().mean().item())


# ---------------------------------------------------------------------------------------------- persistent GEMM: per-sample weights
@pytest.mark.parametrize("B,S,N,K", [(2, 1000, 1280, 512), (2, 4276, 1280, 3072), (3, 700, 256, 320)])
def test_gemm_per_sample_weights(hip, B, S, N, K):
    """out[b] = x[b] @ W[b]^T as ONE persistent launch (mrag_gemm_args.w_batch_stride; the row-tile grid restarts at every sample, a sample's last tile is
    clamped and masked at the sample's end): bit-equal to a launch per sample, nothing written outside a sample's rows"""
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + S)
    x = bf(torch.randn(B, S, K, generator=g)).to(DEV)
    w = bf(torch.randn(B, N, K, generator=g) * K ** -0.5).to(DEV)
    with ops.dispatched() as d:
        got = ops.linear_per_sample(x, w)
    assert d.counts == {"GEMM_W4_BATCHED_W": 1}, d.counts
    want = torch.stack([ops.linear(x[b], w[b]) for b in range(B)])
    assert torch.equal(got, want)
    close(got[:, :200], torch.einsum("bsk,bnk->bsn", x[:, :200].float(), w.float()))
    # one weight for every sample: a plain GEMM over all rows; samples_per_weight > 1 with several weights: a launch per sample
    with ops.dispatched() as d:
        same = ops.linear_per_sample(x, w[:1].contiguous(), samples_per_weight=B)
    assert "GEMM_W4_BATCHED_W" not in d.counts
    assert torch.equal(same, torch.stack([ops.linear(x[b], w[0]) for b in range(B)]))
    # a shape the batched kernel does not take (N % 128 != 0) falls back to the loop
    w2 = w[:, :N - 64].contiguous()
    with ops.dispatched() as d:
        fb = ops.linear_per_sample(x, w2)
    assert "GEMM_W4_BATCHED_W" not in d.counts
    assert torch.equal(fb, got[..., :N - 64])


# ---------------------------------------------------------------------------------------------- folded motion branch: packed score blocks
@pytest.mark.parametrize("B,S,H,keys,ks", [(2, 1000, 48, 25, 26), (1, 333, 4, 25, 30), (2, 128, 48, 9, 10)])
def test_ip_attn_folded_packed_score_blocks(hip, B, S, H, keys, ks):
    """key k of head h at column ks * h + k instead of 32 * h + k (48 heads x 25 keys: 1 280 columns = five GEMM tiles instead of six): the same softmax . V
    update (to a bf16 ulp: the keys occupy other MFMA slots), whatever sits in the columns between and behind the blocks"""
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(S)
    sc32 = bf(torch.randn(B, S, H, 32, generator=g) * 3).to(DEV)
    v = bf(torch.randn(B, keys, H * 64, generator=g)).to(DEV)
    h0 = bf(torch.randn(B, S, H * 64, generator=g)).to(DEV)
    W = -(-((H - 1) * ks + 32) // 8) * 8
    packed = bf(torch.randn(B, S, W, generator=g) * 50).to(DEV)                 # garbage everywhere ...
    for h in range(H):
        packed[..., h * ks:h * ks + keys] = sc32[:, :, h, :keys]                # ... except the valid keys
    a, b2 = h0.clone(), h0.clone()
    ops.ip_attn_folded_(sc32.view(B, S, H * 32).contiguous(), v, a, H, keys, out_scale=0.7)
    with ops.dispatched() as d:
        ops.ip_attn_folded_(packed, v, b2, H, keys, out_scale=0.7, key_stride=ks)
    assert d.counts == {"IP_ATTN_FOLDED": 1}
    # (key k of head h sits in MFMA slot k + (ks h) mod 8 of the aligned chunks that cover its block: another summation order over the keys than the 32-column
    # layout's, so the two agree to fp32 rounding in front of the bf16 store -- a bf16 ulp on a few elements -- not bit for bit)
    diff = (a.float() - b2.float()).abs()
    assert diff.max().item() <= 2.0 ** -5 and (diff.norm() / a.float().norm()).item() <= 1e-3, (diff.max().item(), (diff.norm() / a.float().norm()).item())
    p = torch.softmax(sc32[..., :keys].float() * 0.125, dim=-1)                 # [B, S, H, keys]
    want = h0.float() + 0.7 * torch.einsum("bshk,bkhd->bshd", p, v.view(B, keys, H, 64).float()).reshape(B, S, H * 64)
    close(a, want)


# ---------------------------------------------------------------------------------------------- GroupNorm: fold by the last-arriving statistics workgroup
@pytest.mark.parametrize("N,HW,C,emb", [(28, 2304, 320, True), (32, 576, 1280, False), (2, 4608, 640, True), (5, 1000, 64, False)])
def test_groupnorm_fold_by_last_arriver(hip, N, HW, C, emb):
    """(opt-in: measured 5 % slower on the UNet steps) <= 128 chunk partials per channel: the sample's last statistics workgroup also folds them -- two launches
    instead of three; longer partial lists keep the parallel fold kernel.  The arrival counters are left at zero: a second call on the same workspace gives the
    same bits; the shipped three-launch form agrees to fp32 rounding of the statistics."""
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(N * 10 + C)
    x = bf(torch.randn(N, HW, C, generator=g) * 1.7 + 0.3).to(DEV)
    w, b = bf(1 + 0.1 * torch.randn(C, generator=g)).to(DEV), bf(0.1 * torch.randn(C, generator=g)).to(DEV)
    e = bf(0.5 * torch.randn(N, C, generator=g)).to(DEV) if emb else None
    chunks = max(1, min(1024, HW // 32, max(64, -(-2048 // N))))
    with ops.dispatched() as d:
        got = ops.groupnorm(x, w, b, 32, 1e-5, silu=True, emb=e, fold=True)
    want_counts = {"GN_STATS_FOLD": 1, "GN_APPLY": 1} if chunks <= 128 else {"GN_STATS": 1, "GN_FOLD": 1, "GN_APPLY": 1}
    assert d.counts == want_counts, (chunks, d.counts)
    with ops.dispatched() as d:
        three = ops.groupnorm(x, w, b, 32, 1e-5, silu=True, emb=e)            # the shipped form: statistics, fold, apply
    assert d.counts == {"GN_STATS": 1, "GN_FOLD": 1, "GN_APPLY": 1}, d.counts
    close(got, three, scale=0.5, rtol=1e-2, atol_frac=1e-2)
    xx = x.float() + (e.float()[:, None] if emb else 0)
    ref = torch.nn.functional.group_norm(xx.permute(0, 2, 1), 32, w.float(), b.float(), 1e-5).permute(0, 2, 1)
    close(got, torch.nn.functional.silu(ref), scale=0.5)
    again = ops.groupnorm(x, w, b, 32, 1e-5, silu=True, emb=e, fold=True)
    assert torch.equal(got, again)
    other = ops.groupnorm(x[:, : HW // 2].contiguous(), w, b, 32, 1e-5, fold=True)        # another shape on the same grow-only workspace: the counters are where they were
    ref2 = torch.nn.functional.group_norm(x[:, : HW // 2].float().permute(0, 2, 1), 32, w.float(), b.float(), 1e-5).permute(0, 2, 1)
    close(other, ref2, scale=0.8)


# ---------------------------------------------------------------------------------------------- persistent AdaLN LayerNorm
@pytest.mark.parametrize("B,S,split,affine", [(2, 4276, 226, True), (2, 4500, 0, False), (1, 9000, 1, True)])
def test_layernorm_stream_kernel(hip, B, S, split, affine):
    """D = 3 072, >= 8 192 rows: persistent waves with the per-column factors gamma (1 + scale), beta (1 + scale) + shift folded into registers and refolded at
    every sample / text-video boundary (layernorm_stream_kernel); equals the fp32 reference and the per-row kernel within fp32 rounding of the fold"""
    from motionrag_amd import ops
    D = 3072
    g = torch.Generator().manual_seed(S)
    x = bf(torch.randn(B, S, D, generator=g) * 1.5 + 0.25).to(DEV)
    w = bf(1 + 0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    b = bf(0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    md = bf(0.3 * torch.randn(B, 4, D, generator=g)).to(DEV)
    with ops.dispatched() as d:
        got = ops.layernorm(x, w, b, 1e-5, shift0=md[:, 0], scale0=md[:, 1], shift1=md[:, 2], scale1=md[:, 3], rows_per_batch=S, split=split, mod_stride=md.stride(0))
    assert d.counts == {"LAYERNORM_STREAM": 1}, d.counts
    ln = torch.nn.functional.layer_norm(x.float(), (D,), w.float() if affine else None, b.float() if affine else None, 1e-5)
    want = ln.clone()
    want[:, :split] = ln[:, :split] * (1 + md[:, 1].float()[:, None]) + md[:, 0].float()[:, None]
    want[:, split:] = ln[:, split:] * (1 + md[:, 3].float()[:, None]) + md[:, 2].float()[:, None]
    close(got, want, scale=1.0)
    with ops.dispatched() as d:
        plain = ops.layernorm(x, w, b, 1e-5)
    assert d.counts == {"LAYERNORM_STREAM": 1}, d.counts
    close(plain, ln, scale=1.0)
    with ops.dispatched() as d:                                               # below 8 192 rows: the per-row kernel -- the SAME BITS (a sequence-sharded rank reproduces the unsharded rows)
        few = ops.layernorm(x[:, :1000].contiguous(), w, b, 1e-5)
        few_mod = ops.layernorm(x[:, :1000].contiguous(), w, b, 1e-5, shift0=md[:, 0], scale0=md[:, 1], shift1=md[:, 2], scale1=md[:, 3], rows_per_batch=1000, split=min(split, 1000), mod_stride=md.stride(0))
    assert d.counts == {"LAYERNORM": 2}, d.counts
    assert torch.equal(few, plain[:, :1000])
    assert torch.equal(few_mod[:, :min(split, 1000)], got[:, :min(split, 1000)]) and (split >= 1000 or torch.equal(few_mod[:, split:], got[:, split:1000]))


# ---------------------------------------------------------------------------------------------- 256x256 convolution tiles on the slim tap cursor
def _conv3x3_ref(x, w, b, stride=1, up=False):
    xx = x.float().permute(0, 3, 1, 2)
    if up:
        xx = torch.nn.functional.interpolate(xx, scale_factor=2, mode="nearest")
    Cout, Cin = w.shape[0], x.shape[-1]
    wt = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    return torch.nn.functional.conv2d(xx, wt, b.float(), stride=stride, padding=1).permute(0, 2, 3, 1)


@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,up", [(6, 96, 112, 128, 256, 1, False), (2, 100, 130, 256, 512, 1, True), (32, 72, 96, 128, 256, 2, False), (70, 30, 30, 64, 256, 1, False)])
def test_conv3x3_256x256_tile_slim_cursor(hip, N, H, W, Cin, Cout, stride, up):
    """3x3 convolutions whose shape selects the 8-wave 256x256 tile (>= 192 tiles, N = 256 / 512: neither the 320-wide nor the 128-wide tile pays): the tap cursor
    as 32-bit offsets against a workgroup base with the tap-independent state parked in LDS (gemm_tile, SLIM) -- samples smaller than a tile (30 x 30), image
    borders, stride 2 and the fused nearest upsample included"""
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(H * W)
    x = bf(torch.randn(N, H, W, Cin, generator=g)).to(DEV)
    w = bf(torch.randn(Cout, 9 * Cin, generator=g) * (9 * Cin) ** -0.5).to(DEV)
    b = bf(torch.randn(Cout, generator=g)).to(DEV)
    with ops.dispatched() as d:
        got = ops.conv_implicit(x, w, b, ops.CONV_3X3, stride=stride, upsample=up)
    assert d.counts == {"CONV3_256x256": 1}, d.counts
    close(got, _conv3x3_ref(x, w, b, stride, up))
    r = bf(torch.randn(*got.shape, generator=g)).to(DEV)
    got_r = ops.conv_implicit(x, w, b, ops.CONV_3X3, stride=stride, upsample=up, resid=r)
    close(got_r, r.float() + got.float(), scale=got.float().abs().mean().item())


def test_conv_temporal_and_causal3d_256x256_tile(hip):
    """the (3,1,1) temporal convolution and the causal 3x3x3 one on the same tile"""
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, HW, C, Co = 2, 9, 3000, 128, 256
    x = bf(torch.randn(B * T, HW, C, generator=g)).to(DEV)
    w = bf(torch.randn(Co, 3 * C, generator=g) * (3 * C) ** -0.5).to(DEV)
    b = bf(torch.randn(Co, generator=g)).to(DEV)
    with ops.dispatched() as d:
        got = ops.conv_implicit(x, w, b, ops.CONV_T3, frames=T)
    assert d.counts == {"CONVT_256x256": 1}, d.counts
    xx = x.float().view(B, T, HW, C)
    pad = torch.nn.functional.pad(xx, (0, 0, 0, 0, 1, 1))
    wt = w.float().view(Co, 3, C)
    want = sum(torch.einsum("bthc,oc->btho", pad[:, k:k + T], wt[:, k]) for k in range(3)) + b.float()
    close(got.view(B, T, HW, Co), want)
    # causal 3x3x3: x [S (Tf + 2), H, W, C] -> [S Tf, H, W, Co]
    S, Tf, H, W = 2, 5, 72, 80
    x3 = bf(torch.randn(S * (Tf + 2), H, W, C, generator=g)).to(DEV)
    w3 = bf(torch.randn(Co, 27 * C, generator=g) * (27 * C) ** -0.5).to(DEV)
    with ops.dispatched() as d:
        got3 = ops.conv_implicit(x3, w3, b, ops.CONV_3X3, t_frames=Tf)
    assert d.counts == {"CONV3_256x256": 1}, d.counts
    vol = x3.float().view(S, Tf + 2, H, W, C).permute(0, 4, 1, 2, 3)                                   # [S, C, Tf + 2, H, W]
    wt3 = w3.float().view(Co, 3, 3, 3, C).permute(0, 4, 1, 2, 3)
    want3 = torch.nn.functional.conv3d(vol, wt3, b.float(), padding=(0, 1, 1)).permute(0, 2, 3, 4, 1).reshape(S * Tf, H, W, Co)
    close(got3, want3)


# ---------------------------------------------------------------------------------------------- LayerNorm in the few-row GEMM's A load
@pytest.mark.parametrize("M,N,K,epi", [(250, 1024, 1024, "none"), (250, 4096, 1024, "gelu"), (25, 1024, 1024, "gelu"), (16, 3072, 768, "none"), (251, 2048, 2048, "none")])
def test_gemm_few_rows_layernorm_in_a_load(hip, M, N, K, epi):
    """`to_q(norm2(latents))` / `gelu(ff1(ln(latents)))` of CAMA's Perceiver layers as ONE launch (mrag_gemm_args.a_ln): the rows' statistics and the normalised
    bf16 operands are computed inside the few-row kernel with layernorm_kernel's arithmetic, so the result is BIT-identical to LayerNorm kernel + GEMM"""
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = bf(torch.randn(M, K, generator=g) * 2 + 0.3).to(DEV)
    lw, lb = bf(1 + 0.2 * torch.randn(K, generator=g)).to(DEV), bf(0.1 * torch.randn(K, generator=g)).to(DEV)
    w = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    e = ops.EPI_GELU_ERF if epi == "gelu" else ops.EPI_NONE
    with ops.dispatched() as d:
        got = ops.linear_ln(x, lw, lb, 1e-5, w, epilogue=e)
    assert d.counts == {"GEMM_SKINNY_LNA": 1}, d.counts
    two = ops.linear(ops.layernorm(x, lw, lb, 1e-5), w, epilogue=e)
    assert torch.equal(got, two)
    ref = torch.nn.functional.layer_norm(x.float(), (K,), lw.float(), lb.float(), 1e-5).to(torch.bfloat16).float() @ w.float().t()
    close(got, torch.nn.functional.gelu(ref) if epi == "gelu" else ref, scale=ref.abs().mean().item())
    assert torch.equal(ops.linear_ln(x, None, None, 1e-6, w, epilogue=e), ops.linear(ops.layernorm(x, None, None, 1e-6), w, epilogue=e))
    # more than 256 rows: the two launches, same bits
    xl = x.repeat(300 // M + 1, 1)[:300].contiguous()
    with ops.dispatched() as d:
        big = ops.linear_ln(xl, lw, lb, 1e-5, w, epilogue=e)
    assert "GEMM_SKINNY_LNA" not in d.counts and d.counts.get("LAYERNORM") == 1, d.counts
    close(big[:min(M, 300)], two[:300], scale=ref.abs().mean().item())              # (another GEMM kernel above 256 rows: another K summation order, bf16-close)


# ---------------------------------------------------------------------------------------------- stream-copy probe, retrieval second scoring
def test_stream_copy_probe(hip):
    from motionrag_amd import _lib
    L = _lib.lib()
    src = torch.randint(0, 255, (64 * 1024 * 1024 + 4096,), dtype=torch.uint8, device=DEV)
    dst = torch.zeros_like(src)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.mrag_probe_stream_copy(st, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel() - 16, 0), "copy")
    torch.cuda.synchronize()
    assert torch.equal(src[:-16], dst[:-16]) and dst[-16:].abs().max().item() == 0
    assert L.mrag_probe_stream_copy(st, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), 24, 0) == _lib.MRAG_EINVAL


def test_topk_fanout_l2_distances_are_the_scan_forms(hip):
    """a batch of >= 16 queries (fan-out kernel: selection through |q|^2 + |x|^2 - 2 q.x) returns, for metric l2, the distances of the scan form --
    the merge step scores the 16 selected candidates again with the direct sum of (q - x)^2: a row's distance to itself is exactly 0, nothing is negative, and
    rows AND distances equal the single-query call shape's on unnormalised 768-d data with a near-duplicate in the table"""
    from motionrag_amd import ops
    from oracle import topk_ref
    rng = np.random.default_rng(11)
    db = rng.standard_normal((3000, 768)).astype(np.float32)
    db[1] = db[0] + np.float32(9.2e-5) * rng.standard_normal(768).astype(np.float32)
    q = db[:32].copy()
    dbd, qd = torch.from_numpy(db).to(DEV), torch.from_numpy(q).to(DEV)
    with ops.dispatched() as d:
        rows, dist = ops.topk(dbd, qd, 12)
    assert d.counts == {"TOPK_DENSE": 1}, d.counts
    rs, ds = ops.topk(dbd, qd, 12, order="mfma_stream")                        # the streaming shape of the fan-out form: the same rows and bits
    assert torch.equal(rs, rows) and torch.equal(ds, dist)
    rows, dist = rows.cpu().numpy(), dist.cpu().numpy()
    assert np.all(dist[:, 0] == 0.0) and np.all(rows[:, 0] == np.arange(32)) and np.all(dist >= 0.0)
    wr, wd = topk_ref.topk(db, q, 12, "l2", mode="f32mfma")
    assert np.array_equal(rows, wr) and np.array_equal(dist, wd.astype(np.float32))
    r1, d1 = ops.topk(dbd, qd[:3], 12)                                          # the scan kernel (fewer than 16 queries): the same rows and the same bits
    assert np.array_equal(r1.cpu().numpy(), rows[:3]) and np.array_equal(d1.cpu().numpy(), dist[:3])
    raw_r, raw_d = topk_ref.topk(db, q, 12, "l2", mode="f32mfma_raw")          # what the expansion alone gives: self-distance is rounding noise
    assert np.abs(raw_d[:, 0]).max() > 1e-5


@pytest.mark.parametrize("dups", [100, 800])
def test_topk_one_launch_with_many_equal_scores(hip, dups):
    """the one-launch fan-out form orders the scores under its bound by counting ranks in one LDS array (usually ~16 of them); `dups` identical rows next to
    the queries put 100 (several 64-candidate chunks) / 800 (more than the array holds: the streaming fall-back) equal scores under the bound -- the answer is
    still the top of the strict order (distance, row), equal to the oracle's and to the streaming form's, in both filter orders"""
    from motionrag_amd import ops
    from oracle import topk_ref
    rng = np.random.default_rng(dups)
    N, Q, D = 3000, 32, 64
    db = rng.standard_normal((N, D)).astype(np.float32)
    db[100:100 + dups] = db[100]
    q = (db[100][None] + 0.05 * rng.standard_normal((Q, D))).astype(np.float32)
    group = (np.arange(N) // 4).astype(np.int32)
    excl = np.full(Q, 25, dtype=np.int32)                                       # rows 100..103 carry the excluded id
    dbd, qd, gd, ed = (torch.from_numpy(a).to(DEV) for a in (db, q, group, excl))
    for metric in ("l2", "dot"):
        for post in (False, True):
            want_r, want_d = topk_ref.topk(db, q, 12, metric, group, excl, mode="f32mfma", postfilter=post)
            for order in ("mfma", "mfma_stream", "mfma_nowait"):
                with ops.dispatched() as d:
                    rows, dist = ops.topk(dbd, qd, 12, metric=metric, group=gd, exclude=ed, postfilter=post, order=order)
                assert ("TOPK_DENSE" in d.counts) == (order != "mfma_stream"), (order, d.counts)
                np.testing.assert_array_equal(rows.cpu().numpy(), want_r, err_msg=f"{metric} {post} {order}")
                np.testing.assert_array_equal(dist.cpu().numpy(), want_d.astype(np.float32), err_msg=f"{metric} {post} {order}")
    assert np.array_equal(want_r[0, :8], np.arange(104, 112))                   # post-filter, dot: the 12 nearest are rows 100..111, the first four leave


def test_topk_one_launch_counters_survive_replays_and_mixed_shapes(hip):
    """the one-launch fan-out form keeps a sequence number and two alternating counter sets in the workspace's first 64 bytes: a prepared batch plan replayed as a
    HIP graph many times (odd and even sequence numbers), with new queries between replays, and calls of the OTHER shapes on the same workspace in between
    (single query: the fused scan + merge launch with its own counters in words 0..3; the no-wait diagnostic; the streaming form) keep giving the oracle's rows"""
    from motionrag_amd import ops, _lib
    from oracle import topk_ref
    rng = np.random.default_rng(5)
    N, Q, D = 6000, 48, 96
    db = rng.standard_normal((N, D)).astype(np.float32)
    dbd = torch.from_numpy(db).to(DEV)
    plan = ops.TopkPlan(dbd, Q, 12, graph=True)
    for it in range(7):
        q = rng.standard_normal((Q, D)).astype(np.float32)
        plan.queries.copy_(torch.from_numpy(q).to(DEV))
        rows, dist = plan.replay()
        torch.cuda.synchronize()
        want_r, want_d = topk_ref.topk(db, q, 12, "l2", mode="f32mfma")
        np.testing.assert_array_equal(rows.cpu().numpy(), want_r, err_msg=f"replay {it}")
        np.testing.assert_array_equal(dist.cpu().numpy(), want_d.astype(np.float32), err_msg=f"replay {it}")
    counters = plan._ws[:64].view(torch.int32).cpu().numpy()
    assert counters[8] == 8 and not counters[:8].any(), counters          # word 8 = the sequence number: the plan's eager run + 7 replays (a capture launches nothing); words 0..7 untouched
    # the shared (per device, N, Q) workspace of ops.topk under alternating shapes
    qd = torch.from_numpy(q).to(DEV)
    for order in ("mfma", "mfma_nowait", "mfma", "mfma_stream", "mfma", "mfma"):
        r, d_ = ops.topk(dbd, qd, 12, order=order)
        np.testing.assert_array_equal(r.cpu().numpy(), want_r, err_msg=order)
        np.testing.assert_array_equal(d_.cpu().numpy(), want_d.astype(np.float32), err_msg=order)
        r1, _ = ops.topk(dbd, qd[:1], 12)                                    # (its own workspace key: Q = 1)
        assert np.array_equal(r1.cpu().numpy()[0], topk_ref.topk(db, q[:1], 12, "l2", mode="f32chain")[0][0])
