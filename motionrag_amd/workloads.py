"""Synthetic BASELINE.json workloads: random-init models of the named architectures and device-resident inputs of the named
shapes.  Host plumbing shared by bench.py (the measured configs), tests/ (full-size checks) and tools/microbench.py; every forward runs
through libmrag_hip.so.  No dataset, no checkpoint: weights are N(0, 0.02), norm scales 1, biases N(0, 0.02) (zero-init layers re-randomised)."""
from __future__ import annotations

import contextlib

import torch

from . import ops


@contextlib.contextmanager
def _bf16_on(dev):
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            yield
    finally:
        torch.set_default_dtype(old)


def random_init_(module: torch.nn.Module) -> torch.nn.Module:
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02)
            elif n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02)
    return module


# ---------------------------------------------------------------------------------------------- DynamiCrafter-1024 (BASELINE config #5)
def dynamicrafter1024_unet(dev="cuda", seed=0):
    """configs/dynamicrafter/MotionRAG_open.yml:206-238"""
    from . import dynamicrafter as dc
    torch.manual_seed(seed)
    with _bf16_on(dev):
        net = dc.UNetModel(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
                           channel_mult=(1, 2, 4, 4), num_head_channels=64, transformer_depth=1, context_dim=1024, use_linear=True,
                           temporal_conv=True, temporal_attention=True, temporal_self_att_only=True, use_relative_position=False,
                           temporal_length=16, addition_attention=True, image_cross_attention=True, action_cross_attention=True,
                           default_fs=10, fs_condition=True)
    return random_init_(net).eval()


def dynamicrafter1024_inputs(dev="cuda", seed=1, frames=16, h=72, w=128):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 8, frames, h, w, generator=g).to(dev, torch.bfloat16)
    ctx = {"prompt": torch.randn(2, 77, 1024, generator=g).to(dev, torch.bfloat16), "image": torch.randn(2, 16 * frames, 1024, generator=g).to(dev, torch.bfloat16),
           "action": torch.randn(2, 25, 1024, generator=g).to(dev, torch.bfloat16)}
    ts = torch.tensor([481.0, 481.0], device=dev)
    fs = torch.tensor([15, 15], device=dev)
    return x, ts, ctx, fs


DC1024_STEP_TFLOP = 105.7      # counted on the reference's own UNetModel on the meta device


# ---------------------------------------------------------------------------------------------- SVD img2vid (BASELINE config #2)
def svd_unet(dev="cuda", seed=0):
    from . import svd as svd_glue, svd_unet as su
    torch.manual_seed(seed)
    with _bf16_on(dev):
        net = su.UNetSpatioTemporalConditionModel()
        names = [n for n in net.attn_processors if "temporal_transformer_blocks" not in n and n.endswith("attn2.processor")]
        hidden = {n: dict(net.named_modules())[n[: -len(".processor")]].to_q.in_features for n in names}
        svd_glue.set_attention_processors(net, names, 1024, hidden)
    return random_init_(net).eval(), names


def svd_step(net, dev="cuda", seed=1, frames=14, h=72, w=128):
    """returns step(): one CFG UNet evaluation + the fused per-frame-guidance Euler update on `lat` (returned too)"""
    from . import svd as svd_glue, svd_unet as su
    g = torch.Generator().manual_seed(seed)
    B = 2
    x = torch.randn(B, frames, 8, h, w, generator=g).to(dev, torch.bfloat16)
    ehs = svd_glue.TupleTensor([torch.randn(B, 1, 1024, generator=g).to(dev, torch.bfloat16), torch.randn(B, 25, 1024, generator=g).to(dev, torch.bfloat16)])
    ids = torch.tensor([[6.0, 127.0, 0.02]] * B, device=dev)
    sch = su.EulerDiscreteScheduler()
    sch.set_timesteps(25)
    lat0 = torch.randn(1, frames, 4, h, w, generator=g).to(dev, torch.bfloat16)
    lat = lat0.clone()
    gs = torch.linspace(1.0, 3.0, frames, device=dev)

    def step():
        v = net(x, float(sch.timesteps[3]), ehs, ids).sample
        sch.step_(v.view(2, 1, frames, 4, h, w), lat, 3, gs)
        return v

    def reset():
        lat.copy_(lat0)

    return step, lat, reset


SVD_STEP_TFLOP = 88.7          # this package's analytic counter on the restated architecture (diffusers is absent here: SURVEY 8d)


# ---------------------------------------------------------------------------------------------- algorithmic FLOP counter
def count_flops(fn):
    """algorithmic FLOPs of one call: 2 M N K of every GEMM (true K, before padding), 2 M Cout taps Cin of every implicit-GEMM convolution,
    4 B H Sq Skv 64 of every attention launch"""
    tot = [0.0]
    lin, att, cimp = ops.linear, ops.attention, ops.conv_implicit

    def linear(x, w, *a, **k):
        tot[0] += 2.0 * (x.numel() // x.shape[-1]) * w.shape[0] * min(x.shape[-1], w.shape[1])
        return lin(x, w, *a, **k)

    def attention(q, k_, v, *a, **k):
        tot[0] += 4.0 * q.shape[0] * q.shape[2] * q.shape[1] * k_.shape[1] * 64
        return att(q, k_, v, *a, **k)

    def conv_implicit(x, wk, *a, **k):
        y = cimp(x, wk, *a, **k)
        tot[0] += 2.0 * (y.numel() // y.shape[-1]) * wk.shape[0] * wk.shape[1]
        return y

    ops.linear, ops.attention, ops.conv_implicit = linear, attention, conv_implicit
    try:
        fn()
    finally:
        ops.linear, ops.attention, ops.conv_implicit = lin, att, cimp
    return tot[0]
