"""Full-size CFG denoise steps of the two UNet BASELINE configs (#2 SVD 14x576x1024, #5 DynamiCrafter-1024 16x576x1024) on the GPU.
The fp32 oracle cannot run these sizes in seconds, so the checks are the size-independent ones: shape, finiteness, bit-determinism across
two runs, sensitivity to the motion tokens (the adapter branch is live at every site), and the algorithmic FLOP count of the launches the
step actually makes against the figures BASELINE / DESIGN quote (105.7 TFLOP counted on the reference's own UNetModel; 88.7 TFLOP
for the restated SVD architecture).  Parity at reduced sizes lives in test_gpu_unet.py / test_gpu_svd.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_dynamicrafter1024_full_size_cfg_step(hip):
    from motionrag_amd import workloads as W
    net = W.dynamicrafter1024_unet(DEV)
    x, ts, ctx, fs = W.dynamicrafter1024_inputs(DEV)
    out = {}
    fl = W.count_flops(lambda: out.setdefault("a", net(x, ts, context=ctx, fs=fs)))
    a = out["a"]
    b = net(x, ts, context=ctx, fs=fs)
    assert a.shape == (2, 4, 16, 72, 128) and a.dtype == torch.bfloat16
    assert torch.isfinite(a.float()).all() and a.float().abs().mean().item() > 1e-3
    assert torch.equal(a, b), "two runs of the same step differ (non-deterministic reduction order somewhere)"
    assert abs(fl / 1e12 - W.DC1024_STEP_TFLOP) / W.DC1024_STEP_TFLOP < 0.015, f"step makes {fl / 1e12:.2f} TFLOP of launches, expected {W.DC1024_STEP_TFLOP}"
    ctx2 = dict(ctx, action=ctx["action"] * 0.5)
    c = net(x, ts, context=ctx2, fs=fs)
    assert (c.float() - a.float()).abs().max().item() > 0, "the motion tokens do not reach the output"
    # BASELINE config #5: the same step with the spatial self-attention on the fp8 (e4m3) MFMA path.  15 of the 16 spatial transformers
    # qualify (S = 9216 / 2304 / 576 ... S % 128 == 0 and S >= 512); the UNet output moves by the quantisation error of those attentions,
    # damped by the residual stream: stated bound 5 % relative Frobenius against the bf16 step (single fp8 attention: <= 8 %, test_gpu_fp8.py)
    from motionrag_amd import dynamicrafter as dc
    dc.set_attention_precision(net, "fp8")
    try:
        f8 = net(x, ts, context=ctx, fs=fs)
    finally:
        dc.set_attention_precision(net, "bf16")
    rel = ((f8.float() - a.float()).norm() / a.float().norm()).item()
    print(f"fp8-attention step vs bf16 step: relative Frobenius difference {rel:.4f}")
    assert torch.isfinite(f8.float()).all() and 0.0 < rel <= 0.05, rel


def test_svd_full_size_cfg_step(hip):
    from motionrag_amd import workloads as W
    net, names = W.svd_unet(DEV)
    assert len(names) == 16                                  # adapter sites: configs/svd/MotionRAG_open.yml:115-131
    step, lat, reset = W.svd_step(net, DEV)
    out = {}
    fl = W.count_flops(lambda: out.setdefault("v", step()))
    v1, l1 = out["v"].clone(), lat.clone()
    reset()
    v2 = step()
    assert v1.shape[-3:] == (4, 72, 128) and v1.numel() == 2 * 14 * 4 * 72 * 128
    assert torch.isfinite(v1.float()).all() and torch.isfinite(l1.float()).all() and v1.float().abs().mean().item() > 1e-3
    assert torch.equal(v1, v2) and torch.equal(l1, lat), "two runs of the same step differ"
    assert abs(fl / 1e12 - W.SVD_STEP_TFLOP) / W.SVD_STEP_TFLOP < 0.015, f"step makes {fl / 1e12:.2f} TFLOP of launches, expected {W.SVD_STEP_TFLOP}"
