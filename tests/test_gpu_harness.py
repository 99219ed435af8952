"""Lightning-free evaluation harness (src/projects/base_module.py:129-189) on the GPU: `denormalize` bit-exact against the reference's torch ops on the host, and
the validation_step -> batch-end bookkeeping around a project `eval_pipeline`."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_denormalize(t):          # src/utils/pipeline.py:178-184, verbatim arithmetic on the host tensor (same dtype, hence same rounding points)
    t = (t + 1.0) / 2.0
    t = torch.clip(t, 0.0, 1.0) * 255
    return t.to(torch.uint8)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_denormalize_bit_exact(hip, dtype):
    from motionrag_amd.eval_harness import denormalize
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(100003, generator=g) * 0.8, torch.tensor([-1.0, 1.0, 0.0, -1.5, 1.5, 0.999, -0.999, 1 / 255, -1 / 255, 0.5]),
                   torch.linspace(-1.2, 1.2, 4097)]).to(dtype)
    if dtype == torch.bfloat16:      # every bf16 value in [-1.25, 1.25]: exhaustive over the range that matters
        bits = torch.arange(0, 1 << 16, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
        x = torch.cat([x, bits[torch.isfinite(bits.float()) & (bits.float().abs() <= 1.25)]])
    got = denormalize(x.to(DEV)).cpu()
    want = ref_denormalize(x)
    assert got.dtype == torch.uint8 and torch.equal(got, want), int((got != want).sum())


def test_harness_contract(hip):
    from motionrag_amd.eval_harness import VideoEvalHarness
    calls = {}

    def eval_pipeline(image, positive_prompt, negative_prompt, dtype, ref_videos, metadata, num_inference_steps=50, guidance_scale=6.0, **kw):
        calls.update(n=num_inference_steps, g=guidance_scale, neg=negative_prompt, pos=positive_prompt, dtype=dtype)
        b = image.shape[0]
        return (torch.rand(b, 5, 3, 8, 8, device=image.device) * 2 - 1).to(dtype)           # [b f c h w] in [-1, 1]

    h = VideoEvalHarness(eval_pipeline, {"num_inference_steps": "25", "guidance_scale": "7.5", "tag": "abc"})
    assert h.eval_pipeline_call_kwargs == {"num_inference_steps": 25, "guidance_scale": 7.5, "tag": "abc"}       # YAML strings parsed like base_module.py:114-125
    batch = {"metadata": [{"raw_prompt": "a cat", "id": 7, "save_name": "v7"}, {"raw_prompt": "a dog", "id": 8, "save_name": "v8"}],
             "ref_frame": torch.zeros(2, 3, 8, 8, device=DEV), "ref_videos": torch.zeros(2, 3, 4, 3, 8, 8, device=DEV),
             "video": (torch.rand(2, 5, 3, 8, 8, device=DEV) * 2 - 1).to(torch.bfloat16)}
    recs = h.run([batch, batch])
    assert calls == {"n": 25, "g": 7.5, "neg": ["", ""], "pos": ["a cat", "a dog"], "dtype": torch.bfloat16}
    assert len(recs) == 4 and recs[0]["id"] == 7 and recs[1]["save_name"] == "v8" and recs[0]["prompt"] == "a cat"
    assert recs[0]["video"].shape == (1, 5, 3, 8, 8) and recs[0]["video"].dtype == torch.uint8 and recs[0]["video"].device.type == "cpu"
    assert torch.equal(recs[0]["gt_video"][0], ref_denormalize(batch["video"].cpu())[0])
    with pytest.raises(AssertionError):
        h.output_assertions(torch.zeros(2, 5, 3, 8, 8), batch)                                 # not uint8
    with pytest.raises(AssertionError):
        h.output_assertions(torch.zeros(3, 5, 3, 8, 8, dtype=torch.uint8), batch)              # batch size mismatch
