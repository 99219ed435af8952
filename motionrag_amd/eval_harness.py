"""Lightning-free evaluation harness with the reference's `VideoBaseModule` contracts (second half):

    validation_step / test_step        src/projects/base_module.py:129-150
    output_assertions                  :152-161
    on_validation_batch_end / start    :163-189
    denormalize                        src/utils/pipeline.py:178-184

A project module (cogvideox / svd / dynamicrafter `eval_pipeline`) plugs in as the `eval_pipeline` callable; the harness is the loop `main.py test` would drive
through Lightning: batch -> `eval_pipeline(image=, positive_prompt=, negative_prompt=, dtype=, ref_videos=, metadata=, **eval_pipeline_call_kwargs)` -> uint8 videos on
the host + the `generated_videos` records the reference's callbacks consume.  `denormalize` runs on the GPU (`mrag_denormalize_u8`) with the reference's rounding."""
from typing import Any, Callable, Iterable, Optional

import torch

from . import ops


def denormalize(tensor: torch.Tensor) -> torch.Tensor:
    """[-1, 1] -> uint8 [0, 255] (src/utils/pipeline.py:178-184), same shape, on the tensor's device"""
    if not tensor.is_cuda:
        raise ops.HipOnly("denormalize: GPU tensors only")
    if tensor.dtype not in (torch.bfloat16, torch.float32):
        tensor = tensor.to(torch.float32)
    x = tensor.contiguous()
    y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    ops.check(ops._lib.lib().mrag_denormalize_u8(ops._stream(), ops._p(x), ops._p(y), x.numel(), 1 if x.dtype == torch.float32 else 0), "mrag_denormalize_u8")
    return y


def _parse_kwargs(kw: Optional[dict]) -> dict:
    """base_module.py:114-125: YAML strings that look like numbers become numbers"""
    out = {}
    for k, v in (kw or {}).items():
        if isinstance(v, str):
            try:
                v = int(v)
            except ValueError:
                try:
                    v = float(v)
                except ValueError:
                    pass
        out[k] = v
    return out


class VideoEvalHarness:
    def __init__(self, eval_pipeline: Callable[..., torch.Tensor], eval_pipeline_call_kwargs: Optional[dict] = None, dtype: torch.dtype = torch.bfloat16):
        self.eval_pipeline = eval_pipeline
        self.eval_pipeline_call_kwargs = _parse_kwargs(eval_pipeline_call_kwargs)
        self.dtype = dtype
        self.generated_videos: list = []

    @torch.no_grad()
    def validation_step(self, batch: dict, batch_idx: int = 0) -> torch.Tensor:
        metadata = batch["metadata"]
        positive_prompt = [b["raw_prompt"] for b in metadata]
        generate_videos = self.eval_pipeline(image=batch["ref_frame"], positive_prompt=positive_prompt, negative_prompt=[""] * len(positive_prompt), dtype=self.dtype,
                                             ref_videos=batch["ref_videos"], metadata=metadata, **self.eval_pipeline_call_kwargs)
        return denormalize(generate_videos).cpu()          # [b f c h w] uint8 on the host

    test_step = validation_step

    @staticmethod
    def output_assertions(outputs: torch.Tensor, batch: Any) -> None:
        assert isinstance(outputs, torch.Tensor), f"Expected outputs to be a tensor, got {type(outputs)}"
        assert outputs.dtype == torch.uint8, f"Expected outputs to be uint8, got {outputs.dtype}"
        assert outputs.device == torch.device("cpu"), f"Expected outputs to be on CPU, got {outputs.device}"
        assert len(outputs.shape) == 5, f"Expected outputs to be 5D, got {len(outputs.shape)}D"
        assert "metadata" in batch, f"Expected batch to have a 'metadata' attribute, got {batch}"
        assert len(batch["metadata"]) == outputs.size(0), \
            f"Metadata length does not match outputs batch size, got {len(batch['metadata'])}, expected {outputs.size(0)}"

    def on_validation_batch_end(self, outputs: torch.Tensor, batch: Any, batch_idx: int = 0, dataloader_idx: int = 0) -> None:
        self.output_assertions(outputs, batch)
        gt_videos = denormalize(batch["video"]).cpu() if "video" in batch else None
        for i, item in enumerate(batch["metadata"]):
            self.generated_videos.append({"video": outputs[i][None], "gt_video": gt_videos[i][None] if gt_videos is not None else None, "id": item["id"],
                                          "prompt": item["raw_prompt"], "save_name": item["save_name"]})

    on_test_batch_end = on_validation_batch_end

    def on_validation_start(self) -> None:
        self.generated_videos.clear()

    on_test_start = on_validation_start

    def run(self, batches: Iterable[dict]) -> list:
        """the loop Lightning's `trainer.test` runs over the dataloader"""
        self.on_test_start()
        for i, batch in enumerate(batches):
            self.on_test_batch_end(self.test_step(batch, i), batch, i)
        return self.generated_videos
