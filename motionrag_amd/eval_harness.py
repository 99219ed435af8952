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
    """`eval_pipeline`: the project's callable (cogvideox / svd / dynamicrafter `eval_pipeline`); `eval_pipeline_call_kwargs`: the YAML's extra keyword arguments."""

    # the reference checks these properties of a step's output one by one (base_module.py:152-161); here they are one table
    _OUTPUT_CONTRACT = (
        ("a tensor", lambda o, b: isinstance(o, torch.Tensor)),
        ("uint8", lambda o, b: o.dtype == torch.uint8),
        ("on the host", lambda o, b: o.device.type == "cpu"),
        ("5-D [b, f, c, h, w]", lambda o, b: o.dim() == 5),
        ("accompanied by batch['metadata']", lambda o, b: "metadata" in b),
        ("one video per metadata entry", lambda o, b: len(b["metadata"]) == o.shape[0]),
    )

    def __init__(self, eval_pipeline: Callable[..., torch.Tensor], eval_pipeline_call_kwargs: Optional[dict] = None, dtype: torch.dtype = torch.bfloat16):
        self.eval_pipeline = eval_pipeline
        self.eval_pipeline_call_kwargs = _parse_kwargs(eval_pipeline_call_kwargs)
        self.dtype = dtype
        self.generated_videos: list = []

    @torch.no_grad()
    def validation_step(self, batch: dict, batch_idx: int = 0) -> torch.Tensor:
        """base_module.py:129-147: prompts from the metadata, empty negative prompts, the first frame as the image condition, the retrieved clips; uint8 [b f c h w] on the host"""
        meta = batch["metadata"]
        prompts = [m["raw_prompt"] for m in meta]
        call = dict(image=batch["ref_frame"], positive_prompt=prompts, negative_prompt=[""] * len(prompts), dtype=self.dtype, ref_videos=batch["ref_videos"], metadata=meta)
        call.update(self.eval_pipeline_call_kwargs)
        return denormalize(self.eval_pipeline(**call)).cpu()

    test_step = validation_step

    @classmethod
    def output_assertions(cls, outputs: torch.Tensor, batch: Any) -> None:
        for what, holds in cls._OUTPUT_CONTRACT:
            assert holds(outputs, batch), f"a step's output must be {what} (got {type(outputs).__name__}" + (
                f" {tuple(outputs.shape)} {outputs.dtype} on {outputs.device})" if isinstance(outputs, torch.Tensor) else ")")

    def on_validation_batch_end(self, outputs: torch.Tensor, batch: Any, batch_idx: int = 0, dataloader_idx: int = 0) -> None:
        """base_module.py:163-178: one record per clip for the saving / metric callbacks"""
        self.output_assertions(outputs, batch)
        truth = denormalize(batch["video"]).cpu() if "video" in batch else None
        self.generated_videos.extend(
            {"video": outputs[i:i + 1], "gt_video": None if truth is None else truth[i:i + 1], "id": m["id"], "prompt": m["raw_prompt"], "save_name": m["save_name"]}
            for i, m in enumerate(batch["metadata"]))

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
