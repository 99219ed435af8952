"""SVD glue of the motion-injection path.

The SVD UNet, scheduler and pipeline body live in the third-party diffusers package; what MotionRAG adds in-tree is
  * `TupleTensor` (src/projects/svd/pipelines/pipeline.py:25-57): a tuple masquerading as a tensor so that the pair
    (image_embedding, action_emb) survives diffusers' `.to / .repeat_interleave / [idx] / .shape / .dtype` and reaches every
    spatial `attn2`, where `APAdapterAttnProcessor2_0` unpacks it (attn_processor.py:34-41);
  * `_encode_image` / `__call__` overrides (pipeline.py:113-119,147-160) that put the CAMA tokens into that tuple;
  * `set_attention_processors` (src/projects/svd/module.py:145-165).
This module mirrors those three pieces on top of `motionrag_amd.attn_processor.APAdapterAttnProcessor2_0`.
"""
from __future__ import annotations

from typing import Dict, Iterable

import torch

from .attn_processor import APAdapterAttnProcessor2_0


class TupleTensor(tuple):
    """pipeline.py:25-57: tensor-like forwarding to both members for movement / repetition, to the FIRST member for indexing,
    shape, dtype and size."""

    def to(self, *args, **kwargs):
        return TupleTensor([t.to(*args, **kwargs) for t in self])

    def cuda(self, *args, **kwargs):
        return TupleTensor([t.cuda(*args, **kwargs) for t in self])

    def cpu(self, *args, **kwargs):
        return TupleTensor([t.cpu(*args, **kwargs) for t in self])

    def repeat_interleave(self, *args, **kwargs):
        return TupleTensor([t.repeat_interleave(*args, **kwargs) for t in self])

    def __getitem__(self, item):
        return super().__getitem__(0).__getitem__(item)

    @property
    def dtype(self):
        return super().__getitem__(0).dtype

    @property
    def shape(self):
        return super().__getitem__(0).shape

    def size(self, dim):
        return super().__getitem__(0).size(dim)

    def to_tuple(self):
        return tuple(self)


class SVDMotionMixin:
    """Mix into a diffusers `StableVideoDiffusionPipeline` subclass: `class SVDCTPipeline(SVDMotionMixin, StableVideoDiffusionPipeline)`.
    Reproduces SVDCTPipeline.__call__ / SVDActionPipeline._encode_image (pipeline.py:113-119,147-160)."""

    condition_transformer = None

    def prepare_action_embeddings(self, ref_videos: torch.Tensor, image: torch.Tensor) -> torch.Tensor:
        """image: [b, c, h, w] already normalised to [-1, 1] (`image / 127.5 - 1.0`, pipeline.py:155)"""
        batch_ = {"ref_videos": ref_videos, "video": image[:, None].expand(-1, ref_videos.size(2), -1, -1, -1)}
        self.action_emb = self.condition_transformer.predict(batch_, do_classifier_free_guidance=True)
        return self.action_emb

    def _encode_image(self, *args, **kwargs) -> TupleTensor:
        image_embedding = super()._encode_image(*args, **kwargs)
        return TupleTensor([image_embedding, self.action_emb])


def set_attention_processors(unet, adapter_modules: Iterable[str], cross_attention_dim: int, hidden_sizes: Dict[str, int]) -> None:
    """src/projects/svd/module.py:145-165: install `APAdapterAttnProcessor2_0` on the listed `...attn2.processor` sites
    (configs/svd/MotionRAG_open.yml:115-131); `hidden_sizes[name]` is the block's channel count (320 / 640 / 1280)."""
    attn = {}
    for name, orig in unet.attn_processors.items():
        attn[name] = APAdapterAttnProcessor2_0(hidden_sizes[name], cross_attention_dim) if name in adapter_modules else orig
    unet.set_attn_processor(attn)
