"""Multi-GPU layer of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference scales by Lightning DDP over clips only (configs/cogvideox/MotionRAG_open.yml:4-8): ranks never
exchange tensors inside the denoising loop.  Here the clips of a job are sharded over ranks with no data-path collective,
and the ranks' final latents are all-gathered ONCE at the end of the loop (2.2 MB per clip) so rank 0 can hand them to the
VAE / metrics stage -- the only exchange step the path has.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def shard_clips(n_clips: int, world: int, rank: int) -> range:
    """contiguous, balanced shard of clip indices for `rank` (first n_clips % world ranks get one more)"""
    q, r = divmod(n_clips, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def gather_latents(latents: torch.Tensor, world: int) -> torch.Tensor:
    """[b, ...] per rank -> [world * b, ...] on every rank (rank-major order); identity for world == 1"""
    if world == 1:
        return latents
    x = latents.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    try:
        dist.all_gather_into_tensor(out, x)
    except (RuntimeError, NotImplementedError):       # backends without the flat variant (older gloo)
        parts: List[torch.Tensor] = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        out = torch.cat(parts, dim=0)
    return out
