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
    if x.is_cuda and dist.get_backend() == "gloo":    # developer runs of the N > 1 path on one GPU (bench.py MRAG_BENCH_ONE_GPU): stage through the host
        return gather_latents(x.cpu(), world).to(x.device)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    try:
        dist.all_gather_into_tensor(out, x)
    except (RuntimeError, NotImplementedError):       # backends without the flat variant (older gloo)
        parts: List[torch.Tensor] = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        out = torch.cat(parts, dim=0)
    return out


class SequenceParallel:
    """Tier 2: ONE clip's joint [text ; video] token sequence sharded over the ranks of a node (BASELINE config
    "CogVideoX-5B-I2V 49x720x480 frame-sharded across 8xMI355X, RCCL all-gather over xGMI").  Every per-token op of the DiT stays
    local to a rank's rows; the only exchange is one all-gather of the rank's K and V rows per block (after qk-norm + RoPE), so a
    rank's queries attend to the whole sequence, plus one all-gather of the output rows at the end of the forward.

    xGMI is point-to-point (7 links per GPU): the gather is a single RCCL all-gather of `S/world x B x 2D` bf16 per rank
    (54.6 MB per rank and block at S = 17 776, B = 2, D = 3072), not a ring of small messages.  `all_gather` may be replaced
    (tests run two 'ranks' as threads on one GPU)."""

    def __init__(self, rank: int, world: int, all_gather=None, group=None):
        self.rank, self.world, self.group = rank, world, group
        self._ag = all_gather

    def shard(self, n_rows: int):
        if n_rows % self.world:
            raise ValueError(f"sequence of {n_rows} rows does not split evenly over {self.world} ranks")
        s = n_rows // self.world
        return self.rank * s, (self.rank + 1) * s

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """[n, ...] per rank -> [world * n, ...] (rank-major) on every rank"""
        if self._ag is not None:
            return self._ag(x)
        x = x.contiguous()
        if x.is_cuda and dist.get_backend(self.group) == "gloo":   # developer runs on one GPU: stage through the host
            parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(self.world)]
            dist.all_gather(parts, x.cpu(), group=self.group)
            return torch.cat(parts, dim=0).to(x.device)
        out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out
