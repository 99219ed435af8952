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


def gather_latents(latents: torch.Tensor, world: int, force: bool = False) -> torch.Tensor:
    """[b, ...] per rank -> [world * b, ...] on every rank (rank-major order); identity for world == 1 unless `force` (a one-rank process
    group started by a launcher: the collective then runs through the backend all the same -- bench.py under `--nproc-per-node 1`)"""
    if world == 1 and not (force and dist.is_initialized()):
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

    def layout(self, n_rows: int, text_len: int) -> "RowLayout":
        """this rank's slice of the joint [text ; video] sequence: rows [r0, r1) of the joint order, of which the first `text` are text rows
        [t0, t0 + text) of the prompt and the rest video tokens [v0, v1) of the patch grid.  At the headline geometry (17 776 rows, 226 text
        rows, 8 ranks of 2 222) only rank 0 holds text and its split point is 226; every other rank's split is 0."""
        r0, r1 = self.shard(n_rows)
        text = min(max(text_len - r0, 0), r1 - r0)
        return RowLayout(r0, r1, text, r0, max(r0, text_len) - text_len, max(r1, text_len) - text_len)

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

    def all_gather_rows_async(self, x: torch.Tensor) -> "_Pending":
        """x [B, n, W] (this rank's rows of every sample, contiguous) -> pending [B, world * n, W]: one all-gather per sample straight into
        that sample's slice of the result (rank-major == global row order: no permute, no copy), issued with async_op=True so that RCCL runs
        them on its own stream while the caller keeps launching work on the current stream; `.wait()` orders the current stream behind them.

        With a host-side transport (gloo: the one-GPU developer / test configuration, MRAG_BENCH_ONE_GPU) the exchange is asynchronous too and
        keeps the same ordering structure: an event on the producing stream, a SIDE stream that stages the rows to pinned host memory, a helper
        thread that runs the collective and uploads the gathered rows on the side stream, and a second event that `.wait()` puts in front of
        the consumer -- so the stream / event discipline of the RCCL path is what the two-process GPU test executes, not a synchronous copy."""
        B, n, W = x.shape
        if not x.is_contiguous():
            raise ValueError("all_gather_rows_async: contiguous [B, n, W] required")
        if self._ag is not None:                                                               # in-process test harness (ranks as threads)
            return _Pending(torch.stack([self.all_gather(x[b]) for b in range(B)]), [])
        out = torch.empty(B, self.world * n, W, dtype=x.dtype, device=x.device)
        if dist.get_backend(self.group) == "gloo":
            return _HostStagedGather(self, x, out) if x.is_cuda else _gather_rows_host(self, x, out)
        works = [dist.all_gather_into_tensor(out[b], x[b], group=self.group, async_op=True) for b in range(B)]
        return _Pending(out, works)


class RowLayout(tuple):
    """(r0, r1, text, t0, v0, v1): see SequenceParallel.layout"""
    __slots__ = ()

    def __new__(cls, r0, r1, text, t0, v0, v1):
        return tuple.__new__(cls, (r0, r1, text, t0, v0, v1))

    r0 = property(lambda s: s[0]); r1 = property(lambda s: s[1]); text = property(lambda s: s[2])
    t0 = property(lambda s: s[3]); v0 = property(lambda s: s[4]); v1 = property(lambda s: s[5])
    rows = property(lambda s: s[1] - s[0])


class _Pending:
    def __init__(self, out, works):
        self.out, self.works = out, works

    def wait(self) -> torch.Tensor:
        for w in self.works:
            w.wait()
        return self.out


def _gather_rows_host(sp, x_host: torch.Tensor, out_host: torch.Tensor) -> "_Pending":
    """[B, n, W] host rows of every rank -> out_host [B, world * n, W]; ONE collective for all samples (rank-major parts re-sliced per sample)"""
    B, n, W = x_host.shape
    parts = [torch.empty(x_host.shape, dtype=x_host.dtype) for _ in range(sp.world)]
    dist.all_gather(parts, x_host, group=sp.group)
    for r, p in enumerate(parts):
        out_host[:, r * n:(r + 1) * n].copy_(p)
    return _Pending(out_host, [])


class _HostStagedGather:
    """asynchronous K/V row exchange over a host transport; see SequenceParallel.all_gather_rows_async"""

    def __init__(self, sp, x: torch.Tensor, out: torch.Tensor):
        import threading
        self.out = out
        self._side = _side_stream(x.device)
        self._done = torch.cuda.Event()
        self._err = None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(x.device))                 # the K|V projection that produced x
        host_in = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
        host_out = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        x.record_stream(self._side); out.record_stream(self._side)

        def run():
            try:
                with torch.cuda.device(x.device), torch.cuda.stream(self._side):      # the current device is per thread
                    self._side.wait_event(ready)
                    host_in.copy_(x, non_blocking=True)
                    self._side.synchronize()                              # blocks this helper thread only
                    _gather_rows_host(sp, host_in, host_out)
                    out.copy_(host_out, non_blocking=True)
                    self._done.record(self._side)
                    self._side.synchronize()                              # host_out must outlive the upload
            except BaseException as e:                                   # surfaced by wait()
                self._err = e

        self._thread = threading.Thread(target=run, name="mrag-kv-gather", daemon=True)
        self._thread.start()

    def wait(self) -> torch.Tensor:
        self._thread.join()
        if self._err is not None:
            raise self._err
        torch.cuda.current_stream(self.out.device).wait_event(self._done)
        return self.out


_SIDE = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class RcclComm:
    """`mrag_allgather` (include/mrag_hip.h): RCCL through the C ABI on a stream of the caller's choice, for exchanges that should overlap
    compute explicitly (a side stream + events) instead of going through torch.distributed's process group.  The 128-byte unique id is
    created on rank 0 and carried by the launcher's store (any initialised torch.distributed backend, gloo included)."""

    def __init__(self, rank: int, world: int, group=None):
        import ctypes
        from . import _lib
        self._lib, self.rank, self.world = _lib.lib(), rank, world
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(self._lib.mrag_comm_unique_id(buf), "mrag_comm_unique_id")
        if world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = ctypes.create_string_buffer(box[0], 128)
        self._comm = ctypes.c_void_p()
        _lib.check(self._lib.mrag_comm_init(buf, rank, world, ctypes.byref(self._comm)), "mrag_comm_init")

    def all_gather(self, x: torch.Tensor, out: torch.Tensor = None, stream: torch.cuda.Stream = None) -> torch.Tensor:
        """[n, ...] -> [world * n, ...], enqueued on `stream` (default: the current one)"""
        import ctypes
        from . import _lib
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("RcclComm.all_gather: contiguous GPU tensor required")
        if out is None:
            out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        st = stream if stream is not None else torch.cuda.current_stream(x.device)
        _lib.check(self._lib.mrag_allgather(ctypes.c_void_p(st.cuda_stream), self._comm, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                            x.numel() * x.element_size()), "mrag_allgather")
        return out

    def close(self):
        if self._comm:
            self._lib.mrag_comm_destroy(self._comm)
            self._comm = None


class CFGParallel:
    """SURVEY 8e tier 1: the two classifier-free-guidance branches of ONE clip on a PAIR of GPUs.  Rank parity picks the branch (even =
    unconditional, odd = conditional: the [uncond ; cond] batch order of the pipelines and of ActionTransformer.predict, module.py:329);
    each rank runs the denoiser at batch 1 and the pair exchanges its velocity prediction once per step -- [1, F, 16, h, w] bf16 = 2.2 MB
    at 49x480x720 -- before both apply the CFG + scheduler update on identical inputs (latents stay bit-identical on the pair without a
    broadcast)."""

    def __init__(self, rank: int, world: int, group=None, all_gather=None):
        if world % 2:
            raise ValueError("CFG parallelism pairs ranks: world size must be even")
        self.rank, self.world, self.branch, self.clip = rank, world, rank % 2, rank // 2
        self.group, self._ag = group, all_gather            # `group`: this rank's pair {2i, 2i + 1}

    @staticmethod
    def pair_groups(world: int, rank: int):
        """every rank creates every pair group (torch.distributed requires it) and keeps its own"""
        mine = None
        for i in range(0, world, 2):
            g = dist.new_group([i, i + 1])
            if rank in (i, i + 1):
                mine = g
        return mine

    def gather_branches(self, v_local: torch.Tensor) -> torch.Tensor:
        """[1, ...] of this rank's branch -> [2, ...] = [uncond ; cond] on both ranks of the pair"""
        x = v_local.contiguous()
        if self._ag is not None:
            return self._ag(x)
        if x.is_cuda and dist.get_backend(self.group) == "gloo":
            parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(2)]
            dist.all_gather(parts, x.cpu(), group=self.group)
            return torch.cat(parts, dim=0).to(x.device)
        out = torch.empty((2 * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out


# ---------------------------------------------------------------------------------------------- UNet backbones: frame <-> pixel transposes
def frames_to_pixels(x: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """Ulysses-style transpose in front of a temporal layer of the UNets (TemporalTransformer, attention.py:395-445; TemporalConvBlock,
    openaimodel3d.py:233-236): spatial layers run with the FRAMES sharded, temporal layers need all frames of a pixel.
    x [b, t_loc, hw, c] (this rank's frames, all pixels) -> [b, t_loc * world, hw / world, c] (all frames, this rank's pixel slab);
    one all-to-all of hw / world x t_loc x c elements per peer (xGMI: all-pairs, every link busy)."""
    b, tl, hw, c = x.shape
    if hw % world:
        raise ValueError(f"{hw} pixels do not split over {world} ranks")
    if world == 1:
        return x
    send = x.view(b, tl, world, hw // world, c).permute(2, 0, 1, 3, 4).contiguous()        # [peer, b, t_loc, hw_loc, c]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)                                          # recv[r] = rank r's frames of MY pixel slab
    return recv.permute(1, 0, 2, 3, 4).reshape(b, world * tl, hw // world, c)                # frame index = r * t_loc + i (rank-major == frame order)


def pixels_to_frames(y: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """inverse of `frames_to_pixels`: [b, t, hw_loc, c] -> [b, t / world, hw_loc * world, c]"""
    b, t, hwl, c = y.shape
    if t % world:
        raise ValueError(f"{t} frames do not split over {world} ranks")
    if world == 1:
        return y
    send = y.view(b, world, t // world, hwl, c).permute(1, 0, 2, 3, 4).contiguous()         # [peer, b, t_loc, hw_loc, c]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)                                          # recv[r] = MY frames of rank r's pixel slab
    return recv.permute(1, 2, 0, 3, 4).reshape(b, t // world, world * hwl, c)
