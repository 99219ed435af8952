"""CPU, world_size 2, gloo: the clip sharding + end-of-loop all-gather used by bench.py --gpus N."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_clips, ret):
    sys.path.insert(0, ROOT)
    from motionrag_amd.dist import gather_latents, shard_clips
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_clips(n_clips, world, rank)
        # weak-scaling layout of bench.py: every rank holds the same number of clips
        lat = torch.stack([torch.full((2, 3), float(100 * rank + i)) for i in range(2)]).to(torch.bfloat16)
        out = gather_latents(lat, world)
        ret[rank] = (list(mine), out.float().tolist())
    finally:
        dist.destroy_process_group()


def test_shard_clips_covers_everything():
    from motionrag_amd.dist import shard_clips
    for n in (1, 7, 8, 9, 64):
        for w in (1, 2, 4, 8):
            got = [i for r in range(w) for i in shard_clips(n, w, r)]
            assert got == list(range(n))
            sizes = [len(shard_clips(n, w, r)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_gather_latents_world2_gloo():
    world, port = 2, 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 5, ret), nprocs=world, join=True)
    assert ret[0][0] == [0, 1, 2] and ret[1][0] == [3, 4]
    want = [[[float(100 * r + i)] * 3] * 2 for r in range(world) for i in range(2)]
    assert ret[0][1] == want and ret[1][1] == want


def test_gather_latents_world1_is_identity():
    from motionrag_amd.dist import gather_latents
    x = torch.randn(2, 3)
    assert gather_latents(x, 1) is x


def _sp_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from motionrag_amd.dist import SequenceParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp = SequenceParallel(rank, world)
        r0, r1 = sp.shard(82)
        # the K/V exchange of one block: [s_loc, B, 2, H, 64] rows of this rank -> [S, B, 2, H, 64] on every rank, rank-major == row order
        rows = torch.arange(r0, r1, dtype=torch.float32).view(-1, 1, 1, 1, 1).expand(-1, 2, 2, 1, 4).contiguous()
        out = sp.all_gather(rows)
        ret[rank] = ((r0, r1), tuple(out.shape), out[:, 0, 0, 0, 0].tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sequence_parallel_gather_world2_gloo():
    """tier-2 sharding: row ranges tile the sequence and the gathered K/V rows come back in global row order on every rank"""
    world, port = 2, 31500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sp_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][0] == (0, 41) and ret[1][0] == (41, 82)
    for r in range(world):
        assert ret[r][1] == (82, 2, 2, 1, 4) and ret[r][2] == [float(i) for i in range(82)]


def _cfg_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from motionrag_amd.dist import CFGParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cp = CFGParallel(rank, world, group=CFGParallel.pair_groups(world, rank))
        v = torch.full((1, 2, 3), float(10 * rank))                    # this rank's branch of its clip
        out = cp.gather_branches(v)
        ret[rank] = (cp.branch, cp.clip, out[:, 0, 0].tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_cfg_parallel_pairs_world4_gloo():
    """tier 1: ranks (0, 1) and (2, 3) each own one clip; even rank = unconditional branch; the exchange returns [uncond ; cond] on both"""
    world, port = 4, 33500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cfg_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret[r][:2] for r in range(4)] == [(0, 0), (1, 0), (0, 1), (1, 1)]
    assert ret[0][2] == ret[1][2] == [0.0, 10.0] and ret[2][2] == ret[3][2] == [20.0, 30.0]


def _a2a_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from motionrag_amd.dist import frames_to_pixels, pixels_to_frames
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, t, hw, c = 2, 4, 6, 3
        full = torch.arange(b * t * hw * c, dtype=torch.float32).view(b, t, hw, c)      # the whole activation [b, t, hw, c]
        tl, hwl = t // world, hw // world
        mine = full[:, rank * tl:(rank + 1) * tl].contiguous()                           # frames sharded (spatial layers)
        px = frames_to_pixels(mine, world)                                                # all frames of my pixel slab (temporal layers)
        back = pixels_to_frames(px, world)
        ret[rank] = (torch.equal(px, full[:, :, rank * hwl:(rank + 1) * hwl]), torch.equal(back, mine), tuple(px.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_unet_frame_pixel_all_to_all_world2_gloo():
    """the UNets' frame <-> pixel transpose around temporal layers (attention.py:395-445, openaimodel3d.py:233-236): every rank ends
    up with ALL frames of its pixel slab in frame order, and the inverse restores the frame shard"""
    world, port = 2, 35500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_a2a_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == (True, True, (2, 4, 3, 3))
