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
