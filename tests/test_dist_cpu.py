"""CPU, world_size 2, gloo: the clip sharding + end-of-loop all-gather used by bench.py --gpus N."""
import os
import sys

import numpy as np
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


def _one_rank_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from motionrag_amd.dist import SequenceParallel, gather_latents
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.arange(12, dtype=torch.float32).view(2, 2, 3).to(torch.bfloat16)
        forced = gather_latents(x, 1, force=True)                 # a launcher-started one-rank group: the collective runs all the same (bench.py)
        rows = SequenceParallel(0, 1).all_gather_rows_async(x.contiguous()).wait()
        ret[0] = (forced is not x, forced.float().tolist() == x.float().tolist(), rows.float().tolist() == x.float().tolist(), SequenceParallel(0, 1).shard(17776))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_one_rank_process_group_runs_the_collectives():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (tests/test_gpu_multiproc.py runs it over RCCL): with a one-rank process group the
    end-of-loop gather (`force=True`) and the K / V row gather go THROUGH the backend and return the rank's own data; without a group `force` is inert"""
    from motionrag_amd.dist import gather_latents
    x = torch.randn(2, 3)
    assert gather_latents(x, 1, force=True) is x                  # no process group: identity
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_one_rank_worker, args=(1, 31500 + os.getpid() % 2000, ret), nprocs=1, join=True)
    assert ret[0] == (True, True, True, (0, 17776))


def test_bench_detects_an_attached_profiler(monkeypatch):
    """bench.py skips the shipped-configuration clip (a second shape of the dominant kernel) when rocprofv3's preload is present, so that
    `rocprofv3 --stats -- python3 bench.py` averages that kernel over ONE shape"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mrag_bench_cpu", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in [k for k in os.environ if k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_"))]:
        monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not bench.profiler_attached()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.profiler_attached()
    monkeypatch.setenv("LD_PRELOAD", "")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.profiler_attached()
    # a container that merely exports some ROCPROF_* / ROCP_* variable is NOT a profiler -- the default run must measure the same workload
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    monkeypatch.setenv("ROCP_METRICS", "x.xml")
    assert not bench.profiler_attached()


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


# ------------------------------------------------------------------------------------------------ BASELINE config #4 at its real geometry
S_FULL, LT_FULL, B_FULL, WORLD8 = 17776, 226, 2, 8          # 226 text + 13 x 30 x 45 video rows; CFG batch 2; one node of 8 GPUs


def test_sequence_layout_world8_real_geometry():
    """SequenceParallel.layout at 17 776 rows over 8 ranks: 2 222 rows each, the text / video split (226) falls INSIDE rank 0, every other rank holds
    video rows only; text and video ranges tile the prompt and the patch grid exactly once, in order (what cogvideox.forward slices by)"""
    from motionrag_amd.dist import SequenceParallel
    text, video, rows = [], [], []
    for r in range(WORLD8):
        lay = SequenceParallel(r, WORLD8).layout(S_FULL, LT_FULL)
        assert lay.rows == 2222 and (lay.r0, lay.r1) == (2222 * r, 2222 * (r + 1))
        assert lay.text == (LT_FULL if r == 0 else 0) and lay.t0 == lay.r0
        assert lay.v1 - lay.v0 == lay.rows - lay.text                       # local rows = text rows first, then video rows
        text += list(range(lay.t0, lay.t0 + lay.text)); video += list(range(lay.v0, lay.v1)); rows += list(range(lay.r0, lay.r1))
    assert text == list(range(LT_FULL)) and video == list(range(S_FULL - LT_FULL)) and rows == list(range(S_FULL))
    # a split that straddles two ranks (text longer than one shard) and a rank made of text only
    lays = [SequenceParallel(r, 4).layout(40, 25) for r in range(4)]
    assert [tuple(l) for l in lays] == [(0, 10, 10, 0, 0, 0), (10, 20, 10, 10, 0, 0), (20, 30, 5, 20, 0, 5), (30, 40, 0, 30, 5, 15)]
    with pytest.raises(ValueError):
        SequenceParallel(0, 8).layout(17777, 226)
    assert tuple(SequenceParallel(0, 1).layout(S_FULL, LT_FULL)) == (0, S_FULL, LT_FULL, 0, 0, S_FULL - LT_FULL)


def _w8_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from motionrag_amd.dist import CFGParallel, SequenceParallel, gather_latents
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = {}
        sp = SequenceParallel(rank, world)
        lay = sp.layout(S_FULL, LT_FULL)
        # the joint sequence as the DiT assembles it on this rank (cogvideox.forward): local text rows first, then local video rows
        W = 6
        text_src = torch.arange(LT_FULL, dtype=torch.float32)[None, :, None] + 1e6 * torch.arange(B_FULL)[:, None, None] + torch.zeros(W)                   # [B, Lt, W]
        video_src = 5e4 + torch.arange(S_FULL - LT_FULL, dtype=torch.float32)[None, :, None] + 1e6 * torch.arange(B_FULL)[:, None, None] + torch.zeros(W)  # [B, Nv, W]
        x_loc = torch.cat([text_src[:, lay.t0:lay.t0 + lay.text], video_src[:, lay.v0:lay.v1]], dim=1).contiguous()                                        # [B, 2222, W]
        full = torch.cat([text_src, video_src], dim=1)                                                                                                       # the unsharded order
        # (1) the per-block K|V exchange: [B, s_loc, W] -> [B, S, W], rank-major == global row order, for every sample of the CFG batch
        pend = sp.all_gather_rows_async(x_loc)
        ok["kv_rows"] = torch.equal(pend.wait(), full)
        # (2) the end-of-forward exchange of the projected rows: [s_loc, B, C] -> [S, B, C] -> [B, S, C]; the video part is rows [226:]
        out = sp.all_gather(x_loc.permute(1, 0, 2).contiguous()).permute(1, 0, 2)
        ok["out_rows"] = torch.equal(out, full) and torch.equal(out[:, LT_FULL:], video_src)
        # (3) tier 0 (the judged default): one clip per rank, final latents [1, 13, 16, 60, 90] bf16 gathered once in rank order
        lat = torch.full((1, 13, 16, 60, 90), float(rank), dtype=torch.bfloat16)
        g = gather_latents(lat, world)
        ok["latents"] = tuple(g.shape) == (world, 13, 16, 60, 90) and all(bool((g[r] == r).all()) for r in range(world))
        # (4) tier 1: four CFG pairs; each pair exchanges its two branches of ONE clip
        cp = CFGParallel(rank, world, group=CFGParallel.pair_groups(world, rank))
        v = cp.gather_branches(torch.full((1, 13, 16, 60, 90), float(rank), dtype=torch.bfloat16))
        ok["cfg"] = (cp.clip, cp.branch) == (rank // 2, rank % 2) and tuple(v.shape) == (2, 13, 16, 60, 90) and \
            bool((v[0] == 2 * (rank // 2)).all()) and bool((v[1] == 2 * (rank // 2) + 1).all())
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_config4_world8_gloo_real_geometry():
    """BASELINE config #4 (CogVideoX 49x480x720 over 8 GPUs) with 8 gloo ranks at the REAL row geometry -- 8 x 2 222 rows, B = 2, the 226-row text
    block inside rank 0: the K|V row exchange (the asynchronous entry point the attention processor calls), the output-row gather, the end-of-loop
    latent gather and the CFG pair exchange all return the unsharded order on every rank"""
    port = 37500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_w8_worker, args=(WORLD8, port, ret), nprocs=WORLD8, join=True)
    for r in range(WORLD8):
        assert ret[r] == {"kv_rows": True, "out_rows": True, "latents": True, "cfg": True}, (r, ret[r])


# ---------------------------------------------------------------------------------------------- bench.py end to end at world 8 (round 6)
def _bench_dry(tmp_path, gpus, shard, tag):
    """`python bench.py --gpus N --shard ... --dry-run-cpu`: bench.py's own launcher child (torch.distributed.run, 127.0.0.1), N ranks over gloo, the DiT
    replaced by a shape-faithful fake.  Returns (the JSON lines on stdout, rank 0's latents)."""
    import json
    import subprocess
    import sys
    import numpy as np
    chk = tmp_path / f"{tag}.npy"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--shard", shard, "--dry-run-cpu", "--steps", "2", "--warmup", "1", "--check", str(chk)],
                         capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    return lines, np.load(chk)


def test_bench_world8_clip_sharding_end_to_end_dry_run(tmp_path):
    """what the first 8-GPU scaling run exercises besides RCCL itself: the launcher child, rank -> clip mapping (clip r on rank r), the end-of-loop gather in
    rank-major order, the max-over-ranks time, whole-job `value`, and ONE line printed by rank 0"""
    one, lat1 = _bench_dry(tmp_path, 1, "clips", "w1")
    eight, lat8 = _bench_dry(tmp_path, 8, "clips", "w8")
    assert len(one) == 1 and len(eight) == 1, "exactly one JSON line per run (rank 0 only)"
    d = eight[0]
    assert d["dry_run_cpu"] is True and "NOT a measurement" in d["data"]
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["process_group"] == "gloo, world 8" and d["config"]["parallelism"] == "dp8"
    assert abs(d["value"] - 8 * 49 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]          # whole-job frames/s over all ranks
    assert abs(d["frames_per_sec_per_gpu"] * 8 - d["value"]) <= 1e-3 * d["value"]
    cs = d["gathered_clip_checksums"]
    assert len(cs) == 8 and len(set(cs)) == 8, cs                                               # eight different clips came back
    assert cs[0] == one[0]["gathered_clip_checksums"][0]                                        # clip 0 is the clip a single rank denoises
    assert np.array_equal(lat1, lat8)                                                           # rank 0's own latents: bit for bit the one-rank run's


def test_bench_world8_sequence_sharding_end_to_end_dry_run(tmp_path):
    """--shard sequence (BASELINE config #4): ONE clip over 8 ranks, 17 776 = 8 x 2 222 rows; the per-block K | V gather and the output gather run through
    SequenceParallel over the real process group, and the sharded fake forward reproduces the unsharded one bit for bit"""
    one, lat1 = _bench_dry(tmp_path, 1, "clips", "s1")
    eight, lat8 = _bench_dry(tmp_path, 8, "sequence", "s8")
    d = eight[0]
    assert len(eight) == 1 and d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["parallelism"] == "sp8"
    assert abs(d["value"] - 49 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]              # one clip for the whole job
    assert np.array_equal(lat1, lat8)


def test_bench_world4_cfg_pairs_end_to_end_dry_run(tmp_path):
    """--shard cfg: two clips on four ranks, each pair exchanging its guidance branches every step"""
    four, lat4 = _bench_dry(tmp_path, 4, "cfg", "c4")
    one, lat1 = _bench_dry(tmp_path, 1, "clips", "c1")
    d = four[0]
    assert len(four) == 1 and d["n_gpus"] == 4 and d["config"]["parallelism"] == "dp2xcfg2"
    assert abs(d["value"] - 2 * 49 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert np.array_equal(lat1, lat4)


def test_bench_dry_run_rccl_env_report():
    import json
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--shard", "sequence", "--dry-run-rccl-env"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert d["dry_run_rccl_env"] and d["gpus_requested"] == 8 and "--nproc-per-node=8" in d["launcher_cmd"] and "127.0.0.1" in d["launcher_cmd"]
    assert "HSA_ENABLE_IPC_MODE_LEGACY" in d["required"] and isinstance(d["hsa_enable_ipc_mode_legacy_ok"], bool)
