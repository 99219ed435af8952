"""The N > 1 code paths of bench.py on a ONE-GPU box: two ranks started through torch.distributed.run as CHILD processes (nothing here re-execs a
process that touched the GPU), both on cuda:0, collectives through gloo (MRAG_BENCH_ONE_GPU=1) -- the same Python path as the RCCL run except for
the transport.  `--shard sequence` (SURVEY 8e tier 2: K/V all-gather per block, projected K|V first and gathered under the Q projection) and
`--shard cfg` (tier 1: the two guidance branches on a rank pair, 2.2 MB exchange per step) must reproduce the unsharded N = 1 clip."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--layers", "2", "--frames", "9", "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--no-secondary", "--no-attn-split"]


def _run(cmd, env):
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _bench(n, shard, dump=None):
    env = dict(os.environ, MRAG_BENCH_ONE_GPU="1")
    extra = ["--check", dump] if dump else []
    if n == 1:
        return _run([sys.executable, "bench.py", "--gpus", "1"] + COMMON + extra, env)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    return _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port",
                 str(port), "bench.py", "--gpus", str(n), "--shard", shard] + COMMON + extra, env)


@pytest.mark.timeout(1800)
def test_two_rank_sequence_and_cfg_sharding_reproduce_the_single_gpu_clip(hip, tmp_path):
    """two DDIM steps of a 2-layer, full-width (3072) DiT at 9 frames: the sharded runs' final latents against the single-GPU run's.  Every
    GEMM / norm / attention row is computed by the same arithmetic whatever the sharding (the key-split attention tail, whose summation order
    depends on the launch shape, is off in all three runs: --no-attn-split), so the clips agree to bf16 rounding of a few tail rows: <= 0.5 %"""
    import numpy as np
    ref = _bench(1, "clips", str(tmp_path / "ref.npy"))
    assert ref["n_gpus"] == 1 and ref["config"]["tokens"] == 226 + 3 * 1350
    want = np.load(tmp_path / "ref.npy")
    for shard, par in (("sequence", "sp2"), ("cfg", "dp1xcfg2")):
        got = _bench(2, shard, str(tmp_path / f"{shard}.npy"))
        assert got["n_gpus"] == 2 and got["config"]["parallelism"] == par
        x = np.load(tmp_path / f"{shard}.npy")
        assert x.shape == want.shape and np.isfinite(x).all()
        rel = float(np.linalg.norm(x - want) / np.linalg.norm(want))
        assert rel <= 5e-3, f"{shard}: relative Frobenius difference {rel:.4f} against the unsharded clip"
    assert _bench(2, "clips")["n_gpus"] == 2      # the judged default: one clip per rank + the end-of-loop all-gather


@pytest.mark.timeout(1800)
def test_one_rank_launcher_runs_the_collectives_through_rccl(hip, tmp_path):
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` on this one-GPU box: `init_process_group("nccl", device_id=...)`, the end-of-loop
    `all_gather_into_tensor` of the clips (dist.gather_latents) and -- with `--shard sequence` -- the per-block asynchronous K / V
    `all_gather_into_tensor(async_op=True)` + `.wait()` (SequenceParallel.all_gather_rows_async) execute THROUGH RCCL with one rank (no
    MRAG_BENCH_ONE_GPU: the transport is nccl, not gloo).  The clips must equal the launcher-less run's: bit for bit in clip mode (same kernels,
    the gather of one rank is a copy), to bf16 rounding of a few rows in sequence mode (K|V and Q are projected by two GEMM launches there)."""
    import numpy as np
    env = {k: v for k, v in os.environ.items() if k != "MRAG_BENCH_ONE_GPU"}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    plain = _run([sys.executable, "bench.py", "--gpus", "1", "--check", str(tmp_path / "plain.npy")] + COMMON, env)
    assert plain["process_group"] is None
    want = np.load(tmp_path / "plain.npy")
    for shard in ("clips", "sequence"):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        got = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                    "bench.py", "--gpus", "1", "--shard", shard, "--check", str(tmp_path / f"{shard}.npy")] + COMMON, env)
        assert got["process_group"] == "nccl, world 1" and got["n_gpus"] == 1
        assert got["config"]["parallelism"] == ("sp1" if shard == "sequence" else "dp1")
        x = np.load(tmp_path / f"{shard}.npy")
        if shard == "clips":
            np.testing.assert_array_equal(x, want)
        else:
            rel = float(np.linalg.norm(x - want) / np.linalg.norm(want))
            assert rel <= 5e-3, f"sequence mode over RCCL: relative Frobenius difference {rel:.4f}"


def test_bench_refuses_a_mismatched_launcher(hip):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + COMMON, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode != 0 and b"WORLD_SIZE=1" in out.stderr


def test_rccl_allgather_c_abi_single_rank(hip):
    """mrag_comm_* / mrag_allgather (include/mrag_hip.h) resolve RCCL at run time and move bytes on a side stream: with one rank the all-gather is a
    copy, ordered against the producer by an event (multi-rank transport is RCCL's own; the driver's scaling run exercises it)"""
    import torch
    from motionrag_amd.dist import RcclComm
    comm = RcclComm(0, 1)
    x = torch.randn(3, 1000, device="cuda").to(torch.bfloat16)
    side = torch.cuda.Stream()
    ready = torch.cuda.Event()
    y = x * 2                                            # producer on the current stream
    ready.record()
    side.wait_event(ready)
    out = comm.all_gather(y, stream=side)
    done = torch.cuda.Event()
    done.record(side)
    torch.cuda.current_stream().wait_event(done)
    assert torch.equal(out, y)
    comm.close()
