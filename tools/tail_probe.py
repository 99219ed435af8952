#!/usr/bin/env python3
"""How much do partial last rounds of workgroups cost?  (developer probe)
attention: 48 heads x 2 samples x ceil(Sq/256) query tiles on 512 slots; GEMM: ceil(M/256) x N/256 tiles on 256 CUs.
Times the kernels at row counts that fill whole rounds and at the BASELINE row count; prints time per unit of work."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"


def attn():
    B, H, S = 2, 48, 17776
    qkv = torch.randn(B, S, 3, H, 64, device=DEV).to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, device=DEV, dtype=torch.bfloat16)
    for splits in ("0", "auto", "0", "auto"):     # key-split tail of the BASELINE shape (plan_kv_split picks 5 chunks; 2..8 measured within 0.1 %)
        ops.TUNING["attn_no_split"] = splits == "0"
        dt = timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], out=out), iters=20, warm=3)
        print(f"attn S={S} kv_splits={splits}: {dt*1e3:.3f} ms  {4.0*B*H*S*S*64/dt/1e12:.1f} TF/s")
    ops.TUNING["attn_no_split"] = True
    for sq in (64 * 256, 17776 - 112, 17776):
        dt = timeit(lambda: ops.attention(qkv[:, :sq, 0], qkv[:, :, 1], qkv[:, :, 2], out=out[:, :sq]), iters=20, warm=3)
        tiles = -(-sq // 256) * B * H
        print(f"attn Sq={sq:6d} Skv={S}: {dt*1e3:.3f} ms  wgs {tiles} = {tiles/512:.3f} rounds  {4.0*B*H*sq*S*64/dt/1e12:.1f} TF/s  "
              f"{dt*1e6/(sq/256):.2f} us per 256 query rows")


def gemm():
    for name, N, K in (("qkv", 9216, 3072), ("to_out", 3072, 3072), ("ff1", 12288, 3072), ("ff2", 3072, 12288)):
        nt = N // 256
        full = 2 * 17776
        mt = -(-full // 256)
        rounds = mt * nt / 256
        m_even = (int(rounds) * 256 // nt) * 256            # largest M whose tile count fills whole rounds
        for M in (m_even, full):
            x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
            w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            dt = timeit(lambda: ops.linear(x, w, out=out), iters=20, warm=3)
            t = -(-M // 256) * nt
            print(f"gemm {name:7s} M={M:6d}: {dt*1e3:.3f} ms  tiles {t} = {t/256:.3f} rounds  {2.0*M*N*K/dt/1e12:.1f} TF/s")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["attn", "gemm"]):
        globals()[n]()
