#!/usr/bin/env python3
"""Do the two CFG samples run faster as two concurrent streams (half-size launches that fill each other's tails and overlap the
memory-bound kernels with the compute-bound ones) than as one batched launch sequence?  (developer probe)
One DiT-layer-like sequence per sample: LN -> QKV GEMM -> attention -> to_out GEMM -> LN -> FF1 (gelu) -> FF2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402

DEV = "cuda"
S, D, H = 17776, 3072, 48


def make(B):
    g = lambda *s: torch.randn(*s, device=DEV).to(torch.bfloat16)
    return dict(x=g(B, S, D), ln=torch.ones(D, device=DEV, dtype=torch.bfloat16), wqkv=g(3 * D, D) * 0.02, wo=g(D, D) * 0.02, w1=g(4 * D, D) * 0.02,
                w2=g(D, 4 * D) * 0.02, qkv=torch.empty(B, S, 3 * D, device=DEV, dtype=torch.bfloat16), att=torch.empty(B, S, D, device=DEV, dtype=torch.bfloat16),
                h=torch.empty(B, S, D, device=DEV, dtype=torch.bfloat16), f=torch.empty(B, S, 4 * D, device=DEV, dtype=torch.bfloat16),
                y=torch.empty(B, S, D, device=DEV, dtype=torch.bfloat16), B=B)


def layer(t, w):
    B = t["B"]
    ops.layernorm(t["x"].view(B * S, D), t["ln"], t["ln"], 1e-5, out=t["h"].view(B * S, D))
    ops.linear(t["h"].view(B * S, D), w["wqkv"], out=t["qkv"].view(B * S, 3 * D))
    q = t["qkv"].view(B, S, 3, H, 64)
    ops.attention(q[:, :, 0], q[:, :, 1], q[:, :, 2], out=t["att"])
    ops.linear(t["att"].view(B * S, D), w["wo"], out=t["h"].view(B * S, D))
    ops.layernorm(t["h"].view(B * S, D), t["ln"], t["ln"], 1e-5, out=t["y"].view(B * S, D))
    ops.linear(t["y"].view(B * S, D), w["w1"], out=t["f"].view(B * S, 4 * D), epilogue=ops.EPI_GELU_TANH)
    ops.linear(t["f"].view(B * S, 4 * D), w["w2"], out=t["y"].view(B * S, D))


def main():
    w = make(2)
    t2 = w
    t1a, t1b = make(1), make(1)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    layers = 6

    def batched():
        for _ in range(layers):
            layer(t2, w)

    def two_streams():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        for _ in range(layers):
            with torch.cuda.stream(sa):
                layer(t1a, w)
            with torch.cuda.stream(sb):
                layer(t1b, w)
        cur.wait_stream(sa); cur.wait_stream(sb)

    def one_stream_halves():
        for _ in range(layers):
            layer(t1a, w)
            layer(t1b, w)

    for name, fn in (("batched B=2", batched), ("two streams, B=1 each", two_streams), ("one stream, B=1 twice", one_stream_halves), ("batched B=2", batched),
                     ("two streams, B=1 each", two_streams)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:26s}: {e0.elapsed_time(e1) / 3 / layers:.3f} ms per layer (both samples)")


if __name__ == "__main__":
    main()
