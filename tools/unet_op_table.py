#!/usr/bin/env python3
"""developer tool: WHERE a UNet CFG step spends its time, per (op, shape, epilogue) -- every C-ABI call of one step bracketed by HIP events on the
launch stream (serialising nothing: events are recorded in stream order), aggregated and sorted by total time.

    python tools/unet_op_table.py svd|dc [--steps 3] [--out gpurun_out/r5_svd_ops.json]

Columns: calls per step, total ms per step, average us, algorithmic TFLOP/s (GEMM / convolution / attention) or GB/s (norms, pointwise: bytes read + written)."""
import argparse
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from motionrag_amd import ops, workloads as W  # noqa: E402

REC = []


def timed(name, sig, flops, nbytes, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = fn()
    e1.record()
    REC.append((name, sig, flops, nbytes, e0, e1))
    return y


def install():
    lin, att, cimp, gn, ln, addb, axp, add, silu = ops.linear, ops.attention, ops.conv_implicit, ops.groupnorm, ops.layernorm, ops.add_bcast, ops.axpby, ops.add, ops.silu

    def linear(x, w, bias=None, **k):
        M, K, N = x.numel() // x.shape[-1], x.shape[-1], w.shape[0]
        epi = k.get("epilogue", 0)
        return timed("linear", f"[{M} x {N} x {K}] epi {epi}", 2.0 * M * N * K, 2 * (M * K + N * K + M * (N // 2 if epi == ops.EPI_GEGLU else N) * (2 if k.get('resid') is not None else 1)),
                     lambda: lin(x, w, bias, **k))

    def attention(q, k_, v, **k):
        B, Sq, H, _ = q.shape
        Skv = k_.shape[1]
        return timed("attention", f"B {B} H {H} Sq {Sq} Skv {Skv}" + (" fp8" if k.get("fp8") else "") + (" +resid" if k.get("resid") is not None else ""),
                     4.0 * B * H * Sq * Skv * 64, 2 * (2 * B * Sq * H * 64 + 2 * k_.shape[0] * Skv * H * 64), lambda: att(q, k_, v, **k))

    def conv_implicit(x, wk, bias, mode, **k):
        box = {}

        def run():
            box["y"] = cimp(x, wk, bias, mode, **k)
            return box["y"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = run()
        e1.record()
        M = y.numel() // y.shape[-1]
        tag = "conv3x3" if mode == ops.CONV_3X3 else "conv_t3"
        extra = "".join(f" {n}" for n in ("stride", "upsample") if k.get(n) not in (None, 1, False)) + (" +resid" if k.get("resid") is not None else "")
        REC.append((tag, f"[{M} x {wk.shape[0]} x {wk.shape[1]}] in {tuple(x.shape)}{extra}", 2.0 * M * wk.shape[0] * wk.shape[1],
                    2 * (x.numel() + wk.numel() + y.numel() * (2 if k.get("resid") is not None else 1)), e0, e1))
        return y

    def groupnorm(x, *a, **k):
        return timed("groupnorm", f"{tuple(x.shape)} silu {int(bool(k.get('silu')))} emb {int(k.get('emb') is not None)}", 0.0, 2 * 3 * x.numel(), lambda: gn(x, *a, **k))

    def layernorm(x, *a, **k):
        return timed("layernorm", f"[{x.numel() // x.shape[-1]} x {x.shape[-1]}]", 0.0, 2 * 2 * x.numel(), lambda: ln(x, *a, **k))

    def add_bcast(x, *a, **k):
        return timed("add_bcast", f"{x.numel() // x.shape[-1]} x {x.shape[-1]}", 0.0, 2 * 2 * x.numel(), lambda: addb(x, *a, **k))

    def axpby(x, *a, **k):
        return timed("axpby", f"{x.numel()}", 0.0, 2 * 3 * x.numel(), lambda: axp(x, *a, **k))

    def add_(x, *a, **k):
        return timed("add", f"{x.numel()}", 0.0, 2 * 3 * x.numel(), lambda: add(x, *a, **k))

    def silu_(x, *a, **k):
        return timed("silu", f"{x.numel()}", 0.0, 2 * 2 * x.numel(), lambda: silu(x, *a, **k))

    ops.linear, ops.attention, ops.conv_implicit, ops.groupnorm, ops.layernorm, ops.add_bcast, ops.axpby, ops.add, ops.silu = (
        linear, attention, conv_implicit, groupnorm, layernorm, add_bcast, axpby, add_, silu_)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["svd", "dc"])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = "cuda"
    if args.which == "svd":
        net, _ = W.svd_unet(dev)
        step, lat, reset = W.svd_step(net, dev)
    else:
        net = W.dynamicrafter1024_unet(dev)
        x, ts, ctx, fs = W.dynamicrafter1024_inputs(dev)
        step = lambda: net(x, ts, context=ctx, fs=fs)  # noqa: E731
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    install()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / args.steps
    agg = collections.OrderedDict()
    for name, sig, fl, nb, a, b in REC:
        ent = agg.setdefault((name, sig), [0, 0.0, fl, nb])
        ent[0] += 1
        ent[1] += a.elapsed_time(b)
    rows = []
    for (name, sig), (n, ms, fl, nb) in agg.items():
        per = ms / n * 1e3
        rows.append({"op": name, "shape": sig, "calls_per_step": n / args.steps, "ms_per_step": ms / args.steps, "avg_us": per,
                     "tflops": fl / (per * 1e-6) / 1e12 if fl else None, "GBps": nb / (per * 1e-6) / 1e9})
    rows.sort(key=lambda r: -r["ms_per_step"])
    covered = sum(r["ms_per_step"] for r in rows)
    print(f"{args.which}: {total:.2f} ms per step with the event pairs in the stream; {covered:.2f} ms inside the bracketed ops")
    by_op = collections.defaultdict(float)
    for r in rows:
        by_op[r["op"]] += r["ms_per_step"]
    print("  " + "  ".join(f"{k} {v:.2f}" for k, v in sorted(by_op.items(), key=lambda kv: -kv[1])))
    for r in rows[:70]:
        rate = f"{r['tflops']:7.0f} TF/s" if r["tflops"] else f"{r['GBps']:7.0f} GB/s"
        print(f"{r['ms_per_step']:7.3f} ms  {r['calls_per_step']:5.1f} x {r['avg_us']:8.1f} us  {rate}  {r['op']:10s} {r['shape']}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"workload": args.which, "ms_per_step": total, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
