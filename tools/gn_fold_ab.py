"""developer tool: UNet CFG steps with the GroupNorm fold by the sample's last-arriving statistics workgroup (ops.TUNING["gn_fold"]) vs the separate fold launch,
interleaved in ONE process on one box (SVD 14x576x1024, DynamiCrafter-1024 16x576x1024 bf16 / fp8 attention)"""
import sys
sys.path.insert(0, "tools")
import torch, microbench as mb
from motionrag_amd import ops, workloads as W, dynamicrafter as dc

DEV = "cuda"
net_s, _ = W.svd_unet(DEV)
step_s, _, _ = W.svd_step(net_s, DEV)
net_d = W.dynamicrafter1024_unet(DEV)
x, ts, ctx, fs = W.dynamicrafter1024_inputs(DEV)
step_d = lambda: net_d(x, ts, context=ctx, fs=fs)
for r in range(3):
    for fold in (False, True):
        ops.TUNING["gn_fold"] = fold
        with ops.dispatched() as d:
            step_s()
        n = {k: v for k, v in d.counts.items() if k.startswith("GN_")}
        t_s = mb.timeit(step_s, iters=5, warm=2)
        dc.set_attention_precision(net_d, "bf16")
        t_d = mb.timeit(step_d, iters=5, warm=2)
        dc.set_attention_precision(net_d, "fp8")
        t_8 = mb.timeit(step_d, iters=5, warm=2)
        dc.set_attention_precision(net_d, "bf16")
        print(f"gn_fold={int(fold)}: SVD {t_s*1e3:.1f} ms  DynamiCrafter bf16 {t_d*1e3:.1f} ms  fp8 {t_8*1e3:.1f} ms   (SVD step launches {n})", flush=True)
ops.TUNING["gn_fold"] = False
