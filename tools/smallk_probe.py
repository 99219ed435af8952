#!/usr/bin/env python3
"""Small-K GEMMs of the UNets' level-0 transformers (C = 320): 256-wide tiles (one workgroup per CU) against 128x128 tiles (two per CU,
MRAG_GEMM_CFG=2), where a tile's DMA start-up and epilogue are a large share of its 5-10 K-tiles (developer probe)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from motionrag_amd import ops  # noqa: E402
from microbench import timeit  # noqa: E402

DEV = "cuda"
M = 28 * 9216
outs = {}
for name, N, K, epi in (("ff1 geglu C=320", 2560, 320, "geglu"), ("ff2 C=320", 320, 1280, "resid"), ("qkv C=320", 960, 320, "none"), ("proj C=320", 320, 320, "none"),
                        ("ff1 geglu C=640 (M/4)", 5120, 640, "geglu")):
    m = M if "M/4" not in name else M // 4
    x = torch.randn(m, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=DEV).to(torch.bfloat16)
    res = torch.randn(m, N, device=DEV).to(torch.bfloat16) if epi == "resid" else None
    wg, bg = ops.geglu_interleave(w, b) if epi == "geglu" else (None, None)
    for cfg in (os.environ.get("MRAG_PROBE_CFGS", "0,2,0,2").split(",")):
        ops.TUNING["gemm"] = (int(cfg.split(":")[0]) << 4) | (8 if cfg.endswith(":p") else 0)      # "tile[:p]" -- p = persistent workgroups (MRAG_GEMM_TUNE_PERSIST)
        if epi == "geglu":
            fn = lambda: ops.linear(x, wg, bg, epilogue=ops.EPI_GEGLU)
        elif epi == "resid":
            fn = lambda: ops.linear(x, w, b, epilogue=ops.EPI_RESID, resid=res)
        else:
            fn = lambda: ops.linear(x, w, b)
        y = fn().float()
        ref = outs.setdefault(name, y)
        assert torch.equal(y, ref), f"{name} cfg {cfg}: result differs from the first configuration"
        dt = timeit(fn, iters=20, warm=3)
        print(f"{name:24s} cfg={cfg}: {dt*1e3:.3f} ms  {2.0*m*N*K/dt/1e12:.0f} TF/s")
