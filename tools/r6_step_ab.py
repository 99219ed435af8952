#!/usr/bin/env python3
"""developer tool (round 6): the headline CFG denoise step under in-process toggles, interleaved on ONE device in ONE process (same box, same power state):
    python3 tools/r6_step_ab.py [rounds] [steps]
  shipped        everything on
  tail_rect      FF1's 16-tile 27th round as its own launch of 128x128 tiles (opt-in MRAG_GEMM_TUNE_TAIL_RECT)
  loop_scores    the folded score GEMM as one launch per CFG sample (packed 26-column blocks kept)
  r5_scores      round 5's score path: one launch per sample, 32 columns per head
Prints ms per step per configuration and round, then the medians."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from motionrag_amd import attn_processor, ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
torch.cuda.set_device(0)
dit, cam, pipe = bench.build_models(dev, 42, 13)
g = torch.Generator().manual_seed(1234)
latents = torch.randn(1, 13, 16, 60, 90, generator=g).to(dev, torch.bfloat16)
image_latents = torch.randn(1, 13, 16, 60, 90, generator=g).to(dev, torch.bfloat16)
prompt = torch.randn(2, 226, 4096, generator=g).to(dev, torch.bfloat16)
ref_videos = torch.zeros(1, 9, 16, 3, 8, 8, dtype=torch.bfloat16, device=dev)
image = torch.zeros(1, 3, 8, 8, dtype=torch.bfloat16, device=dev)
pipe.action_emb = pipe.prepare_action_embeddings(ref_videos, None, do_classifier_free_guidance=True, image=image)
rope_ip = pipe._prepare_rotary_positional_embeddings(13, 30, 45, dev)
sched = pipe.scheduler
ts = sched.set_timesteps(50)


def step(i):
    t = int(ts[i % 50])
    timestep = torch.full((2,), float(t), dtype=torch.float32, device=dev)
    v = dit(latents, prompt, timestep, image_rotary_emb=rope_ip, image_latents=image_latents, batch=2)
    ops.cfg_ddim_step_(v, latents.clone(), 6.0, *sched.coeffs(t))


CONFIGS = {"shipped": dict(gemm=0, loop=False, pack=True), "tail_rect": dict(gemm=1 << 19, loop=False, pack=True),
           "loop_scores": dict(gemm=0, loop=True, pack=True), "r5_scores": dict(gemm=0, loop=True, pack=False)}
res = {k: [] for k in CONFIGS}
for r in range(rounds + 1):
    for name, c in CONFIGS.items():
        ops.TUNING["gemm"], ops.TUNING["no_batched_w"], attn_processor.PACK_SCORE_BLOCKS = c["gemm"], c["loop"], c["pack"]
        step(0)                                   # rebuilds the folded weights when the layout changed; not timed
        torch.cuda.synchronize()
        with ops.dispatched() as d:
            t0 = time.perf_counter()
            for i in range(steps):
                step(i + 1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
        if r:                                      # round 0 warms up
            res[name].append(ms)
            print(f"round {r} {name:13s} {ms:8.2f} ms/step   launches per step: GEMM_W4 {d.counts.get('GEMM_W4', 0) // steps}, BATCHED_W {d.counts.get('GEMM_W4_BATCHED_W', 0) // steps}, "
                  f"TAIL_RECT {d.counts.get('GEMM_W4_TAIL_RECT', 0) // steps}, LN_STREAM {d.counts.get('LAYERNORM_STREAM', 0) // steps}", flush=True)
print("medians:", {k: round(statistics.median(v), 2) for k, v in res.items()})
