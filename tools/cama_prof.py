#!/usr/bin/env python3
"""developer tool: CAMA predict (2 Resamplers + 4-layer encoder, k = 9 retrieved clips + target, CFG) with fixed random features in place of the frozen encoders --
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split of the ~3.3 ms per clip"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util  # noqa: E402
import torch  # noqa: E402

spec = importlib.util.spec_from_file_location("mrag_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from microbench import timeit  # noqa: E402

dev = "cuda"
_, cam, _ = bench.build_models(dev, 1, 13)
batch = {"ref_videos": torch.zeros(1, 9, 16, 3, 8, 8, dtype=torch.bfloat16, device=dev), "video": torch.zeros(1, 16, 3, 8, 8, dtype=torch.bfloat16, device=dev)}
dt = timeit(lambda: cam.predict(batch, do_classifier_free_guidance=True), iters=int(os.environ.get("ITERS", "20")), warm=3)
print(f"CAMA predict: {dt*1e3:.3f} ms per clip")
from motionrag_amd import ops  # noqa: E402
with ops.dispatched() as d:
    cam.predict(batch, do_classifier_free_guidance=True)
print(f"CAMA predict: {sum(d.counts.values())} library launches per clip: {dict(sorted(d.counts.items()))}")
from motionrag_amd.cama import GraphedPredict  # noqa: E402
gp0 = GraphedPredict(cam, do_classifier_free_guidance=True)
gp0(batch)
print(f"CAMA predict as one HIP graph: {timeit(lambda: gp0(batch), iters=30, warm=3)*1e3:.3f} ms per clip")
if os.environ.get("CAMA_AB"):          # side-stream condition branch on / off, eager and as one HIP graph, interleaved
    from motionrag_amd.cama import GraphedPredict
    for rnd in range(3):
        for par in (True, False):
            cam.parallel_branches = par
            e = timeit(lambda: cam.predict(batch, do_classifier_free_guidance=True), iters=20, warm=3)
            gp = GraphedPredict(cam, do_classifier_free_guidance=True)
            gp(batch)
            g = timeit(lambda: gp(batch), iters=20, warm=3)
            print(f"parallel_branches={par}: eager {e*1e3:.3f} ms, one HIP graph {g*1e3:.3f} ms", flush=True)
