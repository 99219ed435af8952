#!/usr/bin/env python3
"""Why the tiled CogVideoX VAE decode (tiles fanned out over three HIP streams) is slower inside bench.py's process (722 ms) than standalone (597 ms):
HIP maps a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4); streams created earlier in the process (graph captures, plans)
take slots, and tile streams that land on one queue serialise.  Usage: vae_queue_probe.py [n_dummy_streams]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
keep = []
for _ in range(n):                       # streams a long-lived process has created before the decode (graph captures, side streams)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device="cuda") + 1)
torch.cuda.synchronize()
import microbench  # noqa: E402
r = microbench.cogvideox_vae()
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')} dummy_streams={n}: tiled {r['decode_tiled_ms_per_clip']} ms, one stream {r['decode_tiled_one_stream_ms_per_clip']} ms, untiled {r['decode_untiled_ms_per_clip']} ms")
