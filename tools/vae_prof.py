#!/usr/bin/env python3
"""developer tool: one tiled CogVideoX VAE decode at the shipped size under rocprofv3 --kernel-trace --stats (tools/microbench.py times it; this is the
kernel breakdown): python3 tools/vae_prof.py [streams]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from motionrag_amd.cogvideox_vae import AutoencoderKLCogVideoX  # noqa: E402

torch.manual_seed(0)
m = AutoencoderKLCogVideoX().to("cuda", torch.bfloat16)
m.enable_tiling()
m.tile_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 3
z = torch.randn(1, 16, 13, 60, 90, device="cuda").to(torch.bfloat16)
for _ in range(3):
    y = m.decode(z).sample
torch.cuda.synchronize()
print(tuple(y.shape))
