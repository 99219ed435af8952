"""Deterministic stand-ins for the THIRD-PARTY, out-of-scope pieces around the DynamiCrafter hot path (OpenCLIP image embedder, text
encoder, first-stage VAE, the CAMA feature encoders): TEST INFRASTRUCTURE ONLY.  The golden generator (oracle/gen_golden.py) hands
them to the reference's own `image_guided_synthesis`; the GPU tests hand the very same objects to the product pipeline, so both
sides see identical conditioning inputs.  Plain torch ops, device-agnostic."""
from __future__ import annotations

import hashlib

import torch
import torch.nn.functional as F
from torch import nn


class ImageEmbedderStub(nn.Module):
    """img [b, 3, H, W] -> tokens [b, tokens, dim]: depends on the per-sample mean only, so `zeros_like(img)` gives a fixed embedding"""

    def __init__(self, tokens=9, dim=48, seed=401):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("base", torch.randn(tokens, dim, generator=g))
        self.register_buffer("dirn", torch.randn(tokens, dim, generator=g))

    def forward(self, img):
        m = img.reshape(img.shape[0], -1).float().mean(dim=1)
        return (self.base[None] + m[:, None, None] * self.dirn[None]).to(img.dtype)


class TextStub:
    """prompts list[str] -> [b, tokens, dim] seeded by the md5 of each prompt"""

    def __init__(self, tokens=7, dim=64, device="cpu", dtype=torch.float32):
        self.tokens, self.dim, self.device, self.dtype = tokens, dim, device, dtype

    def __call__(self, prompts):
        out = []
        for p in prompts:
            seed = int(hashlib.md5(p.encode()).hexdigest()[:8], 16)
            out.append(torch.randn(self.tokens, self.dim, generator=torch.Generator().manual_seed(seed)))
        return torch.stack(out).to(self.device, self.dtype)


class FirstStageStub(nn.Module):
    """encode: [n, 3, H, W] -> [n, 4, H/8, W/8] (8x8 average pool + fixed channel mix); decode: [b, 4, t, h, w] -> [b, 3, t, 8h, 8w]"""

    def __init__(self, seed=402):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("enc", torch.randn(4, 3, generator=g))
        self.register_buffer("dec", torch.randn(3, 4, generator=g) * 0.5)

    def encode_first_stage(self, x):
        p = F.avg_pool2d(x.float(), 8)
        return torch.einsum("oc,nchw->nohw", self.enc, p).to(x.dtype)

    def decode_first_stage(self, z):
        y = torch.einsum("oc,bcthw->bothw", self.dec, z.float())
        return torch.tanh(F.interpolate(y, scale_factor=(1, 8, 8), mode="nearest"))


class ConditionTransformerStub(nn.Module):
    """CAMA protocol: predict(batch) -> [b, 25, dim]; encode_vision(videos [b, k, t, c, h, w]) -> [b, k, 25, dim]"""

    def __init__(self, dim=64, seed=403):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("base", torch.randn(25, dim, generator=g))
        self.register_buffer("d_ref", torch.randn(25, dim, generator=g))
        self.register_buffer("d_vid", torch.randn(25, dim, generator=g))

    def encode_vision(self, videos):
        b, k = videos.shape[:2]
        m = videos.reshape(b, k, -1).float().mean(dim=2)
        return self.base[None, None] + m[:, :, None, None] * self.d_vid[None, None]

    def predict(self, batch, do_classifier_free_guidance=False):
        r, v = batch["ref_videos"], batch["video"]
        mr = r.reshape(r.shape[0], -1).float().mean(dim=1)
        mv = v.reshape(v.shape[0], -1).float().mean(dim=1)
        return self.base[None] + mr[:, None, None] * self.d_ref[None] + mv[:, None, None] * self.d_vid[None]
