"""TEST INFRASTRUCTURE -- CPU restatement (torch fp32, two stages as the library runs them) of `kornia.geometry.resize(x, (224, 224), interpolation='bicubic',
align_corners=True, antialias=True)` + `kornia.enhance.normalize`, the `preprocess` of the reference's OpenCLIP image embedders
(`src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/encoders/condition.py:158-166, 271-279, 328-336`).

PARITY UNPINNED: kornia (`requirements.txt:15`, no version) is not installed in this image and not vendored; this restates its published algorithm
(`kornia/geometry/transform/affwarp.py: resize`, `kornia/filters/gaussian.py`, `kornia/filters/kernels.py: gaussian`):
  * nothing to do when the size already matches; blur only when downscaling (`max(factors) > 1`), on both axes then;
  * sigma_axis = max((factor_axis - 1) / 2, 0.001); kernel size int(max(4 sigma, 3)) made odd; window exp(-x^2 / (2 sigma^2)) / sum, x = arange(ks) - ks // 2;
  * separable blur with `reflect` border; then `torch.nn.functional.interpolate(mode='bicubic', align_corners=True)`;
  * normalize: (x - mean) / std per channel, after the reference's (x + 1) / 2.

Only tests/ may import this module."""
from typing import Sequence, Tuple

import torch
import torch.nn.functional as F

CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def blur_geometry(in_hw: Sequence[int], out_hw: Sequence[int]) -> Tuple[bool, Tuple[float, float], Tuple[int, int]]:
    factors = (in_hw[0] / out_hw[0], in_hw[1] / out_hw[1])
    if max(factors) <= 1:
        return False, (0.0, 0.0), (1, 1)
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sigmas[0], 3)), int(max(2.0 * 2 * sigmas[1], 3))]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    return True, sigmas, (ks[0], ks[1])


def gaussian_window(ks: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def resize(x: torch.Tensor, size: Sequence[int] = (224, 224), antialias: bool = True) -> torch.Tensor:
    """x [B, C, H, W] fp32 -> [B, C, size]"""
    H, W = x.shape[-2:]
    if (H, W) == tuple(size):
        return x
    blur, sigmas, ks = blur_geometry((H, W), size)
    if antialias and blur:
        C = x.shape[1]
        gy, gx = gaussian_window(ks[0], sigmas[0]), gaussian_window(ks[1], sigmas[1])
        x = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        x = F.conv2d(x, gx.view(1, 1, 1, -1).expand(C, 1, 1, -1), groups=C)
        x = F.conv2d(x, gy.view(1, 1, -1, 1).expand(C, 1, -1, 1), groups=C)
    return F.interpolate(x, size=tuple(size), mode="bicubic", align_corners=True)


def preprocess(x: torch.Tensor, antialias: bool = True, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """condition.py:328-336: x in [-1, 1] -> CLIP-normalised [B, 3, 224, 224]"""
    y = (resize(x.float(), (224, 224), antialias) + 1.0) / 2.0
    m, s = torch.tensor(mean).view(1, -1, 1, 1), torch.tensor(std).view(1, -1, 1, 1)
    return (y - m) / s
