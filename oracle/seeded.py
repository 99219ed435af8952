"""ORACLE TOOLING: regenerate the seeded weights used by oracle/gen_golden.py (same generator walk: sorted keys, N(0, std), +1 on 1-D
`weight`s), so fixtures only store (seed, std, keys, shapes)."""
import torch


def seeded_sd(keys, shapes, seed: int, std: float):
    g = torch.Generator().manual_seed(int(seed))
    sd = {}
    for k, shp in zip(keys, shapes):
        shp = [int(v) for v in shp]
        v = torch.randn(shp, generator=g) * std
        if k.endswith("weight") and len(shp) == 1:
            v = v + 1.0
        sd[str(k)] = v
    return sd
