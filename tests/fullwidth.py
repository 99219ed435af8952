"""Shared loader of the full-width reference fixtures (tests/golden/fullwidth_{cog,svd,dc}.npz; generator oracle/gen_golden_fullwidth.py, which ran the
REFERENCE'S OWN classes).  The fixtures hold (keys, shapes, seed, std) for the weights, the input seeds and the reference's outputs on a sample of
rows / pixels; weights and inputs are regenerated here by the same torch CPU generator walks the generator used."""
import json
import os

import numpy as np
import torch

from oracle import gen_golden_fullwidth as gf          # functions only: nothing under /root/reference is touched at import or by the helpers used here
from oracle.seeded import seeded_sd


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    return g, json.loads(str(g["meta"]))


def weights(m):
    """{key: bf16-representable fp32 tensor} of one seeded module (`meta["attn"]`, `meta["st"]`, ...)"""
    return {k: gf.bf(v) for k, v in seeded_sd(m["keys"], m["shapes"], m["seed"], m["std"]).items()}


cog_inputs, svd_inputs, dc_inputs = gf.cog_inputs, gf.svd_inputs, gf.dc_inputs


def rel_l2(got, want):
    g, w = torch.as_tensor(np.asarray(got)).float(), torch.as_tensor(np.asarray(want)).float()
    assert g.shape == w.shape, (g.shape, w.shape)
    return ((g - w).norm() / w.norm()).item()
