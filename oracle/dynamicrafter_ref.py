"""ORACLE (test infrastructure only -- never imported by the product path).

CPU / fp32 restatement of DynamiCrafter's motion-injected cross attention
(src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/attention.py:171-223, `CrossAttention.efficient_forward`).
Pinned against the reference's own class by tests/golden/dc_cross_attention.npz (oracle/gen_golden.py G7).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _heads(t: torch.Tensor, h: int) -> torch.Tensor:
    b, l, w = t.shape
    return t.view(b, l, h, w // h).transpose(1, 2)


def _sdpa(q, k, v):
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(s, dim=-1), v)


def cross_attention(sd: SD, x: torch.Tensor, context: Optional[dict], heads: int, image_scale: float = 1.0, action_scale: float = 1.0):
    """attention.py:171-223: out = SDPA(q, k_text, v_text) [+ s_img SDPA(q, k_ip, v_ip)]
    [+ s_act SDPA(to_q_a(merge(out)), k_a, v_a)]; to_out.  context None -> spatial self-attention."""
    q = F.linear(x, sd["to_q.weight"])                                             # :177
    src = x if context is None else context["prompt"]                             # :178-183
    k = F.linear(src, sd["to_k.weight"])
    v = F.linear(src, sd["to_v.weight"])
    q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)                # :185-188
    out = _sdpa(q, k, v)                                                          # :189
    if "to_k_ip.weight" in sd and context is not None:                            # :191-204
        k_ip = _heads(F.linear(context["image"], sd["to_k_ip.weight"]), heads)
        v_ip = _heads(F.linear(context["image"], sd["to_v_ip.weight"]), heads)
        out = out + image_scale * _sdpa(q, k_ip, v_ip)
    if "to_q_a.weight" in sd and context is not None:                             # :206-220
        merged = out.transpose(1, 2).reshape(x.shape[0], -1, heads * out.shape[-1])
        q_a = _heads(F.linear(merged, sd["to_q_a.weight"]), heads)
        k_a = _heads(F.linear(context["action"], sd["to_k_a.weight"]), heads)
        v_a = _heads(F.linear(context["action"], sd["to_v_a.weight"]), heads)
        out = out + action_scale * _sdpa(q_a, k_a, v_a)
    out = out.transpose(1, 2).reshape(x.shape[0], -1, heads * out.shape[-1])      # :222
    return F.linear(out, sd["to_out.0.weight"], sd["to_out.0.bias"])              # :223
