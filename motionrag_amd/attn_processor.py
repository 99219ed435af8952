"""Motion-injection attention processors -- the drop-in boundary.

Same class names, constructor arguments, state-dict keys (`to_{q,k,v}_ip.0.weight`) and call
signatures as the reference's src/projects/condition/attn_processor.py, installed with
`model.set_attn_processor({name: processor})` exactly as src/projects/cogvideox/module.py:163-175 and
src/projects/svd/module.py:145-165 do.  The arithmetic runs on hand-written gfx950 kernels
(libmrag_hip.so): fused QKV GEMM, qk-LayerNorm + RoPE in place, flash attention, and the adapter
branch `hidden += scale * SDPA(to_q_ip(hidden), to_k_ip(ip), to_v_ip(ip))` with the residual update fused
into the small-KV attention epilogue.

`Attention` below is the minimal stand-in for diffusers' `Attention` module (diffusers is not
installed in the build image); a real diffusers `Attention` works too -- the processors only read the
fields listed in SURVEY 8b.1.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops


def packed_score_layout(heads: int, keys: int):
    """(key stride, score columns) of the folded motion branch's score GEMM: the heads' blocks of `keys` scores packed at the smallest even pitch that drops a
    256-column GEMM tile against 32 columns per head (48 heads x 25 keys: pitch 26, 1 280 columns = five tiles instead of six), else (32, 32 * heads).
    A block's keys must fit the 32 slots of the four aligned 16-byte chunks that cover it: ((pitch * h) mod 8) + keys <= 32 for every head
    (mrag_ip_attn_folded_bf16); the column count is a whole number of tiles and holds the last block's aligned 32-element window."""
    tiles32 = -(-(heads * 32) // 256)
    for c in range(keys + (keys & 1), 32, 2):
        if -(-((heads - 1) * c + 32) // 256) < tiles32 and all(((c * h) & 7) + keys <= 32 for h in range(min(heads, 8))):
            return c, -(-((heads - 1) * c + 32) // 256) * 256
    return 32, heads * 32


PACK_SCORE_BLOCKS = True   # developer A/B knob (tools/r6_step_ab.py): False keeps the folded scores at 32 columns per head (round 5's layout)
# fold `to_q_ip` into the motion keys once per clip (joint_attention_core); False reproduces the reference's op order literally
FOLD_IP_QUERY = True


class Attention(nn.Module):
    """Field-compatible stand-in for diffusers.models.attention_processor.Attention."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64, bias: bool = False,
                 out_bias: bool = True, qk_norm: Optional[str] = None, eps: float = 1e-5, processor=None):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.inner_dim = heads, inner
        self.is_cross_attention = cross_attention_dim is not None
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])
        if qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps)
        else:
            self.norm_q = self.norm_k = None
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = processor

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, **kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, **kwargs)


def _wkey(t: torch.Tensor):
    """cache key of a weight tensor: storage address + in-place version (load_state_dict copies in place)"""
    return (t.data_ptr(), t._version)


def _cat_weights(mods, attr="weight"):
    ts = [getattr(m, attr) for m in mods]
    if any(t is None for t in ts):
        return None
    return torch.cat([t.detach() for t in ts], dim=0).contiguous()


class _FusedWeights:
    """per-`Attention` cache of concatenated projection weights (built once, memory plumbing only).  One slot per KIND of fused operand (the first element of
    a tuple key): a key whose weight addresses / versions changed REPLACES the slot, so updated weights do not leave stale copies behind."""

    def __init__(self):
        self._cache = {}

    def get(self, key, builder):
        kind = key[0] if isinstance(key, tuple) else key
        ent = self._cache.get(kind)
        if ent is None or ent[0] != key:
            ent = (key, builder())
            self._cache[kind] = ent
        return ent[1]

    def clear(self):
        self._cache.clear()


def joint_attention_core(attn, proc, x: torch.Tensor, text_len: int, rope, ip_hidden_states: Optional[torch.Tensor], scale: float, sp=None):
    """attn_processor.py:209-273 on the joint [text ; video] sequence x [B, S, D] (bf16, contiguous).
    Returns the attention output BEFORE to_out: `o + scale * ip_attention(to_q_ip(o))`.
    `sp` (dist.SequenceParallel): x holds only this rank's rows (text_len / rope already local); K and V rows of all ranks are
    all-gathered after qk-norm + RoPE, everything else stays local."""
    B, S, D = x.shape
    H = attn.heads
    fw = proc._fused
    wqkv, bqkv = fw.get(("qkv", _wkey(attn.to_q.weight), _wkey(attn.to_k.weight), _wkey(attn.to_v.weight)), lambda: (_cat_weights([attn.to_q, attn.to_k, attn.to_v]),
                                                      _cat_weights([attn.to_q, attn.to_k, attn.to_v], "bias")))
    cos = sin = None
    if rope is not None:
        # single slot, valid only for the SAME table objects (the entry holds references, so their addresses cannot be recycled under it): a
        # pipeline that rebuilds image_rotary_emb per call replaces the entry instead of growing the cache
        ent = fw._cache.get("rope")
        if ent is None or ent[0] is not rope[0] or ent[1] is not rope[1] or ent[2] != (S, text_len, rope[0]._version, rope[1]._version):
            ent = (rope[0], rope[1], (S, text_len, rope[0]._version, rope[1]._version),
                   rope[0].to(device=x.device, dtype=torch.float32).contiguous(), rope[1].to(device=x.device, dtype=torch.float32).contiguous())
            fw._cache["rope"] = ent
        cos, sin = ent[3], ent[4]
        if getattr(attn, "is_cross_attention", False):
            raise NotImplementedError("RoPE on Q only (is_cross_attention=True) is not used by CogVideoX attn1")
    nq, nk = getattr(attn, "norm_q", None), getattr(attn, "norm_k", None)
    if sp is not None:
        o = _sharded_attention(attn, x, wqkv, bqkv, H, nq, nk, cos, sin, text_len, sp)
        return _motion_branch(attn, proc, o, ip_hidden_states, scale)
    # :209-211 + :220-231 in ONE GEMM: the projection's epilogue applies norm_q / norm_k and the rotary embedding to the Q and K thirds
    qkv = ops.qkv_linear_qknorm_rope(x, wqkv, bqkv, H, nq.weight if nq is not None else None, nq.bias if nq is not None else None,
                                     nk.weight if nk is not None else None, nk.bias if nk is not None else None, cos, sin, text_len,
                                     eps=nq.eps if nq is not None else 1e-6, q_premul=ops.LOG2E * 64 ** -0.5)
    q5 = qkv.view(B, S, 3, H, 64)
    o = ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], q_prescaled=True)   # :233-237
    return _motion_branch(attn, proc, o, ip_hidden_states, scale)


def _sharded_attention(attn, x, wqkv, bqkv, H, nq, nk, cos, sin, text_len, sp):
    """Tier-2 sequence sharding: x holds this rank's rows.  The [K | V] thirds are projected FIRST (norm_k + RoPE in the GEMM
    epilogue) straight into this rank's row range of the gathered buffer's layout -- per sample a contiguous [s_loc, 2 D] block -- and their
    all-gather is started asynchronously (RCCL runs it on its own stream over xGMI); the Q third is projected while the gather is in
    flight; the attention waits for the gather.  No permute / contiguous copy: the gathered [B, S, 2, H, 64] buffer is read through strides."""
    B, S, D = x.shape
    qn = dict(eps=nq.eps if nq is not None else 1e-6, q_premul=ops.LOG2E * 64 ** -0.5)
    gq, bq, gk, bk = (nq.weight if nq is not None else None, nq.bias if nq is not None else None,
                      nk.weight if nk is not None else None, nk.bias if nk is not None else None)
    try:
        kv = ops.qkv_linear_qknorm_rope(x, wqkv[D:], bqkv[D:] if bqkv is not None else None, H, gq, bq, gk, bk, cos, sin, text_len, first=1, **qn)   # [B, s_loc, 2 D]
    except NotImplementedError:          # toy sizes (128x128 GEMM tiles carry no fused epilogue): whole projection + the norm / RoPE kernel, then split
        qkv = ops.qkv_linear_qknorm_rope(x, wqkv, bqkv, H, gq, bq, gk, bk, cos, sin, text_len, **qn)
        g = sp.all_gather_rows_async(qkv[..., D:].contiguous()).wait().view(B, -1, 2, H, 64)
        return ops.attention(qkv[..., :D].unflatten(-1, (H, 64)), g[:, :, 0], g[:, :, 1], q_prescaled=True)
    pending = sp.all_gather_rows_async(kv)                                        # -> [B, S_total, 2 D], rank-major rows == global row order
    q = ops.qkv_linear_qknorm_rope(x, wqkv[:D], bqkv[:D] if bqkv is not None else None, H, gq, bq, gk, bk, cos, sin, text_len, first=0, **qn)    # [B, s_loc, D]
    g = pending.wait().view(B, -1, 2, H, 64)
    return ops.attention(q.view(B, S, H, 64), g[:, :, 0], g[:, :, 1], q_prescaled=True)


def _motion_branch(attn, proc, o, ip_hidden_states, scale):
    """attn_processor.py:243-273: o += scale * SDPA(to_q_ip(o), to_k_ip(ip), to_v_ip(ip))"""
    B, S, D = o.shape
    H = attn.heads
    fw = proc._fused
    if ip_hidden_states is not None and scale != 0:                               # :243-249
        ip = ip_hidden_states if ip_hidden_states.dtype == torch.bfloat16 else ip_hidden_states.to(torch.bfloat16)
        ip = ip.contiguous()
        r = B // ip.size(0)                                                       # :254
        wkv = fw.get(("ipkv", _wkey(proc.to_k_ip[0].weight), _wkey(proc.to_v_ip[0].weight)), lambda: _cat_weights([proc.to_k_ip[0], proc.to_v_ip[0]]))
        if FOLD_IP_QUERY and ip.size(1) <= 32:
            # The motion tokens are fixed for a clip, so to_q_ip is folded into the keys ONCE per clip:
            #   to_q_ip(o)_h . K_h^T = o . (K_h . Wq_h)^T = o . M_h^T,  M [B', NW, D]: head h's nk keys in rows KS h .. KS h + nk - 1
            # and every step runs a [S, D] x [D, NW] GEMM instead of the [S, D] x [D, D] projection (less than half the flops of :250) plus one
            # kernel that finishes softmax . V_ip and the `o + scale * ip` update (:264-273) in place.  KS = the heads' row pitch: the smallest even
            # value >= nk that lets the GEMM drop a 256-column tile (25 keys x 48 heads: KS = 26, NW = 1 280 = five tiles instead of the six of
            # KS = 32), else 32.  Both CFG samples' score GEMMs are ONE launch with per-sample weights (ops.linear_per_sample): 700 tiles = 3 rounds of
            # the persistent grid where two launches of 420 paid 2 + 2 (round 6).
            nk = ip.size(1)
            KS, NW = packed_score_layout(H, nk) if PACK_SCORE_BLOCKS else (32, H * 32)

            def build():
                kv0 = ops.linear(ip, wkv)                                         # :251-252 (one GEMM)
                Bp, nk = ip.size(0), ip.size(1)
                wq_t = proc.to_q_ip[0].weight.detach().t().contiguous()           # [D_in, D_out]: the contraction index (h, d) must be contiguous
                # all heads in ONE GEMM per sample: the keys as a block-diagonal [32 H, D] operand (head h's keys in rows 32 h.., columns
                # 64 h..; zeros elsewhere add exact zeros to the fp32 accumulators, so every M_h is what its own [nk, 64] x [64, D] product
                # gives) -- 48x the flops of the per-head products, 2.5 ms per clip on the DiT, instead of 96 launches per layer
                k4 = kv0[..., :D].unflatten(-1, (H, 64))                          # [B', nk, H, 64]
                A = torch.zeros(Bp, NW, H, 64, dtype=torch.bfloat16, device=ip.device)
                hidx = torch.arange(H, device=ip.device)
                rows = (hidx[:, None] * KS + torch.arange(nk, device=ip.device)[None, :])   # [H, nk]: row of (head, key)
                A[:, rows, hidx[:, None]] = k4.permute(0, 2, 1, 3)                # [B', H, nk, 64] into rows KS h + k, head block h
                M = torch.empty(Bp, NW, D, dtype=torch.bfloat16, device=ip.device)
                for bp in range(Bp):
                    ops.linear(A[bp].view(NW, D), wq_t, out=M[bp])
                return M, kv0[..., D:], KS
            # cache hit only for the SAME tensor object at the same version: the entry keeps a reference to `ip`, so its address cannot be
            # recycled for another clip's tokens while the entry lives (a data_ptr key alone would go stale silently)
            ent = fw._cache.get("ipfold")
            if ent is None or ent[0] is not ip or ent[1] != ip._version or ent[2] != (_wkey(proc.to_q_ip[0].weight), _wkey(proc.to_k_ip[0].weight), _wkey(proc.to_v_ip[0].weight)) or ent[5] != KS:
                ent = (ip, ip._version, (_wkey(proc.to_q_ip[0].weight), _wkey(proc.to_k_ip[0].weight), _wkey(proc.to_v_ip[0].weight))) + build()
                fw._cache["ipfold"] = ent
            M, v_ip = ent[3], ent[4]
            sc = ops.linear_per_sample(o, M, samples_per_weight=r)                # [B, S, NW]
            ops.ip_attn_folded_(sc, v_ip, o, H, ip.size(1), kv_batch_div=r, scale=0.125, out_scale=float(scale), key_stride=ent[5])
        else:
            ip_q = ops.linear(o, proc.to_q_ip[0].weight)                          # :250  (text tokens included)
            kv = ops.linear(ip, wkv)                                              # :251-252 (one GEMM)
            k_ip = kv[..., :D].unflatten(-1, (H, 64))
            v_ip = kv[..., D:].unflatten(-1, (H, 64))
            # :264-273  o = o + scale * SDPA(ip_q, ip_k, ip_v), residual fused in the attention epilogue (in place)
            ops.attention(ip_q.view(B, S, H, 64), k_ip, v_ip, out=o, resid=o, kv_batch_div=r, out_scale=float(scale))
    return o


class APAdapterCogVideoXAttnProcessor2_0(nn.Module):
    """attn_processor.py:144-283."""

    def __init__(self, hidden_size, cross_attention_dim=None, num_tokens=(4,), scale=1.0):
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        if not isinstance(num_tokens, (tuple, list)):
            num_tokens = [num_tokens]
        self.num_tokens = num_tokens
        if not isinstance(scale, list):
            scale = [scale] * len(num_tokens)
        if len(scale) != len(num_tokens):
            raise ValueError("`scale` should be a list of integers with the same length as `num_tokens`.")
        self.scale = scale
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in num_tokens])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in num_tokens])
        self.to_q_ip = nn.ModuleList([nn.Linear(hidden_size, hidden_size, bias=False) for _ in num_tokens])
        self._fused = _FusedWeights()

    def _unpack(self, image_rotary_emb, action_hidden_states):
        if isinstance(image_rotary_emb, tuple) and isinstance(image_rotary_emb[1], torch.Tensor) and isinstance(image_rotary_emb[0], tuple):
            return image_rotary_emb                                                # ((cos, sin), ip)  :189-190
        assert action_hidden_states is not None, "action_hidden_states must be provided"   # :192
        return image_rotary_emb, action_hidden_states

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor,
                 action_hidden_states: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 image_rotary_emb=None):
        if attention_mask is not None:
            raise NotImplementedError("CogVideoX attn1 is called without a mask (SURVEY 2.2 K1)")
        rope, ip_hidden_states = self._unpack(image_rotary_emb, action_hidden_states)
        if len(self.num_tokens) != 1:
            raise NotImplementedError("one adapter branch (the shipped configs use num_tokens=(4,))")
        text_len = encoder_hidden_states.size(1)
        x = torch.cat([encoder_hidden_states, hidden_states], dim=1)               # :199
        o = joint_attention_core(attn, self, x, text_len, rope, ip_hidden_states, self.scale[0])
        out = ops.linear(o, attn.to_out[0].weight, attn.to_out[0].bias)            # :276 (dropout p=0 :278)
        return out[:, text_len:], out[:, :text_len]                               # :280-283


class APAdapterAttnProcessor2_0(nn.Module):
    """attn_processor.py:10-141 (SVD cross-attention sites; IPAdapterAttnProcessor2_0 + to_q_ip)."""

    def __init__(self, hidden_size, cross_attention_dim=None, num_tokens=(4,), scale=1.0):
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        if not isinstance(num_tokens, (tuple, list)):
            num_tokens = [num_tokens]
        self.num_tokens = num_tokens
        if not isinstance(scale, list):
            scale = [scale] * len(num_tokens)
        if len(scale) != len(num_tokens):
            raise ValueError("`scale` should be a list of integers with the same length as `num_tokens`.")
        self.scale = scale
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in num_tokens])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in num_tokens])
        self.to_q_ip = nn.ModuleList([nn.Linear(hidden_size, hidden_size, bias=False) for _ in num_tokens])
        self._fused = _FusedWeights()

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states=None, action_hidden_states=None,
                 attention_mask=None, temb=None, scale: float = 1.0, ip_adapter_masks=None, block_residual: Optional[torch.Tensor] = None):
        """`block_residual` (not a diffusers argument): the caller's `x` of `x = attn2(norm2(x)) + x`; added in the output projection's epilogue"""
        residual = hidden_states
        ip_hidden_states = None
        if encoder_hidden_states is not None:                                      # :34-41
            if isinstance(encoder_hidden_states, tuple):
                encoder_hidden_states, ip_hidden_states = encoder_hidden_states
            else:
                assert action_hidden_states is not None, "action_hidden_states must be provided"
                ip_hidden_states = action_hidden_states
        if attention_mask is not None or ip_adapter_masks is not None:
            raise NotImplementedError("SVD attn2 sites run without masks")
        if attn.spatial_norm is not None or attn.group_norm is not None or attn.norm_cross:
            raise NotImplementedError("spatial_norm / group_norm / norm_cross are None on the SVD attn2 sites")
        input_ndim = hidden_states.ndim
        if input_ndim == 4:                                                        # :48-50
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        hidden_states = hidden_states.contiguous()
        B, L, C = hidden_states.shape
        H = attn.heads
        if C != H * 64:
            raise NotImplementedError("head_dim 64 only")
        q = ops.linear(hidden_states, attn.to_q.weight, attn.to_q.bias)            # :65
        enc = hidden_states if encoder_hidden_states is None else encoder_hidden_states.contiguous()
        wkv, bkv = self._fused.get(("kv", _wkey(attn.to_k.weight), _wkey(attn.to_v.weight)), lambda: (_cat_weights([attn.to_k, attn.to_v]), _cat_weights([attn.to_k, attn.to_v], "bias")))
        kv = ops.linear(enc, wkv, bkv)                                             # :72-73
        o = ops.attention(q.view(B, L, H, 64), kv[..., :C].unflatten(-1, (H, 64)), kv[..., C:].unflatten(-1, (H, 64)))   # :85-90
        if ip_hidden_states is not None and self.scale[0] != 0:                    # :93-139
            ip = ip_hidden_states.to(torch.bfloat16).contiguous()
            r = B // ip.size(0)
            ip_q = ops.linear(o, self.to_q_ip[0].weight)
            wip = self._fused.get(("ipkv", _wkey(self.to_k_ip[0].weight), _wkey(self.to_v_ip[0].weight)), lambda: _cat_weights([self.to_k_ip[0], self.to_v_ip[0]]))
            ipkv = ops.linear(ip, wip)
            ops.attention(ip_q.view(B, L, H, 64), ipkv[..., :C].unflatten(-1, (H, 64)), ipkv[..., C:].unflatten(-1, (H, 64)),
                          out=o, resid=o, kv_batch_div=r, out_scale=float(self.scale[0]))
        rescale = float(attn.rescale_output_factor)
        if block_residual is not None:
            if attn.residual_connection or input_ndim == 4:
                raise NotImplementedError("block_residual with residual_connection / 4-D input")
            # the reference divides the PROCESSOR's output (:139) and the caller adds its x afterwards: x + out / f.  The division rides in the
            # epilogue's acc_scale -- resid + (1 / f) (o W^T + b) -- and must not touch the block's residual
            out = ops.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, epilogue=ops.EPI_RESID, resid=block_residual, acc_scale=1.0 / rescale)
            rescale = 1.0
        elif attn.residual_connection and input_ndim != 4:
            out = ops.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, epilogue=ops.EPI_RESID, resid=residual.contiguous())
        else:
            out = ops.linear(o, attn.to_out[0].weight, attn.to_out[0].bias)        # :129-131
        if input_ndim == 4:                                                        # :133-134
            out = out.transpose(-1, -2).reshape(b, c, hh, ww)
            if attn.residual_connection:
                out = ops.add(out.contiguous(), residual.contiguous())
        if rescale != 1.0:                                                         # :139 (1.0 on every SVD attn2 site: one extra pass otherwise); with
            out = out.contiguous()                                                 # attn.residual_connection the reference divides (out + residual) too
            out = ops.axpby(out, out, 1.0 / rescale, 0.0)
        return out
