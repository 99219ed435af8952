"""Thin torch-tensor front end of the C ABI (include/mrag_hip.h).

torch is plumbing here: it owns device memory and the HIP stream; every arithmetic op below is a
hand-written gfx950 kernel in libmrag_hip.so.  There is no CPU path: tensors must be bf16 (or fp32 /
int32 where stated) on a ROCm device, otherwise `HipOnly` is raised.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import AttnArgs, GemmArgs, LnArgs, QkNormRopeArgs, check

EPI_NONE, EPI_GELU_TANH, EPI_GELU_ERF, EPI_RESID, EPI_GATE_RESID, EPI_SILU, EPI_GEGLU, EPI_QKNORM_ROPE = range(8)
LOG2E = 1.4426950408889634


# Developer tuning knobs (tools/microbench.py, A/B tests): explicit `tuning` fields of the C-ABI argument structs, 0 = the shipped behaviour.
# Nothing on the launch path reads the environment.
TUNING = {"gemm": 0, "attn": 0, "attn_no_split": False, "no_qkv_fuse": False, "no_batched_w": False, "gn_fold": False}
GEMM_TUNE_NO_WIDE, GEMM_TUNE_NO_STAGED, GEMM_TUNE_GEGLU_NO_STAGED = 1, 2, 4
ATTN_TUNE_NO_TINY, ATTN_TUNE_LEGACY = 1, 8


class HipOnly(RuntimeError):
    """Raised when an op is handed a tensor that is not on the GPU (no CPU fallback exists)."""


def dispatch_counts() -> dict:
    """the library's diagnostic launch counters by kernel name (include/mrag_hip.h: enum mrag_kernel_id) -- which kernel each call dispatched to"""
    L = _lib.lib()
    n = L.mrag_dispatch_counts(None, 0)
    buf = (ctypes.c_uint64 * n)()
    L.mrag_dispatch_counts(buf, n)
    return {L.mrag_dispatch_name(i).decode(): int(buf[i]) for i in range(n)}


class dispatched:
    """`with ops.dispatched() as d: module(x)` -> d.counts = {kernel name: launches inside the block} (non-zero entries only).  Tests use it to
    assert that a full-width shape ran on the production kernel and not on a fallback tile."""

    def __enter__(self):
        self._before = dispatch_counts()
        self.counts = {}
        return self

    def __exit__(self, *exc):
        after = dispatch_counts()
        self.counts = {k: after[k] - self._before[k] for k in after if after[k] != self._before[k]}
        return False


def _dev(t: torch.Tensor, dtype=torch.bfloat16, name="tensor") -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipOnly(f"{name}: motionrag_amd ops run only on a ROCm GPU (got {getattr(t, 'device', type(t))})")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(t: torch.Tensor) -> torch.Tensor:
    """view as [rows, D] with D contiguous"""
    if t.stride(-1) != 1:
        raise ValueError("innermost dimension must be contiguous")
    return t.reshape(-1, t.shape[-1]) if t.dim() != 2 else t


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, out: Optional[torch.Tensor] = None,
           epilogue: int = EPI_NONE, resid: Optional[torch.Tensor] = None, gate0: Optional[torch.Tensor] = None,
           gate1: Optional[torch.Tensor] = None, rows_per_batch: int = 0, split: int = 0, gate_stride: int = 0, geglu_tanh: bool = False,
           acc_scale: float = 1.0) -> torch.Tensor:
    """out = epilogue(x @ weight.T + bias); x [..., K] bf16, weight [N, K] bf16 (nn.Linear layout).  EPI_GEGLU: `geglu_tanh` picks gelu_tanh (T5's gated-gelu) over gelu_erf.
    EPI_RESID: out = resid + acc_scale * (x @ weight.T + bias)."""
    _dev(x, name="x"); _dev(weight, name="weight")
    x2 = _rows(x)
    M, K = x2.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight {tuple(weight.shape)} does not match K={K}")
    if K % 64 != 0:   # the kernel streams whole 64-deep K-tiles: zero-pad odd reduction dims (memory only; no shipped shape needs it)
        pad = 64 - K % 64
        x2 = torch.nn.functional.pad(x2, (0, pad))
        weight = torch.nn.functional.pad(weight, (0, pad))
        K += pad
    if out is None:
        out = torch.empty(*x.shape[:-1], N // 2 if epilogue == EPI_GEGLU else N, dtype=torch.bfloat16, device=x.device)
    o2 = _rows(out)
    if epilogue == EPI_RESID and acc_scale == 0.0:
        # resid + 0 * (...) is the residual itself.  The C struct keeps 0 as "unset = 1" (zero-initialised callers), so an exact zero -- a saturated
        # AlphaBlender, svd_unet: 1 - sigmoid(mix_factor) -- never reaches it: the product is skipped, not silently scaled by 1
        o2.copy_(_rows(_dev(resid, name="resid")))
        return out
    a = GemmArgs()
    a.A, a.W, a.bias, a.C = _p(x2), _p(weight), _p(bias), _p(o2)
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldw, a.ldc = x2.stride(0), weight.stride(0), o2.stride(0)
    a.epilogue = epilogue
    a.tuning = TUNING["gemm"]
    a.geglu_act = 1 if geglu_tanh else 0
    a.acc_scale = acc_scale
    if resid is not None:
        r2 = _rows(_dev(resid, name="resid"))
        a.resid, a.ldr = _p(r2), r2.stride(0)
    if epilogue == EPI_GATE_RESID:
        a.gate0, a.gate1 = _p(_dev(gate0, name="gate0")), _p(_dev(gate1, name="gate1"))
        a.rows_per_batch, a.split, a.gate_stride = rows_per_batch, split, gate_stride
    _gemm_workspace(a, x.device)
    check(_lib.lib().mrag_gemm_bf16(_stream(), ctypes.byref(a)), "mrag_gemm_bf16")
    return out


def linear_ln(x: torch.Tensor, ln_weight: Optional[torch.Tensor], ln_bias: Optional[torch.Tensor], eps: float, weight: torch.Tensor, *,
              epilogue: int = EPI_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epilogue(LayerNorm(x) @ weight.T): ONE launch where the few-row kernel takes it (mrag_gemm_args.a_ln: M <= 256, EPI_NONE / EPI_GELU_ERF; bit-identical
    to layernorm + linear), else the two launches.  CAMA's `to_q(norm2(latents))` and `gelu(ff1(ln(latents)))`."""
    _dev(x, name="x"); _dev(weight, name="weight")
    x2 = _rows(x)
    M, K = x2.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=torch.bfloat16, device=x.device)
    if K % 64 == 0 and weight.shape[1] == K and x2.stride(0) == K and weight.is_contiguous():
        o2 = _rows(out)
        a = GemmArgs()
        a.A, a.W, a.C = _p(x2), _p(weight), _p(o2)
        a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, o2.stride(0)
        a.epilogue, a.tuning = epilogue, TUNING["gemm"]
        a.a_ln, a.a_ln_gamma, a.a_ln_beta, a.a_ln_eps = 1, _p(ln_weight), _p(ln_bias), eps
        rc = _lib.lib().mrag_gemm_bf16(_stream(), ctypes.byref(a))
        if rc == 0:
            return out
        if rc != _lib.MRAG_ENOTSUP:
            check(rc, "mrag_gemm_bf16 (LayerNorm in the A load)")
    return linear(layernorm(x, ln_weight, ln_bias, eps), weight, out=out, epilogue=epilogue)


def _gemm_workspace(a: "GemmArgs", device: torch.device) -> None:
    """with TUNING["gemm"] & GEMM_TUNE_STREAMK: hand mrag_gemm_bf16 the scratch that turns a partial last round of 256x256 tiles into a stream-K tail (include/mrag_hip.h); the same grow-only
    per-(device, stream) buffers as the attention's: launches on one stream are ordered"""
    if M_SK_MIN_ROWS > a.M or not TUNING["gemm"] & GEMM_TUNE_STREAMK:      # opt-in: measured slower than the partial last round
        return
    need = _lib.lib().mrag_gemm_workspace_bytes(a.M, a.N, a.K)
    if need > 0:
        ws = _attn_workspace(device, need, "gemm")
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()


M_SK_MIN_ROWS = 4096          # below this no 256x256 grid reaches a second round: skip the query
GEMM_TUNE_STREAMK = 8
GEMM_TUNE_NO_W4 = 1 << 16      # keep long-K linears on the 8-wave 256x256 tile (A/B knob; default: the persistent four-wave kernel where it applies)
GEMM_TUNE_NO_SKINNY = 1 << 17  # keep few-row problems (M <= 256) on the tiled kernels (tests pin one kernel family with it)
GEMM_TUNE_TAIL_RECT = 1 << 19  # opt-in: a small partial last round of the persistent GEMM as its own launch of 128x128 tiles (measured equal: DESIGN 3.7d)


def ip_attn_folded_(scores: torch.Tensor, v: torch.Tensor, hidden: torch.Tensor, H: int, keys: int, kv_batch_div: int = 1, scale: float = 0.125,
                    out_scale: float = 1.0, key_stride: int = 32) -> torch.Tensor:
    """hidden [B, S, H*64] += out_scale * softmax(scores[..., h, :keys] * scale) @ v[b // kv_batch_div, :, h]; scores [B, S, W] with key k of head h at
    column key_stride * h + k (32: keys padded to 32 per head, W = 32 H; a key_stride in [keys, 32] with ((key_stride h) mod 8) + keys <= 32 packs the heads, W >= (H - 1) key_stride + 32),
    v [B', keys, H*64] (rows may be strided)."""
    _dev(scores, name="scores"); _dev(v, name="v"); _dev(hidden, name="hidden")
    B, S, W = scores.shape
    if W < (H - 1) * key_stride + 32 or not scores.is_contiguous() or hidden.shape != (B, S, H * 64) or not hidden.is_contiguous():
        raise ValueError("ip_attn_folded_: scores [B, S, >= (H - 1) key_stride + 32] and hidden [B, S, H*64], contiguous")
    if v.shape[0] * kv_batch_div != B or v.shape[1] != keys or v.shape[2] != H * 64 or v.stride(2) != 1:
        raise ValueError("ip_attn_folded_: v [B / kv_batch_div, keys, H*64]")
    check(_lib.lib().mrag_ip_attn_folded_bf16(_stream(), _p(scores), _p(v), _p(hidden), B, S, H, keys, kv_batch_div, W, H * 64, v.stride(0), v.stride(1),
                                              float(scale), float(out_scale), key_stride), "mrag_ip_attn_folded_bf16")
    return hidden


def linear_per_sample(x: torch.Tensor, weight: torch.Tensor, *, out: Optional[torch.Tensor] = None, samples_per_weight: int = 1) -> torch.Tensor:
    """out[b] = x[b] @ weight[b // samples_per_weight].T for x [B, S, K], weight [B', N, K]: ONE launch of the persistent GEMM with per-sample weights
    (mrag_gemm_args.w_batch_stride) where the shape allows it, else a launch per sample.  Stands behind the motion branch's folded score GEMM."""
    _dev(x, name="x"); _dev(weight, name="weight")
    B, S, K = x.shape
    Bw, N, Kw = weight.shape
    if Kw != K or Bw * samples_per_weight != B or not x.is_contiguous() or not weight.is_contiguous():
        raise ValueError(f"linear_per_sample: x {tuple(x.shape)} / weight {tuple(weight.shape)} / samples_per_weight {samples_per_weight}")
    if out is None:
        out = torch.empty(B, S, N, dtype=torch.bfloat16, device=x.device)
    if Bw == 1:                                   # one weight for every sample: a plain GEMM over all rows
        return linear(x, weight[0], out=out)
    if samples_per_weight == 1 and B > 1 and K % 64 == 0 and out.is_contiguous() and not TUNING.get("no_batched_w"):
        a = GemmArgs()
        a.A, a.W, a.C = _p(x), _p(weight), _p(out)
        a.M, a.N, a.K = B * S, N, K
        a.lda, a.ldw, a.ldc = K, K, N
        a.epilogue, a.tuning = EPI_NONE, TUNING["gemm"]
        a.rows_per_batch, a.w_batch_stride = S, N * K
        rc = _lib.lib().mrag_gemm_bf16(_stream(), ctypes.byref(a))
        if rc == 0:
            return out
        if rc != _lib.MRAG_ENOTSUP:
            check(rc, "mrag_gemm_bf16 (per-sample weights)")
    for b in range(B):
        linear(x[b], weight[b // samples_per_weight], out=out[b])
    return out


def geglu_interleave(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """re-order the rows of a GEGLU projection ([value rows | gate rows], `chunk(2)` order) into the 16-row [value | gate] groups
    that `linear(..., epilogue=EPI_GEGLU)` expects.  Memory plumbing, done once per weight."""
    n2 = weight.shape[0]
    inner = n2 // 2
    if n2 % 32 != 0:
        raise ValueError("GEGLU inner width must be a multiple of 16")
    idx = torch.arange(inner, device=weight.device).view(-1, 16)
    perm = torch.cat([idx, idx + inner], dim=1).reshape(-1)
    return weight.detach()[perm].contiguous(), (bias.detach()[perm].contiguous() if bias is not None else None)


_ATTN_WS = {}


def _attn_workspace(device: torch.device, nbytes: int, kind: str = "split") -> torch.Tensor:
    """scratch for mrag_attn_fwd_bf16's key-split tail / mrag_attn_fwd_fp8's e4m3 operands, one grow-only buffer per (device, stream, kind):
    launches on one stream are ordered, so consecutive calls may share it"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, kind)
    ws = _ATTN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        # "gn": the first 16 KiB are mrag_groupnorm_bf16's arrival counters -- zero when the buffer is born, left zero by every call (include/mrag_hip.h)
        ws = _ATTN_WS[key] = (torch.zeros if kind == "gn" else torch.empty)(nbytes, dtype=torch.uint8, device=device)
    return ws


def fp8_attention_supported(Sq: int, Skv: int, kv_batch_div: int = 1, mask=None, q_prescaled: bool = False) -> bool:
    """shapes mrag_attn_fwd_fp8 takes (include/mrag_hip.h); everything else stays on the bf16 kernels"""
    return mask is None and kv_batch_div == 1 and not q_prescaled and Skv % 128 == 0 and Skv >= 512 and Sq > 128


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, out: Optional[torch.Tensor] = None,
              resid: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None, kv_batch_div: int = 1,
              scale: Optional[float] = None, out_scale: float = 1.0, q_prescaled: bool = False, fp8: bool = False,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T * scale [masked]) v for head_dim 64.  fp8=True: the e4m3 MFMA path (mrag_attn_fwd_fp8; raises on shapes it does not take).

    q [B, Sq, H, 64], k/v [Bkv, Skv, H, 64] (any strides with the last dim contiguous, e.g. views of a
    fused QKV buffer); out/resid [B, Sq, H*64] with the last two dims packed.  mask: bool/uint8
    [Sq, Skv], True = blocked.  out = resid + out_scale * attention when resid is given.
    """
    for n, t in (("q", q), ("k", k), ("v", v)):
        _dev(t, name=n)
        if t.dim() != 4 or t.shape[-1] != 64 or t.stride(-1) != 1:
            raise ValueError(f"{n}: expected [B, S, H, 64] with contiguous head dim, got {tuple(t.shape)}")
    B, Sq, H, _ = q.shape
    Bkv, Skv = k.shape[0], k.shape[1]
    if Bkv * kv_batch_div != B:
        raise ValueError("kv batch * kv_batch_div must equal q batch")
    if out is None:
        out = torch.empty(B, Sq, H * 64, dtype=torch.bfloat16, device=q.device)
    a = AttnArgs()
    a.Q, a.K, a.V, a.O = _p(q), _p(k), _p(v), _p(out)
    a.q_sb, a.q_ss, a.q_sh = q.stride(0), q.stride(1), q.stride(2)
    a.k_sb, a.k_ss, a.k_sh = k.stride(0), k.stride(1), k.stride(2)
    a.v_sb, a.v_ss, a.v_sh = v.stride(0), v.stride(1), v.stride(2)
    if out.stride(-1) != 1:
        raise ValueError("out must have a contiguous last dim")
    a.o_sb, a.o_ss = out.stride(0), out.stride(1)
    if resid is not None:
        _dev(resid, name="resid")
        if resid.stride() != out.stride():
            raise ValueError("resid must share out's layout")
        a.resid = _p(resid)
    if mask is not None:
        if not mask.is_cuda:
            raise HipOnly("mask must be on the GPU")
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        if mask.dtype != torch.uint8 or tuple(mask.shape) != (Sq, Skv) or not mask.is_contiguous():
            raise ValueError("mask must be a contiguous bool/uint8 [Sq, Skv]")
        a.mask = _p(mask)
    if bias is not None:                                    # softmax(scale q k^T + bias): fp32 [H, Sq, Skv], shared by the batch (T5's relative position bias)
        _dev(bias, torch.float32, "bias")
        if tuple(bias.shape) != (H, Sq, Skv) or not bias.is_contiguous():
            raise ValueError("bias must be a contiguous fp32 [H, Sq, Skv]")
        if fp8:
            raise ValueError("the fp8 path takes no score bias")
        a.bias, a.bias_sh = _p(bias), Sq * Skv
    a.B, a.H, a.Sq, a.Skv, a.kv_batch_div = B, H, Sq, Skv, kv_batch_div
    a.scale = (64 ** -0.5) if scale is None else scale
    a.out_scale = out_scale
    a.q_prescaled = 1 if q_prescaled else 0
    a.tuning = TUNING["attn"]
    if fp8:
        if not fp8_attention_supported(Sq, Skv, kv_batch_div, mask, q_prescaled):
            raise ValueError(f"fp8 attention does not take Sq={Sq} Skv={Skv} kv_batch_div={kv_batch_div} (see fp8_attention_supported)")
        need = _lib.lib().mrag_attn_fp8_workspace_bytes(B, H, Sq, Skv)
        ws = _attn_workspace(q.device, need, "fp8")
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
        timed = KERNEL_TIMING is not None and Skv >= 1024
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(_lib.lib().mrag_attn_fwd_fp8(_stream(), ctypes.byref(a)), "mrag_attn_fwd_fp8")
        if timed:
            e1.record()
            KERNEL_TIMING.append(("attn_fwd_fp8", 4.0 * B * H * Sq * Skv * 64, e0, e1))
        return out
    if mask is None and not TUNING["attn_no_split"]:
        need = _lib.lib().mrag_attn_workspace_bytes(B, H, Sq, Skv)   # > 0: long sequence with a ragged last query tile (key-split tail)
        if need > 0:
            ws = _attn_workspace(q.device, need)
            a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    timed = KERNEL_TIMING is not None and Skv >= 1024
    if timed:  # bench.py's roofline leg: HIP events on the launch stream around this one kernel
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().mrag_attn_fwd_bf16(_stream(), ctypes.byref(a)), "mrag_attn_fwd_bf16")
    if timed:
        e1.record()
        KERNEL_TIMING.append(("attn_fwd", 4.0 * B * H * Sq * Skv * 64, e0, e1))
    return out


# bench.py sets this to a list to collect (name, algorithmic_flops, start_event, end_event) per big attention launch
KERNEL_TIMING = None


def attention_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, scale: Optional[float] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v for short sequences at head dims 32 / 80 / 96 / 128 (mrag_attn_small_bf16): q [B, Sq, H, D], k / v [B, Skv, H, D] with a contiguous
    head dim (views of a fused QKV buffer are fine), Skv * D <= 32 768; out [B, Sq, H * D]"""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _dev(t, name=n)
        if t.dim() != 4 or t.stride(-1) != 1:
            raise ValueError(f"{n}: expected [B, S, H, D] with a contiguous head dim")
    B, Sq, H, D = q.shape
    if out is None:
        out = torch.empty(B, Sq, H * D, dtype=torch.bfloat16, device=q.device)
    a = AttnArgs()
    a.Q, a.K, a.V, a.O = _p(q), _p(k), _p(v), _p(out)
    a.q_sb, a.q_ss, a.q_sh = q.stride(0), q.stride(1), q.stride(2)
    a.k_sb, a.k_ss, a.k_sh = k.stride(0), k.stride(1), k.stride(2)
    a.v_sb, a.v_ss, a.v_sh = v.stride(0), v.stride(1), v.stride(2)
    a.o_sb, a.o_ss = out.stride(0), out.stride(1)
    a.B, a.H, a.Sq, a.Skv, a.kv_batch_div = B, H, Sq, k.shape[1], 1
    a.scale, a.out_scale = (D ** -0.5) if scale is None else scale, 1.0
    check(_lib.lib().mrag_attn_small_bf16(_stream(), ctypes.byref(a), D), "mrag_attn_small_bf16")
    return out


def layernorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float, *,
              out: Optional[torch.Tensor] = None, shift0=None, scale0=None, shift1=None, scale1=None,
              rows_per_batch: int = 0, split: int = 0, mod_stride: int = 0, out_batched: Optional[torch.Tensor] = None, rms: bool = False) -> torch.Tensor:
    """LayerNorm over the last dim (+ AdaLN modulation).  `out_batched` [B, L', D] (a view whose batch
    stride differs from L*D, e.g. a slice of a concat buffer) receives x [B, L, D] row by row."""
    _dev(x, name="x")
    x2 = _rows(x)
    a = LnArgs()
    if out_batched is not None:
        if x.dim() != 3 or out_batched.stride(-1) != 1 or out_batched.shape != x.shape:
            raise ValueError("out_batched must be a [B, L, D] view matching x")
        out = out_batched
        a.y, a.ldy = _p(out_batched), out_batched.stride(1)
        a.y_rows_per_batch, a.y_batch_stride = x.shape[1], out_batched.stride(0)
    else:
        if out is None:
            out = torch.empty_like(x)
        o2 = _rows(out)
        a.y, a.ldy = _p(o2), o2.stride(0)
    a.x, a.gamma, a.beta = _p(x2), _p(gamma), _p(beta)
    a.rows, a.D, a.ldx = x2.shape[0], x2.shape[1], x2.stride(0)
    a.eps = eps
    a.rms = 1 if rms else 0                               # RMSNorm (T5LayerNorm): x * rsqrt(mean(x^2) + eps) * gamma
    if shift0 is not None:
        a.shift0, a.scale0, a.shift1, a.scale1 = _p(shift0), _p(scale0), _p(shift1), _p(scale1)
        a.rows_per_batch, a.split, a.mod_stride = rows_per_batch, split, mod_stride
    check(_lib.lib().mrag_layernorm_bf16(_stream(), ctypes.byref(a)), "mrag_layernorm_bf16")
    return out


def qknorm_rope_(qkv: torch.Tensor, H: int, q_gamma, q_beta, k_gamma, k_beta, cos: Optional[torch.Tensor],
                 sin: Optional[torch.Tensor], text_len: int, eps: float = 1e-6, q_premul: float = 1.0) -> torch.Tensor:
    """in place on a fused [B, S, 3*H*64] buffer: per-head LayerNorm of Q and K, RoPE on tokens >= text_len."""
    _dev(qkv, name="qkv")
    B, S, W = qkv.shape
    if W != 3 * H * 64 or not qkv.is_contiguous():
        raise ValueError("qkv must be a contiguous [B, S, 3*H*64]")
    a = QkNormRopeArgs()
    a.qkv, a.q_gamma, a.q_beta, a.k_gamma, a.k_beta = _p(qkv), _p(q_gamma), _p(q_beta), _p(k_gamma), _p(k_beta)
    if cos is not None:
        _dev(cos, torch.float32, "cos"); _dev(sin, torch.float32, "sin")
        if tuple(cos.shape) != (S - text_len, 64) or not cos.is_contiguous() or not sin.is_contiguous():
            raise ValueError("cos/sin must be contiguous [S - text_len, 64] fp32")
        a.cos, a.sin = _p(cos), _p(sin)
    a.B, a.S, a.H, a.text_len, a.eps, a.q_premul = B, S, H, text_len, eps, q_premul
    check(_lib.lib().mrag_qknorm_rope_bf16(_stream(), ctypes.byref(a)), "mrag_qknorm_rope_bf16")
    return qkv


def qkv_linear_qknorm_rope(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], H: int, q_gamma, q_beta, k_gamma, k_beta,
                           cos: Optional[torch.Tensor], sin: Optional[torch.Tensor], text_len: int, eps: float = 1e-6, q_premul: float = 1.0,
                           first: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fused QKV projection + per-head qk LayerNorm + RoPE: x [B, S, K] -> [B, S, n*H*64] (attn_processor.py:209-231).  One GEMM whose epilogue does
    what `qknorm_rope_` does in a second pass; falls back to the two kernels when the launch cannot take that epilogue (MRAG_ENOTSUP).
    `weight` holds the thirds first .. first + n - 1 of [Wq | Wk | Wv] (first = 1, n = 2: the K | V projection of a sequence-sharded rank)."""
    _dev(x, name="x"); _dev(weight, name="weight")
    B, S, K = x.shape
    N = weight.shape[0]
    D = H * 64
    if N % D or first + N // D > 3 or weight.shape[1] != K or not x.is_contiguous() or K % 64 != 0:
        raise ValueError("qkv_linear_qknorm_rope: x [B, S, K] contiguous, weight [n*H*64, K] with first + n <= 3, K % 64 == 0")
    if out is None:
        out = torch.empty(B, S, N, dtype=torch.bfloat16, device=x.device)
    a = GemmArgs()
    a.A, a.W, a.bias, a.C = _p(x), _p(weight), _p(bias), _p(out)
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = B * S, N, K, K, weight.stride(0), N
    a.epilogue, a.rows_per_batch, a.rope_text_len, a.qk_dmodel, a.qk_eps, a.q_premul, a.qk_first = EPI_QKNORM_ROPE, S, text_len, D, eps, q_premul, first
    a.q_gamma, a.q_beta, a.k_gamma, a.k_beta = _p(q_gamma), _p(q_beta), _p(k_gamma), _p(k_beta)
    if cos is not None:
        _dev(cos, torch.float32, "cos"); _dev(sin, torch.float32, "sin")
        if tuple(cos.shape) != (S - text_len, 64) or not cos.is_contiguous() or not sin.is_contiguous():
            raise ValueError("cos/sin must be contiguous [S - text_len, 64] fp32")
        a.rope_cos, a.rope_sin = _p(cos), _p(sin)
    a.tuning = TUNING["gemm"]
    _gemm_workspace(a, x.device)
    rc = _lib.MRAG_ENOTSUP if TUNING["no_qkv_fuse"] else _lib.lib().mrag_gemm_bf16(_stream(), ctypes.byref(a))
    if rc == _lib.MRAG_ENOTSUP:                                   # small problem / unaligned output: plain GEMM, then the norm + RoPE pass
        if first != 0 or N != 3 * D:
            raise NotImplementedError("the two-kernel fallback needs the whole fused [Q | K | V] buffer")
        linear(x, weight, bias, out=out)
        return qknorm_rope_(out, H, q_gamma, q_beta, k_gamma, k_beta, cos, sin, text_len, eps=eps, q_premul=q_premul)
    check(rc, "mrag_gemm_bf16")
    return out


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    _dev(t, torch.float32, "t")
    out = torch.empty(t.shape[0], dim, dtype=torch.bfloat16, device=t.device)
    check(_lib.lib().mrag_timestep_embedding_bf16(_stream(), _p(t), _p(out), t.shape[0], dim), "mrag_timestep_embedding_bf16")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    _dev(x, name="x")
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().mrag_silu_bf16(_stream(), _p(x), _p(y), x.numel()), "mrag_silu_bf16")
    return y


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(a, name="a"); _dev(b, name="b")
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("add: same-shape contiguous tensors required")
    if out is None:
        out = torch.empty_like(a)
    check(_lib.lib().mrag_add_bf16(_stream(), _p(a), _p(b), _p(out), a.numel()), "mrag_add_bf16")
    return out


def add_rows(x: torch.Tensor, table: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [..., L, D] + table [L, D] broadcast over the leading dims (sinusoid position table)."""
    _dev(x, name="x"); _dev(table, name="table")
    if not x.is_contiguous() or not table.is_contiguous() or x.shape[-2:] != table.shape:
        raise ValueError("add_rows: x [..., L, D], table [L, D], both contiguous")
    if out is None:
        out = torch.empty_like(x)
    L, D = table.shape
    check(_lib.lib().mrag_add_rows_bf16(_stream(), _p(x), _p(table), _p(out), x.numel() // D, D, L), "mrag_add_rows_bf16")
    return out


def add_bcast(x: torch.Tensor, table: torch.Tensor, div: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [rows, D] (any leading shape) + table[(row // div) % len(table)]: a per-frame vector broadcast over `div` pixel rows.  `table` may be a
    column slice of a wider matrix (row stride a multiple of 8 elements, 16-byte aligned)."""
    _dev(x, name="x"); _dev(table, name="table")
    if not x.is_contiguous() or table.dim() != 2 or table.shape[1] != x.shape[-1] or table.stride(1) != 1 or table.stride(0) % 8 or table.data_ptr() % 16:
        raise ValueError("add_bcast: contiguous x [..., D] and table [L, D] with contiguous rows (row stride % 8 == 0, 16-byte aligned)")
    if out is None:
        out = torch.empty_like(x)
    D = x.shape[-1]
    check(_lib.lib().mrag_add_bcast_bf16(_stream(), _p(x), _p(table), _p(out), x.numel() // D, D, int(div), table.shape[0], table.stride(0)), "mrag_add_bcast_bf16")
    return out


def axpby(x: torch.Tensor, y: torch.Tensor, a: float, b: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a * x + b * y (AlphaBlender)"""
    _dev(x, name="x"); _dev(y, name="y")
    if x.shape != y.shape or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("axpby: same-shape contiguous tensors")
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().mrag_axpby_bf16(_stream(), _p(x), _p(y), _p(out), x.numel(), float(a), float(b)), "mrag_axpby_bf16")
    return out


def cfg_euler_step_(v_pred: torch.Tensor, latents: torch.Tensor, guidance: torch.Tensor, c_x: float, c_v: float) -> torch.Tensor:
    """latents [B, F, C, H, W] <- c_x * latents + c_v * (v_u + g[f] (v_c - v_u)); v_pred [2, B, F, C, H, W] (uncond first), g fp32 [F]."""
    _dev(v_pred, name="v_pred"); _dev(latents, name="latents"); _dev(guidance, torch.float32, "guidance")
    if v_pred.shape[0] != 2 or v_pred.shape[1:] != latents.shape or not v_pred.is_contiguous() or not latents.is_contiguous() or latents.dim() != 5:
        raise ValueError("cfg_euler_step_: v_pred [2, B, F, C, H, W], latents [B, F, C, H, W], contiguous")
    F = latents.shape[1]
    if guidance.numel() != F:
        raise ValueError("cfg_euler_step_: one guidance scale per frame")
    fe = latents.shape[2] * latents.shape[3] * latents.shape[4]
    check(_lib.lib().mrag_cfg_euler_step_bf16(_stream(), _p(v_pred), _p(latents), latents.numel(), _p(guidance.contiguous()), F, fe, float(c_x), float(c_v)),
          "mrag_cfg_euler_step_bf16")
    return latents


def weighted_sum(x: torch.Tensor, w: Optional[torch.Tensor], div: float = 1.0) -> torch.Tensor:
    """out[b, ...] = (sum_k w[b, k] x[b, k, ...]) / div with fp32 weights [B, K] (None = ones); x [B, K, ...] bf16 contiguous."""
    _dev(x, name="x")
    if x.dim() < 3 or not x.is_contiguous():
        raise ValueError("weighted_sum: contiguous [B, K, ...] required")
    B, K = x.shape[:2]
    n = x[0, 0].numel()
    if n % 8:
        raise ValueError("weighted_sum: the fused trailing dims must be a multiple of 8 elements")
    if w is not None:
        _dev(w, torch.float32, "w")
        if tuple(w.shape) != (B, K) or not w.is_contiguous():
            raise ValueError("weighted_sum: w must be a contiguous fp32 [B, K]")
    out = torch.empty(B, *x.shape[2:], dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mrag_weighted_sum_bf16(_stream(), _p(x), _p(w), _p(out), B, K, n, float(div)), "mrag_weighted_sum_bf16")
    return out


def blend_tile(tile: torch.Tensor, up: Optional[torch.Tensor], left: Optional[torch.Tensor], extent_v: int, extent_h: int) -> torch.Tensor:
    """in-place seam blend of a VAE tile [T, th, tw, C] with its (already blended) upper / left neighbours (mrag_hip.h)"""
    _dev(tile, name="tile")
    if tile.dim() != 4 or not tile.is_contiguous():
        raise ValueError("blend_tile: contiguous [T, th, tw, C] required")
    T, th, tw, C = tile.shape
    for nb, nm in ((up, "up"), (left, "left")):
        if nb is not None:
            _dev(nb, name=nm)
            if nb.dim() != 4 or not nb.is_contiguous() or nb.shape[0] != T or nb.shape[3] != C:
                raise ValueError(f"blend_tile: {nm} must be a contiguous [T, ., ., C] tile")
    if up is not None and up.shape[2] != tw:
        raise ValueError("blend_tile: the tile above must have the tile's width")
    if left is not None and left.shape[1] != th:
        raise ValueError("blend_tile: the tile to the left must have the tile's height")
    check(_lib.lib().mrag_blend_tile_bf16(_stream(), _p(tile), _p(up), _p(left), T, th, tw, C, up.shape[1] if up is not None else 0,
                                          left.shape[2] if left is not None else 0, extent_v, extent_h), "mrag_blend_tile_bf16")
    return tile


def softmax_rows(x: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(scale * x, dim=-1) of a 2-D bf16 score matrix (fp32 statistics); the row stride may exceed the width"""
    _dev(x, name="x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("softmax_rows: [rows, cols] with contiguous columns required")
    if out is None:                                       # rows start on 16-byte boundaries: pad the row stride to a multiple of 8 elements
        out = torch.empty(x.shape[0], (x.shape[1] + 7) // 8 * 8, dtype=torch.bfloat16, device=x.device)[:, :x.shape[1]]
    if x.stride(0) % 8 or out.stride(0) % 8:
        raise ValueError("softmax_rows: row strides must be multiples of 8 elements")
    check(_lib.lib().mrag_softmax_rows_bf16(_stream(), _p(x), _p(out), x.shape[0], x.shape[1], x.stride(0), out.stride(0), float(scale)), "mrag_softmax_rows_bf16")
    return out


def patchify(src0: torch.Tensor, src1: Optional[torch.Tensor], B: int) -> torch.Tensor:
    """[Bl, F, C0, H, W] (+ [Bl, F, C1, H, W]) -> [B*F*(H/2)*(W/2), (C0+C1)*4]; batch b reads latent b % Bl."""
    _dev(src0, name="src0")
    Bl, F, C0, H, W = src0.shape
    C1 = 0
    if src1 is not None:
        _dev(src1, name="src1")
        C1 = src1.shape[2]
    if not src0.is_contiguous() or (src1 is not None and not src1.is_contiguous()):
        raise ValueError("patchify: contiguous inputs required")
    out = torch.empty(B * F * (H // 2) * (W // 2), (C0 + C1) * 4, dtype=torch.bfloat16, device=src0.device)
    check(_lib.lib().mrag_patchify_bf16(_stream(), _p(src0), _p(src1), _p(out), B, Bl, F, C0, C1, H, W), "mrag_patchify_bf16")
    return out


def unpatchify(rows: torch.Tensor, B: int, F: int, C: int, H: int, W: int) -> torch.Tensor:
    _dev(rows, name="rows")
    if not rows.is_contiguous():
        raise ValueError("unpatchify: contiguous input required")
    out = torch.empty(B, F, C, H, W, dtype=torch.bfloat16, device=rows.device)
    check(_lib.lib().mrag_unpatchify_bf16(_stream(), _p(rows), _p(out), B, F, C, H, W), "mrag_unpatchify_bf16")
    return out


def cfg_ddim_step_(v_pred: torch.Tensor, latents: torch.Tensor, guidance: float, sqrt_alpha_t: float, sqrt_beta_t: float,
                   a_t: float, b_t: float) -> torch.Tensor:
    """latents <- DDIM(v_u + g (v_c - v_u)); v_pred [2, ...] (uncond first), latents [...] in place."""
    _dev(v_pred, name="v_pred"); _dev(latents, name="latents")
    n = latents.numel()
    if v_pred.numel() != 2 * n or not v_pred.is_contiguous() or not latents.is_contiguous():
        raise ValueError("cfg_ddim_step_: v_pred must be [2, *latents.shape], contiguous")
    check(_lib.lib().mrag_cfg_ddim_step_bf16(_stream(), _p(v_pred), _p(latents), n, guidance, sqrt_alpha_t, sqrt_beta_t, a_t, b_t),
          "mrag_cfg_ddim_step_bf16")
    return latents


def cfg_dpm_step_(v_pred: torch.Tensor, latents: torch.Tensor, x0_prev: torch.Tensor, noise: torch.Tensor, guidance: float, sqrt_alpha_t: float, sqrt_beta_t: float,
                  m1: float, m2: float, m3: float, m4: float, m_noise: float, second_order: bool) -> torch.Tensor:
    """latents <- CogVideoXDPMScheduler.step(v_u + g (v_c - v_u)); v_pred [2, ...] (uncond first); latents, x0_prev (the previous step's x0, updated), noise [...]"""
    for t, nm in ((v_pred, "v_pred"), (latents, "latents"), (x0_prev, "x0_prev"), (noise, "noise")):
        _dev(t, name=nm)
    n = latents.numel()
    if v_pred.numel() != 2 * n or x0_prev.numel() != n or noise.numel() != n or not all(t.is_contiguous() for t in (v_pred, latents, x0_prev, noise)):
        raise ValueError("cfg_dpm_step_: v_pred [2, *latents.shape]; x0_prev, noise like latents; all contiguous")
    check(_lib.lib().mrag_cfg_dpm_step_bf16(_stream(), _p(v_pred), _p(latents), _p(x0_prev), _p(noise), n, guidance, sqrt_alpha_t, sqrt_beta_t, m1, m2, m3, m4, m_noise,
                                            1 if second_order else 0), "mrag_cfg_dpm_step_bf16")
    return latents


_TOPK_WS = {}


def topk_workspace(device: torch.device, N: int, Q: int) -> torch.Tensor:
    """scratch of mrag_topk_f32, one ZEROED buffer per (device, stream, N, Q): its first 64 bytes are the arrival counters of the
    single-launch form, zero on first use and left zero by every call (include/mrag_hip.h)"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, N, Q)
    ws = _TOPK_WS.get(key)
    if ws is None:
        if len(_TOPK_WS) > 64:
            _TOPK_WS.clear()
        ws = _TOPK_WS[key] = torch.zeros(_lib.lib().mrag_topk_workspace_bytes(N, Q), dtype=torch.uint8, device=device)
    return ws


TOPK_ORDER = {"auto": 0, "chain16": 1, "mfma": 2, "mfma_stream": 3, "mfma_nowait": 4}   # 3 / 4: the fan-out form's two launch shapes forced (diagnostics; include/mrag_hip.h)


def topk(db: torch.Tensor, queries: torch.Tensor, k: int, *, metric: str = "l2", group: Optional[torch.Tensor] = None,
         exclude: Optional[torch.Tensor] = None, out: Optional[tuple] = None, postfilter: bool = False, order: str = "auto"):
    """flat-scan top-k: returns (rows int32 [Q, k], dist fp32 [Q, k]) sorted by (dist asc, row asc).  `out` = (rows, dist) buffers to fill
    (a caller that searches repeatedly keeps them: no allocation on the call path).  `postfilter`: rows of the excluded group are dropped
    AFTER the k nearest were selected (lancedb's `where(..., prefilter=False)`; possibly < k results, tail row -1) instead of before.
    `order`: the distance's summation order (include/mrag_hip.h): "chain16" (the scan kernel), "mfma" (the fan-out kernel: >= 16 queries, k <= 16: one
    pass over the table per 256 queries on fp32 MFMAs) or "auto" (mfma where it applies)."""
    _dev(db, torch.float32, "db"); _dev(queries, torch.float32, "queries")
    if not db.is_contiguous() or not queries.is_contiguous():
        raise ValueError("topk: contiguous db / queries required")
    N, D = db.shape
    Q = queries.shape[0]
    m = {"l2": 0, "dot": 1}[metric]
    if exclude is not None:
        _dev(group, torch.int32, "group"); _dev(exclude, torch.int32, "exclude")
    ws = topk_workspace(db.device, N, Q)
    if Q <= 4:
        ws[:64].zero_()      # arrival counters of the single-launch form: re-zeroed on THIS call path (an aborted earlier launch cannot poison it); TopkPlan is the no-extra-launch path
    if out is None:
        rows = torch.empty(Q, k, dtype=torch.int32, device=db.device)
        dist = torch.empty(Q, k, dtype=torch.float32, device=db.device)
    else:
        rows, dist = out
        if tuple(rows.shape) != (Q, k) or tuple(dist.shape) != (Q, k) or rows.dtype != torch.int32 or dist.dtype != torch.float32:
            raise ValueError("topk: out = (int32 [Q, k], float32 [Q, k])")
    check(_lib.lib().mrag_topk_f32(_stream(), _p(db), _p(group) if exclude is not None else None, N, D, _p(queries),
                                   _p(exclude), Q, k, m, _p(rows), _p(dist), _p(ws), ws.numel(), int(bool(postfilter)), TOPK_ORDER[order]), "mrag_topk_f32")
    return rows, dist


class TopkPlan:
    """A prepared search over a resident database: every buffer (query tile, exclusion ids, workspace, outputs) is allocated once and every
    ctypes argument is built once, so a query costs ONE C-ABI call -- for <= 4 queries one kernel launch (scan + merge fused) -- and nothing is
    allocated on the call path (the latency-bound interactive search of src/data/rag.py:63-80: 10 k rows stream in ~7 us, everything above
    that was host overhead).  `graph=True` additionally records the launch in a HIP graph (`replay()`)."""

    def __init__(self, db: torch.Tensor, n_queries: int, k: int, *, metric: str = "l2", group: Optional[torch.Tensor] = None, graph: bool = False,
                 postfilter: bool = False, order: str = "auto"):
        _dev(db, torch.float32, "db")
        if not db.is_contiguous():
            raise ValueError("TopkPlan: contiguous db required")
        self.db, self.group, self.k, self.Q = db, group, k, n_queries
        N, D = db.shape
        dev = db.device
        self.queries = torch.zeros(n_queries, D, dtype=torch.float32, device=dev)          # caller fills: plan.queries.copy_(q)
        self.exclude = torch.full((n_queries,), -1, dtype=torch.int32, device=dev) if group is not None else None
        self.rows = torch.empty(n_queries, k, dtype=torch.int32, device=dev)
        self.dist = torch.empty(n_queries, k, dtype=torch.float32, device=dev)
        L = _lib.lib()
        self._ws = torch.zeros(L.mrag_topk_workspace_bytes(N, n_queries), dtype=torch.uint8, device=dev)   # private: counters stay consistent
        self._fn = L.mrag_topk_f32
        self._args = (_p(db), _p(group) if group is not None else None, N, D, _p(self.queries), _p(self.exclude) if group is not None else None,
                      n_queries, k, {"l2": 0, "dot": 1}[metric], _p(self.rows), _p(self.dist), _p(self._ws), self._ws.numel(), int(bool(postfilter)),
                      TOPK_ORDER[order])
        self._graph = None
        if graph:
            self.run()
            torch.cuda.synchronize(dev)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self.run()

    def run(self):
        """launch on the current stream; results land in `.rows` / `.dist`"""
        rc = self._fn(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), *self._args)
        if rc:
            check(rc, "mrag_topk_f32")
        return self.rows, self.dist

    def replay(self):
        self._graph.replay()
        return self.rows, self.dist


# ---------------------------------------------------------------------------------------------- UNet ops (channels-last rows)
def groupnorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], groups: int, eps: float, *, silu: bool = False,
              emb: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, mod: Optional[torch.Tensor] = None,
              mod_geom: Optional[tuple] = None, fold: bool = False) -> torch.Tensor:
    """nn.GroupNorm over x [N, HW, C] (channels-last) [+ per-(n, c) `emb` added before the statistics] [+ SiLU].
    `mod` [N, Tz, hz, wz, 2C] + `mod_geom` = (T, H, W, shift, split): the spatially conditioned form gn(x) * mod[.., :C] + mod[.., C:] with the
    latent-resolution maps read through the nearest-neighbour frame / pixel map (mrag_hip.h); `out` may then be a [N, HW, C] view whose
    sample stride is larger than HW * C (the frames behind a causal convolution's context frames)."""
    from ._lib import GroupNormArgs
    _dev(x, name="x")
    if x.dim() != 3 or not x.is_contiguous():
        raise ValueError("groupnorm: contiguous [N, HW, C] required")
    N, HW, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.stride(2) != 1 or out.stride(1) != C or (mod is None and not out.is_contiguous()):
        raise ValueError("groupnorm: out must have x's shape with contiguous rows")
    chunks = max(1, min(1024, HW // 32, max(64, -(-2048 // N))))        # >= ~2048 statistics workgroups even when N is 1 or 2
    L = _lib.lib()
    ws = _attn_workspace(x.device, L.mrag_groupnorm_workspace_bytes(N, C, chunks), "gn")     # grow-only per (device, stream): no allocation per call
    a = GroupNormArgs()
    a.x, a.y, a.gamma, a.beta, a.workspace = _p(x), _p(out), _p(gamma), _p(beta), _p(ws)
    a.N, a.HW, a.C, a.G, a.chunks, a.silu, a.eps = N, HW, C, groups, chunks, 1 if silu else 0, eps
    a.fold = 1 if (fold or TUNING.get("gn_fold")) else 0                    # opt-in: statistics + fold in one launch (measured slower than the separate fold: include/mrag_hip.h)
    if emb is not None:
        _dev(emb, name="emb")
        if emb.shape != (N, C) or emb.stride(1) != 1:
            raise ValueError("emb must be [N, C] with contiguous channels")
        a.emb, a.emb_stride = _p(emb), emb.stride(0)
    if mod is not None:
        _dev(mod, name="mod")
        T, H, W, shift, split = mod_geom
        if not mod.is_contiguous() or mod.dim() != 5 or mod.shape[0] != N or tuple(mod.shape[2:]) != (H >> shift, W >> shift, 2 * C):
            raise ValueError(f"groupnorm: mod {tuple(mod.shape)} does not match geometry {mod_geom} and C = {C}")
        a.mod, a.mod_T, a.mod_H, a.mod_W, a.mod_Tz, a.mod_shift, a.mod_split = _p(mod), T, H, W, mod.shape[1], shift, 1 if split else 0
        a.y_stride_n = out.stride(0) if N > 1 else 0
    check(L.mrag_groupnorm_bf16(_stream(), ctypes.byref(a)), "mrag_groupnorm_bf16")
    return out


CONV_3X3, CONV_T3 = 1, 2


def conv_implicit(x: torch.Tensor, wk: torch.Tensor, bias: Optional[torch.Tensor], mode: int, *, stride: int = 1, upsample: bool = False,
                  frames: int = 0, resid: Optional[torch.Tensor] = None, asym_pad: bool = False, t_frames: int = 0, acc_scale: float = 1.0) -> torch.Tensor:
    """implicit-GEMM convolution (no materialised im2col), Cin % 64 == 0.
    CONV_3X3: x [N, H, W, Cin] -> [N, Ho, Wo, Cout], wk [Cout, 9 Cin] in (ky, kx, cin) order.
              t_frames = T > 0: causal 3x3x3 -- x [S (T + 2), H, W, Cin] (two context frames in front of each sample's T frames) -> [S T, H, W, Cout],
              wk [Cout, 27 Cin] in (kt, ky, kx, cin) order.
    CONV_T3:  x [(b t), HW, Cin] with `frames` = t -> same rows x Cout, wk [Cout, 3 Cin] in (kt, cin) order."""
    from ._lib import ConvArgs
    _dev(x, name="x"); _dev(wk, name="weight")
    if not x.is_contiguous() or not wk.is_contiguous():
        raise ValueError("conv_implicit: contiguous activation and weight required")
    a = ConvArgs()
    cout = wk.shape[0]
    if mode == CONV_3X3:
        N, H, W, C = x.shape
        taps = 9
        if t_frames:
            if N % (t_frames + 2):
                raise ValueError("conv_implicit: x must hold t_frames + 2 frames per sample")
            N = N // (t_frames + 2) * t_frames
            a.t_taps, a.t_frames, taps = 3, t_frames, 27
        Hi, Wi = (2 * H, 2 * W) if upsample else (H, W)
        front = 0 if asym_pad else 1                         # asym_pad: F.pad(x, (0, 1, 0, 1)) + stride-2 convolution without padding (KL-VAE Downsample)
        out = torch.empty(N, (Hi + front - 2) // stride + 1, (Wi + front - 2) // stride + 1, cout, dtype=torch.bfloat16, device=x.device)
        a.N, a.H, a.Wd, a.stride, a.upsample, a.asym_pad = N, H, W, stride, 1 if upsample else 0, 1 if asym_pad else 0
    else:
        NT, HW, C = x.shape
        if frames <= 0 or NT % frames:
            raise ValueError("conv_implicit: frames must divide the leading dim")
        out = torch.empty(NT, HW, cout, dtype=torch.bfloat16, device=x.device)
        a.N, a.H, a.Wd = NT // frames, frames, HW
        taps = 3
    if wk.shape[1] != taps * C:
        raise ValueError(f"weight {tuple(wk.shape)} does not match {taps} taps x {C} channels")
    if x.numel() >= 2 ** 31:
        raise ValueError("conv_implicit: activation too large for 32-bit tap offsets")
    a.x, a.W, a.bias, a.y = _p(x), _p(wk), _p(bias), _p(out)
    a.Cin, a.Cout, a.mode, a.epilogue = C, cout, mode, EPI_NONE
    if resid is not None:
        _dev(resid, name="resid")
        if resid.numel() != out.numel() or not resid.is_contiguous():
            raise ValueError("conv_implicit: resid must be contiguous with the output's shape")
        if acc_scale == 0.0:                  # see ops.linear: an exact zero means "the residual alone", the C struct's 0 means "unset = 1"
            out.copy_(resid.view(out.shape))
            return out
        a.resid, a.epilogue, a.acc_scale = _p(resid), EPI_RESID, acc_scale
    check(_lib.lib().mrag_conv_bf16(_stream(), ctypes.byref(a)), "mrag_conv_bf16")
    return out


def _kpad(k: int) -> int:
    return (k + 63) // 64 * 64


def im2col3x3(x: torch.Tensor, stride: int = 1, upsample: bool = False) -> torch.Tensor:
    """x [N, H, W, C] -> rows [N*Ho*Wo, Kpad] for the 3x3 / pad-1 convolution GEMM (Kpad = 9C rounded up to 64)."""
    _dev(x, name="x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("im2col3x3: contiguous [N, H, W, C] required")
    N, H, W, C = x.shape
    Hi, Wi = (2 * H, 2 * W) if upsample else (H, W)
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    kp = _kpad(9 * C)
    out = torch.empty(N * Ho * Wo, kp, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mrag_im2col3x3_bf16(_stream(), _p(x), _p(out), N, H, W, C, stride, 1 if upsample else 0, kp), "mrag_im2col3x3_bf16")
    return out


def unfold_t3(x: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """x [(b t), HW, C] -> rows [(b t hw), 3C] for the (3,1,1) temporal convolution GEMM."""
    _dev(x, name="x")
    if x.dim() != 3 or not x.is_contiguous() or x.shape[0] != B * T:
        raise ValueError("unfold_t3: contiguous [(b t), HW, C] required")
    _, HW, C = x.shape
    out = torch.empty(B * T * HW, 3 * C, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mrag_unfold_t3_bf16(_stream(), _p(x), _p(out), B, T, HW, C), "mrag_unfold_t3_bf16")
    return out


def geglu(x: torch.Tensor) -> torch.Tensor:
    _dev(x, name="x")
    if not x.is_contiguous():
        raise ValueError("geglu: contiguous input required")
    inner = x.shape[-1] // 2
    out = torch.empty(*x.shape[:-1], inner, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().mrag_geglu_bf16(_stream(), _p(x), _p(out), x.numel() // (2 * inner), inner), "mrag_geglu_bf16")
    return out


def ddim_v_step_(v_pred: torch.Tensor, x: torch.Tensor, noise: Optional[torch.Tensor], guidance: float, sqrt_alpha_t: float,
                 sqrt_one_minus_alpha_t: float, rescale: float, sqrt_alpha_prev: float, dir_coef: float, sigma: float) -> torch.Tensor:
    """DynamiCrafter DDIM update in place on fp32 latents x; v_pred bf16 [2, *x.shape] with the conditional half FIRST."""
    _dev(v_pred, name="v_pred"); _dev(x, torch.float32, "x")
    if noise is not None:
        _dev(noise, torch.float32, "noise")
    n = x.numel()
    if v_pred.numel() != 2 * n or not v_pred.is_contiguous() or not x.is_contiguous():
        raise ValueError("ddim_v_step_: v_pred must be [2, *x.shape], contiguous")
    check(_lib.lib().mrag_ddim_v_step_f32(_stream(), _p(v_pred), _p(x), _p(noise), n, guidance, sqrt_alpha_t, sqrt_one_minus_alpha_t, rescale,
                                          sqrt_alpha_prev, dir_coef, sigma), "mrag_ddim_v_step_f32")
    return x
