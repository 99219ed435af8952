"""ctypes binding of libmrag_hip.so (the C ABI declared in include/mrag_hip.h).

The library is the product: there is NO CPU fallback anywhere in this package.  `lib()` raises
`HipLibraryMissing` when the shared object has not been built, and every op in `ops.py` raises when
handed a non-GPU tensor.  Build with `python -m motionrag_amd._lib` (or `__graft_entry__.build()`):
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MRAG_HIP_LIB", os.path.join(_HERE, "libmrag_hip.so"))   # env override: A/B builds in tools/
SOURCES = ["api.hip", "gemm_bf16.hip", "attn_flash.hip", "attn16.hip", "attn_fp8.hip", "comm.hip", "norm.hip", "pointwise.hip", "preprocess.hip", "topk.hip", "unet_ops.hip", "cama_seq.hip", "attn_small.hip", "probe.hip"]
ABI_VERSION = 10
# per-file flags: the SLP vectoriser packs the softmax row-sum adds into v_pk_add_f32 + shuffles (slower beside MFMAs)
EXTRA_FLAGS = {"attn_flash.hip": ["-fno-slp-vectorize"],
               "attn16.hip": ["-fno-slp-vectorize"], "attn_fp8.hip": ["-fno-slp-vectorize"]}

# every symbol include/mrag_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "mrag_abi_version", "mrag_target_arch", "mrag_source_hash", "mrag_dispatch_counts", "mrag_dispatch_name", "mrag_probe_mfma_flops", "mrag_probe_mfma_bf16", "mrag_probe_mfma_f32_flops", "mrag_probe_mfma_f32", "mrag_probe_stream_copy", "mrag_gemm_bf16", "mrag_gemm_workspace_bytes", "mrag_attn_fwd_bf16", "mrag_attn_workspace_bytes", "mrag_layernorm_bf16",
    "mrag_qknorm_rope_bf16", "mrag_timestep_embedding_bf16", "mrag_silu_bf16", "mrag_add_rows_bf16", "mrag_add_bf16", "mrag_add_bcast_bf16", "mrag_axpby_bf16", "mrag_cfg_euler_step_bf16", "mrag_conv_bf16", "mrag_ip_attn_folded_bf16",
    "mrag_patchify_bf16", "mrag_unpatchify_bf16", "mrag_cfg_ddim_step_bf16", "mrag_topk_workspace_bytes", "mrag_topk_f32",
    "mrag_groupnorm_workspace_bytes", "mrag_groupnorm_bf16", "mrag_im2col3x3_bf16", "mrag_unfold_t3_bf16", "mrag_geglu_bf16",
    "mrag_ddim_v_step_f32", "mrag_weighted_sum_bf16", "mrag_attn_fp8_workspace_bytes", "mrag_attn_fwd_fp8",
    "mrag_comm_unique_id", "mrag_comm_init", "mrag_comm_destroy", "mrag_allgather",
    "mrag_resize_patchify_bf16", "mrag_assemble_tokens_bf16", "mrag_softmax_rows_bf16", "mrag_denormalize_u8", "mrag_attn_small_bf16", "mrag_blend_tile_bf16", "mrag_cfg_dpm_step_bf16",
    "mrag_resampler_workspace_bytes", "mrag_resampler_fwd", "mrag_cama_encoder_workspace_bytes", "mrag_cama_encoder_fwd",
]


# entry points whose result is not the int32 status code (their restype is set explicitly in lib())
_NON_INT_RESULT = ("mrag_target_arch", "mrag_source_hash", "mrag_dispatch_name", "mrag_probe_mfma_flops", "mrag_probe_mfma_f32_flops", "mrag_gemm_workspace_bytes", "mrag_attn_workspace_bytes", "mrag_attn_fp8_workspace_bytes", "mrag_topk_workspace_bytes", "mrag_groupnorm_workspace_bytes",
                   "mrag_resampler_workspace_bytes", "mrag_cama_encoder_workspace_bytes")


class HipLibraryMissing(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("C", c_void_p), ("resid", c_void_p),
        ("gate0", c_void_p), ("gate1", c_void_p),
        ("M", c_int64), ("N", c_int64), ("K", c_int64),
        ("lda", c_int64), ("ldw", c_int64), ("ldc", c_int64), ("ldr", c_int64),
        ("rows_per_batch", c_int64), ("split", c_int64), ("gate_stride", c_int64),
        ("epilogue", c_int32), ("rope_text_len", c_int32),
        ("q_gamma", c_void_p), ("q_beta", c_void_p), ("k_gamma", c_void_p), ("k_beta", c_void_p), ("rope_cos", c_void_p), ("rope_sin", c_void_p),
        ("qk_dmodel", c_int64), ("qk_eps", c_float), ("q_premul", c_float), ("qk_first", c_int32), ("tuning", c_int32), ("geglu_act", c_int32),
        ("workspace", c_void_p), ("acc_scale", c_float), ("workspace_bytes", c_int64), ("w_batch_stride", c_int64),
        ("a_ln_gamma", c_void_p), ("a_ln_beta", c_void_p), ("a_ln_eps", c_float), ("a_ln", c_int32),
    ]


class AttnArgs(Structure):
    _fields_ = [
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("O", c_void_p), ("resid", c_void_p), ("mask", c_void_p),
        ("q_sb", c_int64), ("q_ss", c_int64), ("q_sh", c_int64),
        ("k_sb", c_int64), ("k_ss", c_int64), ("k_sh", c_int64),
        ("v_sb", c_int64), ("v_ss", c_int64), ("v_sh", c_int64),
        ("o_sb", c_int64), ("o_ss", c_int64),
        ("B", c_int32), ("H", c_int32), ("Sq", c_int32), ("Skv", c_int32), ("kv_batch_div", c_int32),
        ("scale", c_float), ("out_scale", c_float), ("q_prescaled", c_int32),
        ("workspace", c_void_p), ("workspace_bytes", c_int64), ("tuning", c_int32),
        ("bias", c_void_p), ("bias_sh", c_int64),
    ]


class LnArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("y", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
        ("shift0", c_void_p), ("scale0", c_void_p), ("shift1", c_void_p), ("scale1", c_void_p),
        ("rows", c_int64), ("D", c_int64), ("ldx", c_int64), ("ldy", c_int64),
        ("rows_per_batch", c_int64), ("split", c_int64), ("mod_stride", c_int64),
        ("y_rows_per_batch", c_int64), ("y_batch_stride", c_int64),
        ("eps", c_float), ("rms", c_int32),
    ]


class ResizePatchArgs(Structure):
    _fields_ = [
        ("src", c_void_p), ("frame_idx", c_void_p),
        ("wy", c_void_p), ("y0", c_void_p), ("ny", c_void_p), ("wx", c_void_p), ("x0", c_void_p), ("nx", c_void_p),
        ("out", c_void_p),
        ("s_n", c_int64), ("s_t", c_int64), ("s_c", c_int64), ("ldo", c_int64),
        ("N", c_int32), ("T", c_int32), ("C", c_int32), ("H", c_int32), ("W", c_int32), ("OH", c_int32), ("OW", c_int32),
        ("taps_y", c_int32), ("taps_x", c_int32), ("pt", c_int32), ("ph", c_int32), ("pw", c_int32), ("src_fp32", c_int32),
        ("scale", c_float * 4), ("shift", c_float * 4), ("no_tiling", c_int32),
    ]


class ResamplerLayer(Structure):
    _fields_ = [(n, c_void_p) for n in ("norm1_w", "norm1_b", "norm2_w", "norm2_b", "to_q", "to_kv", "to_out", "ff_ln_w", "ff_ln_b", "ff_w1", "ff_w2")]


class ResamplerArgs(Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("latents", c_void_p), ("proj_in_w", c_void_p), ("proj_in_b", c_void_p), ("proj_out_w", c_void_p),
                ("proj_out_b", c_void_p), ("norm_out_w", c_void_p), ("norm_out_b", c_void_p), ("layers", POINTER(ResamplerLayer)),
                ("workspace", c_void_p), ("workspace_bytes", c_int64)] + [(n, c_int32) for n in ("N", "n1", "nq", "embedding_dim", "dim", "output_dim", "heads",
                                                                                                    "depth", "ff_dim")] + [("eps", c_float)]


class EncoderLayer(Structure):
    _fields_ = [(n, c_void_p) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm1_w", "norm1_b",
                                        "norm2_w", "norm2_b")]


class CamaEncoderArgs(Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("mask", c_void_p), ("layers", POINTER(EncoderLayer)), ("workspace", c_void_p), ("workspace_bytes", c_int64)] + [
        (n, c_int32) for n in ("B", "L", "d_model", "nhead", "ff_dim", "num_layers")] + [("eps", c_float)]


class QkNormRopeArgs(Structure):
    _fields_ = [
        ("qkv", c_void_p), ("q_gamma", c_void_p), ("q_beta", c_void_p), ("k_gamma", c_void_p), ("k_beta", c_void_p),
        ("cos", c_void_p), ("sin", c_void_p),
        ("B", c_int32), ("S", c_int32), ("H", c_int32), ("text_len", c_int32),
        ("eps", c_float), ("q_premul", c_float),
    ]


class GroupNormArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("y", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("emb", c_void_p), ("workspace", c_void_p),
        ("N", c_int64), ("HW", c_int64), ("C", c_int64), ("emb_stride", c_int64),
        ("G", c_int32), ("chunks", c_int32), ("silu", c_int32), ("eps", c_float),
        ("mod", c_void_p), ("mod_T", c_int32), ("mod_H", c_int32), ("mod_W", c_int32), ("mod_Tz", c_int32), ("mod_shift", c_int32), ("mod_split", c_int32),
        ("y_stride_n", c_int64), ("fold", c_int32),
    ]


class ConvArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("y", c_void_p), ("resid", c_void_p),
        ("N", c_int32), ("H", c_int32), ("Wd", c_int32), ("Cin", c_int32), ("Cout", c_int32),
        ("stride", c_int32), ("upsample", c_int32), ("mode", c_int32), ("epilogue", c_int32), ("asym_pad", c_int32),
        ("t_taps", c_int32), ("t_frames", c_int32), ("acc_scale", c_float),
    ]


MRAG_OK, MRAG_EINVAL, MRAG_ENOTSUP = 0, -1, -2

_lib = None


def source_hash() -> str:
    """digest of what the library is built from: every file of csrc/, include/mrag_hip.h and the per-file compile flags (sha256, 16 hex digits).
    The build stamps it into the binary (`mrag_source_hash()`); `lib()` refuses a binary whose stamp differs -- modification times are not
    consulted (a checkout, a copy to the GPU box or a variant build can leave a stale binary NEWER than the sources)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(_CSRC) if f.endswith((".hip", ".h")))
    for f in files:
        h.update(f.encode() + b"\0")
        with open(os.path.join(_CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(_HERE, "..", "include", "mrag_hip.h"), "rb") as fh:
        h.update(b"mrag_hip.h\0" + fh.read())
    h.update(repr((SOURCES, sorted(EXTRA_FLAGS.items()), os.environ.get("MRAG_EXTRA_HIPCC_FLAGS", ""))).encode())
    return h.hexdigest()[:16]


def binary_stamp(path: str) -> str:
    """the source digest a built library carries, read from the file's bytes (no dlopen); '' when absent"""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return ""
    i = blob.find(b"MRAG_SOURCE_HASH=")
    if i < 0:
        return ""
    j = blob.find(b"\0", i)
    return blob[i + 17:j].decode("ascii", "replace")


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into motionrag_amd/libmrag_hip.so (in-tree; needs no GPU).  Rebuilds whenever the binary's stamp is not
    the digest of the sources beside it.  MRAG_EXTRA_HIPCC_FLAGS (developer A/B builds, tools/build_variant.sh) is part of the digest."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    want = source_hash()
    if os.path.exists(LIB_PATH) and binary_stamp(LIB_PATH) == want:
        return LIB_PATH
    extra = os.environ.get("MRAG_EXTRA_HIPCC_FLAGS", "").split()
    build_dir = os.path.join(_HERE, "build")
    os.makedirs(build_dir, exist_ok=True)
    import hashlib
    hdr = hashlib.sha256()
    for f in sorted(f for f in os.listdir(_CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "mrag_hip.h")]:
        with open(os.path.join(_CSRC, f), "rb") as fh:
            hdr.update(fh.read())
    procs = []
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"] + EXTRA_FLAGS.get(src, []) + extra + (
            [f'-DMRAG_SOURCE_HASH="{want}"'] if src == "api.hip" else []) + ["-c", os.path.join(_CSRC, src), "-o", obj]
        objs.append(obj)
        # incremental: an object is reused when its source, the shared headers and its command line are what they were when it was compiled
        with open(os.path.join(_CSRC, src), "rb") as fh:
            key = hashlib.sha256(fh.read() + hdr.digest() + " ".join(cmd).encode()).hexdigest()
        tag = obj + ".key"
        if os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == key:
            continue
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if os.path.exists(tag):
            os.remove(tag)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), tag, key))
    failed = []
    for src, pr, tag, key in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out.decode()}")
        else:
            with open(tag, "w") as fh:
                fh.write(key)
    if failed:
        raise RuntimeError("\n".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout.decode()}")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the C-ABI library; fail loudly when it is missing or stale (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m motionrag_amd._lib` (hipcc, gfx950). "
            "motionrag_amd has no CPU fallback.")
    # torch ships its own libamdhip64.so: it must be the HIP runtime of the process (it owns the streams and the
    # device memory handed to the kernels), so make sure it is loaded BEFORE our library resolves the same soname.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.mrag_abi_version.restype = c_int32
    L.mrag_target_arch.restype = c_char_p
    if L.mrag_abi_version() != ABI_VERSION:
        raise HipLibraryMissing(f"{LIB_PATH} has ABI {L.mrag_abi_version()}, expected {ABI_VERSION}: rebuild")
    L.mrag_source_hash.restype = c_char_p
    have = L.mrag_source_hash().decode()
    if os.environ.get("MRAG_HIP_LIB_ANY_SOURCE"):                            # the explicit override is for tools/ A/B runs of archived variant libraries
        want = have
    else:
        try:
            want = source_hash()
        except OSError as e:                                                 # a binary-only install: nothing to compare the stamp with
            raise HipLibraryMissing(f"{LIB_PATH} carries stamp {have} but the sources it must be checked against are not readable ({e}); "
                                    "set MRAG_HIP_LIB_ANY_SOURCE=1 to load a library without its sources") from e
    if have != want:
        raise HipLibraryMissing(f"{LIB_PATH} was built from other sources (stamp {have}, sources beside it {want}): rebuild with "
                                "`python -m motionrag_amd._lib` -- a stale or variant binary is never loaded silently")
    L.mrag_dispatch_name.restype = c_char_p
    L.mrag_dispatch_name.argtypes = [c_int32]
    L.mrag_dispatch_counts.argtypes = [c_void_p, c_int32]
    L.mrag_probe_mfma_flops.argtypes = [c_int32]
    L.mrag_probe_mfma_flops.restype = c_int64
    L.mrag_probe_mfma_bf16.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int32]
    L.mrag_probe_mfma_f32_flops.argtypes = [c_int32]
    L.mrag_probe_mfma_f32_flops.restype = c_int64
    L.mrag_probe_mfma_f32.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int32]
    L.mrag_probe_stream_copy.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int32]
    L.mrag_gemm_bf16.argtypes = [c_void_p, POINTER(GemmArgs)]
    L.mrag_attn_fwd_bf16.argtypes = [c_void_p, POINTER(AttnArgs)]
    L.mrag_gemm_workspace_bytes.argtypes = [c_int64, c_int64, c_int64]
    L.mrag_gemm_workspace_bytes.restype = c_int64
    L.mrag_attn_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int32]
    L.mrag_attn_workspace_bytes.restype = c_int64
    L.mrag_attn_fp8_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int32]
    L.mrag_attn_fp8_workspace_bytes.restype = c_int64
    L.mrag_attn_fwd_fp8.argtypes = [c_void_p, POINTER(AttnArgs)]
    L.mrag_attn_small_bf16.argtypes = [c_void_p, POINTER(AttnArgs), c_int32]
    L.mrag_comm_unique_id.argtypes = [c_void_p]
    L.mrag_comm_init.argtypes = [c_void_p, c_int32, c_int32, POINTER(c_void_p)]
    L.mrag_comm_destroy.argtypes = [c_void_p]
    L.mrag_allgather.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]
    L.mrag_layernorm_bf16.argtypes = [c_void_p, POINTER(LnArgs)]
    L.mrag_qknorm_rope_bf16.argtypes = [c_void_p, POINTER(QkNormRopeArgs)]
    L.mrag_timestep_embedding_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32]
    L.mrag_silu_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int64]
    L.mrag_add_rows_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64]
    L.mrag_add_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]
    L.mrag_conv_bf16.argtypes = [c_void_p, POINTER(ConvArgs)]
    L.mrag_ip_attn_folded_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64,
                                           c_int64, c_float, c_float, c_int32]
    L.mrag_add_bcast_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64]
    L.mrag_axpby_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float]
    L.mrag_cfg_euler_step_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int64, c_float, c_float]
    L.mrag_patchify_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 7
    L.mrag_unpatchify_bf16.argtypes = [c_void_p, c_void_p, c_void_p] + [c_int32] * 5
    L.mrag_cfg_ddim_step_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int64] + [c_float] * 5
    L.mrag_topk_workspace_bytes.argtypes = [c_int64, c_int32]
    L.mrag_topk_workspace_bytes.restype = c_int64
    L.mrag_topk_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32]
    L.mrag_groupnorm_workspace_bytes.argtypes = [c_int64, c_int64, c_int32]
    L.mrag_groupnorm_workspace_bytes.restype = c_int64
    L.mrag_groupnorm_bf16.argtypes = [c_void_p, POINTER(GroupNormArgs)]
    L.mrag_im2col3x3_bf16.argtypes = [c_void_p, c_void_p, c_void_p] + [c_int32] * 7
    L.mrag_unfold_t3_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32]
    L.mrag_geglu_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64]
    L.mrag_weighted_sum_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_float]
    L.mrag_ddim_v_step_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64] + [c_float] * 7
    L.mrag_resize_patchify_bf16.argtypes = [c_void_p, POINTER(ResizePatchArgs)]
    L.mrag_resampler_workspace_bytes.argtypes = [c_int32] * 7
    L.mrag_resampler_workspace_bytes.restype = c_int64
    L.mrag_resampler_fwd.argtypes = [c_void_p, POINTER(ResamplerArgs)]
    L.mrag_cama_encoder_workspace_bytes.argtypes = [c_int32] * 4
    L.mrag_cama_encoder_workspace_bytes.restype = c_int64
    L.mrag_cama_encoder_fwd.argtypes = [c_void_p, POINTER(CamaEncoderArgs)]
    L.mrag_denormalize_u8.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int32]
    L.mrag_cfg_dpm_step_bf16.argtypes = [c_void_p] * 5 + [c_int64] + [c_float] * 8 + [c_int32]
    L.mrag_blend_tile_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 8
    L.mrag_softmax_rows_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_float]
    L.mrag_assemble_tokens_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32]
    for name in SYMBOLS:          # everything that did not declare a 64-bit / pointer result above returns an int status
        fn = getattr(L, name)
        if name not in _NON_INT_RESULT:
            fn.restype = c_int32
    _lib = L
    return L


class HipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "MRAG_EINVAL", -2: "MRAG_ENOTSUP"}.get(rc, f"hipError {rc}")
        raise HipError(f"{what} failed: {kind}")


if __name__ == "__main__":
    print(build(verbose=True))
