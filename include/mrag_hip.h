/*
 * mrag_hip.h -- flat C ABI of libmrag_hip.so (gfx950 / MI355X only).
 *
 * The reference (MCG-NJU/MotionRAG) is 100 % Python: it has no FFI.  Every
 * "kernel" on its hot path is a stock ATen op reached from three Python
 * protocols.  The entry points below are what a HIP
 * backend for those call sites binds; each one cites the reference call site
 * it stands behind.  Host code (the motionrag_amd python modules) reaches them through
 * ctypes with torch.Tensor.data_ptr(); INTEGRATION.md shows the reference-side
 * stubs.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MRAG_E* code on a bad
 *     argument, or the positive hipError_t of a failed launch; nothing throws;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - all pointers are device pointers unless the name ends in _host;
 *   - bf16 tensors are raw uint16 bit patterns; fp32 accumulate everywhere;
 *   - no hidden allocation, no environment reads, and no global mutable state
 *     on the launch path: workspaces are passed in, developer tuning knobs are
 *     explicit `tuning` fields of the argument structs (0 = the shipped
 *     behaviour).  The ONE piece of process-wide state is a table of
 *     diagnostic launch counters (mrag_dispatch_counts) that nothing reads back;
 *   - row-major, innermost dimension contiguous; ld* / stride* are in ELEMENTS.
 */
#ifndef MRAG_HIP_H
#define MRAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRAG_OK 0
#define MRAG_EINVAL (-1)   /* bad shape / alignment / null pointer          */
#define MRAG_ENOTSUP (-2)  /* shape outside what the kernel was built for   */

/* library identity: returns the ABI version (bumped on any signature change) */
int mrag_abi_version(void);
/* returns "gfx950" -- the only code object in the library */
const char* mrag_target_arch(void);
/* hex digest (sha256, first 16 digits) of the sources this binary was built from -- every csrc file + this header + the per-file compile flags, fed in
 * by the build as -DMRAG_SOURCE_HASH: the loader compares it with the digest of the sources beside it and refuses a stale binary
 * (motionrag_amd/_lib.py: source_hash / lib).  "unstamped" for a hand build.                                                                    */
const char* mrag_source_hash(void);

/* ------------------------------------------------------------------------ */
/* Diagnostic launch counters: WHICH KERNEL an entry point dispatched to.     */
/* Every launch site of the library bumps one slot (relaxed atomic add on the */
/* host, nanoseconds); nothing on the launch path reads them.  Tests read     */
/* them around a module call to assert that a full-width shape really ran on   */
/* the production kernel (persistent four-wave GEMM, attn16, the folded motion */
/* branch, the wide / implicit-GEMM tiles) and not on a fallback tile.  A      */
/* launch recorded into a HIP graph counts once, at capture.                   */
/* ------------------------------------------------------------------------ */
enum mrag_kernel_id {
  MRAG_K_GEMM_W4 = 0,          /* gemm_w4_kernel: persistent four-wave 256x256x64, plain / GELU / residual / gate epilogues */
  MRAG_K_GEMM_W4_QKNORM_ROPE,  /* gemm_w4_kernel<MRAG_EPI_QKNORM_ROPE>: the fused QKV projection                           */
  MRAG_K_GEMM_W4_GEGLU,        /* gemm_w4_kernel<GEGLU>                                                                     */
  MRAG_K_GEMM_256x256,         /* gemm_bf16_kernel 8-wave 256x256 tile (16-wave developer variant included)                 */
  MRAG_K_GEMM_256x320,         /* 8-wave 256x320 tile (N = 320 / 960 / 1 600 ...)                                           */
  MRAG_K_GEMM_256x128,
  MRAG_K_GEMM_128x128,         /* the small-problem tile (< 192 tiles of 256x256)                                           */
  MRAG_K_GEMM_STREAMK_TAIL,
  MRAG_K_GEMM_N320K320,        /* gemm_k320_kernel: K = 320, N = 320 / 640 / 960 (k320_applies) with a 320-column weight slice resident in registers (the UNets' level-0 linears) */
  MRAG_K_GEMM_192x256,         /* 8-wave 192x256 tile: long-K problems whose 256-row tile grid leaves the last round mostly empty */
  MRAG_K_CONV3_W4,             /* 3x3 (and causal 3x3x3) implicit-GEMM convolution on the persistent four-wave kernel       */
  MRAG_K_CONV3_256x256, MRAG_K_CONV3_256x320, MRAG_K_CONV3_256x128, MRAG_K_CONV3_128x128, MRAG_K_CONV3_192x256,
  MRAG_K_CONVT_W4,             /* (3,1,1) temporal convolution on the persistent four-wave kernel                           */
  MRAG_K_CONVT_256x256, MRAG_K_CONVT_256x320, MRAG_K_CONVT_128x128, MRAG_K_CONVT_192x256,
  MRAG_K_CONVT_256x128,        /* (never dispatched today: mrag_conv_bf16 takes the 256x128 tile for 3x3 convolutions only; the id keeps launch_cfg's table one-to-one) */
  MRAG_K_ATTN16,               /* attn16_kernel, whole query tiles                                                          */
  MRAG_K_ATTN16_KSPLIT,        /* attn16_kernel with the key-split ragged tail (+ MRAG_K_ATTN_COMBINE)                      */
  MRAG_K_ATTN_FLASH,           /* attn_fwd_kernel (32x32x16): masked / biased / short launches                              */
  MRAG_K_ATTN_FLASH_KSPLIT,
  MRAG_K_ATTN_COMBINE,
  MRAG_K_ATTN_TINY,            /* <= 16 keys, one wave per (row, head) pair                                                 */
  MRAG_K_ATTN_SMALL,           /* mrag_attn_small_bf16                                                                      */
  MRAG_K_ATTN_FP8,             /* attn8_kernel                                                                              */
  MRAG_K_IP_ATTN_FOLDED,       /* ip_attn_folded_kernel                                                                     */
  MRAG_K_LAYERNORM, MRAG_K_LAYERNORM_ROWS /* several narrow rows per wave (C = 320 / 640 / 1 280) */, MRAG_K_QKNORM_ROPE,
  MRAG_K_GN_STATS, MRAG_K_GN_FOLD, MRAG_K_GN_APPLY, MRAG_K_GN_APPLY_MOD,
  MRAG_K_LAYERNORM_STREAM,     /* layernorm_stream_kernel: persistent waves, AdaLN factors folded into registers (D = 3 072, >= 8 192 rows) */
  MRAG_K_GN_STATS_FOLD,        /* gn_stats_kernel<true>: statistics + the fold by the sample's last-arriving workgroup (<= 128 chunks) */
  MRAG_K_TOPK_SCAN, MRAG_K_TOPK_SCAN_FUSED_MERGE, MRAG_K_TOPK_MERGE, MRAG_K_TOPK_MFMA,
  MRAG_K_GEMM_W4_TAIL_RECT,    /* (opt-in, MRAG_GEMM_TUNE_TAIL_RECT) the 128x128-tile launch behind a persistent launch whose last round would be nearly empty */
  MRAG_K_GEMM_W4_BATCHED_W,    /* gemm_w4_kernel<NONE, true>: per-sample weights (w_batch_stride) */
  MRAG_K_GEMM_SKINNY_LNA,      /* gemm_skinny_kernel<.., LNA>: LayerNorm of the A rows fused into the few-row GEMM's A load */
  MRAG_K_TOPK_DENSE,           /* topk_dense_kernel: the fan-out search in ONE launch (tables whose grid is resident at once: dense first scores, bounded grid wait, claimed finishing) */
  MRAG_K_TOPK_DENSE_FINISH,    /* topk_dense_finish_kernel: the finishing phase as its own launch (tables whose dense grid is not resident at once) */
  MRAG_K_GEMM_SKINNY,          /* gemm_skinny_kernel: M <= 256 (CAMA's latents / encoder tokens, the query embedder): eight waves split K, no LDS ring */
  MRAG_K_COUNT
};
/* copies min(n, MRAG_K_COUNT) counters into out_host (HOST memory) and returns MRAG_K_COUNT */
int mrag_dispatch_counts(uint64_t* out_host, int32_t n);
/* the enumerator's name without the MRAG_K_ prefix ("GEMM_W4", ...), NULL outside the range */
const char* mrag_dispatch_name(int32_t id);

/* Measurement probe (NOT on the product path; bench.py prints its result as `roofline.ceilings.mfma_bf16_sustained_tflops`): a register-resident loop of
 * v_mfma_f32_16x16x32_bf16 over the caller's (random) operand bits -- 256 workgroups x 8 waves x `iters` x 32 MFMAs, no LDS, no memory traffic inside the
 * loop.  What the part sustains on a dense bf16 matrix load in its current power state, against the nominal 2.5 PFLOP/s of SURVEY.md section 8(d).
 * operands: >= 16 bytes of bf16 bits, 16-byte aligned; out: 256 * 512 floats (a checksum per lane, written so the loop cannot be elided).               */
int64_t mrag_probe_mfma_flops(int32_t iters);
int mrag_probe_mfma_bf16(void* stream, const void* operands, int64_t operand_bytes, float* out, int32_t iters);
/* the same for the fp32 matrix pipe of the retrieval fan-out kernel (`retrieval_top12_768d.fp32_mfma_sustained_tflops` in bench.py's JSON): 256 workgroups x
 * 4 waves (one per SIMD, the fan-out kernel's shape) x `iters` x 32 v_mfma_f32_32x32x2_f32, nominal 157 TFLOP/s.  operands: >= 4 bytes of fp32 values,
 * 4-byte aligned; out: 256 * 256 floats.                                                                                                            */
int64_t mrag_probe_mfma_f32_flops(int32_t iters);
int mrag_probe_mfma_f32(void* stream, const void* operands, int64_t operand_bytes, float* out, int32_t iters);
/* the HBM stream every "HBM-bound" kernel is priced against (`roofline.ceilings.stream_copy_TBps`: bytes read + bytes written per second): a grid-stride copy
 * of `bytes` (a multiple of 16; both pointers 16-byte aligned) with nontemporal 16-byte loads and stores, four of each in flight per lane over contiguous 16-KiB runs, 16 workgroups per CU.          */
int mrag_probe_stream_copy(void* stream, const void* src, void* dst, int64_t bytes, int32_t variant /* 0 = shipped form; developer sweep: bits 0-3 kernel form, bits 4-11 workgroups per CU */);

/* ------------------------------------------------------------------------ */
/* GEMM: C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N])     bf16 in/out      */
/* Stands behind every nn.Linear on the path:                                 */
/*   attn.to_q/to_k/to_v/to_out, to_q_ip/to_k_ip/to_v_ip                      */
/*     src/projects/condition/attn_processor.py:65-73,103-105,129,209-211,    */
/*     250-252,276;  Resampler / PerceiverAttention / FeedForward             */
/*     src/projects/condition/encoders/resampler.py:45-52,93-105,157-166;     */
/*   nn.TransformerEncoder in_proj/out_proj/linear1/linear2                   */
/*     src/projects/condition/module.py:305; diffusers CogVideoXBlock FF.     */
/* Summation order: fp32 accumulation over K in ascending 32-deep MFMA steps   */
/* (every tiled kernel: the same bits whichever tile a size selects); a        */
/* problem of M <= 256 rows (CAMA's latents / encoder tokens, a query's        */
/* embedder) runs on the few-row kernel, whose eight waves take every eighth   */
/* K-step and add their partial sums in wave order -- bit-reproducible, but a   */
/* row computed inside a batch of <= 256 rows and inside a larger one may      */
/* differ in the last bit of the bf16 result.                                  */
/* ------------------------------------------------------------------------ */
enum mrag_epilogue {
  MRAG_EPI_NONE = 0,        /* C = acc + bias                                         */
  MRAG_EPI_GELU_TANH = 1,   /* C = gelu_tanh(acc + bias)      (CogVideoX FF)          */
  MRAG_EPI_GELU_ERF = 2,    /* C = gelu_erf(acc + bias)       (CAMA / Resampler FF)   */
  MRAG_EPI_RESID = 3,       /* C = resid + (acc + bias)                               */
  MRAG_EPI_GATE_RESID = 4,  /* C = resid + gate[b(m), n] * (acc + bias)  (AdaLN-zero) */
  MRAG_EPI_SILU = 5,        /* C = silu(acc + bias)            (timestep MLP)         */
  MRAG_EPI_GEGLU = 6,       /* C[M, N/2] = v * gelu_erf(g): W / bias rows interleaved in 16-row [value | gate] groups
                             * (GEGLU of lvdm/modules/attention.py:448-455 and diffusers' FeedForward, N % 32 == 0)  */
  MRAG_EPI_QKNORM_ROPE = 7  /* fused QKV projection of the joint attention (attn_processor.py:209-231): C = [Q | K | V] with
                             * N = 3 * qk_dmodel (or a suffix / prefix of the three: column block j is third qk_first + j, so a
                             * sequence-sharded rank can project [K | V] first, start their all-gather, and project Q under it); per-head LayerNorm(64) of the Q and K thirds (computed on the bf16-rounded
                             * projection, as the reference's norm_q / norm_k see it), RoPE on rows whose position inside the
                             * sample (m % rows_per_batch) is >= rope_text_len, Q multiplied by q_premul.  Same arithmetic as
                             * mrag_qknorm_rope_bf16 after a plain GEMM; MRAG_ENOTSUP when the launch cannot take the LDS-staged
                             * epilogue (small problems, unaligned C) -- run the two kernels then.                          */
};

typedef struct mrag_gemm_args {
  const void* A;      /* [M, K] bf16, lda                                   */
  const void* W;      /* [N, K] bf16, ldw  (nn.Linear weight layout)        */
  const void* bias;   /* [N] bf16 or NULL                                   */
  void* C;            /* [M, N] bf16, ldc                                   */
  const void* resid;  /* [M, N] bf16, ldr (EPI_RESID / EPI_GATE_RESID)      */
  /* EPI_GATE_RESID: row m belongs to sample b = m / rows_per_batch; rows with
   * (m % rows_per_batch) < split use gate0, the others gate1 (text vs video
   * tokens of the joint CogVideoX sequence).  gate pointers are [B, N] bf16
   * with batch stride gate_stride.                                          */
  const void* gate0;
  const void* gate1;
  int64_t M, N, K;
  int64_t lda, ldw, ldc, ldr;
  int64_t rows_per_batch, split, gate_stride;
  int32_t epilogue;   /* enum mrag_epilogue */
  /* MRAG_EPI_QKNORM_ROPE only (rows_per_batch = tokens per sample): */
  int32_t rope_text_len;
  const void* q_gamma; const void* q_beta;   /* [64] bf16 or NULL (no norm) */
  const void* k_gamma; const void* k_beta;
  const float* rope_cos;                     /* [rows_per_batch - rope_text_len, 64] fp32 or NULL */
  const float* rope_sin;
  int64_t qk_dmodel;                         /* D = H * 64 */
  float qk_eps, q_premul;
  int32_t qk_first;   /* MRAG_EPI_QKNORM_ROPE: which third the first qk_dmodel columns are (0 = Q, 1 = K, 2 = V); N = (1..3 - qk_first) * qk_dmodel */
  int32_t tuning;     /* developer knobs (tools/microbench.py), 0 = shipped: MRAG_GEMM_TUNE_* bits, bits 4-7 tile choice
                         (1 = 256x256 on 16 waves, 2 = 128x128), bits 8-15 GROUP_M of the tile order (0 = 4)           */
  int32_t geglu_act;  /* MRAG_EPI_GEGLU: 0 = v * gelu_erf(g) (diffusers / lvdm GEGLU), 1 = v * gelu_tanh(g) (T5 v1.1 "gated-gelu": gelu_new) */
  void* workspace;         /* optional scratch, 16-byte aligned, private to the call until it completes on `stream`; NULL = none        */
  float acc_scale;         /* MRAG_EPI_RESID: C = resid + acc_scale * (acc + bias); 0 means 1.  The SVD UNet's AlphaBlender over a residual branch
                              (a s + (1 - a)(s + c) = s + (1 - a) c) rides in the branch's last projection this way                            */
  int64_t workspace_bytes; /* with tuning & MRAG_GEMM_TUNE_STREAMK and >= mrag_gemm_workspace_bytes(M, N, K) bytes: the partial last round
                              of 256x256 tiles (e.g. 132 of 1 668 tiles on 256 CUs for the DiT's to_out / FF2) runs as a stream-K tail
                              launch -- its K-tiles dealt evenly over the CUs, partial sums exchanged through the workspace and summed in
                              K order by the last arriver (bit-reproducible run to run; differs from the plain launch by fp32 summation
                              order only).  OPT-IN: measured slower than the partial round on MI355X               */
  int64_t w_batch_stride;  /* != 0: PER-SAMPLE weights -- the rows of sample b = m / rows_per_batch multiply W + b * w_batch_stride (elements, a multiple of 8).
                              The motion branch's folded score GEMM (attn_processor.py:250-256: `to_q_ip` folded into each CFG sample's own motion keys) as ONE
                              launch for both samples.  MRAG_EPI_NONE, N % 128 == 0, K >= 320, M % rows_per_batch == 0, 16-byte aligned rows of C; any other
                              shape returns MRAG_ENOTSUP (the caller then loops over the samples).                                                          */
  /* a_ln = 1: A := LayerNorm over K of every row of A (eps a_ln_eps, [K] bf16 a_ln_gamma / a_ln_beta or NULL), rounded to bf16, in FRONT of the product -- the
   * `to_q(norm2(latents))` / `ff1(ln(latents))` pairs of CAMA's Perceiver layers (src/projects/condition/encoders/resampler.py:81-105, :52-63) as one launch.
   * Bit-identical to mrag_layernorm_bf16 followed by this GEMM.  Few-row problems only (M <= 256: the K-split kernel reads all of K per row tile anyway),
   * MRAG_EPI_NONE or MRAG_EPI_GELU_ERF; anything else returns MRAG_ENOTSUP and the caller launches the LayerNorm itself.                                       */
  const void* a_ln_gamma; const void* a_ln_beta;
  float a_ln_eps; int32_t a_ln;
} mrag_gemm_args;
enum { MRAG_GEMM_TUNE_NO_WIDE = 1, MRAG_GEMM_TUNE_NO_STAGED = 2, MRAG_GEMM_TUNE_GEGLU_NO_STAGED = 4, MRAG_GEMM_TUNE_STREAMK = 8,
       MRAG_GEMM_TUNE_NO_W4 = 1 << 16, /* keep long-K problems on the 8-wave 256x256 tile instead of the persistent four-wave kernel */
       MRAG_GEMM_TUNE_NO_SKINNY = 1 << 17, /* keep few-row problems (M <= 256) on the 128x128 tile instead of the K-split few-row kernel */
       MRAG_GEMM_TUNE_SKINNY_8 = 1 << 18, /* few-row kernel: eight waves x 64 columns also for K >= 2 048 (shipped there: sixteen waves x 32 columns) */
       MRAG_GEMM_TUNE_TAIL_RECT = 1 << 19 /* persistent kernel: a small partial last round (<= 32 tiles) as its own launch of 128x128 tiles.  OPT-IN: measured equal
                                              to the partial round it replaces (FF1 of the DiT: 2.24 vs 2.25 ms, the step unchanged: profiles/r6_microbench_items.txt) */ };

int mrag_gemm_bf16(void* stream, const mrag_gemm_args* args);
/* scratch bytes that let mrag_gemm_bf16 run its last, partial round of tiles as stream-K; 0 when the shape has nothing to gain */
int64_t mrag_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* ------------------------------------------------------------------------ */
/* Attention, head_dim 64, bf16, flash-style (no S x S matrix in HBM).       */
/*   O[b, sq, h*64+d] = resid + out_scale * softmax(Q K^T * scale [+mask]) V */
/* Stands behind F.scaled_dot_product_attention at                           */
/*   attn_processor.py:85-87,117-119 (SVD), :233-235,264-266 (CogVideoX),    */
/*   resampler.py:102, DynamiCrafter lvdm/modules/attention.py:189,199,215,  */
/*   and torch._native_multi_head_attention inside nn.TransformerEncoder     */
/*   (module.py:305, block-causal bool mask from module.py:131-135).         */
/* With resid != NULL it is the motion-injection update                      */
/*   hidden = hidden + scale * ip_attention   (attn_processor.py:139,273).   */
/* ------------------------------------------------------------------------ */
typedef struct mrag_attn_args {
  const void* Q;   /* element (b,h,s,d) at Q + b*q_sb + s*q_ss + h*q_sh + d */
  const void* K;
  const void* V;
  void* O;         /* element (b,s,h,d) at O + b*o_sb + s*o_ss + h*64 + d   */
  const void* resid;   /* same addressing as O, or NULL                     */
  const uint8_t* mask; /* [Sq, Skv] bytes, nonzero = blocked, or NULL       */
  int64_t q_sb, q_ss, q_sh;
  int64_t k_sb, k_ss, k_sh;
  int64_t v_sb, v_ss, v_sh;
  int64_t o_sb, o_ss;
  int32_t B, H, Sq, Skv;
  int32_t kv_batch_div; /* K/V batch index = b / kv_batch_div (einops repeat
                           'b ... -> (b r) ...' at attn_processor.py:108-110) */
  float scale;          /* softmax scale, 1/sqrt(64) for every call site     */
  float out_scale;      /* multiplies the attention output (adapter scale)   */
  int32_t q_prescaled;  /* nonzero: Q already carries scale*log2(e) (q_premul
                           of mrag_qknorm_rope_bf16); `scale` is then ignored */
  void* workspace;         /* optional scratch, 16-byte aligned, private to the
                              call until it completes on `stream`; NULL = none */
  int64_t workspace_bytes; /* >= mrag_attn_workspace_bytes(B, H, Sq, Skv) enables
                              the key-split tail of long sequences (same
                              result up to fp32 summation order)              */
  int32_t tuning;          /* developer knobs (tools/microbench.py), 0 = shipped:
                              MRAG_ATTN_TUNE_* bits                            */
  const float* bias;       /* additive score bias, fp32 [H, Sq, Skv] (head stride bias_sh elements, shared by the batch), or NULL:
                              softmax(scale * Q K^T + bias [masked]) V.  T5's relative position bias (transformers T5Attention:
                              `scores += position_bias`, scale = 1) -- the prompt encoder of the CogVideoX path.
                              Takes the per-score path of the 32x32x16 kernel, like a mask.                                        */
  int64_t bias_sh;
} mrag_attn_args;
enum { MRAG_ATTN_TUNE_NO_TINY = 1,   /* never take the <= 16-key one-wave-per-pair kernel          */
                                     /* 2, 4: retired (intra-wave pipelined tile, 4-wave long-sequence workgroups: measured slower, code pruned in round 3) */
       MRAG_ATTN_TUNE_LEGACY = 8 };  /* long unmasked sequences through the 32x32x16 kernel (the masked / biased path's kernel, kept
                                        selectable so tests can compare the two families on one input)                          */
                                     /* every other bit: retired A/B variants (ABI 8 removed 128..1024: the attn16 workgroup-shape variants and
                                        the attn32 family now live in tools/exp/ with their measurement tables)                  */

int mrag_attn_fwd_bf16(void* stream, const mrag_attn_args* args);

/* Scratch bytes with which mrag_attn_fwd_bf16 runs the ragged last query tile
 * (Sq % 192 or Sq % 256 rows per (b, h), by kernel family) as several short key-chunk workgroups + a merge
 * instead of B*H full-length stragglers; 0 when the shape has no such tail.  */
int64_t mrag_attn_workspace_bytes(int32_t B, int32_t H, int32_t Sq, int32_t Skv);

/* fp8 (OCP e4m3) attention path -- BASELINE config "DynamiCrafter-1024 UNet 16x576x1024 + CAMA, fp8 MFMA attention path": the
 * spatial self-attention SDPA of the UNets (lvdm/modules/attention.py:189; diffusers BasicTransformerBlock.attn1 of the SVD UNet).
 * Same argument struct and result as mrag_attn_fwd_bf16 (bf16 Q / K / V views in, bf16 O out, fused residual); inside: per-(batch, head)
 * amax -> power-of-two scales -> e4m3 Q (with scale * log2 e folded in), K, V in MFMA operand order in the workspace, then
 * v_mfma_scale_f32_32x32x64_f8f6f4 for Q K^T and P V (e4m3 P, fp32 softmax, fp32 accumulation).  `workspace` is REQUIRED
 * (>= mrag_attn_fp8_workspace_bytes).  MRAG_ENOTSUP unless: no mask, kv_batch_div == 1, Skv % 128 == 0, Skv >= 512, q_prescaled == 0.
 * Precision: 3 mantissa bits per operand -- 5-6 % relative Frobenius error against fp32 attention (tests/test_gpu_fp8.py: <= 8 %);
 * opt-in, never used for the bf16 headline workload.                                                                                  */
int64_t mrag_attn_fp8_workspace_bytes(int32_t B, int32_t H, int32_t Sq, int32_t Skv);
int mrag_attn_fwd_fp8(void* stream, const mrag_attn_args* args);

/* ------------------------------------------------------------------------ */
/* Motion-adapter branch with the query projection folded into the keys:      */
/*   hidden += scale * SDPA(to_q_ip(hidden), K_ip, V_ip)                       */
/* (attn_processor.py:250-273 / :93-127) where, per head h,                    */
/*   to_q_ip(hidden)_h . K_ip,h^T = hidden . (K_ip,h . Wq_h)^T = hidden . M_h^T */
/* M = [H * 32, D] (25 motion keys per head padded to 32, built once per clip   */
/* because the motion tokens do not change over the denoising steps), so the    */
/* [S, D] x [D, D] projection becomes a [S, D] x [D, 32 H] GEMM (mrag_gemm_bf16) */
/* and this kernel finishes: softmax over the `keys` valid scores of every      */
/* (row, head) times `scale`, times V_ip, added into `hidden` in place.         */
/*   scores [B*S, scores_ld] bf16, key k of head h at column key_stride h + k   */
/*   (key_stride 0 = 32: 64-byte aligned blocks; a value in [keys, 32] packs    */
/*   the heads -- 26 puts 48 heads x 25 keys into 1 280 columns, five instead    */
/*   of six 256-column tiles of the score GEMM.  A lane group reads the four     */
/*   aligned 16-byte chunks that cover a block, so ((key_stride h) mod 8) +      */
/*   keys <= 32 must hold for every head, and scores_ld >= the aligned start     */
/*   of the last block + 32);                                                    */
/*   v: element (kv batch, key, h, d) at v + kb*v_batch_stride + key*v_key_stride + 64 h + d; */
/*   hidden [B*S, hidden_ld] bf16; q batch b uses K/V batch b / kv_batch_div.   */
/* ------------------------------------------------------------------------ */
int mrag_ip_attn_folded_bf16(void* stream, const void* scores, const void* v, void* hidden, int32_t B, int64_t S, int32_t H, int32_t keys,
                             int32_t kv_batch_div, int64_t scores_ld, int64_t hidden_ld, int64_t v_batch_stride, int64_t v_key_stride,
                             float scale, float out_scale, int32_t key_stride);

/* ------------------------------------------------------------------------ */
/* LayerNorm (+ optional AdaLN modulation): y = LN(x)*gamma+beta, then       */
/*   y = y*(1+scale[b])+shift[b].  nn.LayerNorm at resampler.py:47,76-77,    */
/*   129; nn.TransformerEncoderLayer norm1/norm2; diffusers                  */
/*   CogVideoXLayerNormZero / AdaLayerNorm.              */
/* Rows with (m % rows_per_batch) < split use shift0/scale0, others 1.       */
/* ------------------------------------------------------------------------ */
typedef struct mrag_ln_args {
  const void* x;      /* [rows, D] bf16, ldx */
  void* y;            /* [rows, D] bf16, ldy */
  const void* gamma;  /* [D] bf16 or NULL    */
  const void* beta;   /* [D] bf16 or NULL    */
  const void* shift0; const void* scale0;   /* [B, D] bf16, stride mod_stride, or NULL */
  const void* shift1; const void* scale1;
  int64_t rows, D, ldx, ldy;
  int64_t rows_per_batch, split, mod_stride;
  /* optional output row remap (0 = off): row m is written at
   *   y + (m / y_rows_per_batch) * y_batch_stride + (m % y_rows_per_batch) * ldy
   * -- writes LN(x) straight into the [x ; latents] K/V input of the Perceiver
   * attention (torch.cat at resampler.py:95) without a copy.                  */
  int64_t y_rows_per_batch, y_batch_stride;
  float eps;
  int32_t rms;        /* nonzero: RMSNorm -- y = x * rsqrt(mean(x^2) + eps) * gamma, no mean subtraction (transformers T5LayerNorm) */
} mrag_ln_args;

int mrag_layernorm_bf16(void* stream, const mrag_ln_args* args);

/* ------------------------------------------------------------------------ */
/* Per-head qk LayerNorm (eps 1e-6, affine over 64) + 3-D RoPE on the video  */
/* tokens of Q and K, in place on a fused QKV buffer [B, S, 3, H, 64].       */
/* attn_processor.py:220-231 (attn.norm_q / norm_k, apply_rotary_emb).       */
/* q_premul multiplies Q after RoPE (folds softmax scale * log2 e).          */
/* ------------------------------------------------------------------------ */
typedef struct mrag_qknorm_rope_args {
  void* qkv;                 /* [B, S, 3*H*64] bf16, in place on the Q and K thirds */
  const void* q_gamma; const void* q_beta;   /* [64] bf16 or NULL (no norm)        */
  const void* k_gamma; const void* k_beta;
  const float* cos;          /* [S - text_len, 64] fp32 or NULL (no RoPE)          */
  const float* sin;
  int32_t B, S, H, text_len;
  float eps, q_premul;
} mrag_qknorm_rope_args;

int mrag_qknorm_rope_bf16(void* stream, const mrag_qknorm_rope_args* args);

/* ------------------------------------------------------------------------ */
/* Pointwise helpers of the denoising loop.                                   */
/* ------------------------------------------------------------------------ */
/* sinusoidal timestep embedding (diffusers Timesteps, flip_sin_to_cos=True,  */
/* freq_shift 0): out[b, :dim] bf16                                           */
int mrag_timestep_embedding_bf16(void* stream, const float* t, void* out, int32_t B, int32_t dim);
/* y = silu(x) elementwise, bf16 */
int mrag_silu_bf16(void* stream, const void* x, void* y, int64_t n);
/* y[r, :] = x[r, :] + table[r % period, :]   (sinusoid PE: position_embeddings.py:172-174) */
int mrag_add_rows_bf16(void* stream, const void* x, const void* table, void* y,
                       int64_t rows, int64_t D, int64_t period);
/* y = a + b elementwise bf16 */
int mrag_add_bf16(void* stream, const void* a, const void* b, void* y, int64_t n);
/* y[r, :] = x[r, :] + table[(r / div) % period, :] -- a per-frame / per-sample vector broadcast over a frame's pixels.
 * SVD (third-party diffusers 0.32.2 TransformerSpatioTemporalModel, reached from src/projects/svd/module.py:38-47):
 * `hidden_states_mix + emb[:, None, :]` (frame-index embedding) and TemporalResnetBlock's `+ temb` per frame;
 * with div = 1 it reproduces the `time_context` row order of the temporal cross-attention (row % batch).
 * `table_stride`: elements between table rows (0 = D; a multiple of 8): the table may be a column slice of a wider matrix -- the per-frame
 * projections of ALL residual blocks come out of one batched GEMM (svd_unet.TembBank).                               */
int mrag_add_bcast_bf16(void* stream, const void* x, const void* table, void* y, int64_t rows, int64_t D, int64_t div, int64_t period, int64_t table_stride);
/* out = a x + b y (fp32 inside, one rounding): diffusers AlphaBlender `alpha * x_spatial + (1 - alpha) * x_temporal`
 * of SpatioTemporalResBlock / TransformerSpatioTemporalModel (SVD UNet sites the reference adapts at
 * src/projects/svd/module.py:145-165).                                                                          */
int mrag_axpby_bf16(void* stream, const void* x, const void* y, void* out, int64_t n, float a, float b);
/* SVD classifier-free guidance with the per-frame scale linspace(min, max, F) + one Euler-discrete step on v-prediction
 * (diffusers StableVideoDiffusionPipeline.__call__ as driven by src/projects/svd/pipelines/pipeline.py:147-160 and the
 * in-tree c_skip / c_out / c_noise convention src/projects/svd/module.py:92-98):
 *   latents <- c_x * latents + c_v * (v_u + g[f] (v_c - v_u)),  f = (i / frame_elems) % F
 * v_pred [2, n] bf16 (uncond first), latents [n] bf16 in place, guidance [F] fp32 on the device.                   */
int mrag_cfg_euler_step_bf16(void* stream, const void* v_pred, void* latents, int64_t n, const float* guidance, int32_t F,
                             int64_t frame_elems, float c_x, float c_v);
/* condition_fusion 'mean' / 'weight' over the k retrieved references (src/projects/condition/utils.py:21-27; stage-1
 * pipelines cogvideox/pipeline.py:71-72, svd/pipelines/pipeline.py:103-104, DynamiCrafter inference.py:213-218):
 *   out[b, n] = (sum_k w[b, k] * x[b, k, n]) / div      fp32 weights / accumulation in k order, one rounding to bf16
 * x [B, K, n] bf16, w [B, K] fp32 on the device or NULL (all ones: mean = sum / K with div = K), n % 8 == 0.            */
int mrag_weighted_sum_bf16(void* stream, const void* x, const float* w, void* out, int32_t B, int32_t K, int64_t n, float div);
/* softmax(scale * Q K^T) V for SHORT sequences at head dims other than 64 (32, 80, 96, 128): plain fp32 FMAs, K and V of one (batch, head) resident in LDS
 * (Skv * head_dim * 4 bytes <= 128 KB), one workgroup per (batch, head).  Same argument struct as mrag_attn_fwd_bf16 (strides in elements; no mask / bias /
 * resid / kv_batch_div / q_prescaled: MRAG_ENOTSUP).  First user: the CLIP-ViT-H image encoder of the SVD path (257 tokens, 16 heads of 80; transformers
 * CLIPVisionModelWithProjection behind StableVideoDiffusionPipeline._encode_image, src/projects/svd/pipelines/pipeline.py:113-119) -- once per clip.     */
int mrag_attn_small_bf16(void* stream, const mrag_attn_args* args, int32_t head_dim);

/* ------------------------------------------------------------------------ */
/* CAMA building blocks as native launch sequences (`resampler_fwd`, `cama_encoder_fwd`).  No kernel of their own: they issue the launches
 * of the entry points above in the order motionrag_amd/cama.py does (bit-identical results), from C++, with caller-provided scratch.                */
/* ------------------------------------------------------------------------ */
/* Resampler.forward (src/projects/condition/resampler.py:151-174): x [N, n1, embedding_dim] -> out [N, nq, output_dim]; all weights bf16 in the
 * nn.Linear layout [out, in]; PerceiverAttention / FeedForward linears have no bias, head_dim 64.                                                    */
typedef struct mrag_resampler_layer {
  const void* norm1_w; const void* norm1_b;      /* PerceiverAttention.norm1 (media tokens)      [dim]        */
  const void* norm2_w; const void* norm2_b;      /* PerceiverAttention.norm2 (latents)           [dim]        */
  const void* to_q; const void* to_kv; const void* to_out;   /* [H*64, dim], [2*H*64, dim] (K rows first), [dim, H*64] */
  const void* ff_ln_w; const void* ff_ln_b;      /* FeedForward[0] LayerNorm                                   */
  const void* ff_w1; const void* ff_w2;          /* FeedForward[1] [ff_dim, dim], FeedForward[3] [dim, ff_dim] */
} mrag_resampler_layer;
typedef struct mrag_resampler_args {
  const void* x; void* out;
  const void* latents;                           /* [nq, dim] (the learned queries)                            */
  const void* proj_in_w; const void* proj_in_b; const void* proj_out_w; const void* proj_out_b; const void* norm_out_w; const void* norm_out_b;
  const mrag_resampler_layer* layers;            /* host array [depth]                                         */
  void* workspace; int64_t workspace_bytes;      /* >= mrag_resampler_workspace_bytes(...), 256-byte aligned   */
  int32_t N, n1, nq, embedding_dim, dim, output_dim, heads, depth, ff_dim;
  float eps;                                     /* every LayerNorm of the module (1e-5)                       */
} mrag_resampler_args;
int64_t mrag_resampler_workspace_bytes(int32_t N, int32_t n1, int32_t nq, int32_t dim, int32_t output_dim, int32_t heads, int32_t ff_dim);
int mrag_resampler_fwd(void* stream, const mrag_resampler_args* args);
/* torch.nn.TransformerEncoder of post-norm nn.TransformerEncoderLayer(d_model, nhead, ff_dim, activation gelu, batch_first) with a [L, L] byte mask
 * (nonzero = blocked; CAMA's block-causal mask, src/projects/condition/module.py:131-135,303-305): x [B, L, d_model] -> out [B, L, d_model].        */
typedef struct mrag_encoder_layer {
  const void* in_proj_w; const void* in_proj_b;  /* [3 d, d], [3 d]  (nn.MultiheadAttention in_proj: q | k | v) */
  const void* out_proj_w; const void* out_proj_b;
  const void* lin1_w; const void* lin1_b; const void* lin2_w; const void* lin2_b;
  const void* norm1_w; const void* norm1_b; const void* norm2_w; const void* norm2_b;
} mrag_encoder_layer;
typedef struct mrag_cama_encoder_args {
  const void* x; void* out; const uint8_t* mask;
  const mrag_encoder_layer* layers;              /* host array [num_layers]                                    */
  void* workspace; int64_t workspace_bytes;
  int32_t B, L, d_model, nhead, ff_dim, num_layers;
  float eps;
} mrag_cama_encoder_args;
int64_t mrag_cama_encoder_workspace_bytes(int32_t B, int32_t L, int32_t d_model, int32_t ff_dim);
int mrag_cama_encoder_fwd(void* stream, const mrag_cama_encoder_args* args);

/* ------------------------------------------------------------------------ */
/* Frozen feature encoders in front of CAMA: pixel side.   */
/* ------------------------------------------------------------------------ */
/* VideoMAEEmbedder.forward / preprocess (src/projects/condition/encoders/condition.py:378-400) and DINOImageEmbedder.forward /
 * CLIPImageEmbedder.preprocess (condition.py:503-507,599-604): frame gather (uniform 16-frame sampling), torchvision
 * Resize(antialias) + CenterCrop as precomputed separable tap tables, the (x + 1) / 2 value map and the mean / std normalisation folded
 * into one per-channel affine, written straight as the rows of the patch-embedding GEMM (transformers VideoMAEPatchEmbeddings: Conv3d
 * (pt, ph, pw) = (2, 16, 16); Dinov2PatchEmbeddings: Conv2d 14 x 14 with pt = 1):
 *   out[((n * T/pt + t/pt) * OH/ph + y/ph) * OW/pw + x/pw][((c * pt + t%pt) * ph + y%ph) * pw + x%pw]
 *       = scale[c] * sum_j wy[y][j] * (sum_k wx[x][k] * src[n, frame_idx[t], c, y0[y] + j, x0[x] + k]) + shift[c]
 * fp32 accumulation, one rounding to bf16; columns C*pt*ph*pw .. ldo are zeroed (K padding of the GEMM).                         */
typedef struct mrag_resize_patch_args {
  const void* src;                 /* [N, T_src, C, H, W] bf16 (or fp32 with src_fp32 = 1); rows of W contiguous pixels      */
  const int32_t* frame_idx;        /* [T] source frame of output frame t on the device, or NULL (identity)                    */
  const float* wy; const int32_t* y0; const int32_t* ny;   /* [OH, taps_y] weights, first source row, tap count per output row */
  const float* wx; const int32_t* x0; const int32_t* nx;   /* [OW, taps_x] ... per output column                               */
  void* out;                       /* [N * T/pt * OH/ph * OW/pw, ldo] bf16                                                     */
  int64_t s_n, s_t, s_c, ldo;      /* element strides of src over n, t, c; row stride of out                                   */
  int32_t N, T, C, H, W, OH, OW, taps_y, taps_x, pt, ph, pw, src_fp32;
  float scale[4], shift[4];        /* per channel (C <= 4)                                                                     */
  int32_t no_tiling;               /* developer knob: 1 = the per-pixel kernel instead of the LDS-tiled one (same results)      */
} mrag_resize_patch_args;
int mrag_resize_patchify_bf16(void* stream, const mrag_resize_patch_args* args);
/* ViT token assembly (transformers Dinov2Embeddings.forward: cat(cls, patches) + position table; VideoMAEEmbeddings.forward: P = 0):
 *   out[n, j, :] = (j < P ? prefix[j, :] : x[n, j - P, :]) + pos[j, :]     x [N, L, D], prefix [P, D], pos [L + P, D] or NULL, D % 8 == 0 */
int mrag_assemble_tokens_bf16(void* stream, const void* x, const void* prefix, const void* pos, void* out,
                              int64_t N, int32_t L, int32_t P, int32_t D);
/* Row softmax of a materialised score matrix, y[r, :] = softmax(scale * x[r, :]) (fp32 statistics, bf16 in / out; rows 16-byte aligned, ld % 8 == 0,
 * scale > 0; y may alias x: a thread rewrites only elements it has read itself): the middle of the KL-VAE decoder's single-head head_dim-512 AttnBlock (lvdm/modules/networks/ae_modules.py:54-79: bmm, * c^-0.5,
 * softmax(dim=2), bmm), whose two products run on mrag_gemm_bf16.                                                                                    */
int mrag_softmax_rows_bf16(void* stream, const void* x, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, float scale);
/* denormalize (src/utils/pipeline.py:178-184; VideoBaseModule.validation_step, src/projects/base_module.py:129-147): y = uint8(clip((x + 1) / 2, 0, 1) * 255),
 * x bf16 (or fp32 with src_fp32 = 1) with the rounding points of the torch ops on that dtype, truncating cast -- bit-exact byte output.               */
int mrag_denormalize_u8(void* stream, const void* x, void* y, int64_t n, int32_t src_fp32);
/* Seam blending of the tiled VAE decode / encode that `pipe.vae.enable_tiling()` (cogvideox/module.py:39) switches on -- diffusers
 * AutoencoderKLCogVideoX.tiled_decode / blend_v / blend_h: tile [T, th, tw, C] bf16 is blended in place, first over its top extent_v rows with the bottom rows
 * of `up` [T, up_h, tw, C] (weight y / extent), then over its left extent_h columns with the right columns of `left` [T, th, left_w, C]; extents are clipped
 * to the tiles' sizes as the reference clips them; NULL up / left = no neighbour on that side.  Tiles are visited row by row, so neighbours are already blended. */
int mrag_blend_tile_bf16(void* stream, void* tile, const void* up, const void* left, int32_t T, int32_t th, int32_t tw, int32_t C, int32_t up_h,
                         int32_t left_w, int32_t extent_v, int32_t extent_h);
/* patchify [Bl, F, C0, H, W] (+ [Bl, F, C1, H, W]) -> rows [B*F*(H/2)*(W/2), (C0+C1)*4],
 * batch b reads latent b % Bl (CFG duplication).  Conv2d(k=2,s=2) patch embed as GEMM. */
int mrag_patchify_bf16(void* stream, const void* src0, const void* src1, void* dst,
                       int32_t B, int32_t Bl, int32_t F, int32_t C0, int32_t C1, int32_t H, int32_t W);
/* unpatchify rows [B, F*(H/2)*(W/2), C*4] -> [B, F, C, H, W] */
int mrag_unpatchify_bf16(void* stream, const void* src, void* dst,
                         int32_t B, int32_t F, int32_t C, int32_t H, int32_t W);
/* CFG combine + CogVideoX DDIM step (v-prediction), SURVEY Appendix E:
 *   v = v_u + g (v_c - v_u); x0 = sa*x - sb*v; x_prev = a*x + b*x0
 * v_pred [2, n] bf16 (uncond first), latents [n] bf16 in place. */
int mrag_cfg_ddim_step_bf16(void* stream, const void* v_pred, void* latents, int64_t n,
                            float guidance, float sqrt_alpha_t, float sqrt_beta_t, float a_t, float b_t);
/* CFG combine + CogVideoXDPMScheduler.step (diffusers 0.32.2; `scheduler: "dpm"` of configs/cogvideox/MotionRAG_open.yml:189-194, selected at
 * src/projects/cogvideox/module.py:28-35) -- the SDE form of DPM-Solver++(2M), v-prediction:
 *   v = v_u + g (v_c - v_u); x0 = sa*x - sb*v; d = second_order ? m3*x0 - m4*x0_prev : x0; x <- m1*x - m2*d + m_noise*noise; x0_prev <- x0
 * with the host-side multipliers of get_variables / get_mult (h = lambda_next - lambda, r = h_last / h).  v_pred [2, n] bf16 (uncond first);
 * latents, x0_prev, noise [n] bf16; latents and x0_prev in place (x0_prev is only read when second_order).                                  */
int mrag_cfg_dpm_step_bf16(void* stream, const void* v_pred, void* latents, void* x0_prev, const void* noise, int64_t n, float guidance,
                           float sqrt_alpha_t, float sqrt_beta_t, float m1, float m2, float m3, float m4, float m_noise, int32_t second_order);

/* ------------------------------------------------------------------------ */
/* Retrieval: flat scan top-k (lancedb 0.14.0 `table.search(q).limit(k)`,     */
/* src/data/rag.py:54; metric L2 unless an index was built with 'dot',        */
/* tools/build_rag_database.py:51-52).                                        */
/*   dist(q, r) = sum_d (q_d - x_rd)^2            (metric 0, "l2")            */
/*              = 1 - sum_d q_d x_rd              (metric 1, "dot")           */
/* accumulated in fp32 as 16 interleaved fmaf chains (chain c takes the 4-wide */
/* feature groups g with g % 16 == c, in order) folded by a fixed butterfly    */
/* (c ^ 8, 4, 2, 1): bit-reproducible, the "f32chain" mode of                   */
/* oracle/topk_oracle.c; ranks equal the float64 oracle's whenever neighbouring */
/* distances differ by more than the fp32 rounding of the sum (tests assert the */
/* gap).  Rows with group[r] == exclude[q] are                                  */
/* skipped (the `video != "<self>"` filter, src/data/datamodule.py:235).      */
/* Output sorted by (dist asc, row asc); missing entries are row = -1.        */
/* `postfilter` selects WHEN the filter applies -- lancedb's                    */
/* `LanceQueryBuilder.where(where, prefilter=False)` (0.14.0's default, the    */
/* form src/data/rag.py:57-58 calls): 1 = the k nearest rows are selected      */
/* without the filter, excluded rows are then dropped and the survivors move   */
/* up (fewer than k results possible: the tail is row -1, dist +inf);          */
/* 0 = prefilter, excluded rows never enter the selection.                     */
/* `order` selects the summation order of the distance (both are DEFINED and   */
/* restated by oracle/topk_oracle.c, so rows AND distances are bit-exact):      */
/*   1 = the 16-chain order above (mode 0 of the oracle);                       */
/*   2 = the fan-out form for batches (src/data/datamodule.py:231-236 searches  */
/*       per annotation; rag.attach_ref_videos batches 256): the scan as an     */
/*       fp32 matrix product on v_mfma_f32_32x32x2_f32, the table streamed ONCE  */
/*       per 256 queries.  dist = one fmaf chain per (query, row) over the       */
/*       features in the order 8c, 8c+4, 8c+1, 8c+5, 8c+2, 8c+6, 8c+3, 8c+7;     */
/*       "l2": the matrix product SELECTS through fmaf(-2, q.x, |q|^2 + |x|^2)   */
/*       (squared norms as two half-block chains) -- an expansion whose absolute */
/*       error is an ulp of |q|^2 + |x|^2 (~1e-4 on unnormalised 768-d data: a   */
/*       row's distance to itself is noise) -- and the merge step then scores    */
/*       the 16 nearest candidates AGAIN in the order-1 form (direct sum of      */
/*       (q - x)^2) and re-ranks them: the distances returned are order 1's bit  */
/*       for bit, whatever the call shape (round 6; mode 2 of the oracle restates */
/*       selection + second scoring).  "dot": 1 - chain, no second scoring.       */
/*       n_queries >= 16, k <= 16, else MRAG_ENOTSUP;                            */
/*       Tables of one resident round of workgroups (10 000 x 256: BASELINE config #1) run it as ONE launch: the first scores of every (query,
 *       row) leave as a dense matrix in the workspace, every workgroup waits -- bounded -- until the grid has arrived, and the workgroups then
 *       finish the queries (16 nearest under the first score, second scoring, filter order).  Tables up to 65 536 rows / 64 MB of first scores
 *       whose grid is not resident at once do the same in TWO launches (no wait).  Larger tables stream: a |q|^2 pre-pass, the fan-out kernel
 *       with in-kernel lists, a merge launch.  All three compute the SAME defined result.
 *   3 = order 2, streaming (three-launch) form forced;   4 = order 2, one-launch form forced (MRAG_ENOTSUP where its plan does not apply) and
 *       WITHOUT waiting: every workgroup but the last arriver leaves at once and the last one finishes every query -- diagnostics of the two forms'
 *       equality and of the bounded wait's fall-back.                                                                                             */
/*   0 = automatic: 2 where it applies (faster at every table size measured,     */
/*       1 000 to 10^6 rows), else 1.  "l2": a row gets the same distance from   */
/*       both; WHICH rows make the list can differ only where the k-th and a     */
/*       later candidate are closer than the expansion's error AND the later one */
/*       ranks beyond 16 under it.  "dot": the two orders may differ in the last */
/*       bits of a distance (never in a rank whose gap exceeds the fp32 rounding */
/*       of the sum: tests compare both with the float64 oracle).                */
/* ------------------------------------------------------------------------ */
/* workspace: >= mrag_topk_workspace_bytes, 16-byte aligned.  Its first 64 bytes are arrival counters of the single-launch forms
 * (n_queries <= 4: scan + merge in ONE launch, the last workgroup to arrive merges; the one-launch fan-out form: words 8..14, two
 * counter sets used alternately): they must be ZERO before the first call on a workspace; every call leaves them ready for the next one
 * (of any form, on the same stream), so a workspace is zeroed once when it is allocated and never written by the caller afterwards.   */
int64_t mrag_topk_workspace_bytes(int64_t n_rows, int32_t n_queries);
int mrag_topk_f32(void* stream, const float* db, const int32_t* group, int64_t n_rows, int32_t dim,
                  const float* queries, const int32_t* exclude, int32_t n_queries,
                  int32_t k, int32_t metric,
                  int32_t* out_rows, float* out_dist, void* workspace, int64_t workspace_bytes, int32_t postfilter, int32_t order);

/* ------------------------------------------------------------------------ */
/* Spatio-temporal UNet denoisers (DynamiCrafter lvdm, SVD): channels-last   */
/* rows [(n, y, x), C] bf16, n = b*t frames.                                  */
/* ------------------------------------------------------------------------ */
/* nn.GroupNorm(G, C) over (HW x C/G) per (n, group) [+ per-(n,c) embedding   */
/* pre-add: `h + emb_out` of ResBlock._forward, openaimodel3d.py:216-229]     */
/* [+ SiLU].  lvdm/basics.py:81-88; openaimodel3d.py:152-181,258-268;         */
/* lvdm/modules/attention.py:286,357.  workspace: 16-byte aligned; its first  */
/* 16 KiB are per-sample arrival counters of the opt-in one-launch form        */
/* (`fold`): they must be ZERO before the first call on a workspace; every     */
/* call leaves them zero.  Behind them: fp32 partial sums, per-(n, c) scale /  */
/* shift.                                                                      */
typedef struct mrag_groupnorm_args {
  const void* x; void* y;          /* [N, HW, C] bf16                           */
  const void* gamma; const void* beta;   /* [C] bf16 or NULL                    */
  const void* emb;                 /* [N, C] bf16 (row stride emb_stride) or NULL */
  void* workspace;                 /* mrag_groupnorm_workspace_bytes(N, C, chunks) */
  int64_t N, HW, C, emb_stride;
  int32_t G, chunks, silu;
  float eps;
  /* spatially conditioned GroupNorm -- diffusers' CogVideoXSpatialNorm3D (the decoder norms of the VAE behind cogvideox/module.py:39-40):
   * y = GroupNorm(x) * conv_y(zq_up) + conv_b(zq_up), zq_up = nearest-neighbour F.interpolate of the latent to x's (T, H, W).  The 1x1x1
   * convolutions commute with the upsampling, so `mod` holds them at the LATENT resolution: [N, mod_Tz, mod_H >> mod_shift, mod_W >> mod_shift, 2C]
   * bf16 (conv_y | conv_b along the last axis); x rows are (t, y, x) with HW = mod_T * mod_H * mod_W.  mod_split = the odd-T form (frame 0
   * from latent frame 0, the remaining frames resampled from the remaining latent frames).  y_stride_n: elements between the samples of y
   * (0 = HW * C) so that the result can land behind the two context frames of a causal-convolution stack.  NULL mod = plain GroupNorm. */
  const void* mod;
  int32_t mod_T, mod_H, mod_W, mod_Tz, mod_shift, mod_split;
  int64_t y_stride_n;
  int32_t fold;   /* 1: statistics + fold in ONE launch where the shape allows (<= 128 chunks; the sample's last-arriving workgroup folds).  Opt-in: measured 5 %
                     slower on the UNet CFG steps than the separate fold launch (every workgroup pays a release in front of its ticket). 0 = shipped. */
} mrag_groupnorm_args;
int64_t mrag_groupnorm_workspace_bytes(int64_t N, int64_t C, int32_t chunks);
int mrag_groupnorm_bf16(void* stream, const mrag_groupnorm_args* args);
/* row gather in front of the implicit-GEMM 3x3 convolution (pad 1, stride 1|2,
 * optional nearest x2 upsample of the source: openaimodel3d.py:52-107):
 * dst[(n,yo,xo), (ky,kx,c)] -> [N*Ho*Wo, Kpad], columns >= 9C zero.            */
int mrag_im2col3x3_bf16(void* stream, const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t stride, int32_t upsample, int32_t Kpad);
/* Implicit-GEMM convolutions on channels-last activations (no materialised im2col): the MFMA GEMM's LDS-DMA gathers the
 * tap-shifted pixel rows itself, zero rows outside the image / clip.  Cin % 64 == 0.
 *   MRAG_CONV_3X3: nn.Conv2d(Cin, Cout, 3, stride 1|2, padding 1) [+ nearest x2 upsample in front]
 *                  (ResBlock / Downsample / Upsample, lvdm/modules/networks/openaimodel3d.py:52-107,211-237; diffusers
 *                  ResnetBlock2D / Downsample2D / Upsample2D of the SVD UNet): x [N, H, Wd, Cin] -> y [N, Ho, Wo, Cout];
 *                  W [Cout, (ky, kx, cin)].
 *   MRAG_CONV_T3:  nn.Conv3d(Cin, Cout, (3,1,1), padding (1,0,0)) (TemporalConvBlock, openaimodel3d.py:240-281; diffusers
 *                  TemporalResnetBlock): x [(N = B) * (H = T), Wd = HW, Cin] -> y same rows x Cout; W [Cout, (kt, cin)].
 * epilogue: MRAG_EPI_NONE or MRAG_EPI_RESID (resid has the shape of y).                                                 */
enum { MRAG_CONV_3X3 = 1, MRAG_CONV_T3 = 2 };
typedef struct mrag_conv_args {
  const void* x; const void* W; const void* bias; void* y; const void* resid;
  int32_t N, H, Wd, Cin, Cout;
  int32_t stride, upsample;        /* MRAG_CONV_3X3 only */
  int32_t mode, epilogue;
  int32_t asym_pad;                /* MRAG_CONV_3X3, stride 2: 0 = padding 1 on every side; 1 = zero row / column at the bottom / right only --
                                      `F.pad(x, (0, 1, 0, 1))` + Conv2d(3, stride 2, padding 0), the KL-VAE encoder's Downsample
                                      (lvdm/modules/networks/ae_modules.py:93-113): Ho = H / 2                                       */
  int32_t t_taps, t_frames;        /* MRAG_CONV_3X3, stride 1: t_taps = 3 turns the convolution into the causal 3x3x3 one of diffusers'
                                      CogVideoXCausalConv3d (the VAE behind cogvideox/module.py:39-40): x holds, per sample, the two context frames
                                      (conv cache, or the first frame twice) followed by t_frames frames -> x [(N / t_frames) (t_frames + 2), H, Wd, Cin],
                                      y [N, H, Wd, Cout], W [Cout, (kt, ky, kx, cin)]; zero padding in space, none in time.  0 = 2-D.            */
  float acc_scale;                 /* MRAG_EPI_RESID: y = resid + acc_scale * (conv + bias); 0 means 1                                          */
} mrag_conv_args;
int mrag_conv_bf16(void* stream, const mrag_conv_args* args);
/* row gather for nn.Conv3d((3,1,1), padding (1,0,0)), openaimodel3d.py:256-268:
 * dst[(b,t,hw), (kt,c)] = src[b, t+kt-1, hw, c]                                */
int mrag_unfold_t3_bf16(void* stream, const void* src, void* dst, int32_t B, int32_t T, int64_t HW, int32_t C);
/* GEGLU (attention.py:448-455): y[r, j] = x[r, j] * gelu(x[r, inner + j])      */
int mrag_geglu_bf16(void* stream, const void* x, void* y, int64_t rows, int64_t inner);
/* DDIMSampler.p_sample_ddim, v-parameterisation (samplers/ddim.py:203-298):
 * v = v_u + s (v_c - v_u) with v_pred = [cond ; uncond] (cond FIRST, :219-237);
 * eps = sa v + sb x; x0 = (sa x - sb v) * rescale; x <- sqrt_aprev x0 + dir eps
 * + sigma noise.  x, noise fp32 [n] (noise pre-generated on the host, NULL = 0) */
int mrag_ddim_v_step_f32(void* stream, const void* v_pred, float* x, const float* noise, int64_t n, float guidance,
                         float sqrt_alpha_t, float sqrt_one_minus_alpha_t, float rescale, float sqrt_alpha_prev,
                         float dir_coef, float sigma);

/* ------------------------------------------------------------------------ */
/* Multi-GPU exchange step: RCCL all-gather over xGMI on a    */
/* caller-chosen stream.  The reference's only collective use is Lightning DDP */
/* (configs/cogvideox/MotionRAG_open.yml:4-8); the hot path's exchanges are: the */
/* ranks' final latents at the end of the loop, a rank's K / V rows per block   */
/* (sequence sharding), a guidance branch's velocity (CFG pairs).               */
/* RCCL is resolved at run time (the process's librccl.so); MRAG_ENOTSUP when   */
/* there is none.  1000 + ncclResult_t on an RCCL failure.                      */
/*   id: 128 opaque bytes on the HOST: rank 0 creates it, the launcher's store  */
/*   carries it to the other ranks (motionrag_amd/dist.py: RcclComm).           */
/* ------------------------------------------------------------------------ */
int mrag_comm_unique_id(void* id128_host);
int mrag_comm_init(const void* id128_host, int32_t rank, int32_t world, void** comm_out);
int mrag_comm_destroy(void* comm);
/* recv[r * bytes_per_rank ..] = rank r's send buffer, on `stream`; returns after ENQUEUEING (asynchronous like a kernel launch) */
int mrag_allgather(void* stream, void* comm, const void* send, void* recv, int64_t bytes_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* MRAG_HIP_H */
