// cama_seq.hip -- native launch sequencers of the CAMA building blocks (SURVEY 8b export list: `resampler_fwd`, `cama_encoder_fwd`):
//   mrag_resampler_fwd      Resampler.forward, src/projects/condition/resampler.py:151-174 (PerceiverAttention :81-105, FeedForward :52-63)
//   mrag_cama_encoder_fwd   torch.nn.TransformerEncoder(4 post-norm layers) under ActionTransformer.forward, src/projects/condition/module.py:303-305
// No kernel lives here: each function issues the SAME launches, in the same order and with the same arguments, as motionrag_amd/cama.py's
// Python-sequenced forms (so the results are bit-identical), but from C++ -- ~60 launches of a Resampler / ~28 of the encoder leave the host without
// a Python frame or a ctypes marshalling step between them, and a non-Python host gets CAMA as two calls.  All scratch comes from the caller
// (mrag_resampler_workspace_bytes / mrag_cama_encoder_workspace_bytes); nothing is allocated, nothing synchronises.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/mrag_hip.h"

namespace {

typedef unsigned short bf16_t;

inline int64_t al(int64_t n) { return (n + 255) / 256 * 256; }

inline int gemm(void* s, const void* A, const void* W, const void* bias, void* C, int64_t M, int64_t N, int64_t K, int epi, const void* resid) {
  mrag_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.W = W; g.bias = bias; g.C = C; g.resid = resid;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldr = N;
  g.epilogue = epi;
  return mrag_gemm_bf16(s, &g);
}

// C = epilogue(LayerNorm(A) . W^T).  Shipped: LayerNorm into `scratch`, then the plain GEMM.  The one-launch form (mrag_gemm_args.a_ln: the LayerNorm inside the
// few-row kernel's A load, bit-identical) was built for the round-5 review's "fuse dependent steps" item and MEASURED SLOWER: CAMA 1.63 ms per clip with it against
// 1.50 ms without, same box, interleaved, although it removes 16 of 114 launches (profiles/r6_cama_ln_in_a_load_ab.txt) -- every one of the 64-512 workgroups of a
// projection recomputes the statistics of its 32 rows (two passes over 64 KB behind a barrier) before its first MFMA, which costs a 250-row problem more than the
// 6 us LayerNorm launch it saves.  -DMRAG_CAMA_LNA builds the fused sequencer (tools/build_variant.sh); the C-ABI feature stays available and tested.
inline int ln(void* s, const void* x, void* y, const void* w, const void* b, int64_t rows, int64_t D, float eps, int64_t y_rows_per_batch, int64_t y_batch_stride);
inline int gemm_ln(void* s, const void* A, const void* lw, const void* lb, float eps, void* scratch, const void* W, void* C, int64_t M, int64_t N, int64_t K, int epi) {
  mrag_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.W = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.epilogue = epi;
  g.a_ln = 1; g.a_ln_gamma = lw; g.a_ln_beta = lb; g.a_ln_eps = eps;
#ifdef MRAG_CAMA_LNA         // developer A/B build (tools/build_variant.sh): the one-launch form
  const int rc = mrag_gemm_bf16(s, &g);
  if (rc != MRAG_ENOTSUP) return rc;
#else
  (void)g;
#endif
  const int rl = ln(s, A, scratch, lw, lb, M, K, eps, 0, 0);
  if (rl != 0) return rl;
  return gemm(s, scratch, W, nullptr, C, M, N, K, epi, nullptr);
}

inline int ln(void* s, const void* x, void* y, const void* w, const void* b, int64_t rows, int64_t D, float eps, int64_t y_rows_per_batch = 0,
              int64_t y_batch_stride = 0);
inline int ln(void* s, const void* x, void* y, const void* w, const void* b, int64_t rows, int64_t D, float eps, int64_t y_rows_per_batch, int64_t y_batch_stride) {
  mrag_ln_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.gamma = w; a.beta = b; a.rows = rows; a.D = D; a.ldx = D; a.ldy = D; a.eps = eps;
  a.y_rows_per_batch = y_rows_per_batch; a.y_batch_stride = y_batch_stride;
  return mrag_layernorm_bf16(s, &a);
}

#define TRY(expr)             \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_ != 0) return rc_; \
  } while (0)

struct ResamplerWs {
  bf16_t *x, *kv_in, *lat_a, *lat_b, *ln_lat, *q, *kv, *o, *h_ln, *h, *proj;
  void* attn_ws; int64_t attn_ws_bytes;   // the attention's own scratch (key-split tail), as ops.attention provides it
  int64_t bytes;
};

ResamplerWs carve_resampler(void* base, int64_t N, int64_t n1, int64_t nq, int64_t dim, int64_t out_dim, int64_t inner, int64_t ff) {
  const int64_t aws = mrag_attn_workspace_bytes((int32_t)N, (int32_t)(inner / 64), (int32_t)nq, (int32_t)(n1 + nq));
  ResamplerWs w;
  int64_t off = 0;
  auto take = [&](int64_t elems) { bf16_t* p = base ? (bf16_t*)((char*)base + off) : nullptr; off += al(elems * 2); return p; };
  w.x = take(N * n1 * dim);
  w.kv_in = take(N * (n1 + nq) * dim);
  w.lat_a = take(N * nq * dim);
  w.lat_b = take(N * nq * dim);
  w.ln_lat = take(N * nq * dim);
  w.q = take(N * nq * inner);
  w.kv = take(N * (n1 + nq) * 2 * inner);
  w.o = take(N * nq * inner);
  w.h_ln = take(N * nq * dim);
  w.h = take(N * nq * ff);
  w.proj = take(N * nq * out_dim);
  w.attn_ws_bytes = aws;
  w.attn_ws = aws > 0 ? (void*)take((aws + 1) / 2) : nullptr;
  w.bytes = off;
  return w;
}

}  // namespace

extern "C" int64_t mrag_resampler_workspace_bytes(int32_t N, int32_t n1, int32_t nq, int32_t dim, int32_t output_dim, int32_t heads, int32_t ff_dim) {
  if (N <= 0 || n1 <= 0 || nq <= 0 || dim <= 0 || output_dim <= 0 || heads <= 0 || ff_dim <= 0) return 0;
  return carve_resampler(nullptr, N, n1, nq, dim, output_dim, (int64_t)heads * 64, ff_dim).bytes;
}

extern "C" int mrag_resampler_fwd(void* stream, const mrag_resampler_args* a) {
  if (!a || !a->x || !a->out || !a->latents || !a->proj_in_w || !a->proj_out_w || !a->layers || !a->workspace) return MRAG_EINVAL;
  if (a->N <= 0 || a->n1 <= 0 || a->nq <= 0 || a->depth <= 0 || a->heads <= 0 || a->dim % 64 || a->embedding_dim % 64 || a->ff_dim % 64) return MRAG_EINVAL;
  if ((uintptr_t)a->workspace & 255) return MRAG_EINVAL;
  const int64_t N = a->N, n1 = a->n1, nq = a->nq, dim = a->dim, inner = (int64_t)a->heads * 64, ff = a->ff_dim, od = a->output_dim;
  const ResamplerWs w = carve_resampler(a->workspace, N, n1, nq, dim, od, inner, ff);
  if (a->workspace_bytes < w.bytes) return MRAG_EINVAL;
  hipStream_t hs = (hipStream_t)stream;
  // latents.repeat(N, 1, 1)  (resampler.py:158): zero + broadcast add of the [nq, dim] table
  if (hipMemsetAsync(w.lat_a, 0, (size_t)(N * nq * dim * 2), hs) != hipSuccess) return (int)hipGetLastError();
  TRY(mrag_add_rows_bf16(stream, w.lat_a, a->latents, w.lat_a, N * nq, dim, nq));
  TRY(gemm(stream, a->x, a->proj_in_w, a->proj_in_b, w.x, N * n1, dim, a->embedding_dim, MRAG_EPI_NONE, nullptr));                     // :159
  bf16_t* lat = w.lat_a;
  bf16_t* lat_next = w.lat_b;
  for (int l = 0; l < a->depth; ++l) {
    const mrag_resampler_layer& L = a->layers[l];
    if (!L.to_q || !L.to_kv || !L.to_out || !L.ff_w1 || !L.ff_w2) return MRAG_EINVAL;
    // PerceiverAttention.forward :81-105 -- LN1(x) and LN2(latents) land directly in the [x ; latents] concat buffer
    TRY(ln(stream, w.x, w.kv_in, L.norm1_w, L.norm1_b, N * n1, dim, a->eps, n1, (n1 + nq) * dim));
    TRY(ln(stream, lat, w.kv_in + n1 * dim, L.norm2_w, L.norm2_b, N * nq, dim, a->eps, nq, (n1 + nq) * dim));
    TRY(gemm_ln(stream, lat, L.norm2_w, L.norm2_b, a->eps, w.ln_lat, L.to_q, w.q, N * nq, inner, dim, MRAG_EPI_NONE));                     // to_q(norm2(latents))
    TRY(gemm(stream, w.kv_in, L.to_kv, nullptr, w.kv, N * (n1 + nq), 2 * inner, dim, MRAG_EPI_NONE, nullptr));                          // K rows first (chunk(2)) :96
    mrag_attn_args at;
    memset(&at, 0, sizeof(at));
    at.Q = w.q; at.K = w.kv; at.V = w.kv + inner; at.O = w.o;
    at.q_sb = nq * inner; at.q_ss = inner; at.q_sh = 64;
    at.k_sb = (n1 + nq) * 2 * inner; at.k_ss = 2 * inner; at.k_sh = 64;
    at.v_sb = at.k_sb; at.v_ss = at.k_ss; at.v_sh = 64;
    at.o_sb = nq * inner; at.o_ss = inner;
    at.B = (int32_t)N; at.H = a->heads; at.Sq = (int32_t)nq; at.Skv = (int32_t)(n1 + nq); at.kv_batch_div = 1;
    at.scale = 0.125f; at.out_scale = 1.0f;
    at.workspace = w.attn_ws; at.workspace_bytes = w.attn_ws_bytes;
    TRY(mrag_attn_fwd_bf16(stream, &at));
    TRY(gemm(stream, w.o, L.to_out, nullptr, lat_next, N * nq, dim, inner, MRAG_EPI_RESID, lat));                                        // attn(...) + latents :162
    { bf16_t* t = lat; lat = lat_next; lat_next = t; }
    TRY(gemm_ln(stream, lat, L.ff_ln_w, L.ff_ln_b, a->eps, w.h_ln, L.ff_w1, w.h, N * nq, ff, dim, MRAG_EPI_GELU_ERF));                     // gelu(ff1(ln(latents)))
    TRY(gemm(stream, w.h, L.ff_w2, nullptr, lat_next, N * nq, dim, ff, MRAG_EPI_RESID, lat));                                            // ff(...) + latents :163
    { bf16_t* t = lat; lat = lat_next; lat_next = t; }
  }
  TRY(gemm(stream, lat, a->proj_out_w, a->proj_out_b, w.proj, N * nq, od, dim, MRAG_EPI_NONE, nullptr));
  return ln(stream, w.proj, a->out, a->norm_out_w, a->norm_out_b, N * nq, od, a->eps);
}

extern "C" int64_t mrag_cama_encoder_workspace_bytes(int32_t B, int32_t L, int32_t d_model, int32_t ff_dim) {
  if (B <= 0 || L <= 0 || d_model <= 0 || ff_dim <= 0) return 0;
  const int64_t rows = (int64_t)B * L;
  return al(rows * 3 * d_model * 2) + 4 * al(rows * d_model * 2) + al(rows * ff_dim * 2) + al(mrag_attn_workspace_bytes(B, d_model / 64, L, L));
}

extern "C" int mrag_cama_encoder_fwd(void* stream, const mrag_cama_encoder_args* a) {
  if (!a || !a->x || !a->out || !a->layers || !a->workspace || a->B <= 0 || a->L <= 0 || a->num_layers <= 0) return MRAG_EINVAL;
  if (a->d_model != a->nhead * 64 || a->d_model % 64 || a->ff_dim % 64 || ((uintptr_t)a->workspace & 255)) return MRAG_EINVAL;
  if (a->workspace_bytes < mrag_cama_encoder_workspace_bytes(a->B, a->L, a->d_model, a->ff_dim)) return MRAG_EINVAL;
  const int64_t rows = (int64_t)a->B * a->L, d = a->d_model, ff = a->ff_dim;
  char* base = (char*)a->workspace;
  bf16_t* qkv = (bf16_t*)base; base += al(rows * 3 * d * 2);
  bf16_t* att = (bf16_t*)base; base += al(rows * d * 2);
  bf16_t* y = (bf16_t*)base; base += al(rows * d * 2);
  bf16_t* x1 = (bf16_t*)base; base += al(rows * d * 2);
  bf16_t* xl = (bf16_t*)base; base += al(rows * d * 2);   // a layer's output = the next layer's input and residual (the last layer writes a->out)
  bf16_t* f = (bf16_t*)base; base += al(rows * ff * 2);
  const int64_t aws = mrag_attn_workspace_bytes(a->B, a->nhead, a->L, a->L);
  const bf16_t* x = (const bf16_t*)a->x;
  for (int l = 0; l < a->num_layers; ++l) {
    const mrag_encoder_layer& E = a->layers[l];
    if (!E.in_proj_w || !E.out_proj_w || !E.lin1_w || !E.lin2_w) return MRAG_EINVAL;
    // post-norm layer: x = LN1(x + MHA(x)); x = LN2(x + W2 gelu(W1 x))
    TRY(gemm(stream, x, E.in_proj_w, E.in_proj_b, qkv, rows, 3 * d, d, MRAG_EPI_NONE, nullptr));
    mrag_attn_args at;
    memset(&at, 0, sizeof(at));
    at.Q = qkv; at.K = qkv + d; at.V = qkv + 2 * d; at.O = att; at.mask = a->mask;
    at.q_sb = (int64_t)a->L * 3 * d; at.q_ss = 3 * d; at.q_sh = 64;
    at.k_sb = at.q_sb; at.k_ss = at.q_ss; at.k_sh = 64;
    at.v_sb = at.q_sb; at.v_ss = at.q_ss; at.v_sh = 64;
    at.o_sb = (int64_t)a->L * d; at.o_ss = d;
    at.B = a->B; at.H = a->nhead; at.Sq = a->L; at.Skv = a->L; at.kv_batch_div = 1;
    at.scale = 0.125f; at.out_scale = 1.0f;
    at.workspace = aws > 0 ? (void*)base : nullptr; at.workspace_bytes = aws;
    TRY(mrag_attn_fwd_bf16(stream, &at));
    TRY(gemm(stream, att, E.out_proj_w, E.out_proj_b, y, rows, d, d, MRAG_EPI_RESID, x));
    TRY(ln(stream, y, x1, E.norm1_w, E.norm1_b, rows, d, a->eps));
    TRY(gemm(stream, x1, E.lin1_w, E.lin1_b, f, rows, ff, d, MRAG_EPI_GELU_ERF, nullptr));
    TRY(gemm(stream, f, E.lin2_w, E.lin2_b, y, rows, d, ff, MRAG_EPI_RESID, x1));
    bf16_t* dst = (l == a->num_layers - 1) ? (bf16_t*)a->out : xl;   // x (= xl from the previous layer) was last read by out_proj's residual: free to overwrite
    TRY(ln(stream, y, dst, E.norm2_w, E.norm2_b, rows, d, a->eps));
    x = dst;
  }
  return MRAG_OK;
}
