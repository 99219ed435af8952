// norm.hip -- HBM-bound normalisation kernels for gfx950: LayerNorm (+AdaLN modulation) and the
// per-head qk-LayerNorm + 3-D RoPE applied in place on a fused QKV buffer.
// One wavefront (64 lanes) per row, 16-byte bf16 vectors per lane, reductions by wave shuffles.
#include "common.h"
#include "../../include/mrag_hip.h"

#ifndef MRAG_LN_STREAM_WGS
#define MRAG_LN_STREAM_WGS 2        // persistent workgroups per CU of layernorm_stream_kernel (developer knobs: tools/build_variant.sh).  With the next row prefetched the
#define MRAG_LN_STREAM_PREFETCH 1   // kernel holds 144 registers of row + per-column vectors: two workgroups per CU
#endif

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void unpack8(const u32x4 r, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(r[i] << 16);
    f[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return r;
}

struct LnP {
  const bf16_t* x; bf16_t* y; const bf16_t* gamma; const bf16_t* beta;
  const bf16_t* shift0; const bf16_t* scale0; const bf16_t* shift1; const bf16_t* scale1;
  long long rows, D, ldx, ldy, rows_per_batch, split, mod_stride, y_rpb, y_bstride;
  float eps;
  int rms;   // RMSNorm (T5LayerNorm): no mean subtraction, variance = mean(x^2)
};

// y = LN(x) * gamma + beta ; y = y * (1 + scale[b]) + shift[b]
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= p.rows) return;
  const bf16_t* x = p.x + row * p.ldx;
  float v[MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const long long idx = ((long long)c * 64 + lane) * 8;
    if (idx < p.D) {
      unpack8(__builtin_nontemporal_load((const u32x4*)(x + idx)), v[c]);   // streamed once: nontemporal load / store, +6 % (115 vs 122 us at [35552, 3072])
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[c][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  const float mean = p.rms ? 0.f : wave_sum(sum) / (float)p.D;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const long long idx = ((long long)c * 64 + lane) * 8;
    if (idx < p.D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = __fsub_rn(v[c][e], mean); sq = __builtin_fmaf(d, d, sq); }   // (explicit roundings: gemm_skinny_kernel<.., LNA> repeats this arithmetic and must get the same bits)
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);

  const bf16_t* shift = nullptr; const bf16_t* scale = nullptr;
  if (p.shift0) {
    // rows fit 32 bits (host check): a 32-bit unsigned division is ~4x shorter than the 64-bit sequence
    const unsigned b = (unsigned)row / (unsigned)p.rows_per_batch, pos = (unsigned)row - b * (unsigned)p.rows_per_batch;
    shift = ((long long)pos < p.split ? p.shift0 : p.shift1) + (long long)b * p.mod_stride;
    scale = ((long long)pos < p.split ? p.scale0 : p.scale1) + (long long)b * p.mod_stride;
  }
  bf16_t* y = p.y_rpb > 0 ? p.y + (long long)((unsigned)row / (unsigned)p.y_rpb) * p.y_bstride + (long long)((unsigned)row % (unsigned)p.y_rpb) * p.ldy : p.y + row * p.ldy;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const long long idx = ((long long)c * 64 + lane) * 8;
    if (idx >= p.D) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = __fmul_rn(__fsub_rn(v[c][e], mean), rstd);
    if (p.gamma) {
      float g[8]; unpack8(*(const u32x4*)(p.gamma + idx), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = __fmul_rn(o[e], g[e]);
    }
    if (p.beta) {
      float bb[8]; unpack8(*(const u32x4*)(p.beta + idx), bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = __fadd_rn(o[e], bb[e]);
    }
    if (shift) {
      float sc[8], sh[8];
      unpack8(*(const u32x4*)(scale + idx), sc);
      unpack8(*(const u32x4*)(shift + idx), sh);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(o[e], __fadd_rn(1.0f, sc[e]), sh[e]);   // (explicit: layernorm_stream_kernel repeats this arithmetic bit for bit)
    }
    __builtin_nontemporal_store(pack8(o), (u32x4*)(y + idx));
  }
}

// Wide rows, many of them (round 6; the DiT's two AdaLN-modulated LayerNorms per block over [35 552, 3 072]): PERSISTENT waves.  layernorm_kernel above gives
// every row a fresh wave that loads the row AND, per row again, gamma / beta / scale / shift of its columns from L2 -- three bytes through the CU's
// vector-memory path for every byte of the row -- and retires after one row (3.8 TB/s alone, 4.6 in the step).  Here a wave keeps its lanes' columns for its
// whole life: the per-column vectors stay in registers as the packed bf16 they were loaded as (reloaded when the wave crosses a sample or the text / video
// boundary, i.e. a handful of times), rows are dealt to the waves round-robin (the rows in flight at any moment are neighbours in HBM), and row i + 1 is
// requested before row i is reduced: 5.2 TB/s.  The arithmetic is layernorm_kernel's operation for operation (same per-lane summation order, same explicit
// roundings), so the two kernels give the SAME BITS: a sequence-sharded rank (fewer rows: the per-row kernel) reproduces the unsharded model's rows.
// D = 512 MAXC exactly (every lane busy), no output row remap.
template <int MAXC>
__global__ __launch_bounds__(256, MRAG_LN_STREAM_WGS) void layernorm_stream_kernel(const LnP p) {
  const int lane = threadIdx.x & 63;
  const long long nw = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= p.rows) return;
  u32x4 pg[MAXC], pb[MAXC], psc[MAXC], psh[MAXC];     // gamma, beta, scale, shift of this lane's columns, packed bf16
  long long key_now = -1;
  u32x4 cur[MAXC], nxt[MAXC];
  auto load_row = [&](const long long row, u32x4 (&dst)[MAXC]) {
    const bf16_t* x = p.x + row * p.ldx + lane * 8;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) dst[c] = __builtin_nontemporal_load((const u32x4*)(x + c * 512));
  };
  load_row(gw, nxt);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int idx = (c * 64 + lane) * 8;
    pg[c] = p.gamma ? *(const u32x4*)(p.gamma + idx) : u32x4{0u, 0u, 0u, 0u};
    pb[c] = p.beta ? *(const u32x4*)(p.beta + idx) : u32x4{0u, 0u, 0u, 0u};
    psc[c] = u32x4{0u, 0u, 0u, 0u}; psh[c] = u32x4{0u, 0u, 0u, 0u};
  }
  for (long long row = gw; row < p.rows; row += nw) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) cur[c] = nxt[c];
    if (MRAG_LN_STREAM_PREFETCH && row + nw < p.rows) load_row(row + nw, nxt);
    if (p.shift0) {                              // the modulation vectors of this row's (sample, segment)
      const unsigned b = (unsigned)row / (unsigned)p.rows_per_batch, pos = (unsigned)row - b * (unsigned)p.rows_per_batch;
      const bool second = (long long)pos >= p.split;
      const long long key = 2LL * b + (second ? 1 : 0);
      if (key != key_now) {                      // (wave-uniform)
        key_now = key;
        const bf16_t* shift = (second ? p.shift1 : p.shift0) + (long long)b * p.mod_stride;
        const bf16_t* scale = (second ? p.scale1 : p.scale0) + (long long)b * p.mod_stride;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          const int idx = (c * 64 + lane) * 8;
          psc[c] = *(const u32x4*)(scale + idx);
          psh[c] = *(const u32x4*)(shift + idx);
        }
      }
    }
    // ---- statistics: two passes over the row in registers (the packed row is unpacked on the fly), layernorm_kernel's order and roundings
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      float v[8]; unpack8(cur[c], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[e];
    }
    const float mean = p.rms ? 0.f : wave_sum(sum) / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      float v[8]; unpack8(cur[c], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = __fsub_rn(v[e], mean); sq = __builtin_fmaf(d, d, sq); }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
    bf16_t* y = p.y + row * p.ldy + lane * 8;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      float v[8], o[8]; unpack8(cur[c], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = __fmul_rn(__fsub_rn(v[e], mean), rstd);
      if (p.gamma) {
        float g[8]; unpack8(pg[c], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __fmul_rn(o[e], g[e]);
      }
      if (p.beta) {
        float bb[8]; unpack8(pb[c], bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __fadd_rn(o[e], bb[e]);
      }
      if (p.shift0) {
        float sc[8], sh[8]; unpack8(psc[c], sc); unpack8(psh[c], sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(o[e], __fadd_rn(1.0f, sc[e]), sh[e]);
      }
      __builtin_nontemporal_store(pack8(o), (u32x4*)(y + c * 512));
    }
    if (!MRAG_LN_STREAM_PREFETCH && row + nw < p.rows) load_row(row + nw, nxt);
  }
}

// Narrow rows (round 5): the UNets' LayerNorms are over C = 320 / 640 / 1 280 channels of 258 048 / 64 512 / 16 128 pixel rows.  One wave per row leaves 24 of
// 64 lanes idle at C = 320 and retires a wave per 640 bytes (92 us = 3.6 TB/s at [258 048, 320]).  Here a row belongs to LPR = 8 / 16 / 32 lanes (CH = 5 sixteen-byte
// chunks each, chunk index = sub-lane + LPR c: the LPR lanes of a row read 16 LPR contiguous bytes per instruction), so a wave normalises 8 / 4 / 2 rows with every
// lane busy.  Plain LayerNorm only (gamma / beta, output row remap; no AdaLN modulation, no RMS form: those stay on layernorm_kernel); the same two-pass
// statistics in registers, summed over another lane partition -- results agree with layernorm_kernel to fp32 rounding of the statistics.
template <int LPR, int CH>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const LnP p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, rs = lane / LPR;
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * RPW + rs;
  const long long row = row0 < p.rows ? row0 : p.rows - 1;           // rows past the end re-read the last row; their stores are masked
  const bf16_t* x = p.x + row * p.ldx;
  float v[CH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    unpack8(__builtin_nontemporal_load((const u32x4*)(x + (sub + LPR * c) * 8)), v[c]);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[c][e];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)p.D;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)p.D + p.eps);
  bf16_t* y = p.y_rpb > 0 ? p.y + (long long)((unsigned)row / (unsigned)p.y_rpb) * p.y_bstride + (long long)((unsigned)row % (unsigned)p.y_rpb) * p.ldy : p.y + row * p.ldy;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int idx = (sub + LPR * c) * 8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd;
    if (p.gamma) {
      float g[8]; unpack8(*(const u32x4*)(p.gamma + idx), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= g[e];
    }
    if (p.beta) {
      float bb[8]; unpack8(*(const u32x4*)(p.beta + idx), bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += bb[e];
    }
    if (row0 < p.rows) __builtin_nontemporal_store(pack8(o), (u32x4*)(y + idx));
  }
}

struct QkP {
  bf16_t* qkv; const bf16_t* qg; const bf16_t* qb; const bf16_t* kg; const bf16_t* kb;
  const float* cos; const float* sin;
  int B, S, H, text_len; float eps, q_premul;
};

// one wave = 8 heads x 64 dims of one token's Q (or K); LN over the 8 lanes of a head, RoPE pairs lane-local
__global__ __launch_bounds__(256) void qknorm_rope_kernel(const QkP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hgroups = (p.H + 7) / 8;
  const long long units = (long long)p.B * p.S * 2 * hgroups;
  const int sub = lane >> 3, d0 = (lane & 7) * 8;
  for (long long u = (long long)blockIdx.x * 4 + wave; u < units; u += (long long)gridDim.x * 4) {
    const int hg = (int)(u % hgroups);
    const int which = (int)((u / hgroups) & 1);
    const long long tok = u / (2 * hgroups);
    const int s = (int)(tok % p.S);
    const int head = hg * 8 + sub;
    if (head >= p.H) continue;
    bf16_t* ptr = p.qkv + tok * (3LL * p.H * 64) + (long long)which * p.H * 64 + head * 64 + d0;
    float v[8];
    unpack8(*(const u32x4*)ptr, v);
    const bf16_t* g = which ? p.kg : p.qg;
    const bf16_t* bt = which ? p.kb : p.qb;
    if (g) {
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[e];
      sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
      const float mean = sum * (1.0f / 64.0f);
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] -= mean; sq += v[e] * v[e]; }
      sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
      const float rstd = rsqrtf(sq * (1.0f / 64.0f) + p.eps);
      float gg[8], bb[8];
      unpack8(*(const u32x4*)(g + d0), gg);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * gg[e];
      if (bt) {
        unpack8(*(const u32x4*)(bt + d0), bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bb[e];
      }
    }
    if (p.cos && s >= p.text_len) {
      const long long ro = (long long)(s - p.text_len) * 64 + d0;
      const f32x4 c0 = *(const f32x4*)(p.cos + ro), c1 = *(const f32x4*)(p.cos + ro + 4);
      const f32x4 s0 = *(const f32x4*)(p.sin + ro), s1 = *(const f32x4*)(p.sin + ro + 4);
      const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      const float ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
      float o[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // x_rotated = stack([-x_imag, x_real]); out = x*cos + x_rotated*sin   (diffusers apply_rotary_emb)
        o[2 * i] = v[2 * i] * cc[2 * i] - v[2 * i + 1] * ss[2 * i];
        o[2 * i + 1] = v[2 * i + 1] * cc[2 * i + 1] + v[2 * i] * ss[2 * i + 1];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = o[e];
    }
    if (!which && p.q_premul != 1.0f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.q_premul;
    }
    *(u32x4*)ptr = pack8(v);
  }
}

}  // namespace

extern "C" int mrag_layernorm_bf16(void* stream, const mrag_ln_args* a) {
  if (!a || !a->x || !a->y || a->rows <= 0 || a->D <= 0) return MRAG_EINVAL;
  if (a->D % 8 != 0 || a->ldx % 8 != 0 || a->ldy % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)a->x | (uintptr_t)a->y) & 15) return MRAG_EINVAL;
  if (a->D > 8192 || a->rows >= (1LL << 31) || a->rows_per_batch >= (1LL << 31) || a->y_rows_per_batch >= (1LL << 31)) return MRAG_ENOTSUP;
  if (a->shift0 && (!a->scale0 || !a->shift1 || !a->scale1 || a->rows_per_batch <= 0 || a->mod_stride % 8 != 0))
    return MRAG_EINVAL;
  LnP p{};
  p.x = (const bf16_t*)a->x; p.y = (bf16_t*)a->y; p.gamma = (const bf16_t*)a->gamma; p.beta = (const bf16_t*)a->beta;
  p.shift0 = (const bf16_t*)a->shift0; p.scale0 = (const bf16_t*)a->scale0;
  p.shift1 = (const bf16_t*)a->shift1; p.scale1 = (const bf16_t*)a->scale1;
  p.rows = a->rows; p.D = a->D; p.ldx = a->ldx; p.ldy = a->ldy;
  p.rows_per_batch = a->rows_per_batch; p.split = a->split; p.mod_stride = a->mod_stride; p.eps = a->eps; p.rms = a->rms;
  p.y_rpb = a->y_rows_per_batch; p.y_bstride = a->y_batch_stride;
  if (p.y_rpb < 0 || (p.y_rpb > 0 && p.y_bstride % 8 != 0)) return MRAG_EINVAL;
  const dim3 grid((unsigned)((a->rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (!a->shift0 && !a->rms && (a->D == 320 || a->D == 640 || a->D == 1280 || a->D == 1024) && a->rows >= 1024) {   // narrow rows: several rows per wave
    if (a->D == 320) MRAG_LAUNCH((layernorm_rows_kernel<8, 5>), dim3((unsigned)((a->rows + 31) / 32)), block, 0, s, p);
    else if (a->D == 640) MRAG_LAUNCH((layernorm_rows_kernel<16, 5>), dim3((unsigned)((a->rows + 15) / 16)), block, 0, s, p);
    else if (a->D == 1024) MRAG_LAUNCH((layernorm_rows_kernel<16, 8>), dim3((unsigned)((a->rows + 15) / 16)), block, 0, s, p);   // CAMA's media rows (15 680 x 1 024, into the K/V concat buffer): 36 -> ? us
    else MRAG_LAUNCH((layernorm_rows_kernel<32, 5>), dim3((unsigned)((a->rows + 7) / 8)), block, 0, s, p);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_LAYERNORM_ROWS);
    return MRAG_OK;
  }
#ifndef MRAG_LN_NO_STREAM
  if (a->D == 3072 && a->rows >= 8192 && p.y_rpb == 0) {       // the DiT's AdaLN LayerNorms: persistent waves, per-column factors in registers
    MRAG_LAUNCH(layernorm_stream_kernel<6>, dim3(256 * MRAG_LN_STREAM_WGS), block, 0, s, p);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_LAYERNORM_STREAM);
    return MRAG_OK;
  }
#endif
  // the row lives in MAXC x 8 registers per lane: the tightest instantiation keeps the most waves in flight (D = 3072: 76 VGPRs and
  // 88-90 us at [35552, 3072] with MAXC = 6 against 100 VGPRs and 112-115 us with MAXC = 8 -- 4.9 TB/s, the device's copy rate)
  if (a->D <= 1024) MRAG_LAUNCH(layernorm_kernel<2>, grid, block, 0, s, p);
  else if (a->D <= 2048) MRAG_LAUNCH(layernorm_kernel<4>, grid, block, 0, s, p);
  else if (a->D <= 3072) MRAG_LAUNCH(layernorm_kernel<6>, grid, block, 0, s, p);
  else if (a->D <= 4096) MRAG_LAUNCH(layernorm_kernel<8>, grid, block, 0, s, p);
  else MRAG_LAUNCH(layernorm_kernel<16>, grid, block, 0, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_LAYERNORM);
  return MRAG_OK;
}

extern "C" int mrag_qknorm_rope_bf16(void* stream, const mrag_qknorm_rope_args* a) {
  if (!a || !a->qkv || a->B <= 0 || a->S <= 0 || a->H <= 0) return MRAG_EINVAL;
  if ((uintptr_t)a->qkv & 15) return MRAG_EINVAL;
  if ((a->cos == nullptr) != (a->sin == nullptr)) return MRAG_EINVAL;
  if (a->cos && (((uintptr_t)a->cos | (uintptr_t)a->sin) & 15)) return MRAG_EINVAL;
  if (a->text_len < 0 || a->text_len > a->S) return MRAG_EINVAL;
  QkP p{};
  p.qkv = (bf16_t*)a->qkv; p.qg = (const bf16_t*)a->q_gamma; p.qb = (const bf16_t*)a->q_beta;
  p.kg = (const bf16_t*)a->k_gamma; p.kb = (const bf16_t*)a->k_beta;
  p.cos = a->cos; p.sin = a->sin; p.B = a->B; p.S = a->S; p.H = a->H; p.text_len = a->text_len;
  p.eps = a->eps; p.q_premul = a->q_premul;
  const long long units = (long long)a->B * a->S * 2 * ((a->H + 7) / 8);
  long long blocks = (units + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  MRAG_LAUNCH(qknorm_rope_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_QKNORM_ROPE);
  return MRAG_OK;
}
