// attn_flash.hip -- head_dim-64 bf16 attention forward for gfx950 (flash-style, online softmax).
//
//   O[b, q, h, :] = resid + out_scale * softmax(Q K^T * scale  [masked])  V
//
// One kernel serves every F.scaled_dot_product_attention call site of the MotionRAG hot path
// (include/mrag_hip.h): the 17 776-token CogVideoX joint attention, the 25-key motion
// ("ip") cross-attention with its fused `hidden + scale * ip` update, the Perceiver resampler
// (25 queries x 1 593 keys) and the block-causal CAMA encoder (bool mask).
//
// CDNA4 design:
//   * v_mfma_f32_32x32x16_bf16 with SWAPPED products: S^T = K . Q^T and O^T = V^T . P^T, so a lane
//     owns one query row -- row max / row sum are lane-local plus one v_permlane32_swap, and the S
//     accumulator registers are already the B operand of the PV product (no LDS round trip for P);
//   * each wave owns 32 query rows (Q fragments live in registers, pre-multiplied by
//     scale*log2 e so the softmax is a bare v_exp_f32); a workgroup is NW waves;
//   * K/V tiles of 64 keys are staged by 16-byte LDS-DMA (global_load_lds) into two LDS stages;
//     K is XOR-swizzled on the source side for conflict-free ds_read_b128, V keeps row-major
//     [key][d] with its 64-byte halves swapped on odd key pairs and is consumed through
//     ds_read_b64_tr_b16 (hardware transpose) as the A operand of O^T = V^T . P^T;
//   * the running max is carried as the MFMA's C operand (S' = K.Q^T - m comes out of the chain,
//     no per-score subtract) and only moved when a tile raises it by more than THR (deferred
//     rescale): the rare path rescales O, l and the pending S' together;
//   * grid = q-tiles x (b, h) with an XCD-aware order: all q-tiles of one (b, h) run on one XCD,
//     so its K/V (4.5 MB at S = 17 776) streams from that XCD's L2.
#include "common.h"
#include "../../include/mrag_hip.h"

namespace {

struct AttnP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; const bf16_t* resid; const uint8_t* mask;
  long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss;
  int B, H, Sq, Skv, kv_div, n_qtiles;
  float qscale, out_scale;
};

constexpr float kThr = 5.0f;  // deferred-rescale threshold in log2 units (P <= 32)
constexpr int KVB = 64;       // keys per tile
constexpr int STAGE = 2 * KVB * 128;  // K tile + V tile, bytes

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8 scale_frag(u32x4 raw, float s) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(raw[i] << 16) * s;
    const float hi = __uint_as_float(raw[i] & 0xffff0000u) * s;
    r[i] = pack_bf2(lo, hi);
  }
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ float max3_asm(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int NW, bool HAS_MASK>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const AttnP p) {
  constexpr int PPW = 16 / NW;  // 1 KiB DMA pieces per wave per tile (8 K pieces + 8 V pieces)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r32 = lane & 31;

  // ---- XCD-aware block -> (q-tile, b, h)
  const int nbh = p.B * p.H;
  int bh, qt;
  if ((nbh & 7) == 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    bh = (j / p.n_qtiles) * 8 + x;
    qt = j % p.n_qtiles;
  } else {
    bh = blockIdx.x / p.n_qtiles;
    qt = blockIdx.x % p.n_qtiles;
  }
  const int b = bh / p.H, h = bh % p.H;
  const int bkv = b / p.kv_div;

  const int q0 = qt * (NW * 32) + wave * 32;
  const bool wave_active = q0 < p.Sq;
  const int qrow = q0 + r32;
  const int qrow_c = qrow < p.Sq ? qrow : p.Sq - 1;

  // ---- Q fragments: B operand of S^T = K.Q^T : lane holds Q[q = lane&31][d = 16 ks + 8 hh + j]
  bf16x8 qf[4];
  {
    const bf16_t* qp = p.Q + (long long)b * p.q_sb + (long long)qrow_c * p.q_ss + (long long)h * p.q_sh + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 raw = *(const u32x4*)(qp + ks * 16);
      qf[ks] = p.qscale == 1.0f ? __builtin_bit_cast(bf16x8, raw) : scale_frag(raw, p.qscale);
    }
  }

  // ---- DMA source bookkeeping
  const bf16_t* kbase = p.K + (long long)bkv * p.k_sb + (long long)h * p.k_sh;
  const bf16_t* vbase = p.V + (long long)bkv * p.v_sb + (long long)h * p.v_sh;
  const int prow = lane >> 3, ppos = lane & 7;

  auto issue = [&](int stage, int t) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + i * NW;  // 0..7 -> K, 8..15 -> V
      const bool isv = piece >= 8;
      const int kit = (piece & 7) * 8 + prow;  // key inside the tile
      long long key = (long long)t * KVB + kit;
      key = key < p.Skv ? key : p.Skv - 1;  // tail keys re-read a valid row; their scores are masked
      const int chunk = isv ? (ppos ^ (((kit >> 1) & 1) << 2)) : (ppos ^ ((kit >> 1) & 7));
      const bf16_t* src = isv ? (vbase + key * p.v_ss) : (kbase + key * p.k_ss);
      glds16(src + chunk * 8, base + piece * 1024);
    }
  };

  // ---- fragment read addresses (bytes, relative to the stage base)
  // K (A operand of K.Q^T): lane reads row key = kb*32 + r32, 16-byte chunk (2 ks + hh) ^ ((key>>1)&7)
  const int k_row_off = r32 * 128;
  const int k_swz = (r32 >> 1) & 7;
  // V^T (A operand of V^T.P^T) through ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the
  // address of key row (k0 + q), columns 4p..4p+3 of the block; it receives column (lane & 15).
  const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int v_lane_off = KVB * 128 + (4 * hh + q4) * 128 + (g16 & 1) * 32 + p4 * 8;
  const int v_half0 = (q4 >> 1) * 64;        // d-tile 0: 64-byte half index 0 ^ (key>>1 & 1)
  const int v_half1 = (1 - (q4 >> 1)) * 64;  // d-tile 1

  f32x16 o0, o1, negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; negm[i] = 0.f; }
  float m_run = 0.f, l_run = 0.f;

  const int nt = (p.Skv + KVB - 1) / KVB;
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) issue((t + 1) & 1, t + 1);
    if (!wave_active) continue;
    const char* st = smem + (t & 1) * STAGE;

    // ---- S'^T = K . Q^T - m  (two 32-key blocks)
    f32x16 s0 = negm, s1 = negm;
    {
      bf16x8 k0f[4], k1f[4];  // all 8 K fragments in flight before the first MFMA
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int coff = ((2 * ks + hh) ^ k_swz) * 16;
        k0f[ks] = *(const bf16x8*)(st + k_row_off + coff);
        k1f[ks] = *(const bf16x8*)(st + 32 * 128 + k_row_off + coff);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0f[ks], qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1f[ks], qf[ks], s1, 0, 0, 0);
      }
    }
    // register i of block kb holds key t*64 + kb*32 + (i&3) + 8*(i>>2) + 4*hh for query lane&31
    const int kbase_idx = t * KVB + 4 * hh;
    if (t == nt - 1 && (p.Skv & (KVB - 1))) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kbase_idx + (i & 3) + 8 * (i >> 2);
        if (key >= p.Skv) s0[i] = -INFINITY;
        if (key + 32 >= p.Skv) s1[i] = -INFINITY;
      }
    }
    if constexpr (HAS_MASK) {
      const uint8_t* mrow = p.mask + (long long)qrow_c * p.Skv;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kbase_idx + (i & 3) + 8 * (i >> 2);
        if (key < p.Skv && mrow[key]) s0[i] = -INFINITY;
        if (key + 32 < p.Skv && mrow[key + 32]) s1[i] = -INFINITY;
      }
    }

    // ---- tile max over this lane's 32 scores and the other half-wave's 32
    // v_max3_f32 through asm: plain fmaxf() on MFMA outputs makes hipcc emit a canonicalising
    // v_max per operand (3x the VALU work on the softmax critical path)
    float tm = max3_asm(s0[0], s1[0], s0[1]);
    tm = max3_asm(tm, s1[1], s0[2]);
#pragma unroll
    for (int i = 2; i < 15; ++i) tm = max3_asm(tm, s1[i], s0[i + 1]);
    tm = max3_asm(tm, s1[15], s1[15]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tm), __float_as_uint(tm), false, false);
      tm = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    // deferred rescale: move the running max only on the first tile or when a row grew past THR
    const bool first = (t == 0);
    if (first || __any(tm > kThr)) {
      float delta = first ? fmaxf(tm, -1e30f) : fmaxf(tm, 0.f);
      if (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
      }
      m_run += delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s0[i] -= delta; s1[i] -= delta; negm[i] = -m_run; }
    }

    // ---- P = exp2(S'), row-sum, pack to bf16 B fragments (key order is already the MFMA k order)
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s0[i] = __builtin_amdgcn_exp2f(s0[i]);
      s1[i] = __builtin_amdgcn_exp2f(s1[i]);
      lsum += s0[i] + s1[i];
    }
    l_run += lsum;
    bf16x8 pb[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 w0, w1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w0[j] = pack_bf2(s0[8 * s + 2 * j], s0[8 * s + 2 * j + 1]);
        w1[j] = pack_bf2(s1[8 * s + 2 * j], s1[8 * s + 2 * j + 1]);
      }
      pb[s] = __builtin_bit_cast(bf16x8, w0);
      pb[2 + s] = __builtin_bit_cast(bf16x8, w1);
    }

    // ---- O^T += V^T . P^T   (4 key steps of 16, two 32-wide d tiles)
    // The 16 transposed reads go through ONE asm statement: hipcc cannot see that the
    // ds_read_tr16 builtin does not alias the LDS-DMA of the next tile and would drain it
    // (s_waitcnt vmcnt(0)) in the middle of the tile.  EXEC is all ones here (wave-uniform flow).
    u32x2 t0[8], t1[8];
    {
      const unsigned a0 = (unsigned)(size_t)(st + v_lane_off + v_half0);
      const unsigned a1 = (unsigned)(size_t)(st + v_lane_off + v_half1);
      asm volatile(
          "ds_read_b64_tr_b16 %0, %16 offset:0\n\t"
          "ds_read_b64_tr_b16 %1, %16 offset:1024\n\t"
          "ds_read_b64_tr_b16 %8, %17 offset:0\n\t"
          "ds_read_b64_tr_b16 %9, %17 offset:1024\n\t"
          "ds_read_b64_tr_b16 %2, %16 offset:2048\n\t"
          "ds_read_b64_tr_b16 %3, %16 offset:3072\n\t"
          "ds_read_b64_tr_b16 %10, %17 offset:2048\n\t"
          "ds_read_b64_tr_b16 %11, %17 offset:3072\n\t"
          "ds_read_b64_tr_b16 %4, %16 offset:4096\n\t"
          "ds_read_b64_tr_b16 %5, %16 offset:5120\n\t"
          "ds_read_b64_tr_b16 %12, %17 offset:4096\n\t"
          "ds_read_b64_tr_b16 %13, %17 offset:5120\n\t"
          "ds_read_b64_tr_b16 %6, %16 offset:6144\n\t"
          "ds_read_b64_tr_b16 %7, %16 offset:7168\n\t"
          "ds_read_b64_tr_b16 %14, %17 offset:6144\n\t"
          "ds_read_b64_tr_b16 %15, %17 offset:7168\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(t0[0]), "=&v"(t0[1]), "=&v"(t0[2]), "=&v"(t0[3]), "=&v"(t0[4]), "=&v"(t0[5]), "=&v"(t0[6]),
            "=&v"(t0[7]), "=&v"(t1[0]), "=&v"(t1[1]), "=&v"(t1[2]), "=&v"(t1[3]), "=&v"(t1[4]), "=&v"(t1[5]),
            "=&v"(t1[6]), "=&v"(t1[7])
          : "v"(a0), "v"(a1)
          : "memory");
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const u32x4 w0 = {t0[2 * kk][0], t0[2 * kk][1], t0[2 * kk + 1][0], t0[2 * kk + 1][1]};
      const u32x4 w1 = {t1[2 * kk][0], t1[2 * kk][1], t1[2 * kk + 1][0], t1[2 * kk + 1][1]};
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), pb[kk], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), pb[kk], o1, 0, 0, 0);
    }
  }

  if (!wave_active) return;
  // ---- epilogue: combine the two half-waves' row sums, normalise, fused residual, 8-byte stores
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if (qrow >= p.Sq) return;
  const float inv = p.out_scale / l_run;
  const long long obase = (long long)b * p.o_sb + (long long)qrow * p.o_ss + h * 64 + 4 * hh;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (dt ? o1[4 * g + e] : o0[4 * g + e]) * inv;
      const long long off = obase + dt * 32 + 8 * g;
      if (p.resid) {
        const u32x2 rr = *(const u32x2*)(p.resid + off);
        v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
        v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
      }
      u32x2 out;
      out[0] = pack_bf2(v[0], v[1]);
      out[1] = pack_bf2(v[2], v[3]);
      *(u32x2*)(p.O + off) = out;
    }
  }
}

template <int NW>
int launch_attn(hipStream_t s, AttnP p) {
  p.n_qtiles = (p.Sq + NW * 32 - 1) / (NW * 32);
  const dim3 grid(p.n_qtiles * p.B * p.H), block(NW * 64);
  const size_t lds = 2 * STAGE;
  if (p.mask) MRAG_LAUNCH((attn_fwd_kernel<NW, true>), grid, block, lds, s, p);
  else MRAG_LAUNCH((attn_fwd_kernel<NW, false>), grid, block, lds, s, p);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

}  // namespace

extern "C" int mrag_attn_fwd_bf16(void* stream, const mrag_attn_args* a) {
  if (!a || !a->Q || !a->K || !a->V || !a->O) return MRAG_EINVAL;
  if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Skv <= 0 || a->kv_batch_div <= 0) return MRAG_EINVAL;
  if (a->B % a->kv_batch_div != 0) return MRAG_EINVAL;
  // 16-byte fragment / DMA loads and 8-byte stores
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V) & 15) return MRAG_EINVAL;
  if ((a->q_sb | a->q_ss | a->q_sh | a->k_sb | a->k_ss | a->k_sh | a->v_sb | a->v_ss | a->v_sh) % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)a->O & 7) || (a->o_sb | a->o_ss) % 4 != 0) return MRAG_EINVAL;
  if (a->resid && ((uintptr_t)a->resid & 7)) return MRAG_EINVAL;
  AttnP p{};
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V;
  p.O = (bf16_t*)a->O; p.resid = (const bf16_t*)a->resid; p.mask = a->mask;
  p.q_sb = a->q_sb; p.q_ss = a->q_ss; p.q_sh = a->q_sh;
  p.k_sb = a->k_sb; p.k_ss = a->k_ss; p.k_sh = a->k_sh;
  p.v_sb = a->v_sb; p.v_ss = a->v_ss; p.v_sh = a->v_sh;
  p.o_sb = a->o_sb; p.o_ss = a->o_ss;
  p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Skv = a->Skv; p.kv_div = a->kv_batch_div;
  p.qscale = a->q_prescaled ? 1.0f : a->scale * 1.4426950408889634f;
  p.out_scale = a->out_scale;
  hipStream_t s = (hipStream_t)stream;
  if (a->Sq > 128) return launch_attn<8>(s, p);
  if (a->Sq > 32) return launch_attn<2>(s, p);
  return launch_attn<1>(s, p);
}
